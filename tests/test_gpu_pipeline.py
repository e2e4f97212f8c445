"""vsa_pipeline_*: queries from host memory to matches in host memory, three
batches in flight.  The batches of a job, concatenated, must be the reference's
list (query numbers count over the whole job; -mum filters over all batches)."""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import gpu_index

pytestmark = pytest.mark.gpu


def run_job(V, gi, q, mode, L, per):
    m = int(q.length[0])
    p = V.Pipeline(gi, mode, L, m, per)
    sym = q.symbols.reshape(q.nq, m)
    out, first = [], 0
    while first < q.nq:
        buf = p.hostbuffer()
        while buf is None:                 # all slots in flight: take one
            rc, got = p.next()
            assert rc == 0
            out.append(got)
            buf = p.hostbuffer()
        n = min(per, q.nq - first)
        buf[:n * m] = sym[first:first + n].ravel()
        p.submit(n)
        first += n
    while True:
        rc, got = p.next()
        if rc == 1:
            break
        assert rc == 0
        out.append(got)
    res = np.concatenate(out) if out else np.zeros(0, V.MATCH_DTYPE)
    if mode == 3:
        assert len(res) == 0
        res, st = p.finish()
        assert st.count == len(res)
    p.close()
    return res


@pytest.mark.parametrize("per", [10000, 3000, 999, 64])
def test_batches_of_a_job_concatenate_to_the_reference_list(V, per):
    idx, q = H.load_case("c1")
    gi = gpu_index(V, "c1")
    gi.set_queryspeedup(2)
    for mode, key, L in ((0, "complete", 0), (2, "mumcand20", 20),
                         (3, "mum20", 20), (1, "mem20_sp2", 20)):
        got = run_job(V, gi, q, mode, L, per)
        assert np.array_equal(H.matches_as_ref(idx, got),
                              H.expected("c1", key)), (per, key)


def test_slots_are_reused_and_errors_surface_per_batch(V):
    idx, q = H.load_case("c1")
    gi = gpu_index(V, "c1")
    m = 100
    p = V.Pipeline(gi, 0, 0, m, 500)
    sym = q.symbols.reshape(q.nq, m)
    # three submissions fill the pipeline; the fourth buffer needs a slot
    for b in range(3):
        buf = p.hostbuffer()
        assert buf is not None
        buf[:500 * m] = sym[b * 500:(b + 1) * 500].ravel()
        p.submit(500)
    assert p.hostbuffer() is None
    want = H.oracle_complete(idx, H.Queries.uniform(sym[:1500].ravel(), m))
    got = []
    for b in range(3):
        rc, a = p.next()
        assert rc == 0
        got.append(a)
    assert np.array_equal(np.concatenate(got), want)
    assert p.next()[0] == 1
    # a pipeline whose reads are shorter than prefixlength: the reference's
    # hard error, reported for the batch
    p.close()
    p = V.Pipeline(gi, 0, 0, 5, 10)
    buf = p.hostbuffer()
    buf[:50] = 0
    p.submit(10)
    rc, a = p.next()
    assert rc < 0 and "must be >= " in V.messagespace()
    p.close()
    with pytest.raises(V.VsaError):
        V.Pipeline(gi, 7, 20, 100, 10)


def run_packed_job(V, gi, sym, m, mode, L, per, compact=False):
    """the same job through a packed pipeline: the caller packs its reads
    into the slot's page-locked rows (vsa_pack_reads)"""
    nq = len(sym) // m
    p = V.Pipeline(gi, mode, L, m, per, packed=True)
    out, first = [], 0
    while first < nq:
        n = min(per, nq - first)
        while not p.pack_into_slot(sym[first * m:(first + n) * m], n):
            rc, got = p.next()             # all slots in flight: take one
            assert rc == 0
            out.append(got)
        first += n
    while True:
        rc, got = p.next()
        if rc == 1:
            break
        assert rc == 0
        out.append(got)
    res = np.concatenate(out) if out else np.zeros(0, V.MATCH_DTYPE)
    if mode == 3:
        res, st = p.finish(compact=compact)
        assert st.count == len(res)
        if compact:
            assert res.dtype == V.MATCH16_DTYPE
            res = V.expand_match16(res)
    p.close()
    return res


@pytest.mark.parametrize("per", [10000, 999])
def test_packed_pipeline_gives_the_reference_lists(V, per):
    idx, q = H.load_case("c1")
    gi = gpu_index(V, "c1")
    gi.set_queryspeedup(2)
    m = int(q.length[0])
    for mode, key, L in ((0, "complete", 0), (2, "mumcand20", 20),
                         (3, "mum20", 20), (1, "mem20_sp2", 20)):
        got = run_packed_job(V, gi, q.symbols, m, mode, L, per)
        assert np.array_equal(H.matches_as_ref(idx, got),
                              H.expected("c1", key)), (per, key)
    # the MUM list at 16 bytes per match (vsa_pipeline_finish16), and a job
    # without a MUM
    got = run_packed_job(V, gi, q.symbols, m, 3, 20, per, compact=True)
    assert np.array_equal(H.matches_as_ref(idx, got), H.expected("c1", "mum20"))
    got = run_packed_job(V, gi, np.zeros(64 * m, np.uint8), m, 3, 90, per,
                         compact=True)
    assert len(got) == 0
    # reads with wildcards: the side list of every slot, batch after batch
    sym = q.symbols.copy()
    rng = np.random.default_rng(9)
    hit = rng.integers(0, len(sym), 400)
    sym[hit] = H.WILDCARD
    hq = H.Queries.uniform(sym, m)
    for mode, kw in ((2, dict(mum=True, cand=True)), (3, dict(mum=True)),
                     (1, dict())):
        got = run_packed_job(V, gi, sym, m, mode, 20, 2500)
        want = H.oracle_querymatches(idx, hq, 20, speedup=2, **kw)
        assert np.array_equal(got, want), mode
    assert np.array_equal(run_packed_job(V, gi, sym, m, 0, 0, 2500),
                          H.oracle_complete(idx, hq))
