"""Reads at two bits per symbol (vsa_pack_reads, vsa_queries_from_host_packed):
a packed batch must answer every engine call exactly like the byte batch of
the same reads -- -complete, -mum, -mum cand straight from the rows (first
pass, work plan, search kernel), MEM / approximate matching / the reverse
complement through the bytes made on the device -- including reads with
wildcards (side list), reads that end in a repeat (ties: the reference walk
through QSrc), 32- and 64-bit device tables.  Symbol map: the reference's DNA
map (kurtz-basic/alphabet.c:369); wildcard semantics kurtz/maxpref.c:30-41."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def unpack(V, rows, special, nq, m):
    W = int(V.lib.vsa_packed_words(m))
    rows = rows.reshape(nq, W)
    out = np.zeros((nq, m), np.uint8)
    for j in range(m):
        out[:, j] = (rows[:, j // 32] >> np.uint64(62 - 2 * (j % 32))) & \
            np.uint64(3)
    flagged = np.flatnonzero(rows[:, W - 1] & np.uint64(0xFF))
    for i in flagged:
        k = int(rows[i, 0]) >> 8
        out[i] = special[k * m:(k + 1) * m]
    return out.ravel(), flagged


def gpu_index(V, idx, bits=32):
    i = idx.as_width(bits)
    return V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis,
                               i.suf, i.lcp, i.llv, i.bck, i.bwt,
                               i.querysepposition, i.hasqueries)


def both(V, sym, m):
    nq = len(sym) // m
    byte = V.Queries.from_host(sym, np.arange(nq, dtype=np.uint64) * m,
                               np.full(nq, m, np.uint64))
    packed = V.Queries.from_host_packed(sym, m)
    return byte, packed


@pytest.mark.parametrize("bits", [32, 64])
def test_packed_batch_reproduces_the_golden_lists(V, bits):
    idx, q = H.load_case("c1")
    m = int(q.length[0])
    assert (q.length == m).all() and (q.start == np.arange(q.nq) * m).all()
    gi = gpu_index(V, idx, bits)
    byte, packed = both(V, q.symbols, m)
    assert packed.info().numofqueries == q.nq
    got = H.matches_as_ref(idx, V.findcompletematches(gi, packed).fetch())
    assert np.array_equal(got, H.expected("c1", "complete"))
    for key, kw in (("mum20", dict(mum=True)),
                    ("mumcand20", dict(mum=True, cand=True)),
                    ("mem20_sp2", dict())):
        r = V.findquerymatches(gi, packed, 20, **kw)
        got = H.matches_as_ref(idx, r.fetch())
        assert np.array_equal(got, H.expected("c1", key)), key
        rb = V.findquerymatches(gi, byte, 20, **kw)
        sb, sp = rb.stats(), r.stats()
        assert (sb.count, sb.sumlength, sb.candidates) == \
            (sp.count, sp.sumlength, sp.candidates), key
        if kw:      # (MEM counts the plan's own locates by an upper bound)
            assert sb.searches == sp.searches, key
    # approximate matching and the reverse complement go through the bytes
    for key in sorted(H.manifest()["c1"]["runs"]):
        if key.startswith("approx_"):
            spec = key[len("approx_"):]
            doedist, k, pct = spec[0] == "e", int(spec[1:].rstrip("pb")), \
                1 if spec.endswith("p") else (2 if spec.endswith("b") else 0)
            got = V.findapproxcompletematches(gi, packed, doedist, k,
                                              pct).fetch()
            assert np.array_equal(H.matches_as_ref(idx, got),
                                  H.expected("c1", key)), key
    rc = packed.reverse_complement()
    rcb = byte.reverse_complement()
    assert np.array_equal(V.findquerymatches(gi, rc, 20).fetch(),
                          V.findquerymatches(gi, rcb, 20).fetch())


@pytest.mark.parametrize("seed", range(8))
def test_wildcards_repeats_and_short_reads_packed(V, seed):
    """reads with wildcards travel on the side list; reads cut from repeats
    tie on all key symbols (reference walk through QSrc); lengths that are no
    multiple of 4 or 32; a text with wildcards and separators; rows of more
    than four words (150 bp and longer: the first pass looks at them through
    windows) and of more than eight (expanded to bytes on the device)"""
    rng = np.random.default_rng(4200 + seed)
    m = [100, 37, 64, 121, 150, 200, 252, 253][seed]
    L = [20, 12, 16, 25, 20, 30, 18, 22][seed]
    unit = rng.integers(0, 4, 300).astype(np.uint8)
    t = rng.integers(0, 4, 60000).astype(np.uint8)
    for r in range(12):
        p = int(rng.integers(0, len(t) - 300))
        u = unit.copy()
        for e in range(int(rng.integers(0, 3))):
            u[int(rng.integers(0, 300))] = rng.integers(0, 4)
        t[p:p + 300] = u
    t[rng.random(len(t)) < 0.0004] = H.WILDCARD
    t[30000] = H.SEPARATOR
    idx = H.oracle_build_index(t, 4)
    idx.sti1 = H.sti1_from_tables(idx.suf, idx.lcp, idx.prefixlength)
    nq = 3000
    reads = np.zeros((nq, m), np.uint8)
    for i in range(nq):
        p = int(rng.integers(0, len(t) - m))
        reads[i] = t[p:p + m]
        if rng.random() < 0.3:
            reads[i, int(rng.integers(0, m))] = rng.integers(0, 4)
        if rng.random() < 0.1:
            reads[i, int(rng.integers(0, m))] = H.WILDCARD
    reads[reads == H.SEPARATOR] = H.WILDCARD
    sym = reads.ravel()
    rows, special, ns = V.pack_reads(sym, nq, m)
    back, flagged = unpack(V, rows, special, nq, m)
    assert np.array_equal(back, sym) and ns == len(flagged) > 0
    hq = H.Queries.uniform(sym, m)
    for bits in (32, 64):
        gi = gpu_index(V, idx, bits)
        byte, packed = both(V, sym, m)
        assert np.array_equal(V.findcompletematches(gi, packed).fetch(),
                              H.oracle_complete(idx, hq))
        for kw in (dict(mum=True), dict(mum=True, cand=True), dict()):
            got = V.findquerymatches(gi, packed, L, **kw).fetch()
            want = H.oracle_querymatches(idx, hq, L, speedup=2, **kw)
            assert np.array_equal(got, want), (seed, bits, kw)


def test_packed_reads_of_a_multiseq_with_separators(V):
    """stride m + 1: the reads of a reference Multiseq, separators between
    them (kurtz-basic/multiseq.c:129-166), packed where they lie"""
    idx, q = H.load_case("c1")
    m = int(q.length[0])
    nq = 500
    ms = np.full(nq * (m + 1), H.SEPARATOR, np.uint8)
    ms.reshape(nq, m + 1)[:, :m] = q.symbols[:nq * m].reshape(nq, m)
    gi = gpu_index(V, idx)
    packed = V.Queries.from_host_packed(ms[:-1], m, stride=m + 1)
    assert packed.nq == nq
    want = V.findquerymatches(
        gi, V.Queries.from_host(q.symbols[:nq * m], q.start[:nq],
                                q.length[:nq]), 20, mum=True).fetch()
    assert np.array_equal(V.findquerymatches(gi, packed, 20,
                                             mum=True).fetch(), want)
    # a flagged row that names no entry of the side list is refused
    rows, special, ns = V.pack_reads(q.symbols[:nq * m], nq, m)
    W = int(V.lib.vsa_packed_words(m))
    rows[5 * W + W - 1] |= np.uint64(1)
    rows[5 * W] = np.uint64(7)
    h = H.C.c_void_p()
    rc = V.lib.vsa_queries_from_host_packed(V._ptr(rows), nq, m,
                                            V._ptr(special), ns, 0,
                                            H.C.byref(h))
    assert rc == -2 and "side list" in V.messagespace()
