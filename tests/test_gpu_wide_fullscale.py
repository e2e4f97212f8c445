"""An index beyond 2^32 positions (the reference's 64-bit Uint build,
include/types.h:41-61): 4.4 Gbp of synthetic DNA, tables built on the GPU
with 64-bit suf/bck/llv, deep-locate tables in their wide form, searched by
2 M x 100 bp reads.  Checked like the 3 Gbp index of test_gpu_fullscale.py:
  * structure of the tables (suf a permutation, adjacent suffixes in order
    and lcp exact on a sample, bck brackets the q-grams);
  * the CPU oracle on the SAME tables for a sample of the reads: complete /
    MEM / MUM candidates / MUM lists identical, in order;
  * planted answers: every unmodified read is found where it was cut, and
    reads cut from positions >= 2^32 are among them;
  * the reference's MUM filter (CPU) over all GPU candidates == the GPU's
    MUM list; sampled MUMs are maximal exact matches in the text;
  * the scan over the index itself (`vmatch -mum IDX`, the two halves of the
    text as database and query), whole and in 8 ranges, against the oracle.
VSA_WIDE_BP / VSA_WIDE_QUERIES change the size; VSA_WIDE_BP=0 skips the
module (it needs ~230 GB of HBM while the derived tables are made)."""
import os

import numpy as np
import pytest

import helpers as H
import test_gpu_fullscale as F

pytestmark = pytest.mark.gpu

N = int(float(os.environ.get("VSA_WIDE_BP", "4.4e9")))
NQ = int(float(os.environ.get("VSA_WIDE_QUERIES", "2e6")))
M, L = 100, 20


@pytest.fixture(scope="module")
def world(V):
    if N == 0:
        pytest.skip("VSA_WIDE_BP=0")
    import gc
    gc.collect()            # indexes of earlier modules release their HBM
    V.lib.vsa_device_trim(0)
    dg = V.device_malloc(N + 64)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, N, dg, 0))
    pos, sub, step = V.synth_query_plan(N, NQ, M)
    # a separator in the middle (for the scan over the index itself), at a
    # position no read covers
    import ctypes as C
    sep = N // 2
    while ((pos <= sep) & (pos + M > sep)).any():
        sep += 1
    V.device_upload(C.c_void_p(dg.value + sep),
                    np.array([H.SEPARATOR], np.uint8))
    dq = V.device_malloc(NQ * M + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, N, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            NQ, M, dq, 0))
    queries = V.Queries.from_device(dq, NQ, M)
    V.device_free(dq)
    # repeats for the family of the index itself (maximal / supermaximal /
    # tandem repeats), planted behind the reads were cut, in stretches no read
    # touches: a copy of 300 symbols and an array of six 25-symbol units in
    # the upper part of the text (beyond 2^32 for the full-size run)
    spos = np.sort(pos)

    def free_stretch(near, length):
        k = int(np.searchsorted(spos, near))
        while k + 1 < len(spos) and \
                int(spos[k + 1]) - (int(spos[k]) + M) < length + 8:
            k += 1
        return int(spos[k]) + M + 4

    rng = np.random.default_rng(99)
    hi = max(N - N // 40, min(N - 10 ** 6, (1 << 32) + 10 ** 6))
    src, dst = free_stretch(N // 3, 300), free_stretch(hi, 300)
    tan = free_stretch(dst + 10 ** 5, 150)
    assert sep < dst < tan < N - 1000
    piece = np.zeros(300, np.uint8)
    V.device_download(piece, C.c_void_p(dg.value + src))
    V.device_upload(C.c_void_p(dg.value + dst), piece)
    unit = rng.integers(0, 4, 25).astype(np.uint8)
    V.device_upload(C.c_void_p(dg.value + tan), np.tile(unit, 6))
    index = V.Index.build_device(dg, N, 4, 0)
    V.device_free(dg)
    t = index.download()
    info = index.info()
    host = H.Index(N, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                   t["llv"], t["bck"], t["bwt"], None)
    yield dict(index=index, queries=queries, host=host, pos=pos, sub=sub,
               step=step, info=info, sep=sep, planted=(src, dst, tan))
    index.close()
    V.lib.vsa_device_trim(0)


def test_the_tables_are_wide_and_have_the_deep_form(world):
    info = world["info"]
    assert info.totallength == N
    if N + 1 >= 1 << 32:
        assert info.device_integersize == 64
    # (16 = ceil(log4 n), one less where the device lacks room for the slot
    # table and its construction: vsa_index_make_esa8)
    assert info.deepprefix in (15, 16)
    assert info.prefixlength == H.recommended_prefixlength(4, N)
    assert world["host"].suf.dtype == (np.uint64 if N + 1 >= 1 << 32
                                       else np.uint32)


def test_index_structure(world):
    F.test_index_structure(world)
    if N + 1 >= 1 << 32:
        assert int(world["host"].suf.max()) == N     # beyond 32 bits


def test_sample_parity_with_cpu_oracle_on_the_wide_index(V, world):
    # reads from all over the text, the upper end included
    order = np.argsort(world["pos"], kind="stable")
    sel = np.sort(np.concatenate([order[-700:], order[::max(1, NQ // 1300)]
                                  [:1300]]))
    sel = np.unique(sel)
    hq = F.host_queries(world, sel)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    ix, host = world["index"], world["host"]
    got = V.findcompletematches(ix, gq).fetch()
    assert np.array_equal(got, H.oracle_complete(host, hq))
    if N + 1 >= 1 << 32:
        assert (got["dbstart"] >= 1 << 32).sum() >= 300
    for kw in ({}, dict(mum=True, cand=True), dict(mum=True)):
        assert np.array_equal(
            V.findquerymatches(ix, gq, L, speedup=0, **kw).fetch(),
            H.oracle_querymatches(host, hq, L, speedup=0, **kw)), kw


def test_planted_answers_and_global_mum_filter(V, world):
    ix, q = world["index"], world["queries"]
    m = V.findcompletematches(ix, q).fetch()
    exact = world["sub"] == 0xFFFFFFFF
    assert (m["length"] == M).all()
    planted = np.zeros(NQ, bool)
    hit = m["dbstart"] == world["pos"][m["queryseq"]]
    planted[m["queryseq"][hit]] = True
    assert planted[exact].all()
    if N + 1 >= 1 << 32:
        high = world["pos"] >= 1 << 32
        assert (exact & high).sum() > NQ // 200 and planted[exact & high].all()
    assert (np.diff(m["queryseq"].astype(np.int64)) >= 0).all()
    tis = world["host"].tis
    rng = np.random.default_rng(2)
    cand = V.findquerymatches(ix, q, L, mum=True, cand=True).fetch()
    assert (np.diff((cand["queryseq"] * np.uint64(M)
                     + cand["querystart"]).astype(np.int64)) > 0).all()
    mums = V.findquerymatches(ix, q, L, mum=True).fetch()
    import ctypes as C
    out = H.OrcMatches()
    lib = H.oracle_lib()
    lib.orc_matches_init(C.byref(out))
    c2 = np.ascontiguousarray(cand.copy())
    lib.orc_mumuniqueinquery(c2.ctypes.data, len(c2), C.byref(out))
    assert np.array_equal(mums, H._take(out))
    assert (np.diff(mums["dbstart"].astype(np.int64)) >= 0).all()
    if N + 1 >= 1 << 32:
        assert (mums["dbstart"] >= 1 << 32).sum() > len(mums) // 200
    for k in rng.integers(0, len(mums), size=2000):
        ln, s, qi, qo = (int(mums[f][k]) for f in
                         ("length", "dbstart", "queryseq", "querystart"))
        hq = F.host_queries(world, [qi]).symbols
        assert np.array_equal(tis[s:s + ln], hq[qo:qo + ln])
        assert qo + ln == M or s + ln == N or tis[s + ln] != hq[qo + ln]
        assert qo == 0 or s == 0 or tis[s - 1] != hq[qo - 1]


def test_repeats_of_the_index_itself_beyond_2_32(V, world):
    """vmatch -l / -supermax / -tandem on the index itself with suffix array
    positions and text positions beyond 2^32 (the 64-bit position
    instantiation of selfmatch_search.inc): the planted copy and the planted
    array come back, in the reference's order, as the oracle finds them on
    the same tables (include/vdfstrav.c:247, Vmengine/fsuper.c:142,
    ftandem.c:261 know no 32-bit limit either)"""
    ix, host = world["index"], world["host"]
    src, dst, tan = world["planted"]
    got = V.findmaximalrepeats(ix, 40).fetch()
    assert np.array_equal(got, H.oracle_repeats(host, 40))
    assert any(int(r["dbstart"]) == src and int(r["queryseq"]) == dst and
               int(r["length"]) >= 300 for r in got)
    got = V.findsupermaximalrepeats(ix, 40).fetch()
    assert np.array_equal(got, H.oracle_supermax(host, 40))
    assert len(got) >= 1
    got = V.findtandems(ix, 20).fetch()
    assert np.array_equal(got, H.oracle_tandems(host, 20))
    # (u^6: the right branching repeats with units of 75, 50 and 25 symbols)
    assert len(got) >= 3 and (got["dbstart"] >= tan).all() and \
        (got["dbstart"] < tan + 150).all()
    if N + 1 >= 1 << 32:
        assert dst >= 1 << 32 and tan >= 1 << 32


def test_approximate_matches_beyond_2_32(V, world):
    """vmatch -complete -e 2 / -h 2 with text positions beyond 2^32 (the
    pigeonhole path with its positions and sort keys in the width of the
    tables): 2 000 reads, most of them cut from the upper end of the text,
    against the oracle on the same tables (Vmengine/splitesaapm.c:458-558
    knows no 32-bit limit either)"""
    order = np.argsort(world["pos"], kind="stable")
    sel = np.unique(np.concatenate([order[-1500:],
                                    order[::max(1, NQ // 500)][:500]]))
    hq = F.host_queries(world, sel)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    ix, host = world["index"], world["host"]
    for doedist in (True, False):
        got = V.findapproxcompletematches(ix, gq, doedist, 2).fetch()
        want = H.oracle_approx(host, hq, doedist, 2)
        assert len(want) >= len(sel) // 2
        assert np.array_equal(got, want), doedist
        if N + 1 >= 1 << 32:
            assert (got["dbstart"] >= 1 << 32).sum() >= 700
    # reads of different lengths, thresholds in percent (0 for the short
    # ones: the exact search; the others through the pigeonhole path)
    rng = np.random.default_rng(5)
    cut = [hq.symbols[int(s):int(s) + int(rng.integers(30, 101))]
           for s in hq.start[:600]]
    rq = H.Queries.from_list(cut)
    got = V.findapproxcompletematches(
        ix, V.Queries.from_host(rq.symbols, rq.start, rq.length), True, 2,
        True).fetch()
    assert np.array_equal(got, H.oracle_approx(host, rq, True, 2, True))
    # the lcp-interval tree path (approx_tree.inc) with suffix array
    # intervals beyond 2^32: patterns that are not cut (28 symbols, one
    # error: esaapm / esahamming on the whole pattern).  The oracle walks the
    # whole suffix array per pattern as the reference does -- hours at this
    # size -- so the lists are checked through what they must hold: every
    # occurrence with at most one mismatch contains one half of the pattern
    # exactly (found by the oracle's exact search), every reported match is
    # verified in the text, and every Hamming occurrence is an edit distance
    # start.  (The order of these lists is pinned by the golden case c6, on
    # 32- and 64-bit tables.)
    m, k = 28, 1
    top = hq.start[-120:]
    pats = [hq.symbols[int(s):int(s) + m] for s in top]
    tq = H.Queries.from_list(pats)
    gt = V.Queries.from_host(tq.symbols, tq.start, tq.length)
    halves = H.Queries.from_list([p[o:o + m // 2] for p in pats
                                  for o in (0, m // 2)])
    seeds = H.oracle_complete(host, halves)
    tis = host.tis
    expect = set()
    for row in seeds:
        q, o = int(row["queryseq"]) // 2, (int(row["queryseq"]) % 2) * (m // 2)
        s0 = int(row["dbstart"]) - o
        if s0 < 0 or s0 + m > N:
            continue
        w = tis[s0:s0 + m]
        mm = int((w != pats[q]).sum())
        if mm <= k and not (w == H.SEPARATOR).any():
            expect.add((m, s0, q, mm))
    got = V.findapproxcompletematches(ix, gt, False, k).fetch()
    assert set(tuple(int(x) for x in r) for r in got) == expect
    assert len(got) == len(expect) >= len(pats) // 2
    assert (np.diff(got["queryseq"].astype(np.int64)) >= 0).all()
    if N + 1 >= 1 << 32:
        assert (got["dbstart"] >= 1 << 32).sum() >= len(pats) // 2
    edit = V.findapproxcompletematches(ix, gt, True, k).fetch()
    starts = set((int(r["dbstart"]), int(r["queryseq"])) for r in edit)
    assert all((s0, q) in starts for (_, s0, q, _) in expect)
    for r in edit[:: max(1, len(edit) // 400)]:
        ln, s0, q, dist = (int(r[f]) for f in
                           ("length", "dbstart", "queryseq", "querystart"))
        pat, best = pats[q], None
        prev = list(range(m + 1))        # column of the empty text prefix
        for L in range(1, min(m + k, N - s0) + 1):
            c = tis[s0 + L - 1]
            if c == H.SEPARATOR:
                break
            cur = [L] + [0] * m
            for i in range(1, m + 1):
                cur[i] = min(prev[i - 1] + (0 if pat[i - 1] == c and c < 254
                                            else 1), prev[i] + 1,
                             cur[i - 1] + 1)
            if best is None or cur[m] <= best[1]:
                best = (L, cur[m])     # the longest among the best
            prev = cur
        assert best == (ln, dist) and dist <= k, (r, best)


def test_self_index_scan_on_the_wide_index(V, world):
    """vmatch -mum on an index that holds two "genomes" (the first and the
    second half of the text, vsa_index_set_queryseparator): the streaming
    scan over suf/lcp/bwt with positions beyond 2^32, whole and by ranges,
    against the oracle on the same tables"""
    from vstree_amd import sharding as S
    ix, host = world["index"], world["host"]
    sep = world["sep"]
    ix.set_queryseparator(sep)
    host.querysepposition, host.hasqueries = sep, True
    L = 22
    whole = V.findmaximaluniquematches(ix, L).fetch()
    want = H.oracle_selfmum(host, L)
    assert len(want) > 1000
    assert np.array_equal(whole, want)
    if N + 1 >= 1 << 32:
        assert (whole["queryseq"] >= 1 << 32).sum() > len(whole) // 50
    parts = [V.findmaximaluniquematches(ix, L, *S.selfmum_range(N, r, 8))
             .fetch() for r in range(8)]
    assert np.array_equal(np.concatenate(parts), whole)
