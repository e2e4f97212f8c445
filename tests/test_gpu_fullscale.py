"""Parity at BASELINE.json's full size: a 3 Gbp synthetic index built on the
GPU and 10 M x 100 bp queries.  Nothing of that size can be compared list by
list with the CPU, so the test combines
  * the CPU oracle on the SAME 3 Gbp tables (downloaded from the GPU) for a
    sample of the queries: complete / MEM / candidates must agree exactly;
  * the planted answers of the generator: every unmodified query must be
    found at the position it was cut from;
  * the reference's MUM filter (CPU oracle) over ALL 11.6 M GPU candidates
    == the GPU's MUM list;
  * structural properties of the index: suf is a permutation, adjacent
    suffixes are in order and lcp is exact on a large random sample, bck
    brackets the q-grams.
Set VSA_FULLSCALE_BP to run at another size (default 3e9; the driver's GPU box
has 288 GB of HBM and 3 TB of host memory)."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N = int(float(os.environ.get("VSA_FULLSCALE_BP", "3e9")))
NQ = int(float(os.environ.get("VSA_FULLSCALE_QUERIES", "1e7")))
M, L = 100, 20


@pytest.fixture(scope="module")
def world(V):
    dg = V.device_malloc(N + 64)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, N, dg, 0))
    index = V.Index.build_device(dg, N, 4, 0)
    pos, sub, step = V.synth_query_plan(N, NQ, M)
    dq = V.device_malloc(NQ * M + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, N, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            NQ, M, dq, 0))
    queries = V.Queries.from_device(dq, NQ, M)
    V.device_free(dq)
    V.device_free(dg)
    t = index.download()
    info = index.info()
    host = H.Index(N, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                   t["llv"], t["bck"], t["bwt"], None)
    return dict(index=index, queries=queries, host=host, pos=pos, sub=sub,
                step=step, info=info)


def host_queries(w, sel):
    g = w["host"].tis
    out = np.zeros((len(sel), M), np.uint8)
    for k, i in enumerate(sel):
        p = int(w["pos"][i])
        out[k] = g[p:p + M]
        if w["sub"][i] != 0xFFFFFFFF:
            out[k, w["sub"][i]] = (out[k, w["sub"][i]] + w["step"][i]) & 3
    return H.Queries.uniform(out.ravel(), M)


def test_generator_on_device_equals_host_generator(V, world):
    g = world["host"].tis
    assert np.array_equal(g[:100000], V.synth_genome(100000))
    for i in (N - 1, N // 2, 12345678):
        assert g[i] == (V.lib.vsa_splitmix64_at(V.GENOME_SEED, i) >> 62)
    assert world["info"].prefixlength == H.recommended_prefixlength(4, N)


def test_index_structure(world):
    h = world["host"]
    suf, lcp, tis, n = h.suf, h.lcp, h.tis, h.n
    assert int(suf[n]) == n
    # permutation: sum and sum of squares (mod 2^64) of 0..n
    idx = np.arange(n + 1, dtype=np.uint64)
    assert int(suf.astype(np.uint64).sum(dtype=np.uint64)) == int(
        idx.sum(dtype=np.uint64))
    assert int((suf.astype(np.uint64) ** 2).sum(dtype=np.uint64)) == int(
        (idx ** 2).sum(dtype=np.uint64))
    del idx
    # adjacent pairs: order and exact lcp on a sample
    rng = np.random.default_rng(1)
    for j in rng.integers(1, n, size=20000):
        a, b, k = int(suf[j - 1]), int(suf[j]), 0
        while a + k < n and b + k < n and tis[a + k] == tis[b + k]:
            k += 1
        assert int(lcp[j]) == min(k, 255), j
        # suf[j-1] < suf[j]: the smaller suffix still has a (smaller)
        # regular symbol where they part; the end sentinel is the largest
        assert a + k < n and (b + k >= n or tis[a + k] < tis[b + k]), j
    # bck brackets the q-grams of the sampled suffixes
    pl = h.prefixlength
    w4 = 4 ** np.arange(pl - 1, -1, -1, dtype=np.uint64)
    for j in rng.integers(0, n - 5, size=20000):
        s = int(suf[j])
        if s + pl > n:
            continue
        code = int((tis[s:s + pl].astype(np.uint64) * w4).sum())
        assert int(h.bck[2 * code]) <= j < int(h.bck[2 * code + 1])


def test_sample_parity_with_cpu_oracle_on_full_index(V, world):
    sel = np.arange(0, NQ, max(1, NQ // 3000))[:3000]
    hq = host_queries(world, sel)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    ix, host = world["index"], world["host"]
    assert np.array_equal(V.findcompletematches(ix, gq).fetch(),
                          H.oracle_complete(host, hq))
    if host.sti1 is None:
        # the reference's default algorithm 2 reads stitab1
        host.sti1 = H.sti1_from_tables(host.suf, host.lcp, host.prefixlength)
    for kw, sp in (({}, 0), ({}, 2), (dict(mum=True, cand=True), 2),
                   (dict(mum=True), 2)):
        assert np.array_equal(
            V.findquerymatches(ix, gq, L, speedup=sp, **kw).fetch(),
            H.oracle_querymatches(host, hq, L, speedup=sp, **kw)), (kw, sp)


def test_planted_answers_and_global_mum_filter(V, world):
    ix, q = world["index"], world["queries"]
    r = V.findcompletematches(ix, q)
    m = r.fetch()
    exact = world["sub"] == 0xFFFFFFFF
    # every match is a real occurrence: length m, and for unmodified queries
    # the planted position is among the hits
    assert (m["length"] == M).all()
    planted = np.zeros(NQ, bool)
    hit = m["dbstart"] == world["pos"][m["queryseq"]]
    planted[m["queryseq"][hit]] = True
    assert planted[exact].all()
    # queries in file order, suffix-array order inside (positions of one
    # query are distinct)
    assert (np.diff(m["queryseq"].astype(np.int64)) >= 0).all()
    tis = world["host"].tis
    rng = np.random.default_rng(2)
    for k in rng.integers(0, len(m), size=2000):
        s, qi = int(m["dbstart"][k]), int(m["queryseq"][k])
        hq = host_queries(world, [qi])
        assert np.array_equal(tis[s:s + M], hq.symbols)
    # MUM: GPU filter == the reference's filter (CPU) over all candidates
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
    want = H._take(out)
    assert np.array_equal(mums, want)
    assert (np.diff(mums["dbstart"].astype(np.int64)) >= 0).all()
    # each MUM is a maximal exact match (sample)
    for k in rng.integers(0, len(mums), size=2000):
        ln, s, qi, qo = (int(mums[f][k]) for f in
                         ("length", "dbstart", "queryseq", "querystart"))
        hq = host_queries(world, [qi]).symbols
        assert np.array_equal(tis[s:s + ln], hq[qo:qo + ln])
        assert qo + ln == M or s + ln == N or tis[s + ln] != hq[qo + ln]
        assert qo == 0 or s == 0 or tis[s - 1] != hq[qo - 1]


# ---------------------------------------------------------------------------
# BASELINE.json configs[1]: 200 Mbp index, 1 M x 100 bp, -complete -- small
# enough for the CPU oracle to answer EVERY query: the whole list, in order.
# ---------------------------------------------------------------------------

N2 = int(float(os.environ.get("VSA_CONFIG1_BP", "2e8")))
NQ2 = int(float(os.environ.get("VSA_CONFIG1_QUERIES", "1e6")))


def test_config1_complete_matches_of_all_queries_equal_the_oracle(V):
    dg = V.device_malloc(N2 + 64)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, N2, dg, 0))
    index = V.Index.build_device(dg, N2, 4, 0)
    pos, sub, step = V.synth_query_plan(N2, NQ2, M)
    dq = V.device_malloc(NQ2 * M + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, N2, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            NQ2, M, dq, 0))
    queries = V.Queries.from_device(dq, NQ2, M)
    V.device_free(dq)
    V.device_free(dg)
    t = index.download()
    info = index.info()
    assert info.prefixlength == H.recommended_prefixlength(4, N2) == 11
    host = H.Index(N2, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                   t["llv"], t["bck"], t["bwt"], None)
    hq = host_queries(dict(host=host, pos=pos, sub=sub, step=step),
                      np.arange(NQ2))
    got = V.findcompletematches(index, queries).fetch()
    want = H.oracle_complete(host, hq)
    assert len(want) >= NQ2 * 7 // 10          # ~75 % of the reads are exact
    assert np.array_equal(got, want)
    # and the query path on the same index, against the oracle on a sample
    sel = np.arange(0, NQ2, max(1, NQ2 // 20000))[:20000]
    hs = host_queries(dict(host=host, pos=pos, sub=sub, step=step), sel)
    gs = V.Queries.from_host(hs.symbols, hs.start, hs.length)
    host.sti1 = H.sti1_from_tables(host.suf, host.lcp, host.prefixlength)
    for kw, sp in (({}, 2), (dict(mum=True), 2)):
        assert np.array_equal(
            V.findquerymatches(index, gs, L, speedup=sp, **kw).fetch(),
            H.oracle_querymatches(host, hs, L, speedup=sp, **kw)), kw


# ---------------------------------------------------------------------------
# BASELINE.json configs[4]: 3 Gbp index, 10 M x 150 bp, -complete -e 2.
#   * CPU oracle (restatement of approxcompl.c / splitesaapm.c, pinned against
#     the reference) on a sample of >= 3 000 reads: identical lists, in order;
#   * the planted answers of the generator: every read must be reported at the
#     position it was cut from, over its whole length, at the distance the
#     generator gave it (0 or 1 substitution);
#   * every reported match is within 2 errors: unit-cost edit distance of the
#     read and the reported piece of the text, recomputed with numpy for a
#     large random sample of the ~45 M matches, must equal the reported one.
# ---------------------------------------------------------------------------

M5, K5 = 150, 2
NQ5 = int(float(os.environ.get("VSA_CONFIG4_QUERIES", os.environ.get(
    "VSA_FULLSCALE_QUERIES", "1e7"))))


def edit_distances(a, b, la, lb):
    """unit-cost edit distance of a[i,:la[i]] and b[i,:lb[i]] for all rows
    (plain dynamic programming, one numpy row operation per cell column)"""
    rows, wa = a.shape
    wb = b.shape[1]
    inf = np.int32(1 << 20)
    prev = np.tile(np.arange(wb + 1, dtype=np.int32), (rows, 1))
    out = prev[np.arange(rows), lb].copy()
    out[la != 0] = inf
    for i in range(1, wa + 1):
        cur = np.empty_like(prev)
        cur[:, 0] = i
        sub_ = prev[:, :-1] + (a[:, i - 1:i] != b)
        dele = prev[:, 1:] + 1
        best = np.minimum(sub_, dele)
        # insertions: running minimum along the row
        cur[:, 1:] = best
        for j in range(1, wb + 1):
            np.minimum(cur[:, j], cur[:, j - 1] + 1, out=cur[:, j])
        done = la == i
        out[done] = cur[done, lb[done]]
        prev = cur
    return out


def test_config4_approximate_matches_at_full_size(V, world):
    ix, host = world["index"], world["host"]
    dg = V.device_malloc(N + 64)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, N, dg, 0))
    pos, sub, step = V.synth_query_plan(N, NQ5, M5)
    dq = V.device_malloc(NQ5 * M5 + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, N, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            NQ5, M5, dq, 0))
    queries = V.Queries.from_device(dq, NQ5, M5)
    V.device_free(dq)
    V.device_free(dg)
    res = V.findapproxcompletematches(ix, queries, True, K5)
    m = res.fetch()
    res.close()
    g = host.tis
    dist = m["querystart"]                  # the distance travels here
    qn = m["queryseq"].astype(np.int64)
    assert (dist <= K5).all() and len(m) >= NQ5
    assert (np.diff(qn) >= 0).all()         # reads in order

    def reads(sel):
        out = g[pos[sel, None].astype(np.int64) + np.arange(M5)[None, :]]
        hit = np.flatnonzero(sub[sel] != 0xFFFFFFFF)
        c = sub[sel][hit]
        out[hit, c] = (out[hit, c] + step[sel][hit]) & 3
        return out

    # 1. the oracle on a sample, lists in order
    sel = np.arange(0, NQ5, max(1, NQ5 // 3000))[:3000]
    hq = H.Queries.uniform(reads(sel).ravel(), M5)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    got = V.findapproxcompletematches(ix, gq, True, K5).fetch()
    want = H.oracle_approx(host, hq, True, K5)
    assert len(want) >= 3000
    assert np.array_equal(got, want)
    # ... and the same reads inside the big batch gave the same matches
    inbig = m[np.isin(qn, sel)].copy()
    inbig["queryseq"] = np.searchsorted(sel, inbig["queryseq"])
    assert np.array_equal(inbig, want)
    # 2. planted answers: every read is reported from the position it was cut
    # at, with the generator's distance (0, or 1 substitution) -- over its
    # whole length if exact; with one substitution the longest match of that
    # distance may be a symbol shorter or longer (a substituted last symbol
    # can also be read as a deletion or an insertion, longestmatch.c:18-71)
    wantdist = (sub != 0xFFFFFFFF).astype(np.uint64)
    planted = ((m["dbstart"] == pos[qn]) & (dist == wantdist[qn]) &
               (np.abs(m["length"].astype(np.int64) - M5) <= wantdist[qn]))
    seen = np.zeros(NQ5, bool)
    seen[qn[planted]] = True
    assert seen.all()
    # from one start position only one match is reported
    key = qn * (N + 1) + m["dbstart"].astype(np.int64)
    assert len(np.unique(key)) == len(key)
    # 3. reported distance == recomputed edit distance, random sample
    rng = np.random.default_rng(11)
    pick = rng.integers(0, len(m), size=200000)
    a = reads(qn[pick])
    ln = m["length"][pick].astype(np.int64)
    assert (ln >= M5 - K5).all() and (ln <= M5 + K5).all()
    b = g[np.minimum(m["dbstart"][pick, None].astype(np.int64) +
                     np.arange(M5 + K5)[None, :], N - 1)]
    d = edit_distances(a, b, np.full(len(pick), M5), ln)
    assert np.array_equal(d.astype(np.uint64), dist[pick])
