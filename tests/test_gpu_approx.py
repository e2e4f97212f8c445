"""Approximate complete matches (vmatch -complete -e K | -h K, BASELINE.json
configs[4]) on the GPU against the golden output of the real reference and
against the CPU oracle: bit-exact including the reference's order (merged
regions ascending, start positions descending inside a region).  The distance
of a match travels in the querystart field."""
import numpy as np
import pytest

import helpers as H
from test_gpu_parity import gpu_index, gpu_queries

pytestmark = pytest.mark.gpu
M = H.manifest()

APPROX = [(c, k) for c in sorted(M) for k in sorted(M[c]["runs"])
          if k.startswith("approx_")]


def spec(key):
    """approx_e2 / approx_h2p / approx_e5b -> (edit distance?, K, 0 absolute |
    1 percent | 2 best of)"""
    s = key[len("approx_"):]
    return s[0] == "e", int(s[1:].rstrip("pb")), \
        1 if s.endswith("p") else (2 if s.endswith("b") else 0)


def without_special_queries(q):
    keep = np.array([not (q.symbols[int(s):int(s + l)] >= 254).any()
                     for s, l in zip(q.start, q.length)])
    return keep, H.Queries(q.symbols, q.start[keep], q.length[keep])


@pytest.mark.parametrize("wide", [False, True], ids=["narrow", "wide"])
@pytest.mark.parametrize("case,key", APPROX)
def test_gpu_reproduces_reference_approximate_matches(V, case, key, wide):
    idx, q = H.load_case(case)
    doedist, k, pct = spec(key)
    gi = gpu_index(V, case, wide=wide)   # wide: 64-bit tables on the device
    want = H.expected(case, key)
    # (Hamming distance with wildcards in a read -- bytes are compared, a
    # wildcard equals a wildcard -- and the short-pattern configurations of
    # case c6 take the lcp-interval tree path, approx_tree.inc)
    got = V.findapproxcompletematches(gi, gpu_queries(V, q), doedist, k,
                                      pct).fetch()
    got = H.matches_as_ref(idx, got)
    assert len(got) == len(want)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("doedist,qwild", [(True, True), (True, False),
                                           (False, False)])
@pytest.mark.parametrize("m,k", [(100, 2), (150, 2), (64, 1), (33, 1),
                                 (200, 5), (250, 3), (120, 4), (90, 3),
                                 # beyond 256 symbols: five to eight 64-bit
                                 # words per Myers column (round 3)
                                 (300, 3), (400, 6), (500, 4)])
def test_gpu_equals_oracle_on_repeats_and_separators(V, doedist, qwild, m, k):
    """texts with diverged repeats (many overlapping regions), several
    sequences (separators inside regions) and wildcards; reads with up to
    k + 1 edit operations.  qwild: some reads carry a wildcard, which sends
    an edit distance batch through the region scan (Myers columns) instead
    of the banded alignment of the candidate start positions"""
    rng = np.random.default_rng(1000 * m + 10 * k + int(doedist) + 2 * qwild)
    seqs = []
    for s in range(3):
        L = 20000
        t = rng.integers(0, 4, L).astype(np.uint8)
        unit = rng.integers(0, 4, m + 40).astype(np.uint8)
        for r in range(6):
            p = int(rng.integers(0, L - len(unit)))
            u = unit.copy()
            for e in range(int(rng.integers(0, 4))):
                u[int(rng.integers(0, len(u)))] = rng.integers(0, 4)
            t[p:p + len(u)] = u
        t[rng.random(L) < 0.001] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    qs = []
    for i in range(300):
        s = seqs[int(rng.integers(0, 3))]
        p = int(rng.integers(0, len(s) - m))
        q = s[p:p + m].copy()
        q[q == H.WILDCARD] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, k + 2))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0 or not doedist:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4
            elif kind == 1:
                q = np.delete(q, x)
            else:
                q = np.insert(q, x, rng.integers(0, 4))
        if qwild and i % 50 == 0:
            q[int(rng.integers(0, len(q)))] = H.WILDCARD
        qs.append(q.astype(np.uint8))
    q = H.Queries.from_list(qs)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    want = H.oracle_approx(host, q, doedist, k)
    got = V.findapproxcompletematches(gi, gpu_queries(V, q), doedist,
                                      k).fetch()
    assert len(got) == len(want) and len(want) > 100
    assert np.array_equal(got, want)


def test_threshold_not_below_pattern_length_is_the_reference_error(V):
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    short = H.Queries.from_list([q.symbols[:150], q.symbols[200:203],
                                 q.symbols[400:550]])
    with pytest.raises(V.VsaError) as ei:
        V.findapproxcompletematches(gi, gpu_queries(V, short), True, 3)
    assert "threshold=3>=3=patternlen not allowed" in str(ei.value)
    first = H.Queries.from_list([q.symbols[:150]])
    assert np.array_equal(ei.value.partial.fetch(),
                          H.oracle_approx(idx, first, True, 3))


def test_pieces_with_a_threshold_of_their_own(V):
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    # 20 symbols with 3 errors: three pieces of 6 with one error each
    # (declined in round 1, the lcp-interval tree path since round 2)
    tiny = H.Queries.from_list([q.symbols[:20], q.symbols[160:180]])
    got = V.findapproxcompletematches(gi, gpu_queries(V, tiny), True, 3)
    assert np.array_equal(got.fetch(), H.oracle_approx(idx, tiny, True, 3))
    # a batch that mixes threshold 0 (2 % of 20 symbols) with thresholds > 0
    # (declined until round 3): the exact search for the first read,
    # splitesaapm for the second (approxcompl.c:167-191)
    mixed = H.Queries.from_list([q.symbols[:20], q.symbols[150:300]])
    got = V.findapproxcompletematches(gi, gpu_queries(V, mixed), True, 2, True)
    assert np.array_equal(got.fetch(),
                          H.oracle_approx(idx, mixed, True, 2, True))


@pytest.mark.parametrize("doedist", [True, False], ids=["edit", "hamming"])
@pytest.mark.parametrize("pct", [1, 2, 3])
def test_thresholds_zero_and_above_in_one_batch(V, doedist, pct):
    """-e Kp / -h Kp over reads of 12 ... 260 symbols: the short ones have
    threshold 0 and are exact searches, the others go through the pigeonhole
    or the tree path; the list is the oracle's, read by read"""
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    rng = np.random.default_rng(77 + pct)
    text = idx.tis[:idx.n]
    qs = []
    for i in range(400):
        m = int(rng.integers(12, 261))
        p = int(rng.integers(0, idx.n - m))
        r = text[p:p + m].copy()
        r[r >= 254] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, 3))):
            x = int(rng.integers(0, m))
            r[x] = (r[x] + 1 + rng.integers(0, 3)) % 4
        qs.append(r.astype(np.uint8))
    mixed = H.Queries.from_list(qs)
    ks = mixed.length * pct // 100
    assert (ks == 0).sum() > 20 and (ks > 0).sum() > 20
    want = H.oracle_approx(idx, mixed, doedist, pct, True)
    got = V.findapproxcompletematches(gi, gpu_queries(V, mixed), doedist, pct,
                                      True)
    assert len(want) > 200 and np.array_equal(got.fetch(), want)
    st = got.stats()
    assert st.count == len(want) and st.sumlength == int(want["length"].sum())


def test_mixed_thresholds_stop_at_the_reference_errors(V):
    """the first read shorter than prefixlength (threshold 0: the exact
    search's error, exactcompl.c:179-185) ends the run; the reads before it
    are answered"""
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    reads = [q.symbols[:150], q.symbols[160:180], q.symbols[200:203],
             q.symbols[400:550]]
    mixed = H.Queries.from_list(reads)
    with pytest.raises(H.OracleError) as oe:
        H.oracle_approx(idx, mixed, True, 2, True)
    with pytest.raises(V.VsaError) as ei:
        V.findapproxcompletematches(gi, gpu_queries(V, mixed), True, 2, True)
    assert "patternlength=3 must be >=" in str(ei.value)
    assert str(oe.value) in str(ei.value)
    assert np.array_equal(ei.value.partial.fetch(), oe.value.partial)
    assert len(oe.value.partial) >= 1


def test_threshold_zero_is_the_exact_search(V):
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    got = V.findapproxcompletematches(gi, gpu_queries(V, q), True, 0).fetch()
    assert np.array_equal(got, H.oracle_complete(idx, q))


def test_callback_variant_replays_in_order(V):
    idx, q = H.load_case("c5")
    gi = gpu_index(V, "c5")
    want = V.findapproxcompletematches(gi, gpu_queries(V, q), True, 2).fetch()
    rc, got = V.findapproxcompletematches_cb(gi, gpu_queries(V, q), True, 2)
    assert rc == 0 and np.array_equal(np.array(got, dtype=want.dtype), want)


def test_region_scan_and_banded_alignment_agree(V, monkeypatch):
    """the two ways of finding the start positions (VSA_APM_SCAN=1 forces
    the region scan) on the full c1 batch"""
    idx, q = H.load_case("c1")
    gi = gpu_index(V, "c1")
    a = V.findapproxcompletematches(gi, gpu_queries(V, q), True, 2).fetch()
    monkeypatch.setenv("VSA_APM_SCAN", "1")
    b = V.findapproxcompletematches(gi, gpu_queries(V, q), True, 2).fetch()
    assert len(a) == M["c1"]["runs"]["approx_e2"]["lines"]
    assert np.array_equal(a, b)


def test_many_reads_of_different_lengths_are_planned_per_length(V):
    """beyond 4096 reads of mixed lengths the thresholds and piece geometry
    come from per-length tables filled on the device (apm_plan, bylength);
    the list must be the oracle's"""
    rng = np.random.default_rng(2024)
    n = 150000
    tis = rng.integers(0, 4, n).astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(n, gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    reads = []
    for i in range(6000):
        m = int(rng.integers(90, 151))
        p = int(rng.integers(0, n - m))
        q = tis[p:p + m].copy()
        for e in range(int(rng.integers(0, 4))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4
            elif kind == 1:
                q = np.delete(q, x)
            else:
                q = np.insert(q, x, rng.integers(0, 4))
        reads.append(q.astype(np.uint8))
    q = H.Queries.from_list(reads)
    for doedist, k in ((True, 2), (False, 2), (True, 3)):
        want = H.oracle_approx(host, q, doedist, k)
        got = V.findapproxcompletematches(gi, gpu_queries(V, q), doedist,
                                          k).fetch()
        assert len(want) > 1000
        assert np.array_equal(got, want), (doedist, k)


@pytest.mark.parametrize("doedist", [True, False])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_short_patterns_and_piece_thresholds_equal_the_oracle(V, doedist,
                                                              seed):
    """the configurations that reach esaapm / esahamming in the reference
    (patterns that are not cut, pieces with a threshold of their own):
    repetitive multi-sequence texts with wildcards, reads of 6..33 symbols
    with up to K + 1 edit operations, some keeping a wildcard"""
    rng = np.random.default_rng(100 * seed + int(doedist))
    seqs = []
    for s in range(int(rng.integers(1, 4))):
        n = int(rng.integers(3000, 12000))
        t = rng.integers(0, 4, n).astype(np.uint8)
        unit = rng.integers(0, 4, int(rng.integers(20, 80))).astype(np.uint8)
        for r in range(int(rng.integers(0, 30))):
            p = int(rng.integers(0, n - len(unit)))
            u = unit.copy()
            for e in range(int(rng.integers(0, 3))):
                u[int(rng.integers(0, len(u)))] = rng.integers(0, 4)
            t[p:p + len(u)] = u
        for r in range(int(rng.integers(0, 3))):
            ln = int(rng.integers(10, 60))
            a = int(rng.integers(0, n - ln))
            t[a:a + ln] = np.resize(rng.integers(0, 4, int(rng.integers(1, 4))),
                                    ln)
        t[rng.random(n) < 0.002] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    idx = H.oracle_build_index(tis, 4)
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis, idx.suf,
                             idx.lcp, idx.llv, idx.bck, idx.bwt)
    for k in (1, 2, 3):
        reads = []
        for i in range(60):
            m = int(rng.integers(max(k + 3, 6), 34))
            p = int(rng.integers(0, len(tis) - m))
            q = tis[p:p + m].copy()
            q[q == H.SEPARATOR] = rng.integers(0, 4)
            if rng.random() < 0.8:
                q[q == H.WILDCARD] = rng.integers(0, 4)
            for e in range(int(rng.integers(0, k + 2))):
                kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
                if kind == 0 or not doedist:
                    q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4 if q[x] < 4 \
                        else 0
                elif kind == 1 and len(q) > k + 3:
                    q = np.delete(q, x)
                else:
                    q = np.insert(q, x, rng.integers(0, 4))
            reads.append(q.astype(np.uint8))
        hq = H.Queries.from_list(reads)
        gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
        want = H.oracle_approx(idx, hq, doedist, k)
        got = V.findapproxcompletematches(gi, gq, doedist, k).fetch()
        assert len(want) > 20
        assert np.array_equal(got, want), (doedist, seed, k)


@pytest.mark.parametrize("doedist", [True, False])
@pytest.mark.parametrize("seed", range(3))
def test_best_of_thresholds_equal_the_oracle(V, doedist, seed):
    """vmatch -complete -e Kb | -h Kb (Vmengine/initcompl.c:59-77): every read
    at the smallest threshold <= K percent of its length at which it has a
    match -- reads of several lengths, with 0 .. 6 edit operations, some
    without a match within the bound, some exact, a text with repeats; lists
    in the reference's order (the oracle's restatement of best-of is pinned by
    the golden lists approx_e5b / approx_h5b / approx_e4b / approx_h3b)"""
    rng = np.random.default_rng(31000 + 10 * seed + int(doedist))
    unit = rng.integers(0, 4, 500).astype(np.uint8)
    t = rng.integers(0, 4, 80000).astype(np.uint8)
    for r in range(10):
        p = int(rng.integers(0, len(t) - 500))
        u = unit.copy()
        for e in range(int(rng.integers(0, 4))):
            u[int(rng.integers(0, 500))] = rng.integers(0, 4)
        t[p:p + 500] = u
    t[40000] = H.SEPARATOR
    idx = H.oracle_build_index(t, 4)
    reads = []
    for i in range(400):
        m = int(rng.choice([60, 100, 150, 151]))
        p = int(rng.integers(0, len(t) - m))
        r = t[p:p + m].copy()
        r[r >= 254] = 0
        for e in range(int(rng.integers(0, 7))):
            x = int(rng.integers(0, len(r)))
            kind = int(rng.integers(0, 3)) if doedist else 0
            if kind == 0:
                r[x] = (r[x] + 1 + rng.integers(0, 3)) & 3
            elif kind == 1 and len(r) > 40:
                r = np.delete(r, x)
            else:
                r = np.insert(r, x, rng.integers(0, 4))
        reads.append(r.astype(np.uint8))
    q = H.Queries.from_list(reads)
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis, idx.suf,
                             idx.lcp, idx.llv, idx.bck, idx.bwt)
    gq = V.Queries.from_host(q.symbols, q.start, q.length)
    for k in (5, 3, 1):
        got = V.findapproxcompletematches(gi, gq, doedist, k, 2).fetch()
        want = H.oracle_approx(idx, q, doedist, k, percent=2)
        assert np.array_equal(got, want), (seed, doedist, k)
        # fewer reads answer than with the percent threshold itself, and none
        # of them at a larger distance than its best
        pct = V.findapproxcompletematches(gi, gq, doedist, k, 1).fetch()
        assert set(got["queryseq"]) == set(pct["queryseq"])
