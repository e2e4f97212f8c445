"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
golden outputs of the real reference.  Bit-exact, order included."""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
M = H.manifest()

_gpu_index = {}


def gpu_index(V, case, bits=64, wide=False):
    """wide: 64-bit tables ON THE DEVICE whatever the length of the text (what
    an index of 2^32 symbols and more gets: the uint64_t instantiation of
    every kernel, the wide form of the deep slots).  The library reads
    VSA_FORCE_WIDE when an index is created, so the switch is set around the
    upload only."""
    key = (case, bits, wide)
    if key not in _gpu_index:
        idx, _ = H.load_case(case)
        i = idx.as_width(bits)
        before = os.environ.get("VSA_FORCE_WIDE")
        if wide:
            os.environ["VSA_FORCE_WIDE"] = "1"
        try:
            _gpu_index[key] = V.Index.from_tables(
                i.n, i.prefixlength, i.numofchars, i.tis, i.suf, i.lcp, i.llv,
                i.bck, i.bwt, i.querysepposition, i.hasqueries)
        finally:
            if wide:
                if before is None:
                    del os.environ["VSA_FORCE_WIDE"]
                else:
                    os.environ["VSA_FORCE_WIDE"] = before
        if wide:
            assert _gpu_index[key].info().device_integersize == 64
    return _gpu_index[key]


def gpu_queries(V, q):
    return V.Queries.from_host(q.symbols, q.start, q.length)


def run_gpu(V, case, key, bits=64, wide=False):
    idx, q = H.load_case(case)
    gi = gpu_index(V, case, bits, wide)
    if key.startswith("selfmum"):
        r = V.findmaximaluniquematches(gi, int(key[len("selfmum"):]))
        return H.selfmatches_as_ref(idx, r.fetch())
    if key.startswith("supermax"):
        r = V.findsupermaximalrepeats(gi, int(key[len("supermax"):]))
        return H.repeats_as_ref(idx, r.fetch())
    if key.startswith("tandem"):
        r = V.findtandems(gi, int(key[len("tandem"):]))
        return H.repeats_as_ref(idx, r.fetch())
    if key.startswith("palindromic"):
        rq = H.index_as_rc_queries(idx)
        L = int(key[len("palindromic"):].partition("_sp")[0])
        return H.palindromic_as_ref(
            idx, V.findquerymatches(gi, gpu_queries(V, rq), L).fetch())
    if key.startswith("repeats"):
        r = V.findmaximalrepeats(gi, int(key[len("repeats"):]))
        conv = H.selfmatches_as_ref if idx.hasqueries else H.repeats_as_ref
        return conv(idx, r.fetch())
    gq = gpu_queries(V, q)
    if key.startswith("complete"):
        return H.matches_as_ref(idx, V.findcompletematches(gi, gq).fetch())
    name = key.partition("_sp")[0]
    if name.startswith("mumcand"):
        L, kw = int(name[7:]), dict(mum=True, cand=True)
    elif name.startswith("mum"):
        L, kw = int(name[3:]), dict(mum=True)
    else:
        L, kw = int(name[3:].split("_")[0]), {}
    return H.matches_as_ref(idx, V.findquerymatches(gi, gq, L, **kw).fetch())


# MEM order inside one query offset depends on the reference's algorithm
# (vmatch -qspeedup 0 | 2): the GPU reproduces both, every list is compared in
# order
# (mode, -qspeedup level): MEM lists under both algorithms, the lists that do
# not depend on the algorithm once
MODES = [({}, 0), ({}, 2), (dict(mum=True, cand=True), 2), (dict(mum=True), 2)]

CASES = [(c, k) for c in sorted(M) for k in sorted(M[c]["runs"])
         if not k.endswith("_short") and "strands" not in M[c]["runs"][k]
         and not k.startswith("approx_")]   # those: tests/test_gpu_approx.py


@pytest.mark.parametrize("wide", [False, True], ids=["narrow", "wide"])
@pytest.mark.parametrize("case,key", CASES)
def test_gpu_reproduces_reference_output(V, case, key, wide):
    # lists recorded with -qspeedup 0 carry _sp0 in their name, all others come
    # from the reference's default algorithm 2: both are reproduced in order
    gpu_index(V, case, 64, wide).set_queryspeedup(
        0 if key.endswith("_sp0") else 2)
    got = run_gpu(V, case, key, wide=wide)
    want = H.expected(case, key)
    assert len(got) == len(want)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("case", ["micro", "grumbach", "largepat"])
def test_gpu_equals_oracle_32bit_tables(V, case):
    idx, q = H.load_case(case)
    gi = gpu_index(V, case, 32)
    gq = gpu_queries(V, q)
    pl = idx.prefixlength
    assert np.array_equal(V.findcompletematches(gi, gq).fetch()
                          if q.length.min() >= pl else np.zeros(0),
                          H.oracle_complete(idx, q)
                          if q.length.min() >= pl else np.zeros(0))
    L = max(pl, 8 if case != "micro" else 2)
    for kw, sp in MODES:
        a = V.findquerymatches(gi, gq, L, speedup=sp, **kw).fetch()
        b = H.oracle_querymatches(idx, q, L, speedup=sp, **kw)
        assert np.array_equal(a, b), (case, kw, sp)


def test_largepat_known_answer_file_gpu(V):
    idx, q = H.load_case("largepat")
    got = run_gpu(V, "largepat", "complete")
    with open(os.path.join(H.GOLDEN, "LargePat.res")) as f:
        want = H.parse_vmatch_lines([l for l in f.read().splitlines() if l])
    assert np.array_equal(got, want)


def test_short_query_is_the_references_hard_error(V):
    idx, _ = H.load_case("grumbach")
    q = H.fasta_queries(os.path.join(H.GOLDEN, "short.fna"))
    gi, gq = gpu_index(V, "grumbach"), gpu_queries(V, q)
    with pytest.raises(V.VsaError) as e:
        V.findcompletematches(gi, gq)
    assert e.value.message == "patternlength=5 must be >= 6=prefixlen"
    got = H.matches_as_ref(idx, e.value.partial.fetch())
    assert np.array_equal(got, H.expected("grumbach", "complete_short"))
    # -l skips the short query silently
    got = H.matches_as_ref(idx, V.findquerymatches(gi, gq, 8,
                                                   speedup=2).fetch())
    assert np.array_equal(got, H.expected("grumbach", "mem8_short"))
    # searchlength below prefixlength: fquery.c:440-446
    with pytest.raises(V.VsaError) as e:
        V.findquerymatches(gi, gq, 5)
    assert e.value.message == "searchlength=5 must be >= 6=prefixlen"


def test_empty_and_degenerate_batches(V):
    idx, q = H.load_case("micro")
    gi = gpu_index(V, "micro")
    empty = V.Queries.from_host(np.zeros(0, np.uint8), np.zeros(0, np.uint64),
                                np.zeros(0, np.uint64))
    assert V.findcompletematches(gi, empty).count == 0
    assert V.findquerymatches(gi, empty, 3).count == 0
    # queries that are all wildcards / shorter than the search length
    qq = H.Queries.from_list([[254, 254, 254], [0, 1], [3]])
    gq = gpu_queries(V, qq)
    assert V.findquerymatches(gi, gq, 3).count == 0
    assert np.array_equal(V.findcompletematches(gi, gq).fetch(),
                          H.oracle_complete(idx, qq))


def test_callback_delivery_order_and_stop(V):
    idx, q = H.load_case("grumbach")
    gi, gq = gpu_index(V, "grumbach"), gpu_queries(V, q)
    want = H.oracle_querymatches(idx, q, 14, speedup=2)
    rc, got = V.findquerymatches_cb(gi, gq, 14, speedup=2)
    assert rc == 0
    assert got == [tuple(int(x) for x in r) for r in want.tolist()]
    # a non-zero return of the callback stops the run (procexqu.c:61-64)
    rc, got = V.findquerymatches_cb(gi, gq, 14, stop_after=5)
    assert rc != 0 and len(got) == 5
    rc, got = V.findcompletematches_cb(gi, gpu_queries(
        V, H.fasta_queries(os.path.join(H.GOLDEN, "short.fna"))))
    assert rc < 0 and len(got) == 1       # one match, then the hard error
    allidx, _ = H.load_case("grumbach_all")
    rc, got = V.findmaximaluniquematches_cb(gpu_index(V, "grumbach_all"), 14)
    assert rc == 0
    assert got == [tuple(int(x) for x in r)
                   for r in H.oracle_selfmum(allidx, 14).tolist()]


def test_index_open_reads_mkvtree_files(V, tmp_path):
    """the C reader (vsa_index_open) on files in the reference's layout"""
    idx, q = H.load_case("grumbach_all")
    prefix = str(tmp_path / "all")
    i64 = idx.as_width(64)
    for sfx, arr in (("tis", i64.tis), ("suf", i64.suf), ("lcp", i64.lcp),
                     ("llv", i64.llv), ("bck", i64.bck), ("bwt", i64.bwt),
                     ("ssp", i64.ssp.astype(np.uint64))):
        arr.tofile(prefix + "." + sfx)
    prj = M["grumbach_all"]["index"]["prj"]
    with open(prefix + ".prj", "w") as f:
        f.write("dbfile=humhbb.fna 74522 73308\n")
        f.write("queryfile=humdystrop.fna 39422 38770\n")
        for k in ("totallength", "specialcharacters", "specialranges",
                  "lengthofspecialprefix", "lengthofspecialsuffix",
                  "numofsequences", "numofdbsequences",
                  "numofquerysequences", "longest", "prefixlength",
                  "largelcpvalues", "maxbranchdepth", "integersize",
                  "littleendian"):
            f.write("%s=%d\n" % (k, prj[k]))
    with open(prefix + ".al1", "w") as f:
        f.write("aA\ncC\ngG\ntTuU\nnsywrkvbdhmNSYWRKVBDHM\n")
    gi = V.Index.open(prefix)
    info = gi.info()
    assert info.totallength == idx.n and info.prefixlength == 6
    wide = os.environ.get("VSA_FORCE_WIDE") == "1"
    assert info.hasindexedqueries == 1
    assert info.device_integersize == (64 if wide else 32)
    got = H.selfmatches_as_ref(idx, V.findmaximaluniquematches(gi, 14).fetch())
    assert np.array_equal(got, H.expected("grumbach_all", "selfmum14"))


def test_self_index_scan_declines_alphabets_beyond_128_symbols(V):
    """ the streaming pass reads "special symbol" off bit 7 of the
    bwt byte, which holds for symbol codes below 128 only: an index over a
    larger alphabet (mkvtree -smap) is declined, one over 128 symbols is
    searched and equals the oracle"""
    rng = np.random.default_rng(128)
    for numofchars, ok in ((128, True), (129, False), (200, False)):
        db = rng.integers(0, numofchars, 3000).astype(np.uint8)
        qy = db[500:2500].copy()
        qy[rng.random(len(qy)) < 0.02] = numofchars - 1
        tis = np.concatenate([db, [H.SEPARATOR], qy]).astype(np.uint8)
        idx = H.oracle_build_index(tis, numofchars, prefixlength=1)
        idx.querysepposition, idx.hasqueries = len(db), True
        gi = V.Index.from_tables(idx.n, idx.prefixlength, numofchars, idx.tis,
                                 idx.suf, idx.lcp, idx.llv, idx.bck, idx.bwt,
                                 len(db), True)
        if ok:
            got = V.findmaximaluniquematches(gi, 8).fetch()
            assert len(got) > 10
            assert np.array_equal(got, H.oracle_selfmum(idx, 8))
        else:
            with pytest.raises(V.VsaError) as e:
                V.findmaximaluniquematches(gi, 8)
            assert e.value.code == V.NOT_COVERED
            assert "not covered" in e.value.message
        gi.close()


def test_self_index_scan_by_ranges(V):
    """vsa_findmaximaluniquematches_range (third row): any tiling
    of the reference's loop i = 2 .. n-1 (fmumself.c:33) into ranges gives,
    range by range, the matches of that part of the loop, and concatenated
    the list of the whole scan -- cuts at tile boundaries, next to peaks, empty
    and clamped ranges included"""
    from vstree_amd import sharding as S
    rng = np.random.default_rng(99)
    db = rng.integers(0, 4, 150000).astype(np.uint8)
    qy = db[20000:120000].copy()
    qy[rng.random(len(qy)) < 0.01] = rng.integers(0, 4)     # diverged copy
    db[1000:1400] = db[5000:5400]                           # lcp >= 255 inside
    qy[300:700] = db[1000:1400]
    tis = np.concatenate([db, [H.SEPARATOR], qy]).astype(np.uint8)
    synth = H.oracle_build_index(tis, 4)
    synth.querysepposition, synth.hasqueries = len(db), True
    gsynth = V.Index.from_tables(synth.n, synth.prefixlength, 4, synth.tis,
                                 synth.suf, synth.lcp, synth.llv, synth.bck,
                                 synth.bwt, len(db), True)
    real, _ = H.load_case("grumbach_all")
    for idx, gi, L in ((real, gpu_index(V, "grumbach_all"), 14),
                       (synth, gsynth, 12), (synth, gsynth, 300)):
        whole = V.findmaximaluniquematches(gi, L).fetch()
        assert np.array_equal(whole, H.oracle_selfmum(idx, L))
        assert len(whole) > (50 if L < 255 else 0)
        n = idx.n
        peaks = np.flatnonzero(idx.lcp[:n] >= min(L, 255))
        tilings = [[S.selfmum_range(n, r, w) for r in range(w)]
                   for w in (1, 2, 3, 8)]
        cuts = sorted({2, n} | {int(c) for c in rng.integers(2, n, 6)} |
                      {4096, 4097, 8191, 16384, 16385} |
                      {int(p) + d for p in peaks[:3] for d in (0, 1, 2)})
        tilings.append(list(zip(cuts, cuts[1:])))
        for tiling in tilings:
            parts = []
            for first, last in tiling:
                got = V.findmaximaluniquematches(gi, L, first, last).fetch()
                assert np.array_equal(
                    got, H.selfmum_scan_range(idx, L, first, last)), (
                        L, first, last)
                parts.append(got)
            assert np.array_equal(np.concatenate(parts), whole), (L, tiling)
        # clamped and empty ranges
        assert np.array_equal(
            V.findmaximaluniquematches(gi, L, 0, 2 ** 63).fetch(), whole)
        assert len(V.findmaximaluniquematches(gi, L, 500, 500).fetch()) == 0
        assert len(V.findmaximaluniquematches(gi, L, n, n + 9).fetch()) == 0
    with pytest.raises(V.VsaError):
        V.findmaximaluniquematches(gsynth, 12, 10, 5)


def test_random_ragged_queries_with_wildcards(V):
    """seeded random index with separators and wildcards, ragged queries"""
    rng = np.random.default_rng(777)
    parts = []
    for i in range(5):
        s = rng.integers(0, 4, size=int(rng.integers(3000, 9000))).astype(
            np.uint8)
        s[rng.integers(0, len(s), size=4)] = H.WILDCARD
        parts.append(s)
        parts.append(np.array([H.SEPARATOR], np.uint8))
    tis = np.concatenate(parts[:-1])
    idx = H.oracle_build_index(tis, 4)
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis, idx.suf,
                             idx.lcp, idx.llv, idx.bck, idx.bwt)
    seqs = []
    for i in range(2000):
        L = int(rng.integers(idx.prefixlength, 200))
        p = int(rng.integers(0, idx.n - L))
        s = tis[p:p + L].copy()
        if rng.random() < 0.4:
            s[int(rng.integers(0, L))] = int(rng.integers(0, 4))
        seqs.append(s)
    q = H.Queries.from_list(seqs)
    gq = gpu_queries(V, q)
    assert np.array_equal(V.findcompletematches(gi, gq).fetch(),
                          H.oracle_complete(idx, q))
    L = idx.prefixlength + 3
    for kw, sp in MODES:
        assert np.array_equal(
            V.findquerymatches(gi, gq, L, speedup=sp, **kw).fetch(),
            H.oracle_querymatches(idx, q, L, speedup=sp, **kw)), (kw, sp)


def test_repeats_beyond_255_and_many_occurrences(V):
    """long runs: lcp values >= 255 (llv exceptions) and thousands of hits"""
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, size=400).astype(np.uint8)
    tis = np.concatenate([unit, unit, rng.integers(0, 4, 500).astype(np.uint8),
                          unit, np.zeros(3000, np.uint8),
                          rng.integers(0, 4, 2000).astype(np.uint8)])
    idx = H.oracle_build_index(tis, 4, 4)
    assert idx.nllv > 0
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis, idx.suf,
                             idx.lcp, idx.llv, idx.bck, idx.bwt)
    q = H.Queries.from_list([unit, unit[:300], unit[50:350],
                             np.zeros(20, np.uint8), np.zeros(300, np.uint8),
                             np.concatenate([unit[100:], unit[:100]])])
    gq = gpu_queries(V, q)
    assert np.array_equal(V.findcompletematches(gi, gq).fetch(),
                          H.oracle_complete(idx, q))
    for L in (8, 260):
        for kw, sp in MODES:
            a = V.findquerymatches(gi, gq, L, speedup=sp, **kw).fetch()
            b = H.oracle_querymatches(idx, q, L, speedup=sp, **kw)
            assert np.array_equal(a, b), (L, kw, sp)


@pytest.mark.parametrize("mode", ["reference_walk", "deep7", "deep9",
                                  "deep12"])
def test_both_locate_strategies_give_the_same_lists(V, mode, monkeypatch):
    """the deep locate (bck2 + esa8) and the reference walk (bck + binary
    search on suf/tis) must be indistinguishable from outside"""
    if mode == "reference_walk":
        monkeypatch.setenv("VSA_NO_ESA8", "1")
    else:
        monkeypatch.setenv("VSA_DEEP_PREFIX", mode[4:])
    idx, q = H.load_case("c1")
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis,
                             idx.suf.astype(np.uint32), idx.lcp,
                             idx.llv.astype(np.uint32),
                             idx.bck.astype(np.uint32), idx.bwt)
    gq = gpu_queries(V, q)
    got = H.matches_as_ref(idx, V.findcompletematches(gi, gq).fetch())
    assert np.array_equal(got, H.expected("c1", "complete"))
    for key, kw in (("mem20_sp0", dict(speedup=0)), ("mem20_sp2", dict(speedup=2)),
                    ("mumcand20", dict(mum=True, cand=True)),
                    ("mum20", dict(mum=True))):
        got = H.matches_as_ref(idx, V.findquerymatches(gi, gq, 20,
                                                       **kw).fetch())
        assert np.array_equal(got, H.expected("c1", key)), (mode, key)
    # a search length below the deep prefix falls back by itself
    for sp in (0, 2):
        a = V.findquerymatches(gi, gq, 8, speedup=sp).fetch()
        assert np.array_equal(a, H.oracle_querymatches(idx, q, 8, speedup=sp))


def test_deep_locate_on_repeats_and_ties(V):
    """ties on the key (several suffixes share D+11 symbols), wildcards next
    to matches, queries that end inside the key window"""
    rng = np.random.default_rng(21)
    unit = rng.integers(0, 4, size=60).astype(np.uint8)
    parts = [np.tile(unit, 30), rng.integers(0, 4, 4000).astype(np.uint8),
             np.array([H.WILDCARD], np.uint8), np.tile(unit, 3),
             np.array([H.SEPARATOR], np.uint8),
             rng.integers(0, 4, 4000).astype(np.uint8), unit[:40],
             np.array([H.WILDCARD], np.uint8), unit[:33]]
    tis = np.concatenate(parts)
    idx = H.oracle_build_index(tis, 4, 3)
    gi = V.Index.from_tables(idx.n, idx.prefixlength, 4, idx.tis,
                             idx.suf.astype(np.uint32), idx.lcp,
                             idx.llv.astype(np.uint32),
                             idx.bck.astype(np.uint32), idx.bwt)
    assert gi.info().prefixlength == 3
    seqs = [unit, unit[:20], unit[5:17], unit[:9], np.tile(unit, 2),
            np.concatenate([unit[:30], [254], unit[31:]]).astype(np.uint8),
            tis[2000:2100], tis[1790:1830], tis[5990:6110]]
    for L in range(6, 40, 3):
        for p in (0, 7, 1234, 3000, 5000, 9000):
            seqs.append(tis[p:p + L])
    q = H.Queries.from_list(seqs)
    gq = gpu_queries(V, q)
    assert np.array_equal(V.findcompletematches(gi, gq).fetch(),
                          H.oracle_complete(idx, q))
    for L in (6, 7, 9, 12, 18, 25):
        for kw, sp in MODES:
            a = V.findquerymatches(gi, gq, L, speedup=sp, **kw).fetch()
            b = H.oracle_querymatches(idx, q, L, speedup=sp, **kw)
            assert np.array_equal(a, b), (L, kw, sp)


def test_mum_filter_in_dbstart_ranges_equals_whole_filter(V):
    """vsa_mumuniqueinquery_range with the carry of the lower ranges, range by
    range, gives the list of the one-piece filter (multi-GPU path)"""
    import ctypes as C
    idx, q = H.load_case("c1")
    gi, gq = gpu_index(V, "c1"), gpu_queries(V, q)
    cand = V.findquerymatches(gi, gq, 20, mum=True, cand=True).fetch()
    want = V.findquerymatches(gi, gq, 20, mum=True).fetch()
    assert np.array_equal(want, H.oracle_mumfilter(cand))
    pieces, carry, world = [], 0, 3
    dest = (cand["dbstart"] * np.uint64(world)) // np.uint64(idx.n + 1)
    for r in range(world):
        part = np.ascontiguousarray(cand[dest == r])
        dp = V.device_malloc(max(part.nbytes, 16))
        V.device_upload(dp, part)
        res = V.mumuniqueinquery_range(dp, len(part), carry)
        pieces.append(res.fetch())
        assert np.array_equal(pieces[-1], H.oracle_mumfilter(part, carry))
        if len(part):
            carry = max(carry, int((part["dbstart"] + part["length"]).max())
                        - 1)
        V.device_free(dp)
    assert np.array_equal(np.concatenate(pieces), want)


def host_reverse_complement(q):
    """copymultiseqRC (kurtz-basic/readmulti.c:93-125) in numpy"""
    sym = q.symbols.copy()
    for s, l in zip(q.start, q.length):
        s, l = int(s), int(l)
        seg = q.symbols[s:s + l][::-1]
        sym[s:s + l] = np.where(seg == H.WILDCARD, H.WILDCARD, 3 - seg)
    return H.Queries(sym, q.start, q.length)


@pytest.mark.parametrize("case", ["c1", "micro", "grumbach"])
def test_reverse_complement_queries(V, case):
    """vmatch -p at library level: the batch the engine receives with
    rcmode = True; checked against the oracle run on reverse complements
    made on the host (and, for c1, against the number of P lines the
    reference printed for -complete -d -p)"""
    idx, q = H.load_case(case)
    gi = gpu_index(V, case)
    rq = gpu_queries(V, q).reverse_complement()
    hq = host_reverse_complement(q)
    L = max(idx.prefixlength, 14 if case != "micro" else 2)
    got = V.findquerymatches(gi, rq, L, mum=True, cand=True).fetch()
    assert np.array_equal(got, H.oracle_querymatches(idx, hq, L, mum=True,
                                                     cand=True, speedup=0))
    if q.length.min() >= idx.prefixlength:
        both = V.findcompletematches(gi, rq).fetch()
        assert np.array_equal(both, H.oracle_complete(idx, hq))
        if case == "c1":
            fwd = V.findcompletematches(gi, gpu_queries(V, q)).count
            assert fwd + len(both) == M["c1"]["runs"]["complete_dp"]["lines"]


def test_reverse_complement_of_a_protein_batch_is_the_reference_error(V):
    q = H.Queries.from_list([[0, 1, 2, 3], [0, 7, 2]])
    with pytest.raises(V.VsaError) as ei:
        gpu_queries(V, q).reverse_complement()
    assert "reverse complement of 7 undefined" in str(ei.value)


def test_supermaximal_repeats_on_a_repetitive_text(V):
    """many nodes with more than two suffixes, wildcards and separators to
    the left of repeat copies, a copy at the very start of the text"""
    rng = np.random.default_rng(77)
    unit = rng.integers(0, 4, 300).astype(np.uint8)
    seqs = []
    for s in range(3):
        t = rng.integers(0, 4, 20000).astype(np.uint8)
        for r in range(10):
            p = int(rng.integers(0, 20000 - 300))
            u = unit.copy()
            for e in range(int(rng.integers(0, 4))):
                u[int(rng.integers(0, 300))] = rng.integers(0, 4)
            t[p:p + 300] = u
        t[rng.random(20000) < 0.002] = H.WILDCARD
        seqs.append(t)
    seqs[0][:60] = unit[:60]
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    for L in (8, 20, 60):
        got = V.findsupermaximalrepeats(gi, L).fetch()
        want = H.oracle_supermax(host, L)
        assert len(want) > 0 and np.array_equal(got, want)


def test_maximal_repeats_on_a_repetitive_text(V):
    """deep nodes with many children and grandchildren: the order of the
    reference's bottom-up traversal must come out of the slot arithmetic"""
    rng = np.random.default_rng(99)
    unit = rng.integers(0, 4, 200).astype(np.uint8)
    seqs = []
    for s in range(3):
        t = rng.integers(0, 4, 15000).astype(np.uint8)
        for r in range(12):
            p = int(rng.integers(0, 15000 - 200))
            u = unit.copy()
            for e in range(int(rng.integers(0, 5))):
                u[int(rng.integers(0, 200))] = rng.integers(0, 4)
            t[p:p + 200] = u
        t[3000:3300] = np.tile(np.array([0, 1, 0, 2], np.uint8), 75)
        t[rng.random(15000) < 0.002] = H.WILDCARD
        seqs.append(t)
    seqs[0][:50] = unit[:50]
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    for L in (10, 25, 80):
        got = V.findmaximalrepeats(gi, L).fetch()
        want = H.oracle_repeats(host, L)
        assert len(want) > 100 and np.array_equal(got, want)


def test_supermaximal_repeats_refuse_an_index_with_queries(V):
    gi = gpu_index(V, "grumbach_all")
    with pytest.raises(V.VsaError) as ei:
        V.findsupermaximalrepeats(gi, 14)
    assert "does not allow query files in index" in str(ei.value)


def test_empty_results_leave_no_error_behind(V):
    """a call that finds nothing (and so never starts its kernel timer) must
    not poison the next call through the runtime's sticky last error"""
    g = V.synth_genome(50000)
    gi = V.Index.build(g, 4, 0)
    assert V.findmaximalrepeats(gi, 10 ** 6).count == 0
    assert V.findsupermaximalrepeats(gi, 10 ** 6).count == 0
    assert V.findmaximalrepeats(gi, 10 ** 6).count == 0
    empty = V.Queries.from_host(np.zeros(0, np.uint8), np.zeros(0, np.uint64),
                                np.zeros(0, np.uint64))
    assert V.findquerymatches(gi, empty, 20, mum=True).count == 0
    assert V.findapproxcompletematches(gi, empty, True, 2).count == 0
    assert V.findsupermaximalrepeats(gi, 12).count >= 0
    tiny = V.Index.build(np.array([0, 1], np.uint8), 4, 1)
    assert V.findmaximalrepeats(tiny, 1).count == 0
    assert V.findsupermaximalrepeats(tiny, 1).count == 0


@pytest.mark.parametrize("seed", range(7))
def test_mum_work_plan_on_hard_batches(V, seed):
    """the first pass + work plan (offsets that cannot be candidates are not
    searched) on batches built to stress it: repetitive text (non-unique
    longest matches), reads with 0..6 substitutions at random and at chosen
    positions (first and last symbols, just inside and outside the last l
    symbols), wildcards in reads and text, reads that run into a sequence
    boundary, random reads; -mum cand and -mum against the oracle, in
    order"""
    rng = np.random.default_rng(4000 + seed)
    m = [100, 60, 150, 100, 254, 33, 300][seed]
    L = [20, 12, 31, 14, 40, 16, 25][seed]
    unit = rng.integers(0, 4, 3 * m).astype(np.uint8)
    seqs = []
    for s in range(3):
        t = rng.integers(0, 4, 30000).astype(np.uint8)
        for r in range(8):
            p = int(rng.integers(0, 30000 - len(unit)))
            u = unit.copy()
            for e in range(int(rng.integers(0, 4))):
                u[int(rng.integers(0, len(u)))] = rng.integers(0, 4)
            t[p:p + len(u)] = u
        t[rng.random(30000) < 0.0005] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    nq = 3000
    qb = np.zeros(nq * m, np.uint8)
    for i in range(nq):
        kind = i % 10
        p = int(rng.integers(0, len(tis) - m))
        q = tis[p:p + m].copy()
        q[q == H.SEPARATOR] = rng.integers(0, 4)   # read across a boundary
        if kind == 9:
            q = rng.integers(0, 4, m).astype(np.uint8)
        elif kind == 8:
            pos = [0, m - 1, m - L, m - L - 1, L - 1, L][i // 10 % 6]
            q[pos] = (q[pos] + 1) % 4 if q[pos] < 4 else 0
        else:
            for e in range(int(rng.integers(0, 7)) if kind < 6 else 0):
                x = int(rng.integers(0, m))
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4 if q[x] < 4 else 1
        if i % 97 == 0:
            q[int(rng.integers(0, m))] = H.WILDCARD
        qb[i * m:(i + 1) * m] = q
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    hq = H.Queries.uniform(qb, m)
    gq = V.Queries.from_host(qb, np.arange(nq, dtype=np.uint64) * m,
                             np.full(nq, m, np.uint64))
    if L < gi.info().prefixlength:
        L = gi.info().prefixlength
    cand = V.findquerymatches(gi, gq, L, mum=True, cand=True)
    want = H.oracle_querymatches(host, hq, L, mum=True, cand=True, speedup=0)
    assert len(want) > nq // 4
    assert np.array_equal(cand.fetch(), want)
    # the plan really left offsets out (short reads leave little to skip)
    if m >= 4 * L:
        assert cand.stats().kernel_searches < nq * (m - L + 1) // 2
    mum = V.findquerymatches(gi, gq, L, mum=True).fetch()
    assert np.array_equal(mum, H.oracle_querymatches(host, hq, L, mum=True,
                                                     speedup=0))


@pytest.mark.parametrize("case", ["micro", "grumbach", "grumbach_all",
                                  "largepat"])
def test_wide_device_tables(V, case, monkeypatch):
    """VSA_FORCE_WIDE=1: the 64-bit instantiations of every kernel (what an
    index beyond 2^32 positions uses) on the golden cases"""
    monkeypatch.setenv("VSA_FORCE_WIDE", "1")
    idx, q = H.load_case(case)
    i = idx.as_width(64)
    gi = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis, i.suf,
                             i.lcp, i.llv, i.bck, i.bwt, i.querysepposition,
                             i.hasqueries)
    assert gi.info().device_integersize == 64
    for key in sorted(M[case]["runs"]):
        run = M[case]["runs"][key]
        if "strands" in run or key.endswith("_short") or \
                key.startswith("approx_"):
            continue
        want = H.expected(case, key)
        if key.startswith("selfmum"):
            got = H.selfmatches_as_ref(idx, V.findmaximaluniquematches(
                gi, int(key[len("selfmum"):])).fetch())
        elif key.startswith("supermax"):
            got = H.repeats_as_ref(idx, V.findsupermaximalrepeats(
                gi, int(key[len("supermax"):])).fetch())
        elif key.startswith("tandem"):
            got = H.repeats_as_ref(idx, V.findtandems(
                gi, int(key[len("tandem"):])).fetch())
        elif key.startswith("repeats"):
            conv = (H.selfmatches_as_ref if idx.hasqueries
                    else H.repeats_as_ref)
            got = conv(idx, V.findmaximalrepeats(
                gi, int(key[len("repeats"):])).fetch())
        elif key.startswith("palindromic"):
            continue
        else:
            gq = gpu_queries(V, q)
            if key.startswith("complete"):
                got = H.matches_as_ref(idx,
                                       V.findcompletematches(gi, gq).fetch())
            else:
                name = key.partition("_sp")[0]
                L = int("".join(ch for ch in name if ch.isdigit()))
                got = H.matches_as_ref(idx, V.findquerymatches(
                    gi, gq, L, mum=name.startswith("mum"),
                    cand="cand" in name,
                    speedup=0 if key.endswith("_sp0") else 2).fetch())
        assert np.array_equal(got, want), (case, key)


@pytest.mark.parametrize("tune", [2])
def test_without_the_work_reduction_the_lists_are_the_same(V, tune,
                                                           monkeypatch):
    """VSA_TUNE=2 switches first pass and work plan off: every offset of
    every read is searched by the list form of the search kernel
    (esa_search.hip); the lists must not change"""
    monkeypatch.setenv("VSA_TUNE", str(tune))
    idx, q = H.load_case("c1")
    i = idx.as_width(64)
    gi = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis, i.suf,
                             i.lcp, i.llv, i.bck, i.bwt, i.querysepposition,
                             i.hasqueries)
    gq = gpu_queries(V, q)
    r = V.findquerymatches(gi, gq, 20, mum=True, cand=True)
    assert np.array_equal(H.matches_as_ref(idx, r.fetch()),
                          H.expected("c1", "mumcand20"))
    full = q.nq * (100 - 20 + 1)
    if os.environ.get("VSA_FORCE_WIDE") != "1":   # bit 2 is about deep locate
        assert (r.stats().kernel_searches == full) == (tune == 2)
    assert np.array_equal(
        H.matches_as_ref(idx, V.findquerymatches(gi, gq, 20,
                                                 mum=True).fetch()),
        H.expected("c1", "mum20"))


def test_long_reads_on_a_text_whose_largest_suffixes_share_255_symbols(V):
    """the one situation in which the reference's uniqueness test for
    lcp >= 255 (fquery.c:352) accepts a repeated match: the text ends in a
    long run of its largest symbol.  The work reduction steps aside for long
    reads there; lists equal the oracle, which restates the quirk."""
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 20000).astype(np.uint8)
    g[-600:] = 3
    gi = V.Index.build(g, 4, 0)
    t = gi.download()
    host = H.Index(len(g), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    m, nq = 400, 40
    qb = np.concatenate([g[len(g) - m - 7 * i:len(g) - 7 * i] if i else g[-m:]
                         for i in range(nq)]).astype(np.uint8)
    hq = H.Queries.uniform(qb, m)
    gq = V.Queries.from_host(qb, np.arange(nq, dtype=np.uint64) * m,
                             np.full(nq, m, np.uint64))
    for L in (20, 260):
        got = V.findquerymatches(gi, gq, L, mum=True, cand=True).fetch()
        want = H.oracle_querymatches(host, hq, L, mum=True, cand=True,
                                     speedup=0)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(3))
def test_mum_work_plan_on_ragged_batches(V, seed):
    """reads of different lengths (some shorter than the search length, one
    empty) take the first pass + work plan with per-query geometry"""
    rng = np.random.default_rng(7000 + seed)
    L = [20, 14, 30][seed]
    unit = rng.integers(0, 4, 400).astype(np.uint8)
    seqs = []
    for s in range(2):
        t = rng.integers(0, 4, 30000).astype(np.uint8)
        for r in range(6):
            p = int(rng.integers(0, 30000 - 400))
            u = unit.copy()
            for e in range(int(rng.integers(0, 4))):
                u[int(rng.integers(0, 400))] = rng.integers(0, 4)
            t[p:p + 400] = u
        t[rng.random(30000) < 0.0005] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([seqs[0], [H.SEPARATOR], seqs[1]]).astype(np.uint8)
    reads = []
    for i in range(2500):
        m = int(rng.integers(max(1, L - 6), 260))
        if i == 17:
            m = 0
        p = int(rng.integers(0, len(tis) - 300))
        q = tis[p:p + m].copy()
        q[q == H.SEPARATOR] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, 5)) if i % 3 else 0):
            if m:
                x = int(rng.integers(0, m))
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4 if q[x] < 4 else 2
        reads.append(q)
    hq = H.Queries.from_list(reads)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    L = max(L, gi.info().prefixlength)
    gq = gpu_queries(V, hq)
    cand = V.findquerymatches(gi, gq, L, mum=True, cand=True)
    want = H.oracle_querymatches(host, hq, L, mum=True, cand=True, speedup=0)
    assert len(want) > 500
    assert np.array_equal(cand.fetch(), want)
    full = int(np.maximum(hq.length.astype(np.int64) - L + 1, 0).sum())
    assert cand.stats().kernel_searches < full // 2
    assert np.array_equal(
        V.findquerymatches(gi, gq, L, mum=True).fetch(),
        H.oracle_querymatches(host, hq, L, mum=True, speedup=0))


def test_unordered_candidates_and_partition_by_index_range(V):
    """the multi-GPU building blocks: candidates as the kernel left them are
    the ordered list as a set; vsa_result_partition groups them by the range
    of the index their dbstart falls into, equal dbstarts together"""
    idx, q = H.load_case("c1")
    gi, gq = gpu_index(V, "c1"), gpu_queries(V, q)
    ordered = V.findmumcandidates(gi, gq, 20, ordered=True).fetch()
    assert np.array_equal(H.matches_as_ref(idx, ordered),
                          H.expected("c1", "mumcand20"))
    r = V.findmumcandidates(gi, gq, 20, ordered=False)
    loose = r.fetch()
    assert np.array_equal(np.sort(loose, order=list(loose.dtype.names)),
                          np.sort(ordered, order=list(ordered.dtype.names)))
    for nparts in (1, 2, 8, 5):
        buf = V.device_malloc(max(r.count, 1) * 32)
        counts, top = r.partition(nparts, idx.n, buf)
        got = np.zeros(r.count, V.MATCH_DTYPE)
        V.device_download(got, buf)
        V.device_free(buf)
        assert int(counts.sum()) == r.count
        right = ordered["dbstart"] + ordered["length"] - 1
        opart = (ordered["dbstart"] * np.uint64(nparts)) // np.uint64(
            idx.n + 1)
        assert [int(x) for x in top] == [
            int(right[opart == p].max()) if (opart == p).any() else 0
            for p in range(nparts)]
        part = (got["dbstart"] * np.uint64(nparts)) // np.uint64(idx.n + 1)
        assert np.array_equal(part, np.repeat(np.arange(nparts, dtype=np.uint64),
                                              counts.astype(np.int64)))
        assert np.array_equal(np.sort(got, order=list(got.dtype.names)),
                              np.sort(ordered, order=list(ordered.dtype.names)))


def test_tandem_repeats_on_texts_full_of_them(V):
    """short and long units, runs of one symbol (nested intervals with many
    suffixes), copies of tandem arrays, wildcards and separators inside and
    behind arrays, an array at the very end of the text"""
    rng = np.random.default_rng(4711)
    for trial in range(6):
        seqs = []
        for s in range(int(rng.integers(1, 4))):
            parts, total = [], 0
            while total < 3000:
                k = rng.random()
                if k < 0.45:
                    unit = rng.integers(0, 4, int(rng.integers(1, 15)))
                    piece = np.tile(unit, int(rng.integers(2, 9)))
                elif k < 0.5:
                    piece = np.full(int(rng.integers(1, 3)), H.WILDCARD)
                elif k < 0.6 and parts:
                    piece = parts[int(rng.integers(0, len(parts)))]
                elif k < 0.65:
                    piece = np.full(int(rng.integers(5, 70)),
                                    int(rng.integers(0, 4)))
                else:
                    piece = rng.integers(0, 4, int(rng.integers(1, 40)))
                parts.append(piece.astype(np.uint8))
                total += len(piece)
            seqs.append(np.concatenate(parts))
        if trial % 2:
            unit = rng.integers(0, 4, 7).astype(np.uint8)
            seqs[-1] = np.concatenate([seqs[-1], np.tile(unit, 5)])
        tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                              for s in seqs])[:-1].astype(np.uint8)
        gi = V.Index.build(tis, 4, 0)
        t = gi.download()
        host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"],
                       t["suf"], t["lcp"], t["llv"], t["bck"], t["bwt"], None)
        for L in (1, 2, 3, 5, 8, 13, 30):
            got = V.findtandems(gi, L).fetch()
            want = H.oracle_tandems(host, L)
            assert np.array_equal(got, want), (trial, L)
            if L == 1:
                assert len(want) > 100


def test_tandem_repeats_with_units_beyond_255(V):
    """interval depths >= 255 come from the exception table llv"""
    rng = np.random.default_rng(5)
    unit = rng.integers(0, 4, 300).astype(np.uint8)
    tis = np.concatenate([rng.integers(0, 4, 500), unit, unit, unit[:120],
                          rng.integers(0, 4, 400), unit, unit,
                          rng.integers(0, 4, 100)]).astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)
    for L in (100, 255, 256, 300, 301):
        got = V.findtandems(gi, L).fetch()
        want = H.oracle_tandems(host, L)
        assert np.array_equal(got, want), L
    assert len(H.oracle_tandems(host, 300)) >= 2


def test_tandem_repeats_errors_and_empty_results(V):
    gi = gpu_index(V, "grumbach_all")
    with pytest.raises(V.VsaError) as ei:
        V.findtandems(gi, 14)
    assert "does not allow query files in index" in str(ei.value)
    gi = gpu_index(V, "c1")
    assert V.findtandems(gi, 10 ** 6).count == 0
    with pytest.raises(V.VsaError):
        V.findtandems(gi, 0)


def test_packed_candidates_and_filter_in_ranges(V):
    """the multi-GPU form with 16-byte pairs: vsa_findmumcandidates_packed,
    vsa_result_partition on it, vsa_mumuniqueinquery_range_packed range by
    range = the one-piece filter; fetch() of the packed result = the
    candidate set"""
    import ctypes as C
    from vstree_amd import sharding as S
    idx, q = H.load_case("c1")
    gi, gq = gpu_index(V, "c1"), gpu_queries(V, q)
    want = V.findquerymatches(gi, gq, 20, mum=True).fetch()
    cand = V.findquerymatches(gi, gq, 20, mum=True, cand=True).fetch()
    for bits in (0, 7, 11):
        r = V.findmumcandidates_packed(gi, gq, 20, bits)
        used = r.packbits
        assert used == (bits or 7) and r.rowwords == 2
        assert np.array_equal(H.sorted_matches(r.fetch()),
                              H.sorted_matches(cand))
        # (up to 8 parts: tiles of 2 048 pairs counted in registers; beyond:
        # the general kernels; own: that part behind all others)
        for world, own in ((1, -1), (3, -1), (2, 0), (8, 5), (8, 7), (9, 4),
                           (1, 0)):
            dp = V.device_malloc(max(r.count * 16, 16))
            counts, top = r.partition(world, idx.n, dp, own=own)
            rows = np.zeros((r.count, 2), np.uint64)
            V.device_download(rows, dp)
            assert int(counts.sum()) == r.count
            # the form that leaves the numbers on the device and does not wait
            dm, dp2 = V.device_malloc(16 * world), V.device_malloc(
                max(r.count * 16, 16))
            r.partition_device(world, idx.n, dp2, dm, own=own)
            meta, rows2 = np.zeros(2 * world, np.uint64), np.zeros_like(rows)
            V.device_download(meta, dm)
            V.device_download(rows2, dp2)
            V.device_free(dm)
            V.device_free(dp2)
            assert np.array_equal(meta[:world], counts)
            assert np.array_equal(meta[world:], top)
            if world <= 8:      # (the tiled kernels place deterministically)
                assert np.array_equal(rows2, rows)
            assert np.array_equal(
                rows2[np.lexsort((rows2[:, 1], rows2[:, 0]))],
                rows[np.lexsort((rows[:, 1], rows[:, 0]))])
            if own >= 0:
                # the layout: others ascending, the own part last; the two
                # lists of the own-last form give the filter's answer for
                # that range
                order = [p for p in range(world) if p != own] + [own]
                got = S.unpack_candidates(rows, used, V.MATCH_DTYPE)
                dest = (got["dbstart"] * np.uint64(world)) // np.uint64(
                    idx.n + 1)
                assert np.array_equal(
                    dest, np.repeat(np.array(order, np.uint64),
                                    counts[order].astype(np.int64)))
                nown = int(counts[own])
                whole = S.unpack_candidates(rows[r.count - nown:], used,
                                            V.MATCH_DTYPE)
                cut = nown // 3
                res = V.mumuniqueinquery_range_packed2(
                    C.c_void_p(dp.value + 16 * (r.count - nown)), cut,
                    C.c_void_p(dp.value + 16 * (r.count - nown + cut)),
                    nown - cut, used, idx.n, 0)
                one = V.mumuniqueinquery_range_packed(
                    C.c_void_p(dp.value + 16 * (r.count - nown)), nown, used,
                    idx.n, 0)
                assert np.array_equal(res.fetch(), one.fetch())
                assert np.array_equal(one.fetch(),
                                      H.oracle_mumfilter(whole, 0))
                # back to part order for the checks below
                back = np.concatenate(
                    [rows[:int(counts[:own].sum())], rows[r.count - nown:],
                     rows[int(counts[:own].sum()):r.count - nown]])
                V.device_upload(dp, back)
                rows = back
            assert np.array_equal(
                H.sorted_matches(S.unpack_candidates(rows, used,
                                                     V.MATCH_DTYPE)),
                H.sorted_matches(cand))
            pieces, carry, off = [], 0, 0
            for p in range(world):
                cnt = int(counts[p])
                part = S.unpack_candidates(rows[off:off + cnt], used,
                                           V.MATCH_DTYPE)
                dest = (part["dbstart"] * np.uint64(world)) // np.uint64(
                    idx.n + 1)
                assert (dest == p).all()
                if cnt:
                    assert int(top[p]) == int(
                        (part["dbstart"] + part["length"]).max()) - 1
                res = V.mumuniqueinquery_range_packed(
                    C.c_void_p(dp.value + 16 * off), cnt, used, idx.n, carry)
                pieces.append(res.fetch())
                assert res.stats().sumlength == int(
                    pieces[-1]["length"].sum())
                carry = max(carry, int(top[p]))
                off += cnt
            V.device_free(dp)
            assert np.array_equal(np.concatenate(pieces), want)
    # search and grouping in one call, into a buffer the caller keeps
    r = V.findmumcandidates_packed(gi, gq, 20, 7)
    dp, dm = V.device_malloc(r.count * 16), V.device_malloc(16 * 4)
    counts, top = r.partition(4, idx.n, dp, own=2)
    rows = np.zeros((r.count, 2), np.uint64)
    V.device_download(rows, dp)
    dp2 = V.device_malloc(r.count * 16 + 64)
    r2, grouped = V.findmumcandidates_grouped(gi, gq, 20, 7, 4, 2, dp2,
                                              r.count - 1, dm)
    assert not grouped and r2.count == r.count     # no room: nothing grouped
    r3, grouped = V.findmumcandidates_grouped(gi, gq, 20, 7, 4, 2, dp2,
                                              r.count + 4, dm)
    rows3, meta = np.zeros_like(rows), np.zeros(8, np.uint64)
    V.device_download(rows3, dp2)
    V.device_download(meta, dm)
    assert grouped and r3.count == r.count
    assert np.array_equal(meta[:4], counts) and np.array_equal(meta[4:], top)
    part = lambda x: (S.unpack_candidates(x, 7, V.MATCH_DTYPE)["dbstart"]
                      * np.uint64(4)) // np.uint64(idx.n + 1)
    assert np.array_equal(part(rows3), part(rows))
    assert np.array_equal(rows3[np.lexsort((rows3[:, 1], rows3[:, 0]))],
                          rows[np.lexsort((rows[:, 1], rows[:, 0]))])
    for x in (dp, dp2, dm):
        V.device_free(x)
    with pytest.raises(V.VsaError):
        V.findmumcandidates_packed(gi, gq, 20, 5)   # 100 bp need 7 bits


def test_mum_filter_runs_of_equal_dbstart(V):
    """the filter sorts by dbstart alone and settles runs of equal dbstarts
    by looking at the run (k_mum_keyflags_runs); beyond 64 members it sorts on
    all bits instead.  Reads of different lengths from the same start: short
    runs with a unique longest member, with the longest twice, and a run of
    101 -- all against the oracle."""
    rng = np.random.default_rng(8)
    n = 50000
    tis = rng.integers(0, 4, n).astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    t = gi.download()
    host = H.Index(n, gi.info().prefixlength, 4, t["tis"], t["suf"],
                   t["lcp"], t["llv"], t["bck"], t["bwt"], None)

    def read(p, length):
        # ends in a foreign symbol so that the match stops at `length`
        r = np.concatenate([tis[p:p + length],
                            [(int(tis[p + length]) + 1) % 4]])
        return r.astype(np.uint8)

    for runs in ([(1000, [30, 40, 50, 35]), (3000, [60, 60, 45]),
                  (7000, [25]), (7001, [80, 70, 80, 30]),
                  (9000, list(range(30, 40)))],
                 [(2000, list(range(30, 131))), (5000, [50, 40])],
                 [(2000, [90] * 70 + [95]), (2100, [90] * 70 + [95, 95])]):
        reads = [read(p, m) for p, ms in runs for m in ms]
        reads += [tis[p:p + 70] for p in rng.integers(0, n - 200, 300)]
        order = rng.permutation(len(reads))
        q = H.Queries.from_list([reads[i] for i in order])
        got = V.findquerymatches(gi, gpu_queries(V, q), 20, mum=True).fetch()
        want = H.oracle_querymatches(host, q, 20, mum=True, speedup=0)
        assert np.array_equal(got, want), runs[0]


def test_inconsistent_tables_are_refused_on_upload(V):
    """index files anybody may have written: entries that point outside the
    tables are an error return of vsa_index_from_tables, not a GPU fault in
    the search kernels"""
    idx, _ = H.load_case("micro")
    i = idx.as_width(64)

    def upload(**changed):
        t = dict(suf=i.suf.copy(), bck=i.bck.copy(), llv=i.llv.copy())
        for k, f in changed.items():
            f(t[k])
        return V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis,
                                   t["suf"], i.lcp, t["llv"], t["bck"], i.bwt,
                                   i.querysepposition, i.hasqueries)

    upload().close()                                   # the real tables pass

    def beyond(a):
        a[len(a) // 2] = i.n + 5

    def swapped(a):
        a[3], a[4] = a[4] + 2, a[3]

    for what, kw in (("suf", dict(suf=beyond)), ("bck", dict(bck=beyond)),
                     ("bck", dict(bck=swapped))):
        with pytest.raises(V.VsaError) as e:
            upload(**kw)
        assert e.value.code == -2 and what in e.value.message, kw


def test_tandem_kernel_on_the_inverse_suffix_array(V, monkeypatch):
    """VSA_TANDEM_ISA=1 (selfmatch_search.inc): window comparisons by two
    loads of the inverse suffix array instead of the text -- an option for
    callers that ask repeatedly; the lists must not change"""
    monkeypatch.setenv("VSA_TANDEM_ISA", "1")
    for case, key in (("at1mb", "tandem40"), ("at1mb", "tandem5"),
                      ("grumbach", "tandem3"), ("largepat", "tandem8")):
        idx, _ = H.load_case(case)
        got = H.repeats_as_ref(idx, V.findtandems(
            gpu_index(V, case), int(key[len("tandem"):])).fetch())
        assert np.array_equal(got, H.expected(case, key)), (case, key)
