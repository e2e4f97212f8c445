"""The CPU oracle against the reference: golden fixtures written by the real
reference programs (scripts/make_golden.py), the reference's own known-answer
file LargePat.res, and -- when oracle/_ref is built -- the live programs.
No GPU involved."""
import os

import numpy as np
import pytest

import helpers as H

M = H.manifest()


def run_oracle(case, key):
    idx, q = H.load_case(case)
    if key.startswith("approx_"):
        # approx_e2, approx_h2, approx_e2p: vmatch -complete -e 2 | -h 2 |
        # -e 2p; the distance travels in the querystart field
        spec = key[len("approx_"):]
        m = H.oracle_approx(idx, q, spec[0] == "e", int(spec[1:].rstrip("pb")),
                            percent=1 if spec.endswith("p")
                            else (2 if spec.endswith("b") else 0))
        return H.matches_as_ref(idx, m), None
    if key.startswith("complete"):
        try:
            m = H.oracle_complete(idx, q)
        except H.OracleError as e:
            return H.matches_as_ref(idx, e.partial), str(e)
        return H.matches_as_ref(idx, m), None
    if key.startswith("selfmum"):
        L = int(key[len("selfmum"):])
        return H.selfmatches_as_ref(idx, H.oracle_selfmum(idx, L)), None
    if key.startswith("palindromic"):
        L = int(key[len("palindromic"):].partition("_sp")[0])
        sp = int(key.partition("_sp")[2])
        m = H.oracle_querymatches(idx, H.index_as_rc_queries(idx), L,
                                  speedup=sp)
        return H.palindromic_as_ref(idx, m), None
    if key.startswith("repeats"):
        L = int(key[len("repeats"):])
        conv = H.selfmatches_as_ref if idx.hasqueries else H.repeats_as_ref
        return conv(idx, H.oracle_repeats(idx, L)), None
    if key.startswith("supermax"):
        L = int(key[len("supermax"):])
        return H.repeats_as_ref(idx, H.oracle_supermax(idx, L)), None
    if key.startswith("tandem"):
        L = int(key[len("tandem"):])
        return H.repeats_as_ref(idx, H.oracle_tandems(idx, L)), None
    name, _, sp = key.partition("_sp")
    sp = int(sp) if sp else 2
    if name.startswith("mumcand"):
        L, kw = int(name[7:]), dict(mum=True, cand=True)
    elif name.startswith("mum"):
        L, kw = int(name[3:]), dict(mum=True)
    else:
        L, kw = int(name[3:].split("_")[0]), {}
    return H.matches_as_ref(
        idx, H.oracle_querymatches(idx, q, L, speedup=sp, **kw)), None


CASES = [(c, k) for c in sorted(M) for k in sorted(M[c]["runs"])
         if "strands" not in M[c]["runs"][k]]


@pytest.mark.parametrize("case,key", CASES)
def test_oracle_reproduces_reference_output(case, key):
    run = M[case]["runs"][key]
    if case == "grumbach" and key.endswith("_short"):
        # queries come from short.fna, not from the case's query file
        idx, _ = H.load_case(case)
        q = H.fasta_queries(os.path.join(H.GOLDEN, "short.fna"))
        if key == "complete_short":
            with pytest.raises(H.OracleError) as ei:
                H.oracle_complete(idx, q)
            assert "patternlength=5 must be >= 6=prefixlen" in str(ei.value)
            assert "patternlength=5 must be >= 6=prefixlen" in run["stderr"]
            got = H.matches_as_ref(idx, ei.value.partial)
        else:
            got = H.matches_as_ref(idx, H.oracle_querymatches(idx, q, 8,
                                                              speedup=2))
    else:
        got, err = run_oracle(case, key)
        assert err is None
    want = H.expected(case, key)
    assert len(got) == run["lines"]
    # bit-exact INCLUDING the order in which the reference emits
    assert np.array_equal(got, want)


def test_largepat_known_answer_file():
    """src/Vmatch/Itercomplete.sh:25-33: vmatch -complete -d -q LargePat.test
    ychrIII.fna must print Testdir/LargePat.res (positions and lengths)."""
    idx, q = H.load_case("largepat")
    got = H.matches_as_ref(idx, H.oracle_complete(idx, q))
    with open(os.path.join(H.GOLDEN, "LargePat.res")) as f:
        want = H.parse_vmatch_lines([l for l in f.read().splitlines() if l])
    assert np.array_equal(got, want)


def test_index_vs_online_differential():
    """src/Vmatch/Complete.sh:30-43: matching on the index and the
    Boyer-Moore-Horspool scan of the text give the same set."""
    for case in ("largepat", "micro", "c1"):
        idx, q = H.load_case(case)
        a = H.sorted_matches(H.oracle_complete(idx, q))
        b = H.sorted_matches(H.oracle_complete(idx, q, online=True))
        assert np.array_equal(a, b)


def test_algorithm0_and_algorithm2_agree():
    """-qspeedup 0 and the default -qspeedup 2 report the same matches."""
    for case, L in (("grumbach", 14), ("micro", 3), ("wildcards", 2)):
        idx, q = H.load_case(case)
        for kw in ({}, dict(mum=True, cand=True), dict(mum=True)):
            a = H.oracle_querymatches(idx, q, L, speedup=0, **kw)
            b = H.oracle_querymatches(idx, q, L, speedup=2, **kw)
            assert np.array_equal(H.sorted_matches(a), H.sorted_matches(b))


def test_mum_query_vs_selfindex_equivalence():
    """src/Vmatch/Mum.sh:35-61: vmatch -mum on the db+query index and
    vmatch -mum -q query db report the same MUMs."""
    idx, q = H.load_case("grumbach")
    allidx, _ = H.load_case("grumbach_all")
    a = H.matches_as_ref(idx, H.oracle_querymatches(idx, q, 14, mum=True))
    b = H.selfmatches_as_ref(allidx, H.oracle_selfmum(allidx, 14))
    assert np.array_equal(H.sorted_matches(a), H.sorted_matches(b))


def test_32bit_and_64bit_tables_agree():
    idx, q = H.load_case("grumbach")
    i32 = idx.as_width(32)
    assert np.array_equal(H.oracle_complete(idx, q),
                          H.oracle_complete(i32, q))
    assert np.array_equal(H.oracle_querymatches(idx, q, 14),
                          H.oracle_querymatches(i32, q, 14))


def test_index_tables_match_reference_md5():
    """the CPU table builder (oracle/vsindex.c) reproduces mkvtree's files
    byte for byte -- load_case asserts the md5 sums"""
    for case in M:
        H.load_case(case)


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref not built")
def test_live_reference_on_fresh_random_input(tmp_path):
    """fresh seeds every time the reference binary is around: genome with
    wildcards and several sequences, ragged queries"""
    rng = np.random.default_rng(12345)
    wd = str(tmp_path)
    seqs = []
    for i in range(4):
        s = rng.integers(0, 4, size=int(rng.integers(2000, 6000)))
        seqs.append(s.astype(np.uint8))
    letters = np.frombuffer(b"acgt", np.uint8)
    recs = []
    for i, s in enumerate(seqs):
        b = bytearray(letters[s].tobytes())
        for p in rng.integers(0, len(b), size=5):
            b[p] = ord("n")
        recs.append(("s%d" % i, bytes(b)))
    H.write_fasta(wd + "/db.fna", recs)
    qrecs = []
    for i in range(300):
        s = seqs[int(rng.integers(0, 4))]
        L = int(rng.integers(12, 120))
        p = int(rng.integers(0, len(s) - L))
        q = s[p:p + L].copy()
        if rng.random() < 0.3:
            q[int(rng.integers(0, L))] ^= 1
        qrecs.append(("q%d" % i, letters[q].tobytes()))
    H.write_fasta(wd + "/q.fna", qrecs)
    H.run_mkvtree_ref(["-db", "db.fna", "-dna", "-pl", "-allout"], wd)
    idx = H.load_mkvtree_index(wd + "/db.fna")
    tis, ssp, _ = H.fasta_text([wd + "/db.fna"])
    mine = H.oracle_build_index(tis, 4, idx.prefixlength)
    for t in ("suf", "lcp", "bck", "bwt", "sti1", "llv"):
        assert np.array_equal(getattr(idx, t), getattr(mine, t)), t
    q = H.fasta_queries(wd + "/q.fna")
    for args, fn in (
            (["-complete"], lambda: H.oracle_complete(idx, q)),
            (["-l", "12"], lambda: H.oracle_querymatches(idx, q, 12,
                                                         speedup=2)),
            (["-qspeedup", "0", "-l", "12"],
             lambda: H.oracle_querymatches(idx, q, 12, speedup=0)),
            (["-mum", "cand", "-l", "12"],
             lambda: H.oracle_querymatches(idx, q, 12, mum=True, cand=True)),
            (["-mum", "-l", "12"],
             lambda: H.oracle_querymatches(idx, q, 12, mum=True))):
        rc, lines, err = H.run_vmatch_ref(args + ["-q", "q.fna", "db.fna"],
                                          wd)
        assert rc == 0, err
        assert np.array_equal(H.parse_vmatch_lines(lines),
                              H.matches_as_ref(idx, fn())), args


def test_tandem_known_answer_of_the_reference():
    """src/Vmatch/Checktandem.sh: vmatch -l 40 -tandem on the index of
    src/testdata/at1MB must print src/Vmatch/Testdir/Tandem40AT (stored as it
    is): the oracle reproduces the file, order included"""
    known = [l for l in open(os.path.join(H.GOLDEN, "Tandem40AT"))
             .read().splitlines() if l and not l.startswith("#")]
    want = H.parse_vmatch_lines(known)
    assert len(want) == 8
    idx, _ = H.load_case("at1mb")
    got = H.repeats_as_ref(idx, H.oracle_tandems(idx, 40))
    assert np.array_equal(got, want)
    assert np.array_equal(want, H.expected("at1mb", "tandem40"))


def test_tandem_oracle_refuses_an_index_with_queries():
    idx, _ = H.load_case("grumbach_all")
    with pytest.raises(H.OracleError) as ei:
        H.oracle_tandems(idx, 14)
    assert "does not allow query files in index" in str(ei.value)


@pytest.mark.skipif(not H.have_ref(), reason="needs oracle/_ref (built where "
                    "/root/reference exists)")
def test_esaapm_restatement_against_the_live_reference():
    """patterns that reach esaapm / esahamming (splitsize 1, piece thresholds
    > 0): scripts/pin_esaapm_probe.py, a few rounds"""
    import subprocess
    import sys
    p = subprocess.run([sys.executable,
                        os.path.join(H.ROOT, "scripts", "pin_esaapm_probe.py"),
                        "11", "4"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"all 4 ok" in p.stdout, p.stdout.decode()


@pytest.mark.skipif(not H.have_ref(), reason="needs oracle/_ref (built where "
                    "/root/reference exists)")
def test_the_reference_reads_behind_the_text_in_the_approximate_tail(tmp_path):
    """The one documented deviation of `-complete -e K`:
    for a start position in the last m + K symbols the reference hands
    `maxlength` symbols to its longest-match function without looking at the
    end of the mapped text (Vmengine/approxcompl.c:14-66 ->
    longestmatch.c:18-71), i.e. it reads what the kernel maps behind the last
    byte of the .tis file -- zeros ('a') up to the end of the page, anything
    beyond.  Its answer there is a function of memory the index does not hold:
    here it reports a match that ENDS BEHIND THE TEXT.  The oracle (and the
    GPU) end the text where it ends.  scripts/approx_tail_probe.py prints more
    cases (profiles/r03/approx_tail_reference.txt)."""
    rng = np.random.default_rng(7)
    n, m, k = 10000, 40, 2           # n % 4096 != 0: zeros behind the text
    t = rng.integers(0, 4, n).astype(np.uint8)
    pat = np.concatenate([t[n - (m - 2):], np.zeros(2, np.uint8)])
    wd = str(tmp_path)
    H.write_fasta(wd + "/db.fna", [("s0", t)])
    H.write_fasta(wd + "/q.fna", [("q0", pat)])
    H.run_mkvtree_ref(["-db", "db.fna", "-dna", "-pl", "-allout"], wd)
    rc, lines, _ = H.run_vmatch_ref(["-complete", "-e", str(k), "-q", "q.fna",
                                     "db.fna"], wd)
    assert rc == 0 and len(lines) == 1
    f = lines[0].split()
    start, length, dist = int(f[2]), int(f[0]), abs(int(f[7]))
    assert start == n - (m - 2) and start + length > n    # over the end
    assert (length, dist) == (m, 0)      # "matched" the page's zero padding
    idx = H.load_mkvtree_index(wd + "/db.fna")
    got = H.matches_as_ref(idx, H.oracle_approx(idx, H.Queries.from_list([pat]),
                                                True, k))
    assert len(got) == 1
    assert (int(got[0]["dbrel"]), int(got[0]["length"]),
            int(got[0]["querystart"])) == (start, m - 2, 2)
