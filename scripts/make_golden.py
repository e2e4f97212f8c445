#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container (needs /root/reference and the programs built by
`make -f oracle/Makefile.ref`):

    python3 scripts/make_golden.py

For every case it runs oracle/_ref/mkvtree_ref and oracle/_ref/vmatch_ref and
stores DATA only:
  * the input sequences (small FASTA files taken from the reference's own
    test data src/testdata/Grumbach, src/Vmatch/Testdir, or produced by the
    deterministic generator of vstree_amd/csrc/synth.c),
  * md5 sums of the index tables mkvtree wrote and the .prj numbers,
  * the match lists vmatch printed, parsed into integer arrays
    (length, dbseq, dbrel, queryseq, querystart) in output order.
The reference's known-answer file src/Vmatch/Testdir/LargePat.res is stored
as it is (it is a data fixture of the reference's own test suite,
src/Vmatch/Itercomplete.sh:25-33).
"""
import gzip
import hashlib
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import helpers as H  # noqa: E402
import vstree_amd as V  # noqa: E402

REFSRC = "/root/reference/src"
GOLD = os.path.join(ROOT, "tests", "golden")
TABLES = ("tis", "suf", "lcp", "llv", "bck", "bwt", "sti1")


def md5file(p):
    return hashlib.md5(open(p, "rb").read()).hexdigest()


def gzcopy(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        g.write(f.read())


def index_case(wd, indexname, mkvargs):
    H.run_mkvtree_ref(mkvargs, wd)
    prefix = os.path.join(wd, indexname)
    prj = H.read_prj(prefix + ".prj")
    return {"prj": {k: v for k, v in prj.items()
                    if k not in ("dbfile", "queryfile")},
            "md5": {t: md5file(prefix + "." + t) for t in TABLES},
            "mkvargs": mkvargs, "indexname": indexname,
            "md5_allfiles": {t: md5file(prefix + "." + t)
                             for t in ("prj", "al1", "tis", "ois", "des",
                                       "sds", "ssp", "suf", "lcp", "llv",
                                       "bck", "bwt", "sti1", "skp")
                             if os.path.exists(prefix + "." + t)}}


def run_case(wd, args):
    rc, lines, err = H.run_vmatch_ref(args, wd)
    return rc, lines, err


def main():
    if not H.have_ref():
        sys.exit("build the reference first: make -f oracle/Makefile.ref")
    os.makedirs(GOLD, exist_ok=True)
    manifest, arrays = {}, {}

    def record(case, key, args, wd, both=False, approx=False):
        rc, lines, err = run_case(wd, args)
        entry = {"args": args, "rc": rc, "lines": len(lines),
                 "md5_lines": hashlib.md5(
                     ("\n".join(lines) + "\n").encode()).hexdigest()}
        if not both:   # lists with P lines are pinned by their md5 only
            arrays["%s__%s" % (case, key)] = H.parse_vmatch_lines(
                lines, approx=approx)
        else:
            entry["strands"] = "both"
        if rc != 0:
            entry["stderr"] = err.strip()
        manifest[case]["runs"][key] = entry

    # ---- 1. LargePat: the reference's own known-answer test --------------
    wd = tempfile.mkdtemp()
    case = "largepat"
    shutil.copy(REFSRC + "/testdata/Grumbach/ychrIII.fna", wd)
    shutil.copy(REFSRC + "/Vmatch/Testdir/LargePat.test", wd)
    gzcopy(wd + "/ychrIII.fna", GOLD + "/ychrIII.fna.gz")
    shutil.copy(wd + "/LargePat.test", GOLD + "/LargePat.test")
    shutil.copy(REFSRC + "/Vmatch/Testdir/LargePat.res",
                GOLD + "/LargePat.res")
    manifest[case] = {"db": ["ychrIII.fna.gz"], "query": "LargePat.test",
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "ychrIII.fna", ["-db", "ychrIII.fna", "-dna", "-pl", "-allout"])
    record(case, "complete", ["-complete", "-d", "-q", "LargePat.test",
                              "ychrIII.fna"], wd)
    for sp in (0, 2):
        record(case, "mem20_sp%d" % sp,
               ["-qspeedup", str(sp), "-l", "20", "-q", "LargePat.test",
                "ychrIII.fna"], wd)
    record(case, "mem300", ["-l", "300", "-q", "LargePat.test",
                            "ychrIII.fna"], wd)
    # supermaximal repeats of the chromosome itself (Vmengine/fsuper.c)
    record(case, "supermax30", ["-supermax", "-l", "30", "ychrIII.fna"], wd)
    # maximal repeats (Vmengine/vmatfind.c), the default task of vmatch
    record(case, "repeats40", ["-l", "40", "ychrIII.fna"], wd)
    # palindromic self matches: the reverse complement of the index is
    # matched against the index (Vmatch/runself.c:127-178)
    for sp in (0, 2):
        record(case, "palindromic30_sp%d" % sp,
               ["-qspeedup", str(sp), "-p", "-l", "30", "ychrIII.fna"], wd)
    record(case, "repeats_dp40", ["-d", "-p", "-l", "40", "ychrIII.fna"], wd,
           both=True)
    # right branching tandem repeats (Vmengine/ftandem.c)
    record(case, "tandem8", ["-tandem", "-l", "8", "ychrIII.fna"], wd)
    shutil.rmtree(wd)

    # ---- 2. micro: multi-FASTA, wildcards, prefixlength 1 ----------------
    wd = tempfile.mkdtemp()
    case = "micro"
    db = [("s0", b"acgtacgtttgacnnacgtacgtaaccggtt"),
          ("s1", b"ttgacgacgtacgtttga"), ("s2", b"ggggacgtacgt")]
    qs = [("q0", b"acgtacgt"), ("q1", b"acgnacgt"), ("q2", b"gacnnacg"),
          ("q3", b"gtcaa"), ("q4", b"tacgt"), ("q5", b"ttgacgacgtacgtttgat"),
          ("q6", b"nnnnn"), ("q7", b"ggggacgtacgtn")]
    H.write_fasta(wd + "/db.fna", db)
    H.write_fasta(wd + "/q.fna", qs)
    H.write_fasta(wd + "/qshort.fna", qs[:2] + [("tiny", b"")] + qs[2:])
    for f in ("db.fna", "q.fna"):
        shutil.copy(wd + "/" + f, GOLD + "/micro_" + f)
    manifest[case] = {"db": ["micro_db.fna"], "query": "micro_q.fna",
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "db.fna", ["-db", "db.fna", "-dna", "-pl", "-allout"])
    record(case, "complete", ["-complete", "-q", "q.fna", "db.fna"], wd)
    for L in (1, 3, 5):
        for sp in (0, 2):
            record(case, "mem%d_sp%d" % (L, sp),
                   ["-qspeedup", str(sp), "-l", str(L), "-q", "q.fna",
                    "db.fna"], wd)
        record(case, "mumcand%d" % L, ["-mum", "cand", "-l", str(L), "-q",
                                       "q.fna", "db.fna"], wd)
        record(case, "mum%d" % L, ["-mum", "-l", str(L), "-q", "q.fna",
                                   "db.fna"], wd)
    record(case, "supermax2", ["-supermax", "-l", "2", "db.fna"], wd)
    record(case, "repeats2", ["-l", "2", "db.fna"], wd)
    for L in (1, 2):
        record(case, "tandem%d" % L, ["-tandem", "-l", str(L), "db.fna"], wd)
    shutil.rmtree(wd)

    # ---- 3. Wildcards.fna of the reference's test data --------------------
    wd = tempfile.mkdtemp()
    case = "wildcards"
    shutil.copy(REFSRC + "/testdata/Grumbach/Wildcards.fna", wd)
    shutil.copy(wd + "/Wildcards.fna", GOLD + "/Wildcards.fna")
    manifest[case] = {"db": ["Wildcards.fna"], "query": "Wildcards.fna",
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "Wildcards.fna", ["-db", "Wildcards.fna", "-dna", "-pl",
                              "-allout"])
    record(case, "mem2", ["-l", "2", "-q", "Wildcards.fna",
                          "Wildcards.fna"], wd)
    record(case, "mumcand2", ["-mum", "cand", "-l", "2", "-q",
                              "Wildcards.fna", "Wildcards.fna"], wd)
    shutil.rmtree(wd)

    # ---- 4. a Grumbach pair, as in src/Vmatch/Mum.sh / Itermum.sh ---------
    wd = tempfile.mkdtemp()
    case = "grumbach"
    dbf, qf = "humhbb.fna", "humdystrop.fna"
    for f in (dbf, qf):
        shutil.copy(REFSRC + "/testdata/Grumbach/" + f, wd)
        gzcopy(wd + "/" + f, GOLD + "/" + f + ".gz")
    manifest[case] = {"db": [dbf + ".gz"], "query": qf + ".gz", "runs": {}}
    manifest[case]["index"] = index_case(
        wd, dbf, ["-db", dbf, "-dna", "-pl", "-allout"])
    for sp in (0, 2):
        record(case, "mem14_sp%d" % sp, ["-qspeedup", str(sp), "-l", "14",
                                         "-q", qf, dbf], wd)
    record(case, "mumcand14", ["-mum", "cand", "-l", "14", "-q", qf, dbf], wd)
    record(case, "mum14", ["-mum", "-l", "14", "-q", qf, dbf], wd)
    # queries shorter than prefixlength (6 here): a hard error for -complete
    # after the matches of the queries in front of it (exactcompl.c:179-185),
    # silently skipped for -l (matchsub.c:187-190)
    H.write_fasta(wd + "/short.fna",
                  [("a", b"ttttcaacctctttgt"), ("b", b"agacaccatggtgcacctg"),
                   ("c", b"acgta"), ("d", b"gtgcacctgactcctgag")])
    shutil.copy(wd + "/short.fna", GOLD + "/short.fna")
    record(case, "complete_short", ["-complete", "-q", "short.fna", dbf], wd)
    record(case, "mem8_short", ["-l", "8", "-q", "short.fna", dbf], wd)
    record(case, "supermax12", ["-supermax", "-l", "12", dbf], wd)
    record(case, "repeats12", ["-l", "12", dbf], wd)
    record(case, "tandem3", ["-tandem", "-l", "3", dbf], wd)
    # queries inside the index (Mum.sh:35-61): vmatch -mum on db+query index
    manifest["grumbach_all"] = {"db": [dbf + ".gz"], "indexedquery":
                                [qf + ".gz"], "runs": {}}
    manifest["grumbach_all"]["index"] = index_case(
        wd, "all", ["-indexname", "all", "-db", dbf, "-q", qf, "-dna", "-pl",
                    "-allout"])
    record("grumbach_all", "selfmum14", ["-mum", "-l", "14", "all"], wd)
    # maximal repeats between database and query of the same index
    record("grumbach_all", "repeats14", ["-l", "14", "all"], wd)
    shutil.rmtree(wd)

    # ---- 5. C1: synthetic 1 Mbp genome, 10 k x 100 bp queries -------------
    wd = tempfile.mkdtemp()
    case = "c1"
    n, nq, m = 1000000, 10000, 100
    g = V.synth_genome(n)
    qb = V.synth_queries(g, nq, m)
    H.write_fasta(wd + "/genome.fna", [("synthetic_genome seed=42", g)])
    H.write_fasta(wd + "/queries.fna",
                  [("q%d" % i, qb[i * m:(i + 1) * m]) for i in range(nq)],
                  width=1000)
    manifest[case] = {"synthetic": {"n": n, "nq": nq, "m": m,
                                    "genome_seed": 42, "query_seed": 4242},
                      "md5_genome_fna": md5file(wd + "/genome.fna"),
                      "md5_queries_fna": md5file(wd + "/queries.fna"),
                      "md5_genome_codes": hashlib.md5(g.tobytes()).hexdigest(),
                      "md5_query_codes": hashlib.md5(qb.tobytes()).hexdigest(),
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "genome.fna", ["-db", "genome.fna", "-dna", "-pl", "-allout"])
    record(case, "complete", ["-complete", "-q", "queries.fna",
                              "genome.fna"], wd)
    for sp in (0, 2):
        record(case, "mem20_sp%d" % sp, ["-qspeedup", str(sp), "-l", "20",
                                         "-q", "queries.fna", "genome.fna"],
               wd)
    record(case, "mumcand20", ["-mum", "cand", "-l", "20", "-q",
                               "queries.fna", "genome.fna"], wd)
    record(case, "mum20", ["-mum", "-l", "20", "-q", "queries.fna",
                           "genome.fna"], wd)
    # both strands (-d -p): reverse-complemented queries, flag P
    record(case, "complete_dp", ["-complete", "-d", "-p", "-q", "queries.fna",
                                 "genome.fna"], wd, both=True)
    record(case, "mum20_dp", ["-mum", "-l", "20", "-d", "-p", "-q",
                              "queries.fna", "genome.fna"], wd, both=True)
    # approximate complete matches (BASELINE.json configs[4] semantics):
    # edit distance and Hamming distance, threshold 2
    record(case, "approx_e2", ["-complete", "-e", "2", "-q", "queries.fna",
                               "genome.fna"], wd, approx=True)
    record(case, "approx_h2", ["-complete", "-h", "2", "-q", "queries.fna",
                               "genome.fna"], wd, approx=True)
    record(case, "approx_e5b", ["-complete", "-e", "5b", "-q", "queries.fna",
                                "genome.fna"], wd, approx=True)
    record(case, "approx_h5b", ["-complete", "-h", "5b", "-q", "queries.fna",
                                "genome.fna"], wd, approx=True)
    shutil.rmtree(wd)

    # ---- 6. C5 in small: 150 bp (and some 100 bp) reads with up to 3 edit
    # operations against a 3-sequence text with wildcards and planted repeats
    wd = tempfile.mkdtemp()
    case = "c5"
    rng = np.random.default_rng(20240605)
    seqs = []
    for s in range(3):
        L = 70000
        t = rng.integers(0, 4, L).astype(np.uint8)
        unit = rng.integers(0, 4, 400).astype(np.uint8)
        for r in range(8):          # diverged copies of one unit
            p = int(rng.integers(0, L - 400))
            u = unit.copy()
            for e in range(int(rng.integers(0, 6))):
                u[int(rng.integers(0, 400))] = rng.integers(0, 4)
            t[p:p + 400] = u
        t[rng.random(L) < 0.0005] = H.WILDCARD
        seqs.append(t)
    qs = []
    for i in range(2000):
        s = seqs[int(rng.integers(0, 3))]
        m = 150 if i % 4 else 100
        p = int(rng.integers(0, len(s) - m))
        q = s[p:p + m].copy()
        q[q == H.WILDCARD] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, 4))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4
            elif kind == 1:
                q = np.delete(q, x)
            else:
                q = np.insert(q, x, rng.integers(0, 4))
        if i % 97 == 0:
            q[int(rng.integers(0, len(q)))] = H.WILDCARD
        qs.append(q.astype(np.uint8))
    H.write_fasta(wd + "/db.fna", [("s%d" % i, t) for i, t in enumerate(seqs)])
    H.write_fasta(wd + "/reads.fna", [("r%d" % i, q) for i, q in
                                      enumerate(qs)], width=1000)
    gzcopy(wd + "/db.fna", GOLD + "/c5_db.fna.gz")
    gzcopy(wd + "/reads.fna", GOLD + "/c5_reads.fna.gz")
    manifest[case] = {"db": ["c5_db.fna.gz"], "query": "c5_reads.fna.gz",
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "db.fna", ["-db", "db.fna", "-dna", "-pl", "-allout"])
    for k in (1, 2, 3):
        record(case, "approx_e%d" % k, ["-complete", "-e", str(k), "-q",
                                        "reads.fna", "db.fna"], wd,
               approx=True)
    record(case, "approx_h2", ["-complete", "-h", "2", "-q", "reads.fna",
                               "db.fna"], wd, approx=True)
    record(case, "approx_e2p", ["-complete", "-e", "2p", "-q", "reads.fna",
                                "db.fna"], wd, approx=True)
    # "best of" thresholds (Vmengine/initcompl.c:59-77): every read at the
    # smallest threshold <= K percent of its length at which it has a match
    record(case, "approx_e4b", ["-complete", "-e", "4b", "-q", "reads.fna",
                                "db.fna"], wd, approx=True)
    record(case, "approx_h3b", ["-complete", "-h", "3b", "-q", "reads.fna",
                                "db.fna"], wd, approx=True)
    record(case, "complete", ["-complete", "-q", "reads.fna", "db.fna"], wd)
    record(case, "supermax20", ["-supermax", "-l", "20", "db.fna"], wd)
    record(case, "repeats25", ["-l", "25", "db.fna"], wd)
    record(case, "tandem4", ["-tandem", "-l", "4", "db.fna"], wd)
    shutil.rmtree(wd)

    # ---- 6b. short patterns with many errors: the configurations that reach
    # esaapm / esahamming (Vmengine/splitesaapm.c:400-425, 523-543) -- pieces
    # with a threshold of their own (K >= m/10) and patterns that are not cut
    # at all (splitsize 1).  Reads of 8..32 bp against a repetitive 3-sequence
    # text with wildcards.  The text ends in 64 wildcards: the reference's
    # longest match (approxcompl.c:14-66 called with the width of the region,
    # splitesaapm.c:96-105) reads up to m + K symbols from a start position
    # without looking at the end of the mapped text, so that a match starting
    # in the last m + K symbols depends on the bytes behind the file.
    wd = tempfile.mkdtemp()
    case = "c6"
    rng = np.random.default_rng(20261004)
    seqs = []
    for s in range(3):
        L = 16000
        t = rng.integers(0, 4, L).astype(np.uint8)
        unit = rng.integers(0, 4, 60).astype(np.uint8)
        for r in range(25):
            p = int(rng.integers(0, L - 60))
            u = unit.copy()
            for e in range(int(rng.integers(0, 3))):
                u[int(rng.integers(0, 60))] = rng.integers(0, 4)
            t[p:p + 60] = u
        for r in range(3):
            ln = int(rng.integers(15, 50))
            a = int(rng.integers(0, L - ln))
            t[a:a + ln] = np.resize(rng.integers(0, 4, int(rng.integers(1, 4))),
                                    ln)
        t[rng.random(L) < 0.001] = H.WILDCARD
        seqs.append(t)
    seqs[2][-64:] = H.WILDCARD     # nothing matches there
    qs = []
    for i in range(240):
        s = seqs[int(rng.integers(0, 3))]
        m = int(rng.integers(8, 33))
        p = int(rng.integers(0, len(s) - 200))
        q = s[p:p + m].copy()
        if i % 11:
            q[q == H.WILDCARD] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, 4))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4 if q[x] < 4 else 1
            elif kind == 1 and len(q) > 8:
                q = np.delete(q, x)
            else:
                q = np.insert(q, x, rng.integers(0, 4))
        qs.append(q.astype(np.uint8))
    H.write_fasta(wd + "/db.fna", [("s%d" % i, t) for i, t in enumerate(seqs)])
    H.write_fasta(wd + "/reads.fna", [("r%d" % i, q) for i, q in
                                      enumerate(qs)], width=1000)
    gzcopy(wd + "/db.fna", GOLD + "/c6_db.fna.gz")
    gzcopy(wd + "/reads.fna", GOLD + "/c6_reads.fna.gz")
    manifest[case] = {"db": ["c6_db.fna.gz"], "query": "c6_reads.fna.gz",
                      "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "db.fna", ["-db", "db.fna", "-dna", "-pl", "-allout"])
    for k in (1, 2, 3):
        record(case, "approx_e%d" % k, ["-complete", "-e", str(k), "-q",
                                        "reads.fna", "db.fna"], wd,
               approx=True)
    for k in (1, 2):
        record(case, "approx_h%d" % k, ["-complete", "-h", str(k), "-q",
                                        "reads.fna", "db.fna"], wd,
               approx=True)
    shutil.rmtree(wd)

    # ---- 7. the reference's tandem repeat test (src/Vmatch/Checktandem.sh):
    # index of src/testdata/at1MB built as src/bin/Makeindex.sh does, vmatch
    # -l 40 -tandem against the known answer src/Vmatch/Testdir/Tandem40AT
    wd = tempfile.mkdtemp()
    case = "at1mb"
    shutil.copy(REFSRC + "/testdata/at1MB", wd + "/at1MB")
    gzcopy(wd + "/at1MB", GOLD + "/at1MB.gz")
    shutil.copy(REFSRC + "/Vmatch/Testdir/Tandem40AT", GOLD + "/Tandem40AT")
    manifest[case] = {"db": ["at1MB.gz"], "runs": {}}
    manifest[case]["index"] = index_case(
        wd, "atindex", ["-indexname", "atindex", "-db", "at1MB", "-pl",
                        "-dna", "-bwt", "-lcp", "-suf", "-ois", "-tis",
                        "-bck", "-sti1"])
    for L in (40, 12, 5):
        record(case, "tandem%d" % L, ["-tandem", "-l", str(L), "atindex"],
               wd)
    rc, lines, err = run_case(wd, ["-l", "40", "-tandem", "atindex"])
    known = [l for l in open(GOLD + "/Tandem40AT").read().splitlines()
             if l and not l.startswith("#")]
    assert lines == known, "the reference does not reproduce Tandem40AT"
    record(case, "supermax40", ["-supermax", "-l", "40", "atindex"], wd)
    shutil.rmtree(wd)

    np.savez_compressed(GOLD + "/expected.npz", **arrays)
    with open(GOLD + "/manifest.json", "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(arrays), "match lists for", len(manifest), "cases")


if __name__ == "__main__":
    main()
