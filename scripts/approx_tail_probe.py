#!/usr/bin/env python3
"""What does the reference do with approximate matches that start in the last
m + K symbols of the text?  (verdict r2, item 5)

edistprocessstartpos (Vmengine/approxcompl.c:14-66) hands
`sequence + startpos` and the region's maxlength (up to m + K) to
*patternlongestmatch (Vmengine/longestmatch.c), which reads `maxlength`
symbols from there without looking at the end of the text.  The text is the
mapped .tis file (kurtz-basic/readvirt.c), so what lies behind it is whatever
the kernel maps behind a file's last byte: zeros up to the end of the page
(symbol 'a'), nothing beyond.  This probe builds texts whose LENGTH decides
what is there:
  * n not a multiple of the page size, text and pattern end in 'a' runs:
    the reference reports a match that extends into the zero padding
    (length > what the text holds);
  * n a multiple of 4096: the same start position makes the reference read
    an unmapped page (SIGBUS), or stops it -- depending on nothing the
    index holds.
Prints the reference's lines next to the oracle's for each case.
Needs oracle/_ref.  usage: approx_tail_probe.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H

H.build_oracle()
rng = np.random.default_rng(7)


def case(n, taila, m=40, k=2, extra=2):
    """text of n symbols that ends in `taila` symbols 'a'; pattern = the last
    m - extra symbols of the text followed by `extra` more 'a': inside the
    text it occurs at the very end with `extra` errors (the missing symbols);
    with something 'a'-like behind the text it occurs there without any"""
    t = rng.integers(0, 4, n).astype(np.uint8)
    t[n - taila:] = 0
    pat = np.concatenate([t[n - (m - extra):], np.zeros(extra, np.uint8)])
    wd = tempfile.mkdtemp(prefix="tail_")
    H.write_fasta(wd + "/db.fna", [("s0", t)])
    H.write_fasta(wd + "/q.fna", [("q0", pat)])
    H.run_mkvtree_ref(["-db", "db.fna", "-dna", "-pl", "-allout"], wd)
    idx = H.load_mkvtree_index(wd + "/db.fna")
    rc, lines, err = H.run_vmatch_ref(["-complete", "-e", str(k), "-q",
                                       "q.fna", "db.fna"], wd)
    try:
        got = H.matches_as_ref(idx, H.oracle_approx(
            idx, H.Queries.from_list([pat]), True, k))
        olines = ["%d %d %d (dist %d)" % (r["length"], r["dbseq"], r["dbrel"],
                                         r["querystart"]) for r in got]
    except H.OracleError as e:
        olines = ["oracle error: %s" % e]
    print("n = %d (n %% 4096 = %d), text ends in %d x 'a', m = %d, K = %d"
          % (n, n % 4096, taila, m, k))
    print("  reference rc = %d%s" % (rc, (" stderr: " + err.strip()[:120])
                                     if err.strip() else ""))
    for l in lines:
        f = l.split()
        print("    ref   : len %s at %s (dist %s)   -> ends at %d of %d"
              % (f[0], f[2], f[7], int(f[2]) + int(f[0]), n))
    for l in olines:
        print("    oracle: " + l)


case(10000, 8)      # padding of zeros behind the text: "match" runs over the end
case(10000, 8, extra=1)
case(8192, 8)       # the page ends with the text
case(12288, 8, extra=1)
case(10000, 0)      # no 'a' run in the text; the padding is 'a' all the same
