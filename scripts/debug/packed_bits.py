"""debug: which packed kernel differs (VSA_DEBUG_ROWS bits: 1 first, 2 plan, 4 search)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import helpers as H
import vstree_amd as V
idx, q = H.load_case("c1")
i = idx.as_width(32)
gi = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis, i.suf, i.lcp, i.llv, i.bck, i.bwt)
packed = V.Queries.from_host_packed(q.symbols, 100)
for bits in (0, 1, 2, 4, 3, 5, 6, 7):
    os.environ["VSA_DEBUG_ROWS"] = str(bits)
    r = V.findquerymatches(gi, packed, 20, mum=True)
    s = r.stats()
    print("bits", bits, "count", r.count, "searches", s.searches, "kernel", s.kernel_searches, "cand", s.candidates)
