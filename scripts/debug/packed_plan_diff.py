"""debug: plans of the byte and of the packed form of the same reads"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import helpers as H
import vstree_amd as V

idx, q = H.load_case("c1")
i = idx.as_width(32)
gi = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis, i.suf, i.lcp, i.llv, i.bck, i.bwt)
m = 100
byte = V.Queries.from_host(q.symbols, q.start, q.length)
packed = V.Queries.from_host_packed(q.symbols, m)
plans = {}
for name, b in (("byte", byte), ("packed", packed)):
    pf = "/tmp/plan_%s.bin" % name
    os.environ["VSA_DEBUG_PLANFILE"] = pf
    r = V.findquerymatches(gi, b, 20, mum=True)
    print(name, "count", r.count, "searches", r.stats().searches, "kernel", r.stats().kernel_searches)
    plans[name] = np.fromfile(pf, np.uint32).reshape(-1, 5)
del os.environ["VSA_DEBUG_PLANFILE"]
a, b = plans["byte"], plans["packed"]
diff = np.flatnonzero((a != b).any(axis=1))
print("queries with different plans:", len(diff), "of", len(a), "deep prefix", gi.info().deepprefix)
g = idx.tis
sym = q.symbols.reshape(-1, m)
def fmt(row):
    return "count %d " % row[0] + " ".join("[%d,+%d)" % (x & 0xFFFF, x >> 16) for x in row[1:])
for k in diff[:12]:
    # where does the read differ from the genome at its best position?
    want = H.oracle_querymatches(idx, H.Queries.uniform(sym[k], m), 20, mum=True, cand=True, speedup=0)
    print("query", k, "| byte:", fmt(a[k]), "| packed:", fmt(b[k]), "| cand (len, db, q, off):", [tuple(int(x) for x in w) for w in want.tolist()])
