#!/usr/bin/env python3
"""A/B of tuning switches on one GPU: for each setting build the deep tables
anew (they are derived from the same 3 Gbp index) and time -mum -l 20."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vstree_amd as V

n, nq, m, L = int(float(sys.argv[1])), int(float(sys.argv[2])), 100, 20
settings = [s.split(",") for s in sys.argv[3:]]
dg = V.device_malloc(n + 64)
V._check(V.lib.vsa_synth_genome_device(42, n, dg, 0))
pos, sub, step = V.synth_query_plan(n, nq, m)
dq = V.device_malloc(nq * m + 64)
V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data, sub.ctypes.data,
                                        step.ctypes.data, nq, m, dq, 0))
q = V.Queries.from_device(dq, nq, m)
base = None
for st in settings:
    for kv in st:
        k, v = kv.split("=")
        os.environ[k] = v
    t0 = time.time()
    idx = V.Index.build_device(dg, n, 4, 0)
    tb = time.time() - t0
    res = []
    for rep in range(3):
        r = V.findquerymatches(idx, q, L, mum=True)
        s = r.stats()
        res.append(s.search_kernel_ms)
        cnt = s.count
        r.close()
    if base is None:
        base = cnt
    print("%-40s build %.1fs kernel ms %s  count %d %s" % (
        " ".join(st), tb, ["%.1f" % x for x in res], cnt,
        "OK" if cnt == base else "MISMATCH"), flush=True)
    idx.close()
    for kv in st:
        os.environ.pop(kv.split("=")[0])
