#!/usr/bin/env python3
"""vmatch -l L (MEM enumeration) alone at scale, for profiling.
usage: mem_probe.py N NQ [M] [L] [REPS] [QSPEEDUP]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vstree_amd as V  # noqa: E402

n, nq = int(float(sys.argv[1])), int(float(sys.argv[2]))
m = int(sys.argv[3]) if len(sys.argv) > 3 else 100
L = int(sys.argv[4]) if len(sys.argv) > 4 else 20
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 2
speedup = int(sys.argv[6]) if len(sys.argv) > 6 else 2
dg = V.device_malloc(n + 64)
V._check(V.lib.vsa_synth_genome_device(42, n, dg, 0))
idx = V.Index.build_device(dg, n, 4, 0)
pos, sub, step = V.synth_query_plan(n, nq, m)
dq = V.device_malloc(nq * m + 64)
V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data,
                                        sub.ctypes.data, step.ctypes.data,
                                        nq, m, dq, 0))
q = V.Queries.from_device(dq, nq, m)
for rep in range(reps):
    t = time.time()
    r = V.findquerymatches(idx, q, L, speedup=speedup)
    s = r.stats()
    print("mem -l %d -qspeedup %d: call %.1f ms kernel %.2f ms total %.2f ms matches %d "
          "searches %d" % (L, speedup, (time.time() - t) * 1e3, s.search_kernel_ms,
                           s.total_device_ms, s.count, s.searches), flush=True)
    r.close()
