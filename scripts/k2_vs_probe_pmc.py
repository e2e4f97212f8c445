"""One -mum batch (light + heavy pass) and the table-read probes on the same
index, for a rocprofv3 --pmc run that compares their memory counters."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
import vstree_amd as V

n, nq, m, L = 3_000_000_000, 10_000_000, 100, 20
dg = V.device_malloc(n + 64, 0)
V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, 0))
index = V.Index.build_device(dg, n, 4, 0, 0)
pos, sub, step = V.synth_query_plan(n, nq, m)
dq = V.device_malloc(nq * m + 64, 0)
V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data, sub.ctypes.data,
                                        step.ctypes.data, nq, m, dq, 0))
queries = V.Queries.from_device(dq, nq, m, 0)
V.device_free(dq, 0)
V.device_free(dg, 0)
for _ in range(2):
    r = V.findquerymatches(index, queries, L, mum=True)
    s = r.stats()
    print("search kernels %.3f ms, %d matches" % (s.search_kernel_ms, s.count))
    r.close()
g = C.c_double()
for table in (0, 4):
    V._check(V.lib.vsa_measure_table_read(index._h, table, 4, C.byref(g)))
    print("table %d: %.1f G reads/s" % (table, g.value))
