#!/usr/bin/env python3
"""-mum on a batch of reads of different lengths (the usual state of real
reads after trimming): time of the call next to the device time.
usage: ragged_probe.py N NQ [MINLEN] [MAXLEN] [L]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vstree_amd as V  # noqa: E402

n, nq = int(float(sys.argv[1])), int(float(sys.argv[2]))
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 80
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 120
L = int(sys.argv[5]) if len(sys.argv) > 5 else 20
g = V.synth_genome(n)
idx = V.Index.build(g, 4, 0)
rng = np.random.default_rng(3)
length = rng.integers(lo, hi + 1, nq).astype(np.uint64)
start = np.concatenate(([0], np.cumsum(length)[:-1])).astype(np.uint64)
pos = rng.integers(0, n - hi, nq)
sym = np.empty(int(length.sum()), np.uint8)
# reads = pieces of the genome, every fourth with one substitution
for i in range(nq):
    s, l = int(start[i]), int(length[i])
    sym[s:s + l] = g[pos[i]:pos[i] + l]
sub = np.arange(0, nq, 4)
at = (start[sub] + length[sub] // 2).astype(np.int64)
sym[at] = (sym[at] + 1) & 3
q = V.Queries.from_host(sym, start, length)
for kw, name in ((dict(mum=True), "mum"), (dict(mum=True, cand=True),
                                           "mumcand"), ({}, "mem")):
    for rep in range(3):
        t = time.time()
        r = V.findquerymatches(idx, q, L, **kw)
        s_ = r.stats()
        print("%s ragged %d reads: call %.2f ms, device %.2f ms, kernel %.2f "
              "ms, %d matches" % (name, nq, (time.time() - t) * 1e3,
                                  s_.total_device_ms, s_.search_kernel_ms,
                                  s_.count), flush=True)
        r.close()
