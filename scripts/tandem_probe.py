#!/usr/bin/env python3
"""Right branching tandem repeats (vmatch -tandem -l L IDX) at scale: a random
genome with a tandem array planted every SPACING bp (unit 1..60 bp, 2..40
copies, one substitution in every fourth array).
usage: tandem_probe.py N [L] [SPACING]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vstree_amd as V  # noqa: E402

n = int(float(sys.argv[1]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spacing = int(float(sys.argv[3])) if len(sys.argv) > 3 else 10000
tis = V.synth_genome(n)
rng = np.random.default_rng(99)
narrays = 0
for p in range(spacing, n - 3000, spacing):
    u = int(rng.integers(1, 61))
    c = int(rng.integers(2, 41))
    arr = np.tile(rng.integers(0, 4, u).astype(np.uint8), c)
    if narrays % 4 == 3:
        k = int(rng.integers(0, len(arr)))
        arr[k] = (arr[k] + 1) & 3
    tis[p:p + len(arr)] = arr
    narrays += 1
t0 = time.time()
idx = V.Index.build(tis, 4, 0)
print("index %d bp with %d tandem arrays built in %.1fs"
      % (len(tis), narrays, time.time() - t0), flush=True)
for rep in range(3):
    tw = time.time()
    r = V.findtandems(idx, L)
    tw = time.time() - tw
    s = r.stats()
    print("tandem repeats -l %d: %d repeats, %d positions with lcp >= L, "
          "interval kernel %.2f ms, total %.2f ms (call %.2f ms)"
          % (L, s.count, s.candidates, s.search_kernel_ms, s.total_device_ms,
             tw * 1e3), flush=True)
    if rep == 0 and len(tis) <= 3000000:
        import helpers as H
        t = idx.download()
        host = H.Index(len(tis), idx.info().prefixlength, 4, t["tis"],
                       t["suf"], t["lcp"], t["llv"], t["bck"], t["bwt"], None)
        assert np.array_equal(r.fetch(), H.oracle_tandems(host, L))
        print("  == CPU oracle", flush=True)
    r.close()
