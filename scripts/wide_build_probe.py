"""Probe: GPU index builder on a synthetic genome of the given size, tables
forced wide or not; prints the doubling rounds (VSA_BUILD_TRACE) and checks
suf on the device-downloaded tables (permutation sums, sampled order)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
os.environ["VSA_BUILD_TRACE"] = "1"
import vstree_amd as V  # noqa: E402

n = int(float(sys.argv[1]))
dg = V.device_malloc(n + 64)
V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, 0))
t0 = time.time()
ix = V.Index.build_device(dg, n, 4, 0)
print("built %d in %.1f s" % (n, time.time() - t0), ix.info().device_integersize,
      ix.info().deepprefix, ix.info().device_bytes / 1e9, flush=True)
t = ix.download(with_bwt=False)
suf, tis = t["suf"], t["tis"]
idx = np.arange(n + 1, dtype=np.uint64)
print("perm sums", int(suf.astype(np.uint64).sum(dtype=np.uint64)) ==
      int(idx.sum(dtype=np.uint64)), flush=True)
rng = np.random.default_rng(1)
bad = 0
for j in rng.integers(1, n, size=20000):
    a, b, k = int(suf[j - 1]), int(suf[j]), 0
    while a + k < n and b + k < n and tis[a + k] == tis[b + k]:
        k += 1
    ok = a + k < n and (b + k >= n or tis[a + k] < tis[b + k])
    bad += 0 if ok else 1
print("misordered pairs in sample:", bad)
