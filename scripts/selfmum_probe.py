#!/usr/bin/env python3
"""The streaming lcptab scan (vmatch -mum on an index that holds its queries,
Vmengine/fmumself.c) at scale.
  dense : text = genome half + separator + a copy with one substitution every
          97 bp (every suffix pair is a peak, one MUM per 97 bp)
  sparse: two independent random halves (peaks are rare: the pure stream)
usage: selfmum_probe.py N [dense|sparse] [L]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vstree_amd as V

n = int(float(sys.argv[1]))
mode = sys.argv[2] if len(sys.argv) > 2 else "dense"
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20
half = (n - 1) // 2
g = V.synth_genome(half)
if mode == "dense":
    g2 = g.copy()
    g2[::97] = (g2[::97] + 1) & 3
else:
    g2 = V.synth_genome(half, seed=4243)
tis = np.concatenate([g, np.array([255], np.uint8), g2])
t0 = time.time()
idx = V.Index.build(tis, 4, 0)
idx.set_queryseparator(half)
print("%s index %d bp built in %.1fs" % (mode, len(tis), time.time() - t0),
      flush=True)
for rep in range(4):
    tw = time.time()
    r = V.findmaximaluniquematches(idx, L)
    tw = time.time() - tw
    s = r.stats()
    print("selfmum: peak pass %.3f ms total %.3f ms (call %.3f ms) peaks %d "
          "mums %d -> lcp+bwt stream %.1f GB/s" % (
              s.search_kernel_ms, s.total_device_ms, tw * 1e3, s.candidates,
              s.count, 2.0 * len(tis) / (s.search_kernel_ms * 1e-3) / 1e9),
          flush=True)
    if rep == 0 and len(tis) <= 4000000:
        import helpers as H
        t = idx.download()
        host = H.Index(len(tis), idx.info().prefixlength, 4, t["tis"],
                       t["suf"], t["lcp"], t["llv"], t["bck"], t["bwt"], None,
                       querysepposition=half, hasqueries=True)
        assert np.array_equal(r.fetch(), H.oracle_selfmum(host, L))
        print("  == CPU oracle", flush=True)
    r.close()
