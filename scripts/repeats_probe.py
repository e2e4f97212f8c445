#!/usr/bin/env python3
"""Maximal and supermaximal repeats (vmatch -l L IDX, -supermax) at scale: a
random genome with a diverged copy of its first part (one substitution every
STEP bp) inside the same index.
usage: repeats_probe.py N [L] [STEP] [COPY]"""
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
step = int(sys.argv[3]) if len(sys.argv) > 3 else 97
copy = int(float(sys.argv[4])) if len(sys.argv) > 4 else n // 10
g = V.synth_genome(n - copy - 1)
c = g[:copy].copy()
c[::step] = (c[::step] + 1) & 3
tis = np.concatenate([g, np.array([255], np.uint8), c])
t0 = time.time()
idx = V.Index.build(tis, 4, 0)
print("index %d bp (copy of %d bp, one substitution per %d) built in %.1fs"
      % (len(tis), copy, step, time.time() - t0), flush=True)
for name, fn in (("maximal repeats", V.findmaximalrepeats),
                 ("supermaximal repeats", V.findsupermaximalrepeats)):
    for rep in range(3):
        tw = time.time()
        r = fn(idx, L)
        tw = time.time() - tw
        s = r.stats()
        print("%s -l %d: %d matches, %d candidates, total %.2f ms (call "
              "%.2f ms; %.1f M matches/s)" % (name, L, s.count, s.candidates,
                                              s.total_device_ms, tw * 1e3,
                                              s.count / s.total_device_ms
                                              / 1e3), flush=True)
        if rep == 0 and len(tis) <= 3000000:
            import helpers as H
            t = idx.download()
            host = H.Index(len(tis), idx.info().prefixlength, 4, t["tis"],
                           t["suf"], t["lcp"], t["llv"], t["bck"], t["bwt"],
                           None)
            want = (H.oracle_repeats if fn is V.findmaximalrepeats
                    else H.oracle_supermax)(host, L)
            assert np.array_equal(r.fetch(), want)
            print("  == CPU oracle", flush=True)
        r.close()
