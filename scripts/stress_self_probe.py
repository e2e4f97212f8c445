#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison of the scans over the index itself:
maximal repeats, supermaximal repeats, tandem repeats (order included) on
texts with repeats, tandem arrays, low complexity, wildcards, several
sequences.
usage: stress_self_probe.py [ROUNDS] [SEED]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
import vstree_amd as V  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
checked = 0
for rnd in range(rounds):
    seqs = []
    for s in range(int(rng.integers(1, 4))):
        parts, total = [], 0
        target = int(rng.integers(2000, 30000))
        while total < target:
            k = rng.random()
            if k < 0.15:
                unit = rng.integers(0, 4, int(rng.integers(1, 20)))
                piece = np.tile(unit, int(rng.integers(2, 12)))
            elif k < 0.2:
                piece = np.full(int(rng.integers(1, 3)), H.WILDCARD)
            elif k < 0.4 and parts:
                piece = parts[int(rng.integers(0, len(parts)))].copy()
                if len(piece) > 3 and rng.random() < 0.5:
                    piece[int(rng.integers(0, len(piece)))] = rng.integers(0, 4)
            elif k < 0.45:
                piece = np.full(int(rng.integers(5, 300)),
                                int(rng.integers(0, 4)))
            else:
                piece = rng.integers(0, 4, int(rng.integers(1, 200)))
            parts.append(piece.astype(np.uint8))
            total += len(piece)
        seqs.append(np.concatenate(parts))
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    tb = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, tb["tis"], tb["suf"],
                   tb["lcp"], tb["llv"], tb["bck"], tb["bwt"], None)
    for L in sorted({1, int(rng.integers(2, 12)), int(rng.integers(12, 60)),
                     int(rng.integers(60, 400))}):
        for name, gpu, cpu in (
                ("repeats", V.findmaximalrepeats, H.oracle_repeats),
                ("supermax", V.findsupermaximalrepeats, H.oracle_supermax),
                ("tandem", V.findtandems, H.oracle_tandems)):
            if name == "repeats" and L < 6 and len(tis) > 8000:
                continue   # quadratic output
            got = gpu(gi, L).fetch()
            want = cpu(host, L)
            if not np.array_equal(got, want):
                print("MISMATCH round %d %s L=%d: gpu %d oracle %d" % (
                    rnd, name, L, len(got), len(want)), flush=True)
                sys.exit(1)
            checked += 1
    print("round %d ok: %d bp, %.0f s" % (rnd, len(tis), time.time() - t0),
          flush=True)
print("all %d lists equal the oracle's" % checked)
