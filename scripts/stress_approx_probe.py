#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison of approximate complete matching
(-complete -e K | -h K): texts with planted repeats (queries with hundreds of
piece hits), wildcards and several sequences; uniform and mixed read lengths,
small and large batches (the per-length plan), thresholds 1..3, and the
best-of form (Kb) on the same batch.
usage: stress_approx_probe.py [ROUNDS] [SEED]"""
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
checked = declined = 0
for rnd in range(rounds):
    seqs = []
    for s in range(int(rng.integers(1, 4))):
        n = int(rng.integers(30000, 90000))
        t = rng.integers(0, 4, n).astype(np.uint8)
        unit = rng.integers(0, 4, int(rng.integers(100, 400))).astype(np.uint8)
        for r in range(int(rng.integers(0, 40))):       # diverged copies
            p = int(rng.integers(0, n - len(unit)))
            u = unit.copy()
            for e in range(int(rng.integers(0, 5))):
                u[int(rng.integers(0, len(u)))] = rng.integers(0, 4)
            t[p:p + len(u)] = u
        if rng.random() < 0.5:
            t[rng.random(n) < 0.0007] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    gi = V.Index.build(tis, 4, 0)
    tb = gi.download()
    host = H.Index(len(tis), gi.info().prefixlength, 4, tb["tis"], tb["suf"],
                   tb["lcp"], tb["llv"], tb["bck"], tb["bwt"], None)
    uniform = rng.random() < 0.5
    big = rng.random() < 0.4
    doedist = rng.random() < 0.7
    k = int(rng.integers(1, 4))
    short = rnd % 3 == 2     # esaapm / esahamming configurations
    m0 = int(rng.integers(max(k + 3, 6), 34)) if short else \
        int(rng.integers(40 * k + 30, 220))
    nreads = int(rng.integers(4200, 7000)) if big else int(rng.integers(100, 900))
    if short:
        nreads = int(rng.integers(30, 120))
    reads = []
    for i in range(nreads):
        if short:
            m = m0 if uniform else int(rng.integers(max(k + 3, 6), 34))
        else:
            m = m0 if uniform else int(rng.integers(
                max(40 * k + 30, m0 - 40), m0 + 30))
        p = int(rng.integers(0, len(tis) - m))
        q = tis[p:p + m].copy()
        q[q == H.SEPARATOR] = rng.integers(0, 4)
        if not short or rng.random() < 0.8:
            q[q >= H.WILDCARD] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, k + 2))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0 or not doedist or uniform:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4
            elif kind == 1 and len(q) > k + 3:
                q = np.delete(q, x)
            else:
                q = np.insert(q, x, rng.integers(0, 4))
        reads.append(q.astype(np.uint8))
    hq = H.Queries.from_list(reads)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    try:
        want = H.oracle_approx(host, hq, doedist, k)
    except H.OracleNotCovered:
        declined += 1
        continue
    got = V.findapproxcompletematches(gi, gq, doedist, k).fetch()
    if not np.array_equal(got, want):
        print("MISMATCH round %d: %s k=%d m0=%d %s %d reads: gpu %d oracle %d"
              % (rnd, "edist" if doedist else "hamming", k, m0,
                 "uniform" if uniform else "ragged", nreads, len(got),
                 len(want)), flush=True)
        sys.exit(1)
    checked += 1
    # the best-of form on the same batch (-e Kb / -h Kb: the matches at the
    # read's smallest distance within K percent of its length)
    kb, nbest = int(rng.integers(1, 10)), -1
    try:
        wantb = H.oracle_approx(host, hq, doedist, kb, percent=2)
    except (H.OracleNotCovered, H.OracleError):
        wantb = None
    if wantb is not None:
        try:
            gotb = V.findapproxcompletematches(gi, gq, doedist, kb, 2).fetch()
        except V.VsaError as e:
            if e.code != V.NOT_COVERED:
                raise
            gotb = None
        if gotb is not None:
            if not np.array_equal(gotb, wantb):
                print("MISMATCH round %d (best of): %s %db m0=%d %s %d reads: "
                      "gpu %d oracle %d"
                      % (rnd, "edist" if doedist else "hamming", kb, m0,
                         "uniform" if uniform else "ragged", nreads,
                         len(gotb), len(wantb)), flush=True)
                sys.exit(1)
            checked += 1
            nbest = len(wantb)
    print("round %d ok: %s k=%d m~%d %s %d reads, %d matches; %db: %d, %.0f s"
          % (rnd, "edist" if doedist else "hamming", k, m0,
             "uniform" if uniform else "ragged", nreads, len(want), kb, nbest,
             time.time() - t0), flush=True)
print("all %d lists equal the oracle's (%d configurations outside the "
      "restatement)" % (checked, declined))
