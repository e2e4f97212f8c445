#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison of approximate complete matching with
thresholds in PERCENT over reads of very different lengths (-complete -e Kp |
-h Kp): short reads whose threshold is 0 (the exact search), reads of up to
520 symbols (eight Myers words), 32- and 64-bit device tables, texts with
planted repeats, wildcards and several sequences.  What round 3 added to the
engine (batches that mix thresholds 0 and > 0, m <= 512, positions in the
width of the tables) under random configurations.
usage: stress_approx_mixed_probe.py [ROUNDS] [SEED]"""
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
checked = declined = errors = 0
for rnd in range(rounds):
    seqs = []
    for s in range(int(rng.integers(1, 4))):
        n = int(rng.integers(30000, 90000))
        t = rng.integers(0, 4, n).astype(np.uint8)
        unit = rng.integers(0, 4, int(rng.integers(100, 600))).astype(np.uint8)
        for r in range(int(rng.integers(0, 30))):       # diverged copies
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
    built = V.Index.build(tis, 4, 0)
    tb = built.download()
    host = H.Index(len(tis), built.info().prefixlength, 4, tb["tis"],
                   tb["suf"], tb["lcp"], tb["llv"], tb["bck"], tb["bwt"], None)
    wide = rng.random() < 0.5
    gi = built
    if wide:
        i = host.as_width(64)
        os.environ["VSA_FORCE_WIDE"] = "1"
        try:
            gi = V.Index.from_tables(i.n, i.prefixlength, i.numofchars, i.tis,
                                     i.suf, i.lcp, i.llv, i.bck, i.bwt,
                                     i.querysepposition, i.hasqueries)
        finally:
            del os.environ["VSA_FORCE_WIDE"]
        assert gi.info().device_integersize == 64
    doedist = rng.random() < 0.7
    pct = int(rng.integers(1, 4))
    lo = int(rng.integers(8, 60))
    hi = int(rng.integers(lo + 40, 521))
    nreads = int(rng.integers(4200, 6000)) if rng.random() < 0.3 else \
        int(rng.integers(50, 700))
    reads = []
    for i in range(nreads):
        m = int(rng.integers(lo, hi + 1))
        p = int(rng.integers(0, len(tis) - m))
        q = tis[p:p + m].copy()
        q[q == H.SEPARATOR] = rng.integers(0, 4)
        if rng.random() < 0.9:
            q[q >= H.WILDCARD] = rng.integers(0, 4)
        for e in range(int(rng.integers(0, m * pct // 100 + 2))):
            kind, x = int(rng.integers(0, 3)), int(rng.integers(0, len(q)))
            if kind == 0 or not doedist:
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4
            elif kind == 1 and len(q) > 8:
                q = np.delete(q, x)
            elif len(q) < 520:
                q = np.insert(q, x, rng.integers(0, 4))
        reads.append(q.astype(np.uint8))
    hq = H.Queries.from_list(reads)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    what = "round %d: %s %dp m %d..%d %s %d reads" % (
        rnd, "edist" if doedist else "hamming", pct, lo, hi,
        "wide" if wide else "narrow", nreads)
    want, oerr = None, None
    try:
        want = H.oracle_approx(host, hq, doedist, pct, True)
    except H.OracleNotCovered:
        declined += 1
        continue
    except H.OracleError as e:      # the reference's error: same message,
        want, oerr = e.partial, str(e)   # same matches in front of it
    try:
        got = V.findapproxcompletematches(gi, gq, doedist, pct, True).fetch()
        gerr = None
    except V.VsaError as e:
        if e.code == V.NOT_COVERED:
            print(what + ": declined by the engine (%s)" % e, flush=True)
            declined += 1
            continue
        got, gerr = e.partial.fetch(), str(e)
    if (oerr is None) != (gerr is None) or (oerr and oerr not in gerr) or \
            not np.array_equal(got, want):
        print("MISMATCH " + what + ": gpu %d oracle %d, errors %r / %r"
              % (len(got), len(want), gerr, oerr), flush=True)
        sys.exit(1)
    checked += 1
    errors += oerr is not None
    ks = hq.length * pct // 100
    print(what + " ok: %d with threshold 0, %d above, %d matches%s, %.0f s" % (
        (ks == 0).sum(), (ks > 0).sum(), len(want),
        ", stopped by the reference's error" if oerr else "",
        time.time() - t0), flush=True)
print("all %d lists equal the oracle's (%d of them up to the reference's "
      "error; %d configurations outside restatement or engine)"
      % (checked, errors, declined))
