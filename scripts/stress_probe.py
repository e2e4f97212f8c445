#!/usr/bin/env python3
"""Randomised GPU-vs-oracle comparison of the query path beyond the fixed
test suite: texts with planted repeats, low-complexity stretches, wildcards
and several sequences; reads of mixed lengths with substitutions, indels,
wildcards, duplicated and overlapping reads; random least lengths.  Every
list (-complete, MEM, -mum cand, -mum) must equal the oracle's, order
included (MEM: under -qspeedup 0 and under the default -qspeedup 2; every
third round takes a small prefixlength and more low-complexity stretches so
that buckets hold more than 255 suffixes and stitab1 saturates).
VSA_STRESS_TABLES=1 also compares the tables the GPU builder wrote with the
CPU restatement's (with VSA_FORCE_WIDE=1: the 64-bit builder).
usage: stress_probe.py [ROUNDS] [SEED]"""
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
    nseq = int(rng.integers(1, 4))
    seqs = []
    for s in range(nseq):
        n = int(rng.integers(20000, 120000))
        t = rng.integers(0, 4, n).astype(np.uint8)
        for r in range(int(rng.integers(0, 12))):      # planted repeats
            ln = int(rng.integers(20, 600))
            a, b = (int(x) for x in rng.integers(0, n - ln, 2))
            u = t[a:a + ln].copy()
            for e in range(int(rng.integers(0, 4))):
                u[int(rng.integers(0, ln))] = rng.integers(0, 4)
            t[b:b + ln] = u
        for r in range(int(rng.integers(0, 4 if rnd % 3 else 12))):
            ln = int(rng.integers(10, 300 if rnd % 3 else 900))  # low complexity
            a = int(rng.integers(0, n - ln))
            unit = rng.integers(0, 4, int(rng.integers(1, 5)))
            t[a:a + ln] = np.resize(unit, ln)
        if rng.random() < 0.5:
            t[rng.random(n) < 0.0005] = H.WILDCARD
        seqs.append(t)
    tis = np.concatenate([np.concatenate([s, [H.SEPARATOR]])
                          for s in seqs])[:-1].astype(np.uint8)
    gi = V.Index.build(tis, 4, 0 if rnd % 3 else int(rng.integers(1, 6)))
    tb = gi.download()
    if os.environ.get("VSA_STRESS_TABLES") == "1":
        # the builder itself (e.g. under VSA_FORCE_WIDE=1) against the CPU
        # restatement of mkvtree's tables
        want = H.oracle_build_index(tis, 4, gi.info().prefixlength)
        for k in ("suf", "lcp", "llv", "bck", "bwt"):
            assert np.array_equal(tb[k].astype(np.uint64),
                                  getattr(want, k).astype(np.uint64)), (rnd, k)
    host = H.Index(len(tis), gi.info().prefixlength, 4, tb["tis"], tb["suf"],
                   tb["lcp"], tb["llv"], tb["bck"], tb["bwt"],
                   H.sti1_from_tables(tb["suf"], tb["lcp"],
                                      gi.info().prefixlength))
    reads = []
    uniform = rng.random() < 0.4
    m0 = int(rng.integers(30, 160))
    for i in range(int(rng.integers(200, 1500))):
        m = m0 if uniform else int(rng.integers(12, 300))
        p = int(rng.integers(0, len(tis) - m))
        q = tis[p:p + m].copy()
        q[q == H.SEPARATOR] = rng.integers(0, 4)
        k = rng.random()
        if k < 0.3:
            for e in range(int(rng.integers(1, 4))):
                x = int(rng.integers(0, len(q)))
                q[x] = (q[x] + 1 + rng.integers(0, 3)) % 4 if q[x] < 4 else 0
        elif k < 0.4 and not uniform:
            x = int(rng.integers(0, len(q)))
            q = np.delete(q, x) if rng.random() < 0.5 else np.insert(
                q, x, rng.integers(0, 4))
        elif k < 0.45:
            q[int(rng.integers(0, len(q)))] = H.WILDCARD
        elif k < 0.5 and reads:
            q = reads[int(rng.integers(0, len(reads)))]   # duplicate read
            if uniform and len(q) != m0:
                q = tis[p:p + m0].copy()
        reads.append(q.astype(np.uint8))
    if uniform:
        reads = [r for r in reads if len(r) == m0 and
                 not (r == H.SEPARATOR).any()]
    hq = H.Queries.from_list(reads)
    gq = V.Queries.from_host(hq.symbols, hq.start, hq.length)
    # batches of one length also as rows of two bits per symbol (reads with a
    # wildcard on the side list)
    forms = [("bytes", gq)]
    if uniform and hq.nq > 0:
        # (a Multiseq: a separator between two reads, stride m + 1)
        forms.append(("rows", V.Queries.from_host_packed(hq.symbols, m0,
                                                         stride=m0 + 1)))
    pl = gi.info().prefixlength
    lo = max(pl, 6)     # below that every offset matches thousands of suffixes
    for L in sorted({lo, int(rng.integers(lo, 40)), 20}):
        for kw, sp in (({}, 0), ({}, 2), (dict(mum=True, cand=True), 2),
                       (dict(mum=True), 0)):
            want = H.oracle_querymatches(host, hq, L, speedup=sp, **kw)
            for form, q_ in forms:
                got = V.findquerymatches(gi, q_, L, speedup=sp, **kw).fetch()
                if not np.array_equal(got, want):
                    print("MISMATCH round %d L %d %s sp %d (%s): gpu %d "
                          "oracle %d" % (rnd, L, kw, sp, form, len(got),
                                         len(want)), flush=True)
                    sys.exit(1)
                checked += 1
    if hq.length.min() >= pl:
        want = H.oracle_complete(host, hq)
        for form, q_ in forms:
            got = V.findcompletematches(gi, q_).fetch()
            assert np.array_equal(got, want), (rnd, form)
            checked += 1
    print("round %d ok: %d sequences, %d bp, %d reads (%s), %.0f s" % (
        rnd, nseq, len(tis), len(reads), "uniform" if uniform else "ragged",
        time.time() - t0), flush=True)
print("all %d lists equal the oracle's" % checked)
