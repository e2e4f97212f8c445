#!/usr/bin/env python3
"""BASELINE.json configs[4] at scale: approximate complete matches
(vmatch -complete -e K | -h K) of NQ synthetic M-bp reads against an N-bp index
on one GPU; a sample of the reads is checked against the CPU oracle.
usage: approx_probe.py N NQ [M] [K] [e|h] [SAMPLE]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vstree_amd as V  # noqa: E402


def main():
    n = int(float(sys.argv[1]))
    nq = int(float(sys.argv[2]))
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    doedist = (sys.argv[5] if len(sys.argv) > 5 else "e") == "e"
    sample = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    t0 = time.time()
    dg = V.device_malloc(n + 64)
    V._check(V.lib.vsa_synth_genome_device(42, n, dg, 0))
    idx = V.Index.build_device(dg, n, 4, 0)
    info = idx.info()
    print("index n=%d pl=%d built in %.1fs" % (n, info.prefixlength,
                                               time.time() - t0), flush=True)
    pos, sub, step = V.synth_query_plan(n, nq, m)
    dq = V.device_malloc(nq * m + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            nq, m, dq, 0))
    q = V.Queries.from_device(dq, nq, m)
    for rep in range(3):
        t = time.time()
        r = V.findapproxcompletematches(idx, q, doedist, k)
        s = r.stats()
        print("approx -%s %d: wall %.3fs piece search %.2f ms total %.2f ms; "
              "pieces %d hits %d matches %d -> %.2f Mq/s" % (
                  "e" if doedist else "h", k, time.time() - t,
                  s.search_kernel_ms, s.total_device_ms, s.searches,
                  s.candidates, s.count, nq / s.total_device_ms / 1e3),
              flush=True)
        if rep < 2:
            r.close()
    if sample > 0:
        import helpers as H
        t = time.time()
        tb = idx.download(with_bwt=False)
        host = H.Index(n, info.prefixlength, 4, tb["tis"], tb["suf"],
                       tb["lcp"], tb["llv"], tb["bck"], None, None)
        qb = V.synth_queries(tb["tis"], sample, m)
        hq = H.Queries.uniform(qb, m)
        t1 = time.time()
        want = H.oracle_approx(host, hq, doedist, k)
        dt = time.time() - t1
        got = r.fetch()
        got = got[got["queryseq"] < sample]
        assert np.array_equal(got, want), "GPU != oracle on the sample"
        print("first %d reads == CPU oracle (%d matches); oracle %.2f s = "
              "%.1f reads/s on one core" % (sample, len(want), dt,
                                            sample / dt), flush=True)


if __name__ == "__main__":
    main()
