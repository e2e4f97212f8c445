#!/usr/bin/env python3
"""Times index construction and the search kernels at a given scale on one GPU.
usage: scale_probe.py N NQ [M] [L]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vstree_amd as V  # noqa: E402


def main():
    n = int(float(sys.argv[1]))
    nq = int(float(sys.argv[2]))
    m = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    L = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    t0 = time.time()
    dg = V.device_malloc(n + 64)
    V._check(V.lib.vsa_synth_genome_device(42, n, dg, 0))
    t1 = time.time()
    print("genome on device: %.2fs" % (t1 - t0), flush=True)
    idx = V.Index.build_device(dg, n, 4, 0)
    t2 = time.time()
    info = idx.info()
    print("index build n=%d pl=%d: %.2fs, device bytes %.2f GB, nllv=%d" % (
        n, info.prefixlength, t2 - t1, info.device_bytes / 1e9,
        info.largelcpvalues), flush=True)
    pos, sub, step = V.synth_query_plan(n, nq, m)
    t3 = time.time()
    dq = V.device_malloc(nq * m + 64)
    V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            nq, m, dq, 0))
    q = V.Queries.from_device(dq, nq, m)
    t4 = time.time()
    print("queries: plan %.2fs device %.2fs" % (t3 - t2, t4 - t3), flush=True)
    for rep in range(2):
        t = time.time()
        r = V.findcompletematches(idx, q)
        s = r.stats()
        print("complete: wall %.3fs kernel %.2fms total %.2fms count %d -> "
              "%.2f Mq/s" % (time.time() - t, s.search_kernel_ms,
                             s.total_device_ms, s.count,
                             nq / s.total_device_ms / 1e3), flush=True)
        r.close()
    for name, kw in (("mumcand", dict(mum=True, cand=True)),
                     ("mum", dict(mum=True)), ("mem", {})):
        for rep in range(2):
            t = time.time()
            r = V.findquerymatches(idx, q, L, **kw)
            s = r.stats()
            print("%s: wall %.3fs kernel %.2fms total %.2fms count %d cand %d"
                  " searches %d -> %.3f Mq/s" % (
                      name, time.time() - t, s.search_kernel_ms,
                      s.total_device_ms, s.count, s.candidates, s.searches,
                      nq / s.total_device_ms / 1e3), flush=True)
            r.close()
    print("stream read GB/s:", V.measure_stream_read(4 << 30))
    os.system("free -g | head -2; nproc")


if __name__ == "__main__":
    main()
