#!/usr/bin/env python3
"""Does the uniqueness filter of batch i hide under the search of batch i+1?

The -mum step is search (first pass, plan, K2: random 64-byte sectors, 2.8 ms)
followed by the filter (radix sort + tile passes: short streaming kernels,
0.85 ms).  This probe runs the two stages of consecutive batches from two host
threads on two HIP streams (search: the index's stream; filter: the null
stream) and compares steps/s with the same calls made one after the other.

    python scripts/overlap_probe.py [--genome 3e9] [--steps 20]
"""
import argparse
import ctypes as C
import os
import queue
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=float, default=3e9)
    ap.add_argument("--queries", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=20)
    a = ap.parse_args()
    import torch
    import vstree_amd as V
    dev = 0
    torch.cuda.set_device(dev)
    n, nq, m, L = int(a.genome), int(a.queries), 100, 20
    dg = V.device_malloc(n + 64, dev)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, dev))
    index = V.Index.build_device(dg, n, 4, 0, dev)
    pos, sub, step = V.synth_query_plan(n, nq, m)
    dq = V.device_malloc(nq * m + 64, dev)
    V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data,
                                            sub.ctypes.data, step.ctypes.data,
                                            nq, m, dq, dev))
    queries = V.Queries.from_device(dq, nq, m, dev)
    V.device_free(dq, dev)
    V.device_free(dg, dev)
    lenbits = max(1, m.bit_length())

    def search():
        r = V.findmumcandidates_packed(index, queries, L, lenbits)
        rows = torch.empty(max(r.count, 1) * 2, dtype=torch.int64,
                           device="cuda")[:r.count * 2]
        r.partition(1, n, C.c_void_p(rows.data_ptr()))
        r.close()
        return rows

    def filt(rows):
        res = V.mumuniqueinquery_range_packed(
            C.c_void_p(rows.data_ptr()), rows.numel() // 2, lenbits, n, 0, dev)
        st = res.stats()
        res.close()
        return st.count, st.sumlength

    def whole():
        r = V.findquerymatches(index, queries, L, mum=True)
        s = r.stats()
        r.close()
        return s.count, s.sumlength

    for _ in range(3):
        want = filt(search())
    assert want == whole(), (want, whole())
    out = {}
    for name in ("one call", "two calls in turn", "two threads"):
        V.device_synchronize(dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if name == "one call":
            for _ in range(a.steps):
                got = whole()
        elif name == "two calls in turn":
            for _ in range(a.steps):
                got = filt(search())
        else:
            q = queue.Queue(maxsize=2)
            results = []

            def consumer():
                while True:
                    rows = q.get()
                    if rows is None:
                        return
                    results.append(filt(rows))

            t = threading.Thread(target=consumer)
            t.start()
            for _ in range(a.steps):
                q.put(search())
            q.put(None)
            t.join()
            got = results[-1]
            assert len(results) == a.steps
        V.device_synchronize(dev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        assert got == want, (name, got, want)
        out[name] = ms
        print("%-20s %.3f ms per batch" % (name, ms), flush=True)
    print("mums %d sumlength %d" % want)


if __name__ == "__main__":
    main()
