#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native Vmengine query path.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric, configs[2]): a 3 Gbp synthetic DNA index
resident in HBM, 10 M queries of 100 bp per GPU, `vmatch -mum -l 20 -q`
semantics (maximal unique matches of every query suffix, global uniqueness
filter over the queries).  One "step" = one pass of that hot path over the
whole query batch; inputs (index tables, query symbols) are resident in HBM
before the timed region, the match list ends up in HBM.

N > 1 (launched by torch.distributed.run, one rank per GPU): the index is
replicated, every rank owns 10 M queries of a 10*N M query job (weak
scaling).  Phase 1 (search) needs no communication; the MUM uniqueness filter
(kurtz/cleanMUMcand.c of the reference) is one global step: candidates are
range-partitioned by database position over the ranks (RCCL all-to-all), every
rank filters its range with the carry of the lower ranges; one RCCL all-reduce
sums the match counters.

Prints ONE JSON line on rank 0.  Extra objects: "roofline" (dominant kernel
k_query_search, algorithmic bytes from the instrumented CPU restatement) and
"cpu_baseline" (rank 0, N = 1: the CPU oracle = port of the reference's
default algorithm, one core, on a bounded sample of the same batch).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome", type=float, default=3e9,
                    help="index length in bp (default 3 Gbp)")
    ap.add_argument("--queries", type=float, default=1e7,
                    help="queries per GPU (default 10 M)")
    ap.add_argument("--qlen", type=int, default=100)
    ap.add_argument("--minlen", type=int, default=20, help="vmatch -l")
    ap.add_argument("--cpu-sample", type=int, default=1000000,
                    help="queries timed on the CPU baseline (0 = skip)")
    ap.add_argument("--cpu-cores", type=int, default=1)
    # rehearsal of the N > 1 path on a box with ONE GPU: every rank uses
    # device 0 and the collectives run over gloo on host copies
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    # one rank, but through the N > 1 code path with the real backend (RCCL):
    # process group, all-to-all / all-gather / all-reduce on device tensors
    ap.add_argument("--force-distributed", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(H, host_index, sample, minlen, w):
    """bytes the search has to touch per query, counted by the instrumented
    CPU restatement running the GPU's algorithm (bucket + binary search per
    query suffix): per search 2w (bck pair) + probes*w
    (suf) + compared symbols (tis) + lcp entries; per query its m symbols;
    per reported match w (suf) + 1 (left symbol) + 32 (record written)."""
    H.oracle_counters(reset=True)
    H.oracle_querymatches(host_index, sample, minlen, mum=True, cand=True,
                          speedup=0)
    c = H.oracle_counters(reset=True)
    nq = sample.nq
    total = (c["bckreads"] * 2 * w + c["sufprobes"] * w + c["charcomp"] +
             c["lcpreads"] + int(sample.length.sum()) +
             c["emitted"] * (w + 1 + 32))
    return total / nq, c


def main():
    a = parse()
    # stdout carries the one JSON line and nothing else: libraries that
    # print there (RCCL's version banner) go to stderr
    sys.stdout.flush()
    jsonfd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            log("bench.py: --gpus %d needs torch.distributed.run; running the "
                "single-process case" % a.gpus)
        a.gpus = world
    dev = 0 if a.rehearse_on_one_gpu else local_rank

    torch = dist = S = None
    distributed = world > 1 or a.force_distributed
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # torch first: its wheel bundles a HIP runtime, and the library must
        # bind to the one runtime of the process (vstree_amd/__init__.py)
        import torch
        import torch.distributed as dist
    import vstree_amd as V
    if distributed:
        from vstree_amd import sharding as S
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo" if a.rehearse_on_one_gpu else "nccl",
                                rank=rank, world_size=world)

    n, nq, m, L = int(a.genome), int(a.queries), a.qlen, a.minlen

    # ---- setup (untimed): genome, index, this rank's queries, all in HBM --
    t0 = time.time()
    dg = V.device_malloc(n + 64, dev)
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, dev))
    index = V.Index.build_device(dg, n, 4, 0, dev)
    t_index = time.time() - t0
    info = index.info()
    # rank r owns queries [r*nq, (r+1)*nq) of the world*nq query job
    pos, sub, step = V.synth_query_plan(n, nq * world, m)
    sl = slice(rank * nq, (rank + 1) * nq)
    pos, sub, step = (np.ascontiguousarray(x[sl]) for x in (pos, sub, step))
    dq = V.device_malloc(nq * m + 64, dev)
    V._check(V.lib.vsa_synth_queries_device(dg, n, pos.ctypes.data,
                                            sub.ctypes.data,
                                            step.ctypes.data, nq, m, dq, dev))
    queries = V.Queries.from_device(dq, nq, m, dev)
    queries.set_offset(rank * nq)
    V.device_free(dq, dev)
    V.device_free(dg, dev)
    if rank == 0:
        log("setup: index %d bp (prefixlength %d, %.1f GB in HBM) built in "
            "%.1fs, %d queries/GPU" % (n, info.prefixlength,
                                       info.device_bytes / 1e9, t_index, nq))

    lenbits = 0
    if distributed:
        # the pairs of all ranks must be laid out alike: length bits of the
        # longest query of the job (one number, agreed on once per batch)
        t = torch.tensor([m], dtype=torch.int64,
                         device="cpu" if a.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        lenbits = max(1, int(t.item()).bit_length())

    def sync():
        V.device_synchronize(dev)
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    kernel_ms, totals = [], None

    def one_step():
        """the hot path over the whole batch; returns (count, sumlength,
        searches, candidates) of the job"""
        nonlocal totals
        if not distributed:
            r = V.findquerymatches(index, queries, L, mum=True)
            s = r.stats()
            kernel_ms.append(s.search_kernel_ms)
            totals = (s.count, s.sumlength, s.searches, s.candidates,
                      s.kernel_searches)
            r.close()
            return
        # phase 1: candidates of this rank's queries (no communication), in
        # the order the kernel left them, grouped by the rank that filters
        # their range of the index
        # (as pairs of sort key and value, 16 bytes each: half the exchange)
        r = V.findmumcandidates_packed(index, queries, L, lenbits)
        s = r.stats()
        kernel_ms.append(s.search_kernel_ms)
        mine = torch.empty(max(r.count, 1) * 2, dtype=torch.int64,
                           device="cuda")[:r.count * 2]
        send, top = r.partition(world, n, C.c_void_p(mine.data_ptr()))
        r.close()
        cdev = "cuda"
        if a.rehearse_on_one_gpu:
            mine, cdev = mine.cpu(), "cpu"

        # phase 2: the one exchange step (RCCL all-to-all) -- every rank runs
        # the uniqueness filter on its range with the carry of the lower ones
        def filter_fn(part, carry):
            part = part.cuda().contiguous()
            res = V.mumuniqueinquery_range_packed(
                C.c_void_p(part.data_ptr()), part.numel() // 2, lenbits, n,
                carry, dev)
            st = res.stats()
            res.close()
            return st.count, st.sumlength

        nmum, sumlen, ncand, searches, ksearches = \
            S.partitioned_mum_filter_presorted(
                dist, torch, mine, send, top, cdev, filter_fn, words=2,
                extra=[s.searches, s.kernel_searches])
        totals = (nmum, sumlen, searches, ncand, ksearches)

    for _ in range(a.warmup):
        one_step()
    kernel_ms.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        e = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if a.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    count, sumlength, searches, candidates, kernel_searches = totals
    total_queries = nq * world
    qps = total_queries * a.steps / elapsed
    kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")

    out = {
        "metric": "queries/sec (100 bp queries, vmatch -mum -l 20 "
                  "semantics, 3 Gbp ESA index resident in HBM)",
        "value": qps, "unit": "queries/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "3 Gbp synthetic DNA index, 10 M x 100 bp "
                               "queries per GPU, -mum -l 20 "
                               "(BASELINE.json configs[2]; configs[3] "
                               "sharding for N > 1)",
                   "index_bp": n, "queries_per_gpu": nq, "query_len": m,
                   "minlen": L, "prefixlength": info.prefixlength,
                   "index_bytes_hbm": info.device_bytes,
                   "index_build_s": round(t_index, 2),
                   "parallelism": "index replicated, queries sharded x%d"
                                  % world},
        "gbp_matched_per_s": sumlength * a.steps / elapsed / 1e9
        if world == 1 else sumlength / (elapsed / a.steps) / 1e9,
        "matches": count, "candidates": candidates,
        "query_suffix_searches": searches,
    }

    if rank == 0:
        import helpers as H  # test infrastructure: the CPU oracle
        w = info.device_integersize // 8
        t = index.download()
        host = H.Index(n, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                       t["llv"], t["bck"], t["bwt"], None)
        qsym = np.zeros(min(nq, max(a.cpu_sample, 20000)) * m, np.uint8)
        g = t["tis"]
        ns = qsym.shape[0] // m
        for i in range(ns):   # the same queries the GPU has, from the plan
            p = int(pos[i])
            qsym[i * m:(i + 1) * m] = g[p:p + m]
            if sub[i] != V.NO_SUBST:
                k = i * m + int(sub[i])
                qsym[k] = (qsym[k] + step[i]) & 3
        small = H.Queries.uniform(qsym[:20000 * m], m)
        bytes_per_query, counters = algorithmic_bytes(H, host, small, L, w)
        # SURVEY 8d / BASELINE.md: bytes of the reference's per-suffix
        # algorithm (one bucket lookup + binary search for EVERY query
        # suffix) x queries per launch
        alg0_bytes_launch = bytes_per_query * nq
        full_searches = nq * (m - L + 1)
        # the dominant kernel only runs the searches the anchor pass and the
        # work plan left over; price it on that work, not on work they
        # proved unnecessary
        main_searches = kernel_searches
        executed_bytes_launch = alg0_bytes_launch * (
            (main_searches / world) / full_searches)
        achieved = executed_bytes_launch / (kms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if (tj.get("index_bp") == n and tj.get("queries") == nq):
                traffic = tj.get("hbm_bytes_per_launch")
        out["roofline"] = {
            "kernel": "k_query_search<uint32_t, MUM, deep, 256>",
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "kernel_ms": kms,
            "algorithmic_bytes_per_launch": executed_bytes_launch,
            "searches_per_launch": main_searches / world,
            "algorithmic_bytes_per_query_all_suffixes": bytes_per_query,
            "achieved_if_priced_on_all_suffixes":
                alg0_bytes_launch / (kms * 1e-3) / 1e9,
            "note": "algorithmic bytes = SURVEY 8d formula counted by the "
                    "instrumented CPU restatement (7.1 kB per 100 bp query "
                    "for all 81 suffixes) scaled to the %.1f%% of the "
                    "suffix searches this kernel executes after the first "
                    "pass and the work plan; the path is random 8/16-byte "
                    "reads, one 64-byte sector each (traffic/algorithmic "
                    "~ %.2f), see DESIGN.md"
                    % (100.0 * main_searches / world / full_searches,
                       (traffic or 0) / executed_bytes_launch)}
        if world == 1 and a.cpu_sample > 0:
            host.sti1 = index.make_sti1()
            sample = H.Queries.uniform(qsym[:a.cpu_sample * m], m)
            t0 = time.perf_counter()
            ref = H.oracle_querymatches(host, sample, L, mum=True, speedup=2)
            dt = time.perf_counter() - t0
            # the same sample as a batch of its own on the GPU: identical list
            gsample = V.Queries.from_host(sample.symbols, sample.start,
                                          sample.length, dev)
            gres = V.findquerymatches(index, gsample, L, mum=True)
            same = bool(np.array_equal(gres.fetch(), ref))
            gres.close()
            if not same:
                raise RuntimeError("bench.py: GPU and CPU oracle disagree on "
                                   "the %d-query sample" % sample.nq)
            out["cpu_baseline"] = {
                "value": sample.nq / dt, "unit": "queries/s", "cores": 1,
                "kind": "port",
                "sample": "first %d queries of the same batch, same 3 Gbp "
                          "index (32-bit tables), oracle/vsoracle.c "
                          "algorithm 2 = the reference's default -qspeedup 2 "
                          "incl. the MUM filter, %.1f s, %d MUMs"
                          % (sample.nq, dt, len(ref)),
                "gpu_list_equal_on_sample": same}
            # the real reference program, timed once on a GPU box on the
            # same index and sample size (scripts/cpu_reference_probe.py)
            rpath = os.path.join(ROOT, "profiles", "cpu_reference.json")
            if os.path.exists(rpath) and n == 3000000000 and m == 100:
                with open(rpath) as f:
                    rj = json.load(f)
                run = rj["runs"].get("-mum -l %d" % L)
                if run:
                    out["cpu_baseline"]["reference_vmatch"] = {
                        "value": run["queries_per_s"], "unit": "queries/s",
                        "cores": 1, "kind": "reference",
                        "sample": "%d queries, %s" % (rj["queries"],
                                                      rj["source"])}
            out["speedup_vs_cpu_1core"] = qps / (sample.nq / dt)
        sys.stdout.flush()
        os.write(jsonfd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
