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
sums the match counters.  `--mode selfmum` runs the other sharded path of
SURVEY 8e instead: the self-index MUM scan split into suffix-array ranges.

Prints ONE JSON line on rank 0.  Everything in it is measured in this run:
  roofline           the dominant kernel (k_query_search_planned), HIP-event time
  roofline_families  the same for the other kernel families of the path --
                     first pass, -complete (K1), MEM, -complete -e 2 (piece
                     search and banded alignment), the self-index scan (K3)
  cpu_baseline       the reference program (oracle/_ref/vmatch_ref, built from
                     the reference's sources, index files written by
                     vsa_mkvtree) on P = all host cores given to the box
                     (P processes over 1/P of the sample each, wall = slowest)
                     and on one core; the CPU restatement (oracle) on one core
`--quick` leaves the extra families and the reference program out (A/B runs,
profiling passes).
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def human(x, unit=""):
    """3e9 -> '3 G', 1e7 -> '10 M', 150000 -> '150 k' (exact sizes stand next
    to the label in the config, so rounding here hides nothing)"""
    for f, s in ((1e9, "G"), (1e6, "M"), (1e3, "k")):
        if x >= f:
            v = x / f
            return ("%d %s%s" % (round(v), s, unit) if abs(v - round(v)) < 1e-9
                    else "%.3g %s%s" % (v, s, unit))
    return "%d %s" % (x, unit)


def workload_label(n, nq, m, L, mode, tail=""):
    """what this run searched, from the sizes it ran with -- a line cannot
    claim another configuration than the one its arguments gave it"""
    return ("%s synthetic DNA index, %s x %d bp queries per GPU, %s -l %d%s"
            % (human(n, "bp"), human(nq), m, mode, L, tail))


def baseline_config(n, nq, m, L):
    """which BASELINE.json config these sizes are, if any"""
    if (n, nq, m, L) == (3000000000, 10000000, 100, 20):
        return "BASELINE.json configs[2] (configs[3] sharding for N > 1)"
    return "not a BASELINE.json configuration (sizes given on the command line)"


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
    ap.add_argument("--mode", choices=("mum", "selfmum"), default="mum")
    ap.add_argument("--cpu-sample", type=int, default=500000,
                    help="queries timed on one CPU core (0 = no CPU baseline)")
    ap.add_argument("--ref-sample", type=int, default=2000000,
                    help="queries given to the P reference processes together")
    ap.add_argument("--reads", choices=("packed", "bytes"), default="packed",
                    help="how the query batch lies in HBM: the reference's "
                         "Multiseq bytes, or two bits per symbol "
                         "(vsa_pack_reads)")
    ap.add_argument("--quick", action="store_true",
                    help="headline step and its roofline only")
    ap.add_argument("--no-reference", action="store_true",
                    help="do not run oracle/_ref/vmatch_ref")
    ap.add_argument("--workdir", default="/dev/shm")
    # rehearsal of the N > 1 path on a box with ONE GPU: every rank uses
    # device 0 and the collectives run over gloo on host copies
    ap.add_argument("--rehearse-on-one-gpu", action="store_true")
    # one rank, but through the N > 1 code path with the real backend (RCCL):
    # process group, all-to-all / all-gather / all-reduce on device tensors
    ap.add_argument("--force-distributed", action="store_true")
    ap.add_argument("--meta-on-host", action="store_true",
                    help="N > 1 form: split sizes through the host "
                         "(vsa_result_partition_own) instead of staying on "
                         "the device")
    # N > 1: `torch` = one process per GPU (torch.distributed, RCCL), what the
    # driver launches; `c` = ONE process that drives all GPUs through
    # libvstree_amd_multi.so (vsa_multi_findmatches: a host thread per GPU,
    # peer copies for the -mum exchange, ncclAllReduce of the counters) -- the
    # path the drop-in binary takes (VMATCH_GPUS=N)
    ap.add_argument("--path", choices=("torch", "c"), default=None,
                    help="N > 1 (default c): c = one process, a host thread "
                         "per GPU (libvstree_amd_multi.so); torch = one "
                         "process per GPU over torch.distributed")
    ap.add_argument("--host", action="store_true",
                    help="--path c: time the host-memory entries instead of "
                         "the device-resident one: vsa_multi_pipeline_* "
                         "(packed rows in page-locked slots)")
    ap.add_argument("--compat", action="store_true",
                    help="--path c --host: vsa_multi_findmatches (a Multiseq "
                         "in pageable memory, one list back) instead of the "
                         "pipelines")
    # --path c on a box with one GPU: N replicas of the index on device 0
    ap.add_argument("--replicas-on-one-gpu", action="store_true")
    # (tests only: the C path gives up after its setup, so that the fallback
    # to the torch form can be rehearsed on a box with one GPU)
    ap.add_argument("--test-fail-c-path", action="store_true",
                    help=argparse.SUPPRESS)
    return ap.parse_args()


# what the kernels of a default run are compiled from
KERNEL_SOURCES = ("search_query.inc", "esa_device.hpp", "mum_workplan.inc",
                  "mem_workplan.inc", "search_complete.inc",
                  "selfmum_scan.inc", "mum_filter.inc", "esa_search.hip",
                  "selfmum_search.hip", "search_common.hip")


def kernel_source_hash():
    """identifies the kernel sources a PMC profile was taken with"""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "vstree_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def count_bytes(H, run, nq, w, symbols):
    """bytes the reference's algorithm touches per query, counted by the
    instrumented CPU restatement: per search 2w (bck
    pair) + probes*w (suf) + compared symbols (tis) + lcp entries; per query
    its symbols; per reported match w (suf) + 1 (left symbol) + 32 (record)."""
    H.oracle_counters(reset=True)
    run()
    c = H.oracle_counters(reset=True)
    total = (c["bckreads"] * 2 * w + c["sufprobes"] * w + c["charcomp"] +
             c["lcpreads"] + symbols + c["emitted"] * (w + 1 + 32))
    return total / nq, c


_TRAFFIC = None


def pmc_traffic(n, nq):
    """HBM bytes per launch from the committed PMC passes (scripts/
    pmc_passes.sh + pmc_summary.py), quoted only if they were taken at this
    size with these very kernel sources -- a profile of other sources says
    nothing about this run"""
    global _TRAFFIC
    if _TRAFFIC is None:
        _TRAFFIC = {}
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            if (tj.get("index_bp") == n and tj.get("queries") == nq and
                    tj.get("kernel_source_sha16") == kernel_source_hash()):
                _TRAFFIC = tj
    return _TRAFFIC


def family(name, mode, ms, nbytes, note, traffic_key=None, **more):
    ach = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    d = {"kernel": name, "mode": mode, "bound": "hbm", "achieved": ach,
         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
         "traffic": None, "kernel_ms": ms,
         "algorithmic_bytes_per_launch": nbytes, "note": note}
    fam = (_TRAFFIC or {}).get("families", {}).get(traffic_key)
    if fam:
        d["traffic"] = fam["hbm_bytes_per_launch"]
        d["traffic_source"] = _TRAFFIC.get("source")
    d.update(more)
    return d


def counted_search_bytes(a, V, H, index, queries, host, small, m, L, w):
    """algorithmic bytes per search, counted on exactly the searches the
    dominant kernel runs: the plans of the first queries as the engine made
    them (one more call with VSA_DEBUG_PLANFILE, outside the timed region),
    every planned (query, offset) searched by the instrumented restatement.
    -> (bytes per search, searches counted) or (None, None)"""
    pf = os.path.join(a.workdir, "vsa_plans_%d.bin" % os.getpid())
    os.environ["VSA_DEBUG_PLANFILE"] = pf
    try:
        V.findquerymatches(index, queries, L, mum=True).close()
    finally:
        del os.environ["VSA_DEBUG_PLANFILE"]
    if not os.path.exists(pf):
        return None, None
    pl = np.fromfile(pf, dtype=np.uint32).reshape(-1, 5)[:small.nq]
    os.unlink(pf)
    qi, offs = [], []
    for k in range(1, 5):
        first, ln = pl[:, k] & 0xFFFF, pl[:, k] >> 16
        ln = np.where(pl[:, 0] != 0, ln, 0)
        rep = np.repeat(np.arange(len(pl)), ln)
        within = np.arange(ln.sum()) - np.repeat(np.cumsum(ln) - ln, ln)
        qi.append(rep)
        offs.append(first[rep] + within)
    qi, offs = np.concatenate(qi), np.concatenate(offs)
    if not len(qi):
        return None, None
    lens = (m - offs).astype(np.uint64)
    starts = (qi * m + offs).astype(np.uint64)
    sufq = H.Queries(small.symbols, starts, lens)
    cb, _ = count_bytes(H, lambda: H.oracle_complete(host, sufq), len(qi), w,
                        int(lens.sum()))
    # (count_bytes charges every search its whole suffix as "query symbols"; a
    # search reads the ones it compares, which `charcomp` holds already)
    return cb - float(lens.sum()) / len(qi), int(len(qi))


def write_fasta(path, header, symbols, width=1 << 20):
    letters = np.frombuffer(b"acgt", np.uint8)
    with open(path, "wb") as f:
        f.write(header)
        n = len(symbols)
        for i in range(0, n, width << 6):
            chunk = letters[symbols[i:i + (width << 6)]]
            k = (len(chunk) // width) * width
            if k:
                rows = chunk[:k].reshape(-1, width)
                nl = np.full((rows.shape[0], 1), 10, np.uint8)
                f.write(np.hstack([rows, nl]).tobytes())
            if k < len(chunk):
                f.write(chunk[k:].tobytes() + b"\n")


def write_queries(path, qsym, m, first, count):
    """count reads as FASTA, one line each: '>q' + 9 digits (the global
    number), the symbols.  Built as one byte matrix: a Python loop over 10 M
    reads took longer than the reference needs to match them."""
    letters = np.frombuffer(b"acgt", np.uint8)
    rows = np.empty((count, 12 + m + 1), np.uint8)
    rows[:, 0] = ord(">")
    rows[:, 1] = ord("q")
    ids = np.arange(first, first + count, dtype=np.int64)
    for k in range(9):
        rows[:, 10 - k] = (ids // 10 ** k) % 10 + 48
    rows[:, 11] = 10
    rows[:, 12:12 + m] = letters[qsym[first * m:(first + count) * m]
                                 ].reshape(count, m)
    rows[:, 12 + m] = 10
    with open(path, "wb") as f:
        f.write(rows.tobytes())


def reference_baseline(a, V, H, genome, qsym, m, L, ncores):
    """oracle/_ref/vmatch_ref on this box's host cores.  Returns a dict or
    None (binary missing / no room for the 64-bit index files)."""
    n = len(genome)
    if a.no_reference or not os.access(H.VMATCH_REF, os.X_OK):
        return None
    need = 16 * n + (2 << 30)
    if shutil.disk_usage(a.workdir).free < need:
        log("reference baseline skipped: %s has less than %.0f GB free"
            % (a.workdir, need / 1e9))
        return None
    wd = os.path.join(a.workdir, "vsa_bench_%d" % os.getpid())
    os.makedirs(wd)
    try:
        t0 = time.time()
        write_fasta(wd + "/genome.fna", b">synthetic_genome seed=42\n", genome)
        # the index files the reference maps: vsa_mkvtree (GPU build, files
        # byte-identical to mkvtree's, tests/test_gpu_mkvtree.py), 64-bit
        # integers like the reference's LP64 build
        V.mkvtree([wd + "/genome.fna"], wd + "/genome.fna", integersize=64,
                  withskp=False)
        t_index = time.time() - t0
        ns1 = min(a.cpu_sample, len(qsym) // m)
        nsp = min(a.ref_sample, len(qsym) // m)
        per = nsp // ncores
        nsp = per * ncores
        write_queries(wd + "/q1.fna", qsym, m, 0, ns1)
        for p in range(ncores):
            write_queries(wd + "/qp%d.fna" % p, qsym, m, p * per, per)
        # P = the physical cores of the host (lscpu), each process
        # 1/P of the sample -- here the whole batch that is at hand
        pphys, lscpu = physical_cores()
        pphys = min(pphys or 0, len(os.sched_getaffinity(0)))
        perphys = min(125000, (len(qsym) // m) // pphys) if pphys else 0
        if pphys > ncores and perphys >= 10000:
            for p in range(pphys):
                write_queries(wd + "/qf%d.fna" % p, qsym, m, p * perphys,
                              perphys)
        else:
            pphys = 0
        env = dict(os.environ, VMATCHSHOWTIMESPACE="on")
        args = [H.VMATCH_REF, "-mum", "-l", str(L), "-q"]

        def run(files):
            t = time.time()
            procs = [subprocess.Popen(args + [f, "genome.fna"], cwd=wd,
                                      env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.DEVNULL)
                     for f in files]
            counts = []
            for p in procs:
                out = p.communicate()[0]
                if p.returncode != 0:
                    raise RuntimeError("vmatch_ref failed")
                counts.append(sum(1 for l in out.splitlines()
                                  if l and not l.startswith(b"#")))
            return time.time() - t, counts
        # page the index files in (pumpthroughcache, readvirt.c:567-678, by a
        # short run), then time
        write_queries(wd + "/warm.fna", qsym, m, 0, min(20000, ns1))
        run(["warm.fna"])
        t1, c1 = run(["q1.fna"])
        tp, cp = run(["qp%d.fna" % p for p in range(ncores)])
        log("reference vmatch: index files %.0f s, 1 core %.1f s (%d queries),"
            " %d cores %.1f s (%d queries)" % (t_index, t1, ns1, ncores, tp,
                                               nsp))
        allcores = None
        if pphys:
            tf, cf = run(["qf%d.fna" % p for p in range(pphys)])
            log("reference vmatch on all %d physical cores: %.1f s (%d "
                "queries)" % (pphys, tf, pphys * perphys))
            allcores = {
                "value": pphys * perphys / tf, "unit": "queries/s",
                "cores": pphys, "kind": "reference",
                "sample": "%d processes x %d queries (%s), wall of the "
                          "slowest = %.1f s; %d MUMs"
                          % (pphys, perphys, lscpu, tf, sum(cf))}
        # the drop-in binary (reference vmatch + integration/vmengine_shim.c)
        # on the same files: whole process, incl. mapping the index, its
        # upload to HBM, the derived tables, FASTA parsing and printing
        dropin = None
        gpubin = os.path.join(ROOT, "integration", "_build", "vmatch_gpu")
        if os.access(gpubin, os.X_OK):
            with open(wd + "/qall.fna", "wb") as f:
                for p in range(ncores):
                    with open(wd + "/qp%d.fna" % p, "rb") as g:
                        shutil.copyfileobj(g, f)
            genv = dict(os.environ, VMATCH_GPU_TRACE="1", VSA_TRACE="1")
            td = time.time()
            pr = subprocess.run([gpubin, "-mum", "-l", str(L), "-q",
                                 "qall.fna", "genome.fna"], cwd=wd, env=genv,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            td = time.time() - td
            nm = sum(1 for l in pr.stdout.splitlines()
                     if l and not l.startswith(b"#"))
            trace = [l for l in pr.stderr.decode().splitlines()
                     if l.startswith("vstree_amd:")]
            log("drop-in vmatch_gpu: %.1f s for %d queries, %d MUMs; %s"
                % (td, nsp, nm, "; ".join(trace)))
            # (its MUM count differs from the sum over the 16 slices above:
            # uniqueness is judged over ALL queries of a run; the drop-in's
            # output is compared with the reference's in tests/)
            if pr.returncode == 0 and nm > 0:
                dropin = {"value": nsp / td, "unit": "queries/s",
                          "wall_s": td, "queries": nsp, "mums": nm,
                          "what": "integration/_build/vmatch_gpu -mum -l %d "
                                  "-q (the reference program with the GPU "
                                  "engine linked in): whole process, index "
                                  "files -> HBM included" % L,
                          "trace": trace,
                          "reference_same_queries_16_processes_s": tp}
        return {
            "value": nsp / tp, "unit": "queries/s", "cores": ncores,
            "kind": "reference",
            "sample": "vmatch -mum -l %d (default -qspeedup 2) of the "
                      "reference built from its own sources (oracle/_ref), "
                      "%d processes x %d queries of the same batch on the "
                      "same %.2g bp index (files written by vsa_mkvtree, "
                      "64-bit; %d hardware threads visible, %d = the share "
                      "of one GPU used), whole processes incl. FASTA parsing "
                      "and output, wall of the slowest = %.1f s; %d MUMs"
                      % (L, ncores, per, n, len(os.sched_getaffinity(0)),
                         ncores, tp, sum(cp)),
            "reference_1core": {
                "value": ns1 / t1, "unit": "queries/s", "cores": 1,
                "kind": "reference",
                "sample": "%d queries, one process, %.1f s, %d MUMs"
                          % (ns1, t1, c1[0])},
            "reference_physical_cores": allcores,
            "mums_1core_sample": c1[0], "dropin_end_to_end": dropin}
    finally:
        shutil.rmtree(wd, ignore_errors=True)


def launch_ranks(a, jsonfd, have=None, more_args=(), more_keys=None):
    """`python bench.py --gpus N` without a launcher around it: start N ranks
    through torch.distributed.run (what the driver's own command line does),
    relay rank 0's JSON line.  device_count() does not initialise the GPU."""
    import socket
    if have is None:
        import torch
        have = torch.cuda.device_count()
    if not a.rehearse_on_one_gpu and have < a.gpus:
        log("bench.py: --gpus %d, but this node shows %d GPU(s)"
            % (a.gpus, have))
        sys.exit(2)
    if a.rehearse_on_one_gpu and have < 1:
        log("bench.py: no GPU")
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + \
        sys.argv[1:] + list(more_args)
    # (started from inside a rank of another launcher -- the fallback of the C
    # path --: nothing of that launcher's environment reaches the new ranks)
    env = {k: v for k, v in os.environ.items()
           if not (k.startswith(("TORCHELASTIC_", "PET_", "ROLE_", "MASTER_"))
                   or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "GROUP_RANK",
                            "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE"))}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for l in p.stdout.decode().splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
    if p.returncode != 0 or line is None:
        log("bench.py: the %d ranks ended with code %d%s"
            % (a.gpus, p.returncode, "" if line else " and printed no line"))
        sys.exit(p.returncode or 3)
    d = json.loads(line)
    if d.get("n_gpus") != a.gpus:
        log("bench.py: asked for %d GPUs, the ranks report %r"
            % (a.gpus, d.get("n_gpus")))
        sys.exit(3)
    if more_keys:
        d.update(more_keys)
        line = json.dumps(d)
    os.write(jsonfd, (line + "\n").encode())


def c_path_mode(a, jsonfd, rank, world):
    """--path c (the default for N > 1): the product's own N > 1 path.  ONE
    process drives every GPU through libvstree_amd_multi.so, a host thread per
    GPU (what integration/vmengine_shim.c binds for VMATCH_GPUS=N): the index
    replicated device to device, every replica's block of queries resident in
    ITS HBM before the clock starts, the lists left in HBM
    (vsa_multi_findmatches_device); -mum candidates cross between GPUs as
    16-byte rows by peer copies, the counters take the one ncclAllReduce.
    `--host` times the host-memory entry instead (vsa_multi_findmatches:
    queries in host memory, matches back in host memory -- PCIe-inclusive,
    named so in the metric, never the headline).
    Under a launcher with N ranks (the driver's command line) rank 0 is that
    one process; the other ranks hold no GPU and leave at once (the launcher
    waits for all of them, i.e. for rank 0)."""
    if rank != 0:
        # (no torch.distributed here, not even for a barrier: torch brings a
        # HIP runtime of its own, and libvstree_amd_multi.so drives the GPUs
        # through the one it links -- one runtime per process)
        log("bench.py: rank %d of %d: the C path runs in rank 0" % (rank, world))
        return
    import vstree_amd as V
    from vstree_amd import multi as M
    n, nq, m, L = int(a.genome), int(a.queries), a.qlen, a.minlen
    N = a.gpus
    have = V.device_count()
    if a.replicas_on_one_gpu:
        devices = [0] * N
    else:
        if have < N:
            log("bench.py: --path c --gpus %d, but %d GPU(s) here" % (N, have))
            sys.exit(2)
        devices = list(range(N))
    t0 = time.time()
    dg = V.device_malloc(n + 64, devices[0])
    V._check(V.lib.vsa_synth_genome_device(V.GENOME_SEED, n, dg, devices[0]))
    index = V.Index.build_device(dg, n, 4, 0, devices[0])
    info = index.info()
    t_index = time.time() - t0
    # the queries of the whole job: block r of nq reads for replica r (weak
    # scaling), generated on the first device
    pos, sub, step = V.synth_query_plan(n, nq * N, m)
    dq = V.device_malloc(nq * m + 64, devices[0])
    hq = np.empty(nq * N * m, np.uint8) if a.host else None
    blocks, hblock = [], np.empty(nq * m, np.uint8)
    for r in range(N):
        sl = slice(r * nq, (r + 1) * nq)
        ps, sb, st = (np.ascontiguousarray(x[sl]) for x in (pos, sub, step))
        V._check(V.lib.vsa_synth_queries_device(
            dg, n, ps.ctypes.data, sb.ctypes.data, st.ctypes.data, nq, m, dq,
            devices[0]))
        if a.host:
            V.device_download(hq[r * nq * m:(r + 1) * nq * m], dq, devices[0])
        elif a.reads == "packed":
            V.device_download(hblock, dq, devices[0])
            b = V.Queries.from_host_packed(hblock, m, devices[r])
            b.set_offset(r * nq)
            blocks.append(b)
        elif devices[r] == devices[0]:
            b = V.Queries.from_device(dq, nq, m, devices[r])
            b.set_offset(r * nq)
            blocks.append(b)
        else:
            V.device_download(hblock, dq, devices[0])
            b = V.Queries.from_host(hblock, np.arange(nq, dtype=np.uint64) * m,
                                    np.full(nq, m, np.uint64), devices[r])
            b.set_offset(r * nq)
            blocks.append(b)
    V.device_free(dq, devices[0])
    V.device_free(dg, devices[0])
    t1 = time.time()
    multi = M.Multi.replicate(index, devices)
    t_rep = time.time() - t1
    if a.test_fail_c_path:
        multi.close()
        raise RuntimeError("--test-fail-c-path")
    log("setup: index %d bp (%.1f GB in HBM, deep prefix %d) built in %.1fs, "
        "%d replica(s) in %.1fs" % (n, info.device_bytes / 1e9,
                                    info.deepprefix, t_index, N, t_rep))
    start = np.arange(nq * N, dtype=np.uint64) * m
    length = np.full(nq * N, m, np.uint64)
    kernel_ms, first_ms = [], []

    mp, slotns = None, {}
    if a.host and not a.compat:
        # vsa_multi_pipeline_*: a packed pipeline per replica; a step = one
        # -mum job of one batch per replica; the slots' page-locked rows are
        # packed before the clock starts (a slot keeps its reads)
        mp = M.MultiPipeline(multi, M.MUM, L, m, nq, maxspecial=1024)

    def pipeline_job():
        for r in range(N):
            got = mp.hostrows()
            assert got is not None
            rows, special = got
            key = rows.ctypes.data
            if key not in slotns:
                ns = C.c_uint64(0)
                V._check(V.lib.vsa_pack_reads(
                    hq[r * nq * m:].ctypes.data, nq, m, m, rows.ctypes.data,
                    special.ctypes.data, mp.maxspecial, C.byref(ns)))
                slotns[key] = int(ns.value)
            mp.submit(nq, slotns[key])
        while mp.next(copy=False)[0] != 1:
            pass
        lists, st = mp.finish(copy=False)
        assert sum(len(x) for x in lists) == st.count
        return st

    def one_step():
        if mp is not None:
            st, rc, msg = pipeline_job(), 0, ""
        elif a.host:
            mm, st, rc, msg = multi.findmatches(M.MUM, hq, start, length, L)
        else:
            res, st, rc, msg = multi.findmatches_device(M.MUM, blocks, L)
            for r_ in res:
                if r_ is not None:
                    r_.close()
        if rc != 0:
            raise RuntimeError(msg)
        kernel_ms.append(st.search_kernel_ms)
        first_ms.append(st.first_kernel_ms)
        return st

    def sync():
        for d in sorted(set(devices)):
            V.device_synchronize(d)

    st = None
    for _ in range(max(a.warmup, 3) if mp is not None else a.warmup):
        st = one_step()         # (a pipeline: all three slots of a replica)
    kernel_ms.clear()
    first_ms.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        st = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    kms = float(np.mean(kernel_ms))
    what = ("host memory to host memory through vsa_multi_pipeline_* (packed "
            "rows in page-locked slots, three batches in flight per GPU) -- "
            "PCIe-inclusive" if mp is not None else
            "host memory to host memory through vsa_multi_findmatches (a "
            "Multiseq in pageable memory in, one malloc'ed list out) -- "
            "PCIe-inclusive" if a.host else
            "queries and match lists resident in HBM, "
            "vsa_multi_findmatches_device")
    out = {
        "metric": ("queries/sec (%d bp queries, vmatch -mum -l %d semantics, "
                   "%s ESA index replicated in HBM%s)"
                   % (m, L, human(n, "bp"),
                      "; host memory to host memory, PCIe-inclusive"
                      if a.host else "")),
        "value": nq * N * a.steps / elapsed, "unit": "queries/s",
        "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": workload_label(n, nq, m, L, "-mum"),
                   "baseline_config": baseline_config(n, nq, m, L),
                   "path": "c: one process, a host thread per GPU (" + what +
                           ")",
                   "index_bp": n, "queries_per_gpu": nq, "query_len": m,
                   "minlen": L, "prefixlength": info.prefixlength,
                   "deepprefix": info.deepprefix,
                   "reads_in_hbm": a.reads,
                   "index_bytes_hbm": info.device_bytes,
                   "index_build_s": round(t_index, 2),
                   "devices": devices,
                   "replicas_on_one_gpu": bool(a.replicas_on_one_gpu),
                   "index_replication_s": round(t_rep, 2),
                   "parallelism": "index replicated, queries in %d blocks, "
                                  "-mum candidates exchanged by peer copies, "
                                  "counters by one ncclAllReduce" % N},
        "gbp_matched_per_s": st.sumlength / (elapsed / a.steps) / 1e9,
        "rccl_ranks": N if multi.uses_rccl() else 0, "ranks": 1,
        "launcher_ranks": world,
        "matches": int(st.count), "candidates": int(st.candidates),
        "query_suffix_searches": int(st.searches),
        # the dominant kernel on the slowest replica (HIP events); its
        # algorithmic bytes are counted at N = 1 by the single-process line
        "roofline": {"kernel": "k_query_search_planned<uint32_t, 256, deep, "
                               "%s, MUM>" % ("bytes" if (a.host and a.compat) or
                                             a.reads != "packed" else "rows"),
                     "bound": "hbm",
                     "kernel_ms": kms, "first_pass_ms": float(np.mean(first_ms)),
                     "searches_per_launch": int(st.kernel_searches) / N,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "achieved": None, "frac": None, "traffic": None,
                     "note": "slowest replica per step; bytes per search are "
                             "counted by the N = 1 line of the same sources"},
    }
    if a.host:
        out["bytes_over_pcie_per_step"] = int(nq * N * m + 16 * nq * N +
                                              32 * int(st.count))
    borrowed = None
    if not a.quick:
        try:
            # the kernel's algorithmic bytes, counted like the N = 1 line does: on
            # the searches the plans of block 0 hold (test infrastructure: the
            # instrumented CPU restatement), x the searches of one replica
            import helpers as H
            w = info.device_integersize // 8
            # (replica 0, borrowed: the set owns it)
            index = borrowed = V.Index(
                C.c_void_p(M.lib.vsa_multi_index(multi._h, 0)))
            t = index.download()
            host = H.Index(n, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                           t["llv"], t["bck"], t["bwt"], None)
            ns = min(nq, 20000)
            rows = t["tis"][pos[:ns, None].astype(np.int64) +
                            np.arange(m)[None, :]].astype(np.uint8)
            hit = np.flatnonzero(sub[:ns] != V.NO_SUBST)
            rows[hit, sub[hit]] = (rows[hit, sub[hit]] + step[hit]) & 3
            small = H.Queries.uniform(np.ascontiguousarray(rows).ravel(), m)
            block0 = V.Queries.from_host_packed(small.symbols, m, devices[0]) \
                if a.reads == "packed" else V.Queries.from_host(
                    small.symbols, small.start, small.length, devices[0])
            per, items = counted_search_bytes(a, V, H, index, block0, host, small,
                                              m, L, w)
            block0.close()
            del t, host
            if per is not None:
                rf = out["roofline"]
                nbytes = per * rf["searches_per_launch"]
                rf.update({
                    "algorithmic_bytes_per_launch": nbytes,
                    "bytes_per_search": per, "searches_counted": items,
                    "bytes_are": "counted",
                    "achieved": nbytes / (kms * 1e-3) / 1e9,
                    "frac": nbytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "note": "slowest replica per step (HIP events); algorithmic "
                            "bytes = SURVEY 8d formula counted by the "
                            "instrumented CPU restatement on the searches the "
                            "plans of the first %d reads hold, x the searches of "
                            "one replica" % small.nq})
                tj = pmc_traffic(n, nq)
                if tj.get("hbm_bytes_per_launch"):
                    rf["traffic"] = tj["hbm_bytes_per_launch"]
                    rf["traffic_source"] = "%s (the same kernel at N = 1)" % \
                        tj.get("source")
        except Exception as e:     # the roofline must not sink the line
            log("bench.py: counting the kernel's bytes failed: %r" % (e,))
        finally:
            if borrowed is not None:
                borrowed._h = None     # (not ours to close)
    if mp is not None:
        mp.close()
    multi.close()
    os.write(jsonfd, (json.dumps(out) + "\n").encode())


def physical_cores():
    """(sockets x cores per socket, what lscpu said)"""
    try:
        txt = subprocess.run(["lscpu"], stdout=subprocess.PIPE).stdout.decode()
        f = {}
        for l in txt.splitlines():
            if ":" in l:
                k, v = l.split(":", 1)
                f[k.strip()] = v.strip()
        p = int(f["Socket(s)"]) * int(f["Core(s) per socket"])
        return p, "lscpu: %s socket(s) x %s cores, %s threads per core, %s" % (
            f["Socket(s)"], f["Core(s) per socket"],
            f.get("Thread(s) per core", "?"), f.get("Model name", "?"))
    except Exception as e:      # no lscpu, unexpected output
        return None, "lscpu unavailable (%r)" % (e,)


def selfmum_text(V, n):
    """db half + separator + a copy with one substitution every 97 bp: every
    suffix pair is an lcp peak (the dense case of scripts/selfmum_probe.py)"""
    half = (n - 1) // 2
    g = V.synth_genome(half)
    g2 = g.copy()
    g2[::97] = (g2[::97] + 1) & 3
    return np.concatenate([g, np.array([255], np.uint8), g2]), half


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
    path_by_default = a.path is None
    if a.path is None:
        # N > 1: the C path (north_star: host code in C, RCCL for the count
        # reduction); the rehearsal switches belong to the torch form
        a.path = "c" if (a.gpus > 1 and a.mode == "mum" and
                         not a.rehearse_on_one_gpu and
                         not a.force_distributed) else "torch"
    if a.path == "c" and world not in (1, a.gpus):
        log("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a line "
            "for another number of GPUs than asked for" % (a.gpus, world))
        sys.exit(2)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and a.path == "torch":
        # started directly: become the launcher of N ranks (one per GPU) before
        # anything touches a GPU, hand their one JSON line on, leave with
        # their exit code
        return launch_ranks(a, jsonfd)
    if world != a.gpus and a.path == "torch":
        log("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to print a line "
            "for another number of GPUs than asked for" % (a.gpus, world))
        sys.exit(2)
    dev = 0 if a.rehearse_on_one_gpu else local_rank
    if a.path == "c":
        try:
            return c_path_mode(a, jsonfd, rank, world)
        except Exception as e:
            # No node with more than one GPU has run the C path yet.  Where it
            # was chosen by default and fails, the job is not lost: rank 0
            # starts the other N > 1 form (one process per GPU over
            # torch.distributed / RCCL) and says so in the line.
            if not path_by_default or rank != 0:
                raise
            import gc
            import traceback
            why = "%s: %s" % (type(e).__name__, e)
            log("bench.py: the C path failed (%s); falling back to --path "
                "torch\n%s" % (why, traceback.format_exc()))
            e = None
            gc.collect()
            import vstree_amd as V
            have = V.device_count()
            for d in range(have):
                V.lib.vsa_device_trim(d)
            more = ["--path", "torch"]
            if a.replicas_on_one_gpu:     # (the rehearsal of this very step)
                more.append("--rehearse-on-one-gpu")
                a.rehearse_on_one_gpu = True
            return launch_ranks(a, jsonfd, have=have, more_args=more,
                                more_keys={"c_path_failed": why})

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
    if a.mode == "selfmum":
        return selfmum_mode(a, V, S, torch, dist, rank, world, dev, jsonfd)

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
    bytequeries = None
    if a.reads == "packed":
        hq = np.empty(nq * m, np.uint8)
        V.device_download(hq, dq, dev)
        bytequeries = queries      # (timed for comparison behind the steps)
        bytequeries.set_offset(rank * nq)
        queries = V.Queries.from_host_packed(hq, m, dev)
        del hq
    queries.set_offset(rank * nq)
    V.device_free(dq, dev)
    extras = rank == 0 and world == 1 and not a.quick
    q150 = hq150 = None
    if extras:
        # BASELINE configs[4]: 150 bp reads of the same genome
        p5, s5, st5 = V.synth_query_plan(n, nq, 150)
        dq = V.device_malloc(nq * 150 + 64, dev)
        V._check(V.lib.vsa_synth_queries_device(dg, n, p5.ctypes.data,
                                                s5.ctypes.data,
                                                st5.ctypes.data, nq, 150, dq,
                                                dev))
        q150 = V.Queries.from_device(dq, nq, 150, dev)
        hq150 = np.empty(nq * 150, np.uint8)    # (for the rows of -mum below)
        V.device_download(hq150, dq, dev)
        V.device_free(dq, dev)
    if not extras:
        V.device_free(dg, dev)
        dg = None
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

    kernel_ms, first_ms, totals = [], [], None
    rowbuf = [None]     # N > 1 form: the candidate rows, kept between batches
    if distributed and not a.rehearse_on_one_gpu:
        rowbuf[0] = torch.empty(2 * 1024, dtype=torch.int64, device="cuda")

    def one_step():
        """the hot path over the whole batch; returns (count, sumlength,
        searches, candidates) of the job"""
        nonlocal totals
        if not distributed:
            r = V.findquerymatches(index, queries, L, mum=True)
            s = r.stats()
            kernel_ms.append(s.search_kernel_ms)
            first_ms.append(s.first_kernel_ms)
            totals = (s.count, s.sumlength, s.searches, s.candidates,
                      s.kernel_searches)
            r.close()
            return
        # phase 1: candidates of this rank's queries (no communication), in
        # the order the kernel left them, grouped by the rank that filters
        # their range of the index
        # (as pairs of sort key and value, 16 bytes each: half the exchange)
        # (with RCCL the split sizes and right ends stay on the device, where
        # the all-gather reads them, and search and grouping are one call into
        # the library: the row buffer is kept from batch to batch)
        # (the rows for this rank's own range behind the others: they stay
        # where they are, the exchange runs over the rows in front of them)
        ondev = not a.rehearse_on_one_gpu and not a.meta_on_host
        send = top = None
        if ondev:
            meta = S.meta_device_buffer(torch, world, "cuda")
            r, grouped = V.findmumcandidates_grouped(
                index, queries, L, lenbits, world, rank,
                C.c_void_p(rowbuf[0].data_ptr()), rowbuf[0].numel() // 2,
                C.c_void_p(meta.data_ptr()))
            if not grouped:
                rowbuf[0] = torch.empty(int(r.count * 1.1 + 1024) * 2,
                                        dtype=torch.int64, device="cuda")
                r.partition_device(world, n, C.c_void_p(rowbuf[0].data_ptr()),
                                   C.c_void_p(meta.data_ptr()), own=rank)
            mine = rowbuf[0][:r.count * 2]
            s = r.stats()
        else:
            r = V.findmumcandidates_packed(index, queries, L, lenbits)
            s = r.stats()
            mine = torch.empty(max(r.count, 1) * 2, dtype=torch.int64,
                               device="cuda")[:r.count * 2]
            send, top = r.partition(world, n, C.c_void_p(mine.data_ptr()),
                                    own=rank)
            r.close()
        kernel_ms.append(s.search_kernel_ms)
        first_ms.append(s.first_kernel_ms)
        cdev = "cuda"
        if a.rehearse_on_one_gpu:
            mine, cdev = mine.cpu(), "cpu"
        # (ondev: the grouping is still queued; r goes after the exchange)

        # phase 2: the one exchange step (RCCL all-to-all) -- every rank runs
        # the uniqueness filter on its range with the carry of the lower ones
        def filter_fn(own, received, carry):
            own = own.cuda().contiguous()
            received = received.cuda().contiguous()
            res = V.mumuniqueinquery_range_packed2(
                C.c_void_p(own.data_ptr()), own.numel() // 2,
                C.c_void_p(received.data_ptr()), received.numel() // 2,
                lenbits, n, carry, dev)
            st = res.stats()
            res.close()
            return st.count, st.sumlength

        # (the counters stay local: ONE all-reduce at the end of the job)
        nmum, sumlen, ncand, searches, ksearches = \
            S.partitioned_mum_filter_presorted(
                dist, torch, mine, send, top, cdev, filter_fn, words=2,
                extra=[s.searches, s.kernel_searches], reduce=False,
                own_last=True, meta_on_device=ondev)
        if ondev:
            r.close()
        totals = (nmum, sumlen, searches, ncand, ksearches)

    for _ in range(a.warmup):
        one_step()
    kernel_ms.clear()
    first_ms.clear()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_step()
    if distributed:
        # the final match-count reduction of the job (every step produces the
        # same counters here; a real job would have summed them over its
        # batches first)
        totals = tuple(S.all_reduce_counters(
            dist, torch, totals, "cpu" if a.rehearse_on_one_gpu else "cuda"))
    sync()
    elapsed = time.perf_counter() - t0
    if distributed:
        e = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if a.rehearse_on_one_gpu else "cuda")
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    count, sumlength, searches, candidates, kernel_searches = totals
    bytes_form = None
    if bytequeries is not None and not distributed:
        # the same batch as the reference's Multiseq bytes, a few steps
        V.device_synchronize(dev)
        tb = time.perf_counter()
        for _ in range(max(3, a.steps // 4)):
            r = V.findquerymatches(index, bytequeries, L, mum=True)
            sb = r.stats()
            r.close()
        V.device_synchronize(dev)
        tb = (time.perf_counter() - tb) / max(3, a.steps // 4)
        if (sb.count, sb.sumlength, sb.candidates) != (count, sumlength,
                                                       candidates):
            raise RuntimeError("bench.py: the packed and the byte batch of "
                               "the same reads disagree")
        bytes_form = {"ms_per_step": tb * 1e3, "value": nq / tb,
                      "search_kernel_ms": sb.search_kernel_ms,
                      "first_pass_ms": sb.first_kernel_ms,
                      "what": "the same reads resident as one byte per symbol "
                              "(the reference's Multiseq, round 3's form): "
                              "same counters"}
    if bytequeries is not None:
        bytequeries.close()
    total_queries = nq * world
    qps = total_queries * a.steps / elapsed
    kms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    fms = float(np.mean(first_ms)) if first_ms else float("nan")

    out = {
        "metric": "queries/sec (%d bp queries, vmatch -mum -l %d "
                  "semantics, %s ESA index resident in HBM)"
                  % (m, L, human(n, "bp")),
        "value": qps, "unit": "queries/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": workload_label(n, nq, m, L, "-mum"),
                   "baseline_config": baseline_config(n, nq, m, L),
                   "index_bp": n, "queries_per_gpu": nq, "query_len": m,
                   "minlen": L, "prefixlength": info.prefixlength,
                   "index_bytes_hbm": info.device_bytes,
                   "index_build_s": round(t_index, 2),
                   "reads_in_hbm": ("two bits per symbol, %d bytes per read "
                                    "(vsa_pack_reads)"
                                    % (V.lib.vsa_packed_words(m) * 8)
                                    if a.reads == "packed" else
                                    "one byte per symbol (the reference's "
                                    "Multiseq)"),
                   "parallelism": "index replicated, queries sharded x%d"
                                  % world},
        "gbp_matched_per_s": sumlength * a.steps / elapsed / 1e9
        if world == 1 else sumlength / (elapsed / a.steps) / 1e9,
        "matches": count, "candidates": candidates,
        "query_suffix_searches": searches,
        # ranks of the RCCL communicator the counters went through (0: one
        # process, no collective; gloo in the one-GPU rehearsal is not RCCL)
        "rccl_ranks": (dist.get_world_size()
                       if distributed and not a.rehearse_on_one_gpu else 0),
        "ranks": world,
    }
    if bytes_form is not None:
        out["reads_as_bytes"] = bytes_form

    if rank == 0:
        import helpers as H  # test infrastructure: the CPU oracle
        w = info.device_integersize // 8
        t = index.download()
        host = H.Index(n, info.prefixlength, 4, t["tis"], t["suf"], t["lcp"],
                       t["llv"], t["bck"], t["bwt"], None)
        pphys = physical_cores()[0] or 0
        nsample = min(nq, max(a.cpu_sample, a.ref_sample if extras else 0,
                              min(pphys, len(os.sched_getaffinity(0))) *
                              125000 if extras and not a.no_reference else 0,
                              20000))
        qsym = np.zeros(nsample * m, np.uint8)
        g = t["tis"]
        rows = qsym.reshape(nsample, m)
        for c0 in range(0, nsample, 1 << 20):   # the same queries the GPU has
            c1 = min(nsample, c0 + (1 << 20))
            idx = pos[c0:c1, None].astype(np.int64) + np.arange(m)[None, :]
            rows[c0:c1] = g[idx]
            del idx
        hit = np.flatnonzero(sub[:nsample] != V.NO_SUBST)
        rows[hit, sub[hit]] = (rows[hit, sub[hit]] + step[hit]) & 3
        small = H.Queries.uniform(qsym[:20000 * m], m)
        bytes_per_query, counters = count_bytes(
            H, lambda: H.oracle_querymatches(host, small, L, mum=True,
                                             cand=True, speedup=0),
            small.nq, w, int(small.length.sum()))
        # ... and counted on exactly the searches the dominant kernel runs:
        # the plans of the first queries as the engine made them (one more
        # call with VSA_DEBUG_PLANFILE, outside the timed region), every
        # planned (query, offset) searched by the instrumented restatement
        counted_per_search = counted_items = None
        if not distributed:
            counted_per_search, counted_items = counted_search_bytes(
                a, V, H, index, queries, host, small, m, L, w)
        # SURVEY 8d / BASELINE.md: bytes of the reference's per-suffix
        # algorithm (one bucket lookup + binary search for EVERY query
        # suffix) x queries per launch
        alg0_bytes_launch = bytes_per_query * nq
        full_searches = nq * (m - L + 1)
        # the dominant kernel only runs the searches the first pass and the
        # work plan left over; price it on that work, not on work they
        # proved unnecessary
        main_searches = kernel_searches
        executed_bytes_launch = alg0_bytes_launch * (
            (main_searches / world) / full_searches)
        bytes_are = "modelled"
        if counted_per_search is not None:
            executed_bytes_launch = counted_per_search * main_searches / world
            bytes_are = "counted"
        achieved = executed_bytes_launch / (kms * 1e-3) / 1e9
        # the rate at which THIS device delivers random 64-byte sectors of the
        # table every search starts in (vsa_measure_table_read on slot16,
        # disjoint address sequences per lane): the ceiling of a kernel whose
        # loads are random 16-byte slots -- 39 % of the streaming peak
        rnd = C.c_double(0.0)
        rc = V.lib.vsa_measure_table_read(index._h, 0, 4, C.byref(rnd))
        random_gs = rnd.value if rc == 0 else None
        # HBM bytes of the dominant kernel from the PMC passes of this very
        # kernel source (scripts/pmc_passes.sh writes the file; a profile of
        # other sources is not quoted)
        tj = pmc_traffic(n, nq)
        traffic, traffic_source = (tj.get("hbm_bytes_per_launch"),
                                   tj.get("source"))
        out["roofline"] = {
            "kernel": "k_query_search_planned<uint32_t, 256, deep, %s, MUM>"
                      % ("rows" if a.reads == "packed" else "bytes"),
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel_ms": kms,
            "algorithmic_bytes_per_launch": executed_bytes_launch,
            "searches_per_launch": main_searches / world,
            "algorithmic_bytes_per_query_all_suffixes": bytes_per_query,
            "achieved_if_priced_on_all_suffixes":
                alg0_bytes_launch / (kms * 1e-3) / 1e9,
            "bytes_are": bytes_are,
            "bytes_per_search": executed_bytes_launch / max(
                main_searches / world, 1),
            "searches_counted": counted_items,
            "note": "kernel_ms: HIP events around the kernel, this run. "
                    "algorithmic bytes = SURVEY 8d formula (2w bucket + w "
                    "per probe + compared symbols + lcp entries + w + 17 per "
                    "hit) counted by the instrumented CPU restatement on the "
                    "(query, offset) searches this kernel runs for the first "
                    "20 000 queries (their plans dumped by the engine), x "
                    "the kernel's searches (%.1f%% of the %d per query: the "
                    "first pass and the work plan prove the others "
                    "unnecessary; all of them: %.1f kB per query); the path "
                    "is random 16-byte slots, one 128-byte line of HBM each "
                    "(traffic = 2 x FETCH_SIZE + WRITE_SIZE)"
                    % (100.0 * main_searches / world / full_searches,
                       m - L + 1, bytes_per_query / 1e3)}
        if random_gs:
            # round 4: what HBM charges a random read is the aligned 128-byte
            # line (profiles/r04/README.md), so the rate below is a rate of
            # LINES and x 128 B the streaming rate of the device -- rounds 1-3
            # read it as 64-byte sectors (39 % of the peak)
            lines = tj.get("read_requests_per_launch")
            out["roofline"]["random_line_ceiling"] = {
                "measured_G_lines_per_s": random_gs,
                "GBs_of_128B_lines": random_gs * 128,
                "frac_of_hbm_peak": random_gs * 128 / HBM_PEAK_GBS,
                "kernel_G_lines_per_s":
                    lines / (kms * 1e-3) / 1e9 if lines else None,
                "kernel_frac_of_ceiling":
                    lines / (kms * 1e-3) / 1e9 / random_gs if lines else None,
                "useful_bytes_per_line":
                    executed_bytes_launch / lines if lines else None,
                "note": "random 16-byte reads of the slot table, 4 in flight "
                        "per lane, every lane its own address sequence "
                        "(vsa_measure_table_read); the kernel's lines = read "
                        "requests of the PMC pass (FETCH_SIZE x 1024 / 64)"}
        fams = []
        if world == 1:
            # every query once, with the whole query: priced like the
            # reference's complete-match search of a query (exactcompl.c:168)
            cbytes, _ = count_bytes(
                H, lambda: H.oracle_complete(host, small), small.nq, w,
                int(small.length.sum()))
            fams.append(family(
                "k_mum_first<uint32_t, deep>", "-mum -l %d, first pass" % L,
                fms, cbytes * nq,
                "offset 0 of every query located with the whole query; "
                "bytes = %.0f B/query, the reference's -complete search of "
                "the query (bucket, binary search, comparison, lcp)" % cbytes,
                traffic_key="k_mum_first", queries=nq))
        if extras:
            fams += extra_families(a, V, H, index, queries, q150, host, small,
                                   w, nq, m, L, bytes_per_query, cbytes)
            out["mum_150bp"] = mum_other_length(V, index, q150, hq150, nq,
                                                150, L, dev)
            hq150 = None
            out["end_to_end"] = end_to_end(V, index, dg, n, nq, m, L, dev)
            V.device_free(dg, dev)
        out["roofline_families"] = fams
        if world == 1 and a.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baselines(
                a, V, H, index, host, qsym, m, L, dev, qps,
                free_index=lambda: index.close())
        if extras:
            index.close()   # (if the CPU baselines have not done so: room for
            #                  the second 3 Gbp index)
            k3 = selfmum_family(a, V, n, L, dev)
            fams.append(k3)
            # north_star's "suftab scan": where the driver's record keeps it
            out["roofline"]["suftab_scan"] = {
                k: k3[k] for k in ("kernel", "mode", "bound", "achieved",
                                   "peak", "unit", "frac", "traffic",
                                   "kernel_ms", "algorithmic_bytes_per_launch",
                                   "note") if k in k3}
            # ... and as scalars: a record that keeps only the flat keys of
            # `roofline` (BENCH_r03.json.parsed dropped the nested dict)
            out["roofline"]["suftab_scan_frac"] = k3["frac"]
            out["roofline"]["suftab_scan_ms"] = k3["kernel_ms"]
            out["roofline"]["suftab_scan_bytes"] = \
                k3["algorithmic_bytes_per_launch"]
            out["roofline"]["suftab_scan_achieved_GBs"] = k3["achieved"]
        tj = pmc_traffic(n, nq)
        if tj.get("step_hbm_bytes"):
            sb = tj["step_hbm_bytes"]
            out["roofline"]["step"] = {
                "hbm_bytes_per_step": sb,
                "achieved": sb / (out["ms_per_step"] * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": sb / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "HBM bytes of ALL kernels of one step (PMC FETCH_SIZE "
                        "+ WRITE_SIZE of every dispatch between the first "
                        "pass of one step and the first pass of the next, "
                        "%s) / ms_per_step of this run" % tj.get("source")}
        sys.stdout.flush()
        os.write(jsonfd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


def extra_families(a, V, H, index, queries, q150, host, small, w, nq, m, L,
                   bytes_per_query, cbytes):
    """the other kernel families of the path on the same index: a few calls
    each, HIP-event kernel times from the library's statistics"""
    fams = []

    def best(fn, reps=4):
        got = []
        for _ in range(reps):
            r = fn()
            got.append(r.stats())
            r.close()
        return min(got, key=lambda s: s.total_device_ms)

    s = best(lambda: V.findcompletematches(index, queries))
    fams.append(family(
        "k_complete_search<uint32_t, deep>", "-complete (BASELINE configs[1] "
        "semantics on the %s index)" % human(host.n, "bp"), s.search_kernel_ms, cbytes * nq,
        "bytes = %.0f B/query (SURVEY 8d: m + 2w + probes*(w + c) + lcp + "
        "occ*(w + 16))" % cbytes, traffic_key="k_complete_search", queries=nq,
        matches=s.count, call_device_ms=s.total_device_ms))
    s = best(lambda: V.findquerymatches(index, queries, L), reps=2)
    # since round 4 the MEM plan (mem_workplan.inc) leaves the search kernel
    # the offsets it cannot answer from the repeat bits: the kernel is priced
    # on the searches it RUNS (the per-search average over all suffixes x its
    # work-items: modelled, like K2 before round 3), the call on all of them
    allbytes = bytes_per_query * nq
    ran = float(s.kernel_searches)
    persearch = bytes_per_query / (m - L + 1)
    fams.append(family(
        "k_query_search_planned<uint32_t, 256, deep, MEM>", "-l %d (MEM)" % L,
        s.search_kernel_ms, persearch * ran,
        "the MEM plan leaves %d of the %d query suffixes to the search "
        "kernel (%.1f %%); bytes = %.0f B per search (average over all "
        "suffixes, modelled) x the searches it runs; the whole call does the "
        "work of %.0f B/query x %d queries = %.1f GB in %.2f ms"
        % (s.kernel_searches, (m - L + 1) * nq,
           100.0 * ran / ((m - L + 1) * nq), persearch, bytes_per_query, nq,
           allbytes / 1e9, s.total_device_ms),
        traffic_key="k_query_search_mem",
        queries=nq, matches=s.count, call_device_ms=s.total_device_ms))
    # BASELINE configs[4]: -complete -e 2 on 150 bp reads
    s = best(lambda: V.findapproxcompletematches(index, q150, True, 2), reps=3)
    # piece search: every read is cut into exact pieces (splitesaapm.c:317),
    # each searched like a complete match; counted on a sample of 30-mers
    piece = H.Queries.uniform(np.ascontiguousarray(
        small.symbols.reshape(small.nq, m)[:, :30]).ravel(), 30)
    pbytes, _ = count_bytes(H, lambda: H.oracle_complete(host, piece),
                            piece.nq, w, int(piece.length.sum()))
    fams.append(family(
        "k_complete_search<uint32_t, deep> (pieces)",
        "-complete -e 2, %s x 150 bp (BASELINE configs[4] semantics): piece "
        "search" % human(nq),
        s.search_kernel_ms, pbytes * s.searches,
        "%d pieces of 30 bp, %.0f B each" % (s.searches, pbytes),
        queries=nq, matches=s.count, call_device_ms=s.total_device_ms))
    band = 150 + 2 * 2 + 150   # text window + pattern per start position
    fams.append(family(
        "k_apm_banded<2>",
        "-complete -e 2, %s x 150 bp: banded alignment of the start "
        "positions" % human(nq), s.first_kernel_ms, float(band) * s.kernel_searches,
        "%d start positions x (154 text + 150 pattern symbols)"
        % s.kernel_searches, traffic_key="k_apm_banded", queries=nq))
    return fams


def mum_other_length(V, index, qbytes, hostsymbols, nq, m, L, dev):
    """the headline step on reads of another length (150 bp, the other common
    read length: rows of five words, which the first pass looks at through
    windows): the same timing, a few steps; not the headline"""
    packed = V.Queries.from_host_packed(hostsymbols, m, dev)
    out = {}
    for name, q in (("rows", packed), ("bytes", qbytes)):
        for _ in range(2):
            V.findquerymatches(index, q, L, mum=True).close()
        V.device_synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(5):
            r = V.findquerymatches(index, q, L, mum=True)
            st = r.stats()
            r.close()
        V.device_synchronize(dev)
        dt = (time.perf_counter() - t0) / 5
        out[name] = {"ms_per_step": dt * 1e3, "value": nq / dt,
                     "search_kernel_ms": st.search_kernel_ms,
                     "first_pass_ms": st.first_kernel_ms,
                     "matches": int(st.count)}
    packed.close()
    if out["rows"]["matches"] != out["bytes"]["matches"]:
        raise RuntimeError("bench.py: %d bp reads as rows and as bytes "
                           "disagree" % m)
    out["what"] = "vmatch -mum -l %d, %s reads of %d bp, results in HBM" % (
        L, human(nq), m)
    return out


def end_to_end(V, index, dg, n, nq, m, L, dev):
    """queries in (page-locked) host memory -> matches in host memory through
    vsa_pipeline_*: three batches in flight, upload / search / download
    overlapped.  The reads travel at two bits per symbol (vsa_pack_reads into
    the slots' rows: 32 bytes per 100 bp read instead of 100).  Two jobs:
    `-mum` over 3 different batches (30 M reads; the global filter over all
    candidates and the download of the MUM list are inside the timed region),
    and `-mum cand` in steady state (12 batches, every batch's list
    downloaded).  The page-locked rows are filled before the clock starts (a
    slot keeps its reads between jobs): what is timed is host memory -> host
    memory, not the production of the reads; the packer's own rate (one host
    thread, and the threads of this GPU's share of the box) is reported next
    to it, and `mumcand_incl_packing` runs the steady state with the packing
    of every batch inside the clock.  `bytes`: the same two jobs with the
    reads as bytes (the form of round 3), for comparison."""
    pos, sub, step = V.synth_query_plan(n, 3 * nq, m, seed=777)
    dq = V.device_malloc(nq * m + 64, dev)
    hbuf = np.empty(nq * m, np.uint8)

    def reads(b):
        sl = slice(b * nq, (b + 1) * nq)
        ps, sb, st = (np.ascontiguousarray(x[sl]) for x in (pos, sub, step))
        V._check(V.lib.vsa_synth_queries_device(
            dg, n, ps.ctypes.data, sb.ctypes.data, st.ctypes.data, nq, m, dq,
            dev))
        V.device_download(hbuf, dq, dev)
        return hbuf

    packrate = [0.0]
    packthreads = max(1, min(16, len(os.sched_getaffinity(0))))
    hbatches = []     # (filled by the first job: the three batches as bytes)

    def job(p, batches, refill, packthreads_=0):
        """refill: the slots get their reads (untimed callers); packthreads_
        > 0: every batch is packed from its bytes into the slot INSIDE this
        call, on that many host threads (the slots' rows do not count as given)"""
        sub_, got, total = 0, 0, 0
        while got < batches:
            slot = None
            if sub_ < batches:
                slot = p.hostrows() if p.packed else p.hostbuffer()
            if slot is not None:
                if p.packed:
                    rows, special = slot
                    key = rows.ctypes.data
                    if refill or packthreads_:
                        ns = C.c_uint64(0)
                        if len(hbatches) < 3:
                            hbatches.append(reads(len(hbatches)).copy())
                        src = hbatches[sub_ % 3]
                        t0 = time.perf_counter()
                        V._check(V.lib.vsa_pack_reads_mt(
                            src.ctypes.data, nq, m, m, rows.ctypes.data,
                            special.ctypes.data, p.maxspecial, C.byref(ns),
                            max(packthreads_, 1)))
                        if not packthreads_:
                            packrate[0] = max(
                                packrate[0],
                                nq / (time.perf_counter() - t0))
                        job.ns[key] = int(ns.value)
                    V._check(V.lib.vsa_pipeline_submit_packed(
                        p._h, nq, job.ns[key]))
                else:
                    if refill:
                        slot[:nq * m] = reads(sub_ % 3)
                    p.submit(nq)
                sub_ += 1
            else:
                rc, mm = p.next(copy=False)
                total += len(mm)
                got += 1
        return total
    job.ns = {}

    def both(packed):
        out = {}
        p = V.Pipeline(index, 3, L, m, nq, packed=packed, maxspecial=1024)
        job(p, 3, True)                    # fills the three slots, warms up
        p.finish()
        t0 = time.perf_counter()
        job(p, 3, False)
        t1 = time.perf_counter()
        mums, st = p.finish(copy=False)   # a view of the page-locked list
        dt = time.perf_counter() - t0
        nmums = int(len(mums))
        log("end to end -mum (%s): batches %.1f ms, filter + list to the "
            "host %.1f ms" % ("packed" if packed else "bytes",
                              (t1 - t0) * 1e3, (dt - (t1 - t0)) * 1e3))
        p.close()
        up = nq * (V.lib.vsa_packed_words(m) * 8 if packed else m)
        out["mum"] = {
            "end_to_end_queries_per_s": 3 * nq / dt, "queries": 3 * nq,
            "ms": dt * 1e3, "mums": nmums,
            "candidates": int(st.candidates),
            "what": "vmatch -mum -l %d: 3 batches of %s reads from "
                    "page-locked host memory (%.2f GB each over PCIe) to the "
                    "MUM list of the whole job in host memory (%.2f GB), "
                    "global filter included"
                    % (L, human(nq), up / 1e9, nmums * 32 / 1e9)}
        if packed:
            # the MUM list at 16 bytes per match (vsa_pipeline_finish16): what
            # a -mum job waits for at its end is the list on the host link
            p = V.Pipeline(index, 3, L, m, nq, packed=True, maxspecial=1024)
            job(p, 3, True)
            p.finish(copy=False, compact=True)
            t0 = time.perf_counter()
            job(p, 3, False)
            mums, st = p.finish(copy=False, compact=True)
            dt = time.perf_counter() - t0
            assert int(len(mums)) == nmums
            p.close()
            out["mum16"] = {
                "end_to_end_queries_per_s": 3 * nq / dt, "queries": 3 * nq,
                "ms": dt * 1e3, "mums": nmums,
                "what": "as mum, the list as vsa_match16 (%.2f GB instead of "
                        "%.2f)" % (nmums * 16 / 1e9, nmums * 32 / 1e9)}
        p = V.Pipeline(index, 2, L, m, nq, packed=packed, maxspecial=1024)
        job(p, 3, True)
        t0 = time.perf_counter()
        total = job(p, 12, False)
        dt = time.perf_counter() - t0
        p.close()
        out["mumcand"] = {
            "end_to_end_queries_per_s": 12 * nq / dt, "queries": 12 * nq,
            "ms_per_batch": dt / 12 * 1e3, "matches": int(total),
            "what": "vmatch -mum cand -l %d: 12 batches of %s reads (%.2f GB "
                    "up each), every batch's candidate list (%.2f GB) back "
                    "in host memory"
                    % (L, human(nq), up / 1e9, total / 12 * 32 / 1e9)}
        if packed:
            # the same steady state with the packing of every batch inside
            # the clock: bytes in pageable host memory -> rows in the slot
            # (vsa_pack_reads_mt) -> PCIe -> search -> list in host memory
            p = V.Pipeline(index, 2, L, m, nq, packed=True, maxspecial=1024)
            job(p, 3, True)
            t0 = time.perf_counter()
            total = job(p, 12, False, packthreads_=packthreads)
            dt = time.perf_counter() - t0
            p.close()
            out["mumcand_incl_packing"] = {
                "end_to_end_queries_per_s": 12 * nq / dt,
                "ms_per_batch": dt / 12 * 1e3, "matches": int(total),
                "pack_threads": packthreads,
                "what": "as mumcand, but every batch starts as one byte per "
                        "symbol in pageable host memory and is packed into "
                        "its slot inside the timed region on %d host threads"
                        % packthreads}
        out["end_to_end_queries_per_s"] = \
            out["mum"]["end_to_end_queries_per_s"]
        return out

    out = both(True)
    out["reads"] = "two bits per symbol (vsa_pack_reads, %d bytes per read)" \
        % (V.lib.vsa_packed_words(m) * 8)
    out["pack_reads_per_s_one_host_thread"] = packrate[0]
    # ... and on the host threads of this GPU's share of the box
    W = int(V.lib.vsa_packed_words(m))
    scratch, sp = np.zeros(nq * W, np.uint64), np.zeros(1024 * m, np.uint8)
    best = 0.0
    for _ in range(3):
        ns = C.c_uint64(0)
        t0 = time.perf_counter()
        V._check(V.lib.vsa_pack_reads_mt(
            hbatches[0].ctypes.data, nq, m, m, scratch.ctypes.data,
            sp.ctypes.data, 1024, C.byref(ns), packthreads))
        best = max(best, nq / (time.perf_counter() - t0))
    out["pack_reads_per_s"] = {"threads": packthreads, "value": best}
    del scratch, sp
    out["bytes"] = both(False)
    V.device_free(dq, dev)
    return out


def selfmum_family(a, V, n, L, dev):
    """K3, the suftab/lcptab scan: an index of its own (database + separator +
    diverged copy as query part), built after the query index is gone"""
    tis, half = selfmum_text(V, n)
    t0 = time.time()
    idx = V.Index.build(tis, 4, 0, dev)
    idx.set_queryseparator(half)
    log("self-index text %d bp built in %.1fs" % (len(tis), time.time() - t0))
    got = []
    for _ in range(5):
        r = V.findmaximaluniquematches(idx, L)
        got.append(r.stats())
        r.close()
    s = min(got[1:], key=lambda x: x.search_kernel_ms)
    idx.close()
    return family(
        "k_selfmum_peaks<nontemporal, uint32_t>",
        "-mum -l %d on an index that holds its queries (the suftab scan, "
        "fmumself.c)" % L, s.search_kernel_ms, 2.0 * (len(tis) + 1),
        "streams lcptab and bwttab once: 2(n+1) bytes; the whole call "
        "(peaks -> suf gathers -> MUM list) %.2f ms, %d MUMs"
        % (s.total_device_ms, s.count), traffic_key="k_selfmum_peaks",
        matches=s.count, call_device_ms=s.total_device_ms)


def cpu_baselines(a, V, H, index, host, qsym, m, L, dev, qps, free_index):
    ns = min(a.cpu_sample, len(qsym) // m)
    host.sti1 = H.sti1_from_tables(host.suf, host.lcp, host.prefixlength)
    sample = H.Queries.uniform(qsym[:ns * m], m)
    t0 = time.perf_counter()
    ref = H.oracle_querymatches(host, sample, L, mum=True, speedup=2)
    dt = time.perf_counter() - t0
    # the same sample as a batch of its own on the GPU: identical list
    if a.reads == "packed":
        gsample = V.Queries.from_host_packed(sample.symbols, m, dev)
    else:
        gsample = V.Queries.from_host(sample.symbols, sample.start,
                                      sample.length, dev)
    gres = V.findquerymatches(index, gsample, L, mum=True)
    same = bool(np.array_equal(gres.fetch(), ref))
    gres.close()
    gsample.close()
    if not same:
        raise RuntimeError("bench.py: GPU and CPU oracle disagree on the "
                           "%d-query sample" % sample.nq)
    port = {"value": sample.nq / dt, "unit": "queries/s", "cores": 1,
            "kind": "port",
            "sample": "first %d queries of the same batch, same index "
                      "(32-bit tables), oracle/vsoracle.c algorithm 2 = the "
                      "reference's default -qspeedup 2 incl. the MUM filter, "
                      "engine only (no FASTA parsing, no output), %.1f s, %d "
                      "MUMs" % (sample.nq, dt, len(ref)),
            "gpu_list_equal_on_sample": same}
    # P: the host cores that go with one GPU of the box (gpurun: 16 per GPU;
    # the machine shows all its hardware threads to every box)
    ncores = min(len(os.sched_getaffinity(0)),
                 int(os.environ.get("VSA_BENCH_CORES", "16")))
    refb = None
    if not a.quick:
        genome = host.tis
        free_index()      # vsa_mkvtree builds its own copy on the GPU
        try:
            refb = reference_baseline(a, V, H, genome, qsym, m, L, ncores)
        except Exception as e:          # the baseline must not sink the line
            log("reference baseline failed: %r" % (e,))
        if refb is not None and refb["mums_1core_sample"] != len(ref):
            raise RuntimeError("bench.py: vmatch_ref reports %d MUMs on the "
                               "sample, GPU and port %d"
                               % (refb["mums_1core_sample"], len(ref)))
    if refb is None:
        port["speedup_gpu_vs_this"] = qps / port["value"]
        return port
    refb["port_1core"] = port
    refb["gpu_over_reference_all_cores"] = qps / refb["value"]
    if refb.get("reference_physical_cores"):
        refb["gpu_over_reference_physical_cores"] = \
            qps / refb["reference_physical_cores"]["value"]
    refb["gpu_over_reference_1core"] = qps / refb["reference_1core"]["value"]
    return refb


def selfmum_mode(a, V, S, torch, dist, rank, world, dev, jsonfd):
    """SURVEY 8e, third row: the self-index scan split into suffix-array
    ranges.  Every rank holds the whole index and scans its range (the entries
    around the range's ends come from its own replica); the lists stay
    distributed in suffix-array order; one all-reduce sums the counters.
    Strong scaling: the text is fixed, the ranges shrink with N."""
    from vstree_amd import sharding as Sh
    n, L = int(a.genome), a.minlen
    tis, half = selfmum_text(V, n)
    t0 = time.time()
    idx = V.Index.build(tis, 4, 0, dev)
    idx.set_queryseparator(half)
    t_index = time.time() - t0
    first, last = Sh.selfmum_range(len(tis), rank, world)
    distributed = world > 1 or a.force_distributed
    totals = None

    def sync():
        V.device_synchronize(dev)
        if distributed:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    kms = []

    def one_step():
        nonlocal totals
        r = V.findmaximaluniquematches(idx, L, first, last)
        s = r.stats()
        kms.append(s.search_kernel_ms)
        totals = [s.count, s.sumlength]
        r.close()
        if distributed:
            totals = Sh.all_reduce_counters(
                dist, torch, totals,
                "cpu" if a.rehearse_on_one_gpu else "cuda")

    for _ in range(a.warmup):
        one_step()
    kms.clear()
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
    if rank == 0:
        ms = float(np.mean(kms))
        nbytes = 2.0 * (last - first)
        out = {
            "metric": "suffix-array positions scanned per second (vmatch -mum "
                      "-l %d on an index that holds its queries, %s)"
                      % (L, human(len(tis), "bp")),
            "value": len(tis) * a.steps / elapsed, "unit": "positions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "self-index MUM scan (fmumself.c), %d bp "
                                   "text = database + separator + diverged "
                                   "copy, -l %d, suffix-array ranges x%d"
                                   % (len(tis), L, world),
                       "index_build_s": round(t_index, 2)},
            "matches": totals[0],
            "roofline": family("k_selfmum_peaks<nontemporal, uint32_t>",
                               "rank 0's range", ms, nbytes,
                               "lcptab + bwttab of the range, once")}
        sys.stdout.flush()
        os.write(jsonfd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
