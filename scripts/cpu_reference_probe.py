#!/usr/bin/env python3
"""The REAL reference vmatch (oracle/_ref/vmatch_ref, built from the reference
sources in the build container) timed on the GPU box's host cores on the
headline index: the synthetic genome is written as FASTA, `vsa_mkvtree` builds
the index on the GPU and writes the reference's files (byte-identical to
mkvtree's, tests/test_gpu_mkvtree.py), and vmatch_ref answers a sample of the
same queries.  Next to it: the CPU port (oracle) and the GPU on that sample.
usage: cpu_reference_probe.py N NQ [WORKDIR]"""
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vstree_amd as V  # noqa: E402
import helpers as H  # noqa: E402

n = int(float(sys.argv[1]))
nq = int(float(sys.argv[2]))
base = sys.argv[3] if len(sys.argv) > 3 else "/dev/shm"
wd = os.path.join(base, "vsa_cpuref_%d" % os.getpid())
need = 16 * n + (1 << 30)
free = shutil.disk_usage(base).free
print("work directory %s: %.0f GB free, %.0f GB needed" % (
    base, free / 1e9, need / 1e9), flush=True)
if free < need or not os.access(H.VMATCH_REF, os.X_OK):
    sys.exit("not enough space or no oracle/_ref/vmatch_ref")
os.makedirs(wd)
try:
    m, L = 100, 20
    t0 = time.time()
    g = V.synth_genome(n)
    letters = np.frombuffer(b"acgt", np.uint8)
    with open(wd + "/genome.fna", "wb") as f:
        f.write(b">synthetic_genome seed=42\n")
        width = 1 << 20
        for i in range(0, n, width << 6):
            chunk = letters[g[i:i + (width << 6)]]
            k = (len(chunk) // width) * width
            if k:
                rows = chunk[:k].reshape(-1, width)
                nl = np.full((rows.shape[0], 1), 10, np.uint8)
                f.write(np.hstack([rows, nl]).tobytes())
            if k < len(chunk):
                f.write(chunk[k:].tobytes() + b"\n")
    qb = V.synth_queries(g, nq, m)
    rows = letters[qb].reshape(nq, m)
    with open(wd + "/queries.fna", "wb") as f:
        for i in range(nq):
            f.write(b">q%d\n" % i)
            f.write(rows[i].tobytes())
            f.write(b"\n")
    print("FASTA files written in %.1fs" % (time.time() - t0), flush=True)
    t0 = time.time()
    V.mkvtree([wd + "/genome.fna"], wd + "/genome.fna", integersize=64,
              withskp=False)
    print("vsa_mkvtree (GPU build + %d-bit files): %.1fs" % (
        64, time.time() - t0), flush=True)
    env = dict(os.environ, VMATCHSHOWTIMESPACE="on")
    for name, args in (
            ("-mum -l 20", ["-mum", "-l", str(L)]),
            ("-mum cand -l 20", ["-mum", "cand", "-l", str(L)]),
            ("-complete", ["-complete"])):
        for rep in range(2):    # first run pages the index in
            t0 = time.time()
            p = subprocess.run([H.VMATCH_REF] + args + [
                "-q", "queries.fna", "genome.fna"], cwd=wd, env=env,
                stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            dt = time.time() - t0
            lines = p.stdout.decode().splitlines()
            nm = sum(1 for l in lines if l and not l.startswith("#"))
            tl = [l for l in lines if "TIME" in l or "overall" in l]
            print("vmatch_ref %s, run %d: rc %d, %d matches, wall %.2fs = "
                  "%.1f k queries/s %s" % (name, rep, p.returncode, nm, dt,
                                           nq / dt / 1e3, tl[-1:] ), flush=True)
    # the same sample on the GPU (index built again from the text)
    idx = V.Index.build(g, 4, 0)
    q = V.Queries.from_host(qb, np.arange(nq, dtype=np.uint64) * m,
                            np.full(nq, m, np.uint64))
    for rep in range(2):
        r = V.findquerymatches(idx, q, L, mum=True)
        s = r.stats()
        print("GPU -mum -l 20: %d matches, %.2f ms" % (s.count,
                                                       s.total_device_ms),
              flush=True)
finally:
    shutil.rmtree(wd, ignore_errors=True)
