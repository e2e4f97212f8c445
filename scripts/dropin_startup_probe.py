#!/usr/bin/env python3
"""Wall clock of the drop-in binary (integration/_build/vmatch_gpu: the
reference's vmatch with the GPU engine linked in) on a 3 Gbp index for 0.2 M,
2 M and 10 M reads of 100 bp, `-mum -l 20`, next to the unmodified reference
on 16 processes (1/16 of the reads each, wall = slowest): where the GPU
program's start-up (index files -> HBM + derived tables) is paid back.
VSA_TRACE=1 prints the phases of the upload.  Needs oracle/_ref and
integration/_build (built where /root/reference exists).
usage: dropin_startup_probe.py [genome bp] [workdir]"""
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench as B
import helpers as H
import vstree_amd as V

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000_000
base = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
m, L, P = 100, 20, 16
counts = [200_000, 2_000_000, 10_000_000]
wd = os.path.join(base, "vsa_dropin_%d" % os.getpid())
os.makedirs(wd)
out = {"index_bp": n, "runs": []}
try:
    g = V.synth_genome(n)
    B.write_fasta(wd + "/genome.fna", b">synthetic_genome seed=42\n", g)
    t0 = time.time()
    V.mkvtree([wd + "/genome.fna"], wd + "/genome.fna", integersize=64,
              withskp=False)
    out["mkvtree_gpu_s"] = round(time.time() - t0, 1)
    nq = max(counts)
    pos, sub, step = V.synth_query_plan(n, nq, m)
    qsym = np.zeros(nq * m, np.uint8)
    rows = qsym.reshape(nq, m)
    for c0 in range(0, nq, 1 << 20):
        c1 = min(nq, c0 + (1 << 20))
        rows[c0:c1] = g[pos[c0:c1, None].astype(np.int64) + np.arange(m)]
    hit = np.flatnonzero(sub != V.NO_SUBST)
    rows[hit, sub[hit]] = (rows[hit, sub[hit]] + step[hit]) & 3
    del g
    gpubin = os.path.join(ROOT, "integration", "_build", "vmatch_gpu")
    args = ["-mum", "-l", str(L), "-q"]
    # page the index files in once (both programs map them)
    B.write_queries(wd + "/warm.fna", qsym, m, 0, 20000)
    subprocess.run([H.VMATCH_REF] + args + ["warm.fna", "genome.fna"], cwd=wd,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for c in counts:
        B.write_queries(wd + "/q.fna", qsym, m, 0, c)
        env = dict(os.environ, VMATCH_GPU_TRACE="1", VSA_TRACE="1")
        t0 = time.time()
        p = subprocess.run([gpubin] + args + ["q.fna", "genome.fna"], cwd=wd,
                           env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE)
        tg = time.time() - t0
        mums = sum(1 for l in p.stdout.splitlines()
                   if l and not l.startswith(b"#"))
        trace = [l for l in p.stderr.decode().splitlines()
                 if l.startswith("vstree_amd:")]
        per = c // P
        for k in range(P):
            B.write_queries(wd + "/qp%d.fna" % k, qsym, m, k * per, per)
        t0 = time.time()
        procs = [subprocess.Popen([H.VMATCH_REF] + args +
                                  ["qp%d.fna" % k, "genome.fna"], cwd=wd,
                                  stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL) for k in range(P)]
        for q in procs:
            q.wait()
        tr = time.time() - t0
        out["runs"].append({"queries": c, "dropin_wall_s": round(tg, 2),
                            "dropin_rc": p.returncode, "mums": mums,
                            "reference_16_processes_wall_s": round(tr, 2),
                            "trace": trace})
        print("%9d reads: drop-in %.2f s (rc %d, %d MUMs), reference on 16 "
              "processes %.2f s" % (c, tg, p.returncode, mums, tr), flush=True)
        for l in trace:
            print("           " + l)
finally:
    shutil.rmtree(wd, ignore_errors=True)
print(json.dumps(out))
