#!/bin/bash
# round 4, job 1: the unit of a random HBM read (64-B sector or 128-B line?),
# rate against table size, translation counters; a quick bench line of the box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b01
mkdir -p $O
cd $R
P=scripts/probes/_bin/line_probe
rocprofv3 -L > $O/counters.txt 2>&1
timeout -k 10 240 $P 64 all > $O/line_probe_64G.log 2>&1 && tail -25 $O/line_probe_64G.log | cut -c1-200 &&
timeout -k 10 240 $P 200 sizes > $O/line_probe_sizes.log 2>&1 && tail -30 $O/line_probe_sizes.log | cut -c60-200
for gb in 1 8 64; do
  for shape in one:1:64:0:0 one:2:128:0:64 one:2:256:0:128; do
    for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE"; do
      tag=$(echo "$gb $shape $grp" | tr ' :' '__' | cut -c1-60)
      timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$tag -- $P $gb $shape > $O/pmc_$tag.log 2>&1 || echo "pmc failed: $tag"
    done
  done
done
python3 - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4_b01"
rows = collections.OrderedDict()
for f in sorted(glob.glob(O + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    tag = f.split("/pmc_")[1].split("/")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_line"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        rows[(tag, k)] = (sum(v) / len(v), len(v))
with open(O + "/pmc_line_probe_summary.txt", "w") as out:
    for (tag, k), (v, n) in rows.items():
        out.write("%-62s %-34s %.4e (%d launches)\n" % (tag, k, v, n))
print(open(O + "/pmc_line_probe_summary.txt").read()[-3000:])
PY
timeout -k 10 400 python bench.py --quick > $O/bench_quick.json 2> $O/bench_quick.err
echo "bench rc=$?"; tail -3 $O/bench_quick.err | cut -c1-200
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4_b01"
d = json.loads(open(O + "/bench_quick.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("step %.3f ms  K2 %.3f ms frac %.3f" % (d["ms_per_step"], r["kernel_ms"], r["frac"]))
PY
