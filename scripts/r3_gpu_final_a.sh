#!/bin/bash
# round 3, final A: counter passes on the final sources (summarised on the
# box, so that the bench that follows quotes them), the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -rf $O/r3_pmc_last
bash scripts/pmc_passes.sh r3_pmc_last --quick | tail -7
python3 scripts/pmc_summary.py $O/r3_pmc_last profiles/r03/bench_pmc_summary.txt --traffic profiles/hbm_traffic.json > /dev/null
cp profiles/hbm_traffic.json $O/r3_hbm_traffic_last.json
cp profiles/r03/bench_pmc_summary.txt $O/r3_bench_pmc_summary_last.txt
timeout -k 10 480 python bench.py > $O/r3_bench_last.json 2> $O/r3_bench_last.err
echo "bench rc=$?"; tail -4 $O/r3_bench_last.err | cut -c1-250
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_last.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("step %.3f ms  K2 %.3f ms frac %.3f traffic %s" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"]))
print(r.get("random_sector_ceiling")); print(r.get("step")); print({k:r["suftab_scan"][k] for k in ("kernel_ms","frac","traffic")})
for f in d["roofline_families"]: print("  %-48s %.3f ms frac %.3f traffic %s" % (f["kernel"][:48], f["kernel_ms"], f["frac"], f["traffic"]))
PY
