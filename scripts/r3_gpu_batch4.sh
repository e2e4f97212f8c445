#!/bin/bash
# one GPU call on the final sources: the whole -m gpu suite, the counter
# passes (summarised on the box so that the bench line can quote them), the
# default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=5 > $O/r3_gputests4.log 2>&1
rc=$?
tail -10 $O/r3_gputests4.log
if [ $rc -ne 0 ]; then exit 1; fi
rm -rf $O/r3_pmc_final2
bash scripts/pmc_passes.sh r3_pmc_final2 --quick | tail -9
python3 scripts/pmc_summary.py $O/r3_pmc_final2 profiles/r03/bench_pmc_summary.txt --traffic profiles/hbm_traffic.json > /dev/null
cp profiles/hbm_traffic.json $O/r3_hbm_traffic.json
cp profiles/r03/bench_pmc_summary.txt $O/r3_bench_pmc_summary.txt
timeout -k 10 420 python bench.py > $O/r3_bench_final.json 2> $O/r3_bench_final.err
echo "bench rc=$?"; tail -8 $O/r3_bench_final.err | cut -c1-300
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_final.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("step %.3f ms  K2 %.3f ms frac %.3f traffic %s" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"]))
print(r.get("random_sector_ceiling")); print(r.get("step")); print({k:r["suftab_scan"][k] for k in ("kernel_ms","frac","traffic")})
PY
