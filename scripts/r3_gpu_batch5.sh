#!/bin/bash
# one GPU call on the final sources: K3 tests and timing, counter passes
# (summarised on the box), the default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_fullscale.py tests/test_gpu_multi.py -x -q > $O/r3_gputests5.log 2>&1
rc=$?
tail -4 $O/r3_gputests5.log
if [ $rc -ne 0 ]; then exit 1; fi
for i in 1 2; do timeout -k 10 200 python bench.py --mode selfmum --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('K3 kernel %.3f ms frac %.3f step %.3f ms matches %d' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['matches']))"; done
rm -rf $O/r3_pmc_final3
bash scripts/pmc_passes.sh r3_pmc_final3 --quick | tail -7
python3 scripts/pmc_summary.py $O/r3_pmc_final3 profiles/r03/bench_pmc_summary.txt --traffic profiles/hbm_traffic.json > /dev/null
cp profiles/hbm_traffic.json $O/r3_hbm_traffic.json
cp profiles/r03/bench_pmc_summary.txt $O/r3_bench_pmc_summary.txt
timeout -k 10 420 python bench.py > $O/r3_bench_final2.json 2> $O/r3_bench_final2.err
echo "bench rc=$?"; tail -6 $O/r3_bench_final2.err | cut -c1-250
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_bench_final2.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("step %.3f ms  K2 %.3f ms frac %.3f traffic %s" % (d["ms_per_step"], r["kernel_ms"], r["frac"], r["traffic"]))
print(r.get("random_sector_ceiling")); print(r.get("step")); print({k:r["suftab_scan"][k] for k in ("kernel_ms","frac","traffic")})
for f in d["roofline_families"]: print("  %-48s %.3f ms frac %.3f traffic %s" % (f["kernel"][:48], f["kernel_ms"], f["frac"], f["traffic"]))
PY
