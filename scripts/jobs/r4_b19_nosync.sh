#!/bin/bash
# round 4, job 19: the step without the host wait behind the first pass --
# parity tests, the step's time, then the counter passes of these sources
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b19
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_fullscale.py tests/test_gpu_multi.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
  timeout -k 10 200 python bench.py --quick --cpu-sample 0 > $O/quick$i.json 2> $O/quick$i.err
  python3 -c "
import json
d=json.loads(open('$O/quick$i.json').read().strip().splitlines()[-1])
print('quick $i: step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step']))"
done
mkdir -p $R/gpurun_out/r4_pmc2
cd /tmp
bash $R/scripts/pmc_passes.sh r4_pmc2 --quick
cd $R
python3 scripts/pmc_summary.py gpurun_out/r4_pmc2 gpurun_out/r4_pmc2/bench_pmc_summary.txt --traffic gpurun_out/r4_pmc2/hbm_traffic.json > gpurun_out/r4_pmc2/summary.out 2>&1
echo "summary rc=$?"
rm -rf gpurun_out/r4_pmc2/p*/
