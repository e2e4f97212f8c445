#!/bin/bash
# round 4, job 26: rows of more than four words in the first pass (150 bp)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b26
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for m in 150 100; do
  timeout -k 10 240 python bench.py --quick --cpu-sample 0 --qlen $m > $O/q$m.json 2> $O/q$m.err
  python3 -c "
import json
d=json.loads(open('$O/q$m.json').read().strip().splitlines()[-1])
print('m=$m: step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f  searches %d  matches %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step'], d['roofline']['searches_per_launch'], d['matches']))"
done
