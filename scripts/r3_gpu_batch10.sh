#!/bin/bash
# round 3, batch 10: approximate matching on wide tables incl. the tree path,
# the N > 1 form with page-locked metadata staging
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_approx.py -x -q > $O/r3_b10_tests.log 2>&1
rc=$?
tail -3 $O/r3_b10_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 800 python -m pytest tests/test_gpu_wide_fullscale.py -x -v --durations=4 > $O/r3_b10_wide.log 2>&1
rc=$?
tail -8 $O/r3_b10_wide.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in single dist; do
  extra=""; [ $v != single ] && extra="--force-distributed"
  timeout -k 10 400 python3 bench.py --quick --cpu-sample 0 --steps 30 --warmup 3 $extra > $O/r3_b10_$v.json 2> $O/r3_b10_$v.err
  echo "$v rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('$O/r3_b10_$v.json').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], d['n_gpus'], d.get('rccl_ranks'), d['matches'] if 'matches' in d else '')"
done
