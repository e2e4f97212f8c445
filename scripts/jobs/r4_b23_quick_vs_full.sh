#!/bin/bash
# round 4, job 23: does the timed region of the default run differ from the
# --quick run's?  alternating, one box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b23
mkdir -p $O
cd $R
for i in 1 2; do
  timeout -k 10 200 python bench.py --quick --cpu-sample 0 > $O/quick$i.json 2> $O/quick$i.err
  timeout -k 10 300 python bench.py --no-reference --cpu-sample 0 > $O/full$i.json 2> $O/full$i.err
done
python3 -c "
import json
for f in ('quick1','full1','quick2','full2'):
    d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    print(f, 'step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step']))"
