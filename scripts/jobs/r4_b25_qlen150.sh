#!/bin/bash
# round 4, job 25: -mum -l 20 on 10 M reads of 150 symbols (packed for the host
# link, expanded on the device: rows of five words) and of 124 (the longest
# read the kernels take from its row)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b25
mkdir -p $O
cd $R
for m in 150 124 100; do
  timeout -k 10 240 python bench.py --quick --cpu-sample 0 --qlen $m > $O/q$m.json 2> $O/q$m.err
  python3 -c "
import json
d=json.loads(open('$O/q$m.json').read().strip().splitlines()[-1])
print('m=$m: step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f  searches %d  matches %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step'], d['roofline']['searches_per_launch'], d['matches']))"
done
