#!/bin/bash
# round 3, batch 14: the N > 1 form with the rows placed on the index's
# stream (1) or on the default stream (0), same box, alternating
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
: > $O/r3_b14_ab.txt
for rep in 1 2 3; do
for v in single dist_stream1 dist_stream0; do
  extra="--force-distributed"; [ $v = single ] && extra=""
  own=1; [ $v = dist_stream0 ] && own=0
  VSA_GROUP_ON_INDEX_STREAM=$own timeout -k 10 400 python3 bench.py --quick --cpu-sample 0 --steps 30 --warmup 3 $extra > $O/r3_b14_$v.json 2> $O/r3_b14_$v.err
  echo "$v rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('$O/r3_b14_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep ms_per_step %.4f n_gpus %d rccl_ranks %s mums %d' % (d['ms_per_step'], d['n_gpus'], d.get('rccl_ranks'), d['matches']))" | tee -a $O/r3_b14_ab.txt
done
done
