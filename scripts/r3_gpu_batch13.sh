#!/bin/bash
# round 3, batch 13: the N > 1 form with rows placed on the index's stream while the split sizes travel;
# same-box A/B of the three metadata forms against the single-process step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "partition or packed" > $O/r3_b13_tests.log 2>&1
rc=$?
if [ $rc -eq 0 ]; then timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q >> $O/r3_b13_tests.log 2>&1; rc=$?; fi
tail -3 $O/r3_b13_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
: > $O/r3_b13_ab.txt
for rep in 1 2 3; do
for v in single dist_device; do
  extra=""; staging=1
  [ $v = dist_device ] && extra="--force-distributed"
  [ $v = dist_host_staged ] && extra="--force-distributed --meta-on-host"
  [ $v = dist_host_fresh ] && extra="--force-distributed --meta-on-host" && staging=0
  VSA_META_STAGING=$staging timeout -k 10 400 python3 bench.py --quick --cpu-sample 0 --steps 30 --warmup 3 $extra > $O/r3_b13_$v.json 2> $O/r3_b13_$v.err
  echo "$v rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('$O/r3_b13_$v.json').read().strip().splitlines()[-1])
print('$v rep $rep ms_per_step %.4f n_gpus %d rccl_ranks %s mums %d' % (d['ms_per_step'], d['n_gpus'], d.get('rccl_ranks'), d['matches']))" | tee -a $O/r3_b13_ab.txt
done
done
