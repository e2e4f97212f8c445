#!/bin/bash
# round 3, batch 8: the N > 1 form with the own range kept out of the
# exchange and the partition in tiles -- tests, same-box A/B, timeline
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "partition or packed" > $O/r3_b8_tests.log 2>&1
rc=$?
if [ $rc -eq 0 ]; then timeout -k 10 600 python -m pytest tests/test_gpu_multi.py -x -q >> $O/r3_b8_tests.log 2>&1; rc=$?; fi
tail -4 $O/r3_b8_tests.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in single dist dist_oldpartition; do
  extra=""; [ $v != single ] && extra="--force-distributed"
  small=1; [ $v = dist_oldpartition ] && small=0
  VSA_PARTITION_SMALL=$small timeout -k 10 400 python3 bench.py --quick --cpu-sample 0 --steps 30 --warmup 3 $extra > $O/r3_b8_$v.json 2> $O/r3_b8_$v.err
  echo "$v rc=$?"
  python3 -c "
import json,sys
d=json.loads(open('$O/r3_b8_$v.json').read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], d['n_gpus'], d.get('rccl_ranks'), d['matches'] if 'matches' in d else '')"
done
cd /tmp
rm -rf /tmp/tl_d
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_d -- python3 $R/bench.py --quick --cpu-sample 0 --steps 4 --warmup 2 --force-distributed > $O/r3_b8_tl.json 2> $O/r3_b8_tl.err
f=$(ls /tmp/tl_d/*/*kernel_trace.csv | head -1)
python3 $R/scripts/step_timeline.py $f > $O/r3_b8_step_timeline_distributed.txt
tail -1 $O/r3_b8_step_timeline_distributed.txt
