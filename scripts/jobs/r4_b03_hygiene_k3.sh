#!/bin/bash
# round 4, job 3: the whole GPU suite on the pruned sources; the one-tile K3
# kernel at 4/5/6/8 wavefronts per SIMD against the two-tile kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b03
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -9 $O/tests.log | cut -c1-200
VSA_PEAK_ROLLING=5 timeout -k 10 300 python -m pytest tests -x -q -m gpu -k "selfmum or self_index or scan" > $O/tests_rolling5.log 2>&1
echo "rolling tests rc=$?"; tail -3 $O/tests_rolling5.log | cut -c1-200
for rep in 1 2; do
for w in 0 4 5 6 8; do
  VSA_PEAK_ROLLING=$w timeout -k 10 200 python bench.py --mode selfmum --steps 10 --warmup 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('rolling $w  K3 %.3f ms  frac %.3f  step %.3f ms  matches %d' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['matches']))" | tee -a $O/k3_rolling_ab.txt
done
done
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/single.json 2> $O/single.err
python3 -c "
import json
d=json.loads(open('$O/single.json').read().strip().splitlines()[-1])
print('single step %.3f ms K2 %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))"
