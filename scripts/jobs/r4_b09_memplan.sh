#!/bin/bash
# round 4, job 9: the MEM work plan -- the suite (MEM lists in order: golden,
# stress, drop-in, 3 Gbp sample), then the families of the bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b09
mkdir -p $O
cd $R
timeout -k 10 800 python -m pytest tests -x -q -m gpu --ignore tests/test_gpu_wide_fullscale.py --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-220
timeout -k 10 300 python scripts/stress_probe.py 60 77 > $O/stress.log 2>&1; echo "stress rc=$?"; tail -3 $O/stress.log | cut -c1-200
timeout -k 10 500 python bench.py --no-reference --cpu-sample 0 > $O/bench_full.json 2> $O/bench_full.err
echo "full bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1])
print('step %.3f ms' % d['ms_per_step'])
for f in d['roofline_families']: print('  %-50s %.3f ms frac %.3f  call %.3f ms  matches %s' % (f['kernel'][:50], f['kernel_ms'], f['frac'], f.get('call_device_ms', -1), f.get('matches')))"
