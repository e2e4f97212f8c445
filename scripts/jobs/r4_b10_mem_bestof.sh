#!/bin/bash
# round 4, job 10: best-of thresholds (golden lists, stress, drop-in), the MEM
# plan on packed batches, MEM timing
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b10
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests/test_gpu_approx.py tests/test_gpu_packed.py tests/test_gpu_dropin.py tests/test_gpu_multi.py tests/test_gpu_sink.py -x -q -m gpu --durations=5 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-220
timeout -k 10 500 python bench.py --no-reference --cpu-sample 0 > $O/bench_full.json 2> $O/bench_full.err
echo "full bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1])
print('step %.3f ms' % d['ms_per_step'])
for f in d['roofline_families']: print('  %-50s %.3f ms frac %.3f  call %.3f ms  matches %s' % (f['kernel'][:50], f['kernel_ms'], f['frac'], f.get('call_device_ms', -1), f.get('matches')))"
