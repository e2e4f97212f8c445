#!/bin/bash
# round 4, job 21: the kernels of one step and the gaps between them
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b21
mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --quick --cpu-sample 0 --steps 6 --warmup 2 > $O/trace_line.json 2> $O/trace.err
T=$(ls -S $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 $R/scripts/step_timeline.py $T > $O/step_timeline_r04.txt 2>&1
rm -rf $O/trace
cat $O/step_timeline_r04.txt | cut -c1-150
cd $R
for i in 1 2; do
  timeout -k 10 200 python bench.py --quick --cpu-sample 0 > $O/quick$i.json 2> $O/quick$i.err
  python3 -c "
import json
d=json.loads(open('$O/quick$i.json').read().strip().splitlines()[-1])
print('quick $i: step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step']))"
done
