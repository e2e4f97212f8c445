#!/bin/bash
# round 3, final B: the whole -m gpu suite on the final sources, the default
# bench under rocprofv3 --kernel-trace --stats, one batch under the kernel
# tracer in both forms
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests -x -v -m gpu --durations=5 > $O/r3_gputests_last.log 2>&1
rc=$?
tail -9 $O/r3_gputests_last.log
if [ $rc -ne 0 ]; then exit 1; fi
cd /tmp
rm -rf /tmp/st
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py > $O/r3_bench_under_rocprof_last.json 2> $O/r3_bench_under_rocprof_last.err
echo "stats rc=$?"
f=$(ls /tmp/st/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/r3_bench_kernel_stats_last.csv
for v in single distributed; do
  rm -rf /tmp/tl_$v
  extra=""; [ $v = distributed ] && extra="--force-distributed"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -- python3 $R/bench.py --quick --cpu-sample 0 --steps 4 --warmup 2 $extra > $O/r3_tl_$v.json 2> $O/r3_tl_$v.err
  f=$(ls /tmp/tl_$v/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/step_timeline.py $f > $O/r3_step_timeline_${v}_last.txt
  tail -1 $O/r3_step_timeline_${v}_last.txt
done
