#!/bin/bash
# one GPU call: counter passes, kernel statistics and timelines of the final
# sources
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
rm -rf $O/r3_pmc_final
bash scripts/pmc_passes.sh r3_pmc_final --quick | tail -12
cd /tmp
rm -rf /tmp/st
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $R/bench.py > $O/r3_bench_under_rocprof.json 2> $O/r3_bench_under_rocprof.err
echo "stats rc=$?"
f=$(ls /tmp/st/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/r3_bench_kernel_stats.csv && head -12 $f | cut -c1-160
for v in single distributed; do
  rm -rf /tmp/tl_$v
  extra=""; [ $v = distributed ] && extra="--force-distributed"
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$v -- python3 $R/bench.py --quick --cpu-sample 0 --steps 4 --warmup 2 $extra > $O/r3_tl_$v.json 2> $O/r3_tl_$v.err
  f=$(ls /tmp/tl_$v/*/*kernel_trace.csv | head -1)
  python3 $R/scripts/step_timeline.py $f > $O/r3_step_timeline_$v.txt
  tail -1 $O/r3_step_timeline_$v.txt
done
cd $R
for v in "" "--force-distributed"; do
  timeout -k 10 200 python3 bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 $v 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['matches'], d['candidates'])"
done
