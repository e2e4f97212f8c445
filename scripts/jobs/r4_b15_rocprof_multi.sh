#!/bin/bash
# round 4, job 15: multi tests after the per-source copy streams; the default
# bench under rocprofv3 --kernel-trace --stats -- the stats file of the bench
# process itself (the tracer also follows the reference / drop-in children)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b15
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_fullscale.py tests/test_gpu_dropin.py -x -q -m gpu --durations=5 > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -10 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 540 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py > $O/bench_line_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "rocprof bench rc=$?"
ls -la $O/prof/*/*kernel_stats.csv | cut -c1-200
S=$(ls -S $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp $S $O/bench_kernel_stats.csv
rm -rf $O/prof
grep -E "k_query_search_planned|k_mum_first|k_mum_plan|k_selfmum_peaks" $O/bench_kernel_stats.csv | cut -c1-80,200-
