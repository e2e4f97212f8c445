#!/bin/bash
# round 4, job 24: the stress probes once more on the very last sources
# (other seeds), then the whole suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b24
mkdir -p $O
cd $R
timeout -k 10 260 python scripts/stress_probe.py 120 777001 > $O/stress_query_last_sources.log 2>&1; echo "query rc=$?"; tail -1 $O/stress_query_last_sources.log | cut -c1-200
timeout -k 10 200 python scripts/stress_approx_probe.py 60 777003 > $O/stress_approx_last_sources.log 2>&1; echo "approx rc=$?"; tail -1 $O/stress_approx_last_sources.log | cut -c1-200
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=5 > $O/gpu_tests_final.log 2>&1
echo "tests rc=$?"; tail -8 $O/gpu_tests_final.log | cut -c1-200
