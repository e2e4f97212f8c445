#!/bin/bash
# round 4, job 16: the three stress probes at length on the final sources
# (random inputs against the oracle, list by list, order included), narrow and
# forced-wide device tables
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b16
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py tests/test_gpu_approx.py -x -q -m gpu > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 330 python scripts/stress_probe.py 150 9001 > $O/stress_query_150_rounds.log 2>&1; echo "query rc=$?"; tail -1 $O/stress_query_150_rounds.log | cut -c1-200
VSA_FORCE_WIDE=1 timeout -k 10 200 python scripts/stress_probe.py 60 9002 > $O/stress_query_forcewide_60_rounds.log 2>&1; echo "query wide rc=$?"; tail -1 $O/stress_query_forcewide_60_rounds.log | cut -c1-200
timeout -k 10 300 python scripts/stress_approx_probe.py 80 9003 > $O/stress_approx_80_rounds.log 2>&1; echo "approx rc=$?"; tail -1 $O/stress_approx_80_rounds.log | cut -c1-200
timeout -k 10 200 python scripts/stress_self_probe.py 100 9004 > $O/stress_self_100_rounds.log 2>&1; echo "self rc=$?"; tail -1 $O/stress_self_100_rounds.log | cut -c1-200
