#!/bin/bash
# round 4, job 17: smoke() and the whole -m gpu suite on the final tree
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b17
mkdir -p $O
cd $R
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 $O/smoke.log | cut -c1-200
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > $O/gpu_tests_final.log 2>&1
echo "tests rc=$?"; tail -12 $O/gpu_tests_final.log | cut -c1-200
