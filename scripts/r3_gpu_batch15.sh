#!/bin/bash
# round 3, batch 15: approximate matching over replicas (C multi-GPU path and
# the drop-in binary)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_dropin.py -x -v > $O/r3_b15_tests.log 2>&1
rc=$?
tail -5 $O/r3_b15_tests.log
exit $rc
