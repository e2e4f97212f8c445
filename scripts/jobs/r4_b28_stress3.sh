#!/bin/bash
# round 4, job 28: the query stress probe on the last sources (uniform batches
# of up to 160 symbols as rows: the windowed first pass and -complete)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b28
mkdir -p $O
cd $R
timeout -k 10 400 python scripts/stress_probe.py 160 31337 > $O/stress_query_windows.log 2>&1; echo "query rc=$?"; tail -1 $O/stress_query_windows.log | cut -c1-200
grep -c "uniform" $O/stress_query_windows.log
