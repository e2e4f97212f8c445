#!/bin/bash
# round 4, job 29: the other stress probes on the last sources (they share
# search_common.hip with the query path: the numbers that come back to the
# host through k_fetch_words), narrow and forced-wide tables
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b29
mkdir -p $O
cd $R
timeout -k 10 200 python scripts/stress_self_probe.py 80 424201 > $O/stress_self_last_sources.log 2>&1; echo "self rc=$?"; tail -1 $O/stress_self_last_sources.log | cut -c1-200
VSA_FORCE_WIDE=1 timeout -k 10 200 python scripts/stress_self_probe.py 40 424202 > $O/stress_self_forcewide_last_sources.log 2>&1; echo "self wide rc=$?"; tail -1 $O/stress_self_forcewide_last_sources.log | cut -c1-200
VSA_FORCE_WIDE=1 timeout -k 10 200 python scripts/stress_probe.py 60 424203 > $O/stress_query_forcewide_last_sources.log 2>&1; echo "query wide rc=$?"; tail -1 $O/stress_query_forcewide_last_sources.log | cut -c1-200
timeout -k 10 240 python scripts/stress_approx_mixed_probe.py 40 424204 > $O/stress_approx_mixed_last_sources.log 2>&1; echo "approx mixed rc=$?"; tail -1 $O/stress_approx_mixed_last_sources.log | cut -c1-200
