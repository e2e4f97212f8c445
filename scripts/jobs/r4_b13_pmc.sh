#!/bin/bash
# round 4, job 13: the counter passes of the final kernel sources (one
# rocprofv3 --pmc run per counter group, scripts/pmc_passes.sh) and their
# summary; profiles/hbm_traffic.json is written here and copied home
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_pmc
mkdir -p $O
cd /tmp
bash $R/scripts/pmc_passes.sh r4_pmc --quick
cd $R
python3 scripts/pmc_summary.py gpurun_out/r4_pmc $O/bench_pmc_summary.txt --traffic $O/hbm_traffic.json > $O/summary.out 2>&1
echo "summary rc=$?"; tail -3 $O/summary.out | cut -c1-600
# the raw csv files are large: keep the summaries only
rm -rf $O/p*/
ls -la $O
