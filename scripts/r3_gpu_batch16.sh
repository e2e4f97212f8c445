#!/bin/bash
# round 3, batch 16: randomised percent-threshold batches against the oracle
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 700 python scripts/stress_approx_mixed_probe.py 150 11 > gpurun_out/r3_stress_approx_mixed.log 2>&1
echo "rc $?"; tail -4 gpurun_out/r3_stress_approx_mixed.log
