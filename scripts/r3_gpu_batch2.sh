#!/bin/bash
# one GPU call: the kernels touched since batch 1 (repeat family on 64-bit
# positions, bucket boundaries), the random probes on wide tables, the
# drop-in's start-up
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_fullscale.py tests/test_gpu_index_build.py tests/test_gpu_mkvtree.py tests/test_gpu_dropin.py -x -q --durations=5 > $O/r3_gputests3.log 2>&1
rc=$?
tail -12 $O/r3_gputests3.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
VSA_FORCE_WIDE=1 timeout -k 10 200 python scripts/stress_self_probe.py 120 7 > $O/r3_stress_self_wide.log 2>&1
echo "stress wide rc=$?"; tail -3 $O/r3_stress_self_wide.log
timeout -k 10 200 python scripts/stress_self_probe.py 80 8 > $O/r3_stress_self.log 2>&1
echo "stress rc=$?"; tail -3 $O/r3_stress_self.log
timeout -k 10 500 python scripts/dropin_startup_probe.py > $O/r3_dropin_startup.log 2>&1
echo "dropin probe rc=$?"; tail -30 $O/r3_dropin_startup.log | cut -c1-200
bash scripts/ab_tune.sh r3_ab10 early=VSA_TUNE=0 lateslot=VSA_TUNE=2097152 plan6=VSA_PLAN_MINBLK=6 plan8=VSA_PLAN_MINBLK=8 early2=VSA_TUNE=0
