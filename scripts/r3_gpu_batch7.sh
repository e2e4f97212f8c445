#!/bin/bash
# round 3, batch 7: approximate matching with thresholds 0 and > 0 in one
# batch; does the filter of batch i hide under the search of batch i+1?
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_approx.py -x -q \
    > gpurun_out/r3_approx_mixed.log 2>&1
echo "approx tests rc $?" && tail -3 gpurun_out/r3_approx_mixed.log
timeout -k 10 600 python scripts/overlap_probe.py --steps 20 \
    > gpurun_out/r3_overlap_probe.log 2>&1
echo "probe rc $?" && tail -6 gpurun_out/r3_overlap_probe.log
