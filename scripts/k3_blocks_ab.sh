#!/bin/bash
# K3 (k_selfmum_peaks): workgroups of the streaming pass, interleaved repeats
# (the first k3_variants.sh run drifted 7 % between its first and last line
# of the same configuration)
R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2 3; do
for blocks in 1024 768 512 896; do
  VSA_PEAKBLOCKS=$blocks timeout -k 10 200 python bench.py --mode selfmum --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rep $rep blocks $blocks: K3 kernel %.3f ms frac %.3f  step %.3f ms  matches %d' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['matches']))" | tee -a gpurun_out/r3_k3_blocks_ab.txt
done
done
