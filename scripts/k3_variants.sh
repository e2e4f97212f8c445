#!/bin/bash
# K3 (k_selfmum_peaks) under its experiment switches: pieces per tile
# (VSA_PEAKVARIANT bit 1) x workgroups (VSA_PEAKBLOCKS), 3 Gbp dense case
R=$GRAFT_REPO_ROOT
cd $R
for spec in "1 1024" "1 768" "1 1280" "3 1024" "3 1536" "3 2048" "1 1024"; do
  set -- $spec
  VSA_PEAKVARIANT=$1 VSA_PEAKBLOCKS=$2 timeout -k 10 200 python bench.py --mode selfmum --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('variant $1 blocks $2: K3 kernel %.3f ms frac %.3f  step %.3f ms  matches %d' % (r['kernel_ms'], r['frac'], d['ms_per_step'], d['matches']))"
done
