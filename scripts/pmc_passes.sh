#!/bin/bash
# usage: pmc_passes.sh OUTDIR -- each pass is its own rocprofv3 run (counter
# collection only, no tracing), bench with 1 timed step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$1
mkdir -p $R/gpurun_out/$OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $R/gpurun_out/$OUT/p$i.json 2> $R/gpurun_out/$OUT/p$i.err || echo "pass $i failed: $line"
  echo "pass $i done: $line"
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_GATE_EN1_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_sum
TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUBBLE_sum
PASSES
