#!/bin/bash
# usage: pmc_passes.sh OUTDIR [bench args...] -- every pass is its own
# rocprofv3 run (counter collection only), each under its own timeout;
# progress goes to gpurun_out/OUTDIR/progress.log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$1
shift
mkdir -p $R/gpurun_out/$OUT
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  # a line that starts with "2 " runs two steps: the difference to the pass
  # with one step is the traffic of exactly one step (pmc_summary.py)
  steps=1
  case "$line" in "2 "*) steps=2; line=${line#2 };; esac
  # a line that starts with "F " runs the WHOLE bench (every kernel family,
  # the self-index scan; no reference program) whatever the arguments say
  args=("$@")
  case "$line" in "F "*) line=${line#F }; args=(--no-reference);; esac
  echo $steps > $R/gpurun_out/$OUT/p$i.steps
  timeout -k 10 300 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/bench.py --steps $steps --warmup 0 --cpu-sample 0 "${args[@]}" > $R/gpurun_out/$OUT/p$i.json 2> $R/gpurun_out/$OUT/p$i.err
  rc=$?
  echo "pass $i rc=$rc : $line" >> $R/gpurun_out/$OUT/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then break; fi
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
F FETCH_SIZE
F TCC_HIT_sum TCC_MISS_sum WRITE_SIZE
PASSES
cat $R/gpurun_out/$OUT/progress.log
