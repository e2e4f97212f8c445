#!/bin/bash
# usage: pmc_quick.sh OUTDIR "ENV=.." "COUNTERS of pass 1" ["COUNTERS of pass 2" ...]
# rocprofv3 --pmc passes (counter collection only) over one batch of the
# headline workload, with the given environment switches
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$1
ENVS=$2
shift 2
mkdir -p $OUT
i=0
for line in "$@"; do
  i=$((i+1))
  export $ENVS
  timeout -k 10 200 rocprofv3 --pmc $line --output-format csv -d $OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $OUT/p$i.json 2> $OUT/p$i.err
  rc=$?
  echo "pass $i rc=$rc : $line" >> $OUT/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cat $OUT/progress.log
