#!/bin/bash
# usage: pmc_ab.sh OUTDIR "COUNTERS of pass 1" ["COUNTERS of pass 2" ...]
# one rocprofv3 --pmc run of `bench.py --steps 1 --warmup 0 --cpu-sample 0`
# per counter group (environment switches such as VSA_SLOT are inherited);
# results under gpurun_out/OUTDIR/pN
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$1
shift
mkdir -p $R/gpurun_out/$OUT
i=0
for line in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $line --output-format csv -d $R/gpurun_out/$OUT/p$i -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --quick > $R/gpurun_out/$OUT/p$i.json 2> $R/gpurun_out/$OUT/p$i.err
  echo "pass $i rc=$? : $line" >> $R/gpurun_out/$OUT/progress.log
done
cat $R/gpurun_out/$OUT/progress.log
