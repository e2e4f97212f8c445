#!/bin/bash
# one GPU call: the whole -m gpu suite, the default bench line, the C path and
# the torch rehearsal on two replicas / ranks of a 1 Gbp index
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=6 > $O/r3_gputests2.log 2>&1
rc=$?
tail -12 $O/r3_gputests2.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 420 python bench.py > $O/r3_bench2.json 2> $O/r3_bench2.err
rc=$?; echo "bench rc=$rc"; tail -14 $O/r3_bench2.err
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 200 python bench.py --gpus 2 --path c --replicas-on-one-gpu --genome 1e9 --steps 3 --warmup 1 > $O/r3_cpath.json 2> $O/r3_cpath.err
rc=$?; echo "c path rc=$rc"; tail -3 $O/r3_cpath.err; cat $O/r3_cpath.json | cut -c1-600
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 200 python bench.py --gpus 2 --rehearse-on-one-gpu --genome 1e9 --steps 5 --warmup 2 --quick --cpu-sample 0 > $O/r3_rehearse2.json 2> $O/r3_rehearse2.err
rc=$?; echo "rehearsal rc=$rc"; tail -3 $O/r3_rehearse2.err; cat $O/r3_rehearse2.json | cut -c1-400
