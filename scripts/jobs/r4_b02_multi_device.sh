#!/bin/bash
# round 4, job 2: the device-resident N > 1 entry -- tests (golden cases, the
# 3 Gbp two-replica case), then A/B of the step: single-process form against
# the C form on one replica, two replicas of a deep-prefix-15 index on the one
# GPU (device-resident and host-memory entry)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b02
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_multi_fullscale.py -x -q -m gpu --durations=8 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/tests.log | cut -c1-200
line() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s n_gpus %d step %.3f ms  %.3f G q/s  K2 %.3f ms  matches %d rccl %s" % (
        sys.argv[1].split("/")[-1], d["n_gpus"], d["ms_per_step"], d["value"] / 1e9,
        d["roofline"]["kernel_ms"], d["matches"], d.get("rccl_ranks")))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
for i in 1 2; do
  timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/single_$i.json 2> $O/single_$i.err; line $O/single_$i.json
  timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 --gpus 1 --path c > $O/c1_$i.json 2> $O/c1_$i.err; line $O/c1_$i.json
done
export VSA_DEEP_PREFIX=15
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/single_d15.json 2> $O/single_d15.err; line $O/single_d15.json
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 --gpus 1 --path c > $O/c1_d15.json 2> $O/c1_d15.err; line $O/c1_d15.json
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --gpus 2 --replicas-on-one-gpu > $O/c2_d15.json 2> $O/c2_d15.err; line $O/c2_d15.json
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --gpus 2 --replicas-on-one-gpu --host > $O/c2_d15_host.json 2> $O/c2_d15_host.err; line $O/c2_d15_host.json
tail -3 $O/c2_d15.err | cut -c1-200
