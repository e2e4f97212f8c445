#!/bin/bash
# round 4, job 6: vsa_multi_pipeline_* (tests), big-index md5 parity (100 Mbp,
# 20 Mbp with repeats), the default bench line with packed reads, the C path on
# two replicas of a deep-prefix-15 index: device-resident, pipelines, compat
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b06
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_index_big.py -x -q -m gpu --durations=6 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -14 $O/tests.log | cut -c1-220
line() { python3 - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s n_gpus %d step %.3f ms  %.3f G q/s  matches %d | %s" % (
        sys.argv[1].split("/")[-1], d["n_gpus"], d["ms_per_step"], d["value"] / 1e9,
        d["matches"], d["config"].get("path", "single")[:70]))
    if "reads_as_bytes" in d: print("   bytes form:", round(d["reads_as_bytes"]["ms_per_step"], 3), "ms")
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/single.json 2> $O/single.err; line $O/single.json
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 --gpus 1 --path c > $O/c1.json 2> $O/c1.err; line $O/c1.json
export VSA_DEEP_PREFIX=15
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/single_d15.json 2> $O/single_d15.err; line $O/single_d15.json
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --gpus 2 --replicas-on-one-gpu > $O/c2_d15.json 2> $O/c2_d15.err; line $O/c2_d15.json
timeout -k 10 400 python bench.py --steps 6 --warmup 3 --gpus 2 --replicas-on-one-gpu --host > $O/c2_d15_pipeline.json 2> $O/c2_d15_pipeline.err; line $O/c2_d15_pipeline.json
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --gpus 2 --replicas-on-one-gpu --host --compat > $O/c2_d15_compat.json 2> $O/c2_d15_compat.err; line $O/c2_d15_compat.json
unset VSA_DEEP_PREFIX
timeout -k 10 400 python bench.py --steps 8 --warmup 3 --gpus 1 --path c --host > $O/c1_pipeline.json 2> $O/c1_pipeline.err; line $O/c1_pipeline.json
tail -2 $O/c2_d15_pipeline.err | cut -c1-200
