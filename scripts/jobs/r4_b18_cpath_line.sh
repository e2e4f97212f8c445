#!/bin/bash
# round 4, job 18: the line of the C path WITH its roofline (not --quick):
# two replicas on the one GPU, started directly and under the driver's launcher;
# then at full per-rank size with D = 15 (2 x 63.5 GB)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b18
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --gpus 2 --replicas-on-one-gpu --genome 3e8 --queries 2e6 --steps 5 --warmup 2 --cpu-sample 0 > $O/c2_small.json 2> $O/c2_small.err
echo "direct rc=$?"; tail -2 $O/c2_small.err | cut -c1-200
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --replicas-on-one-gpu --genome 3e8 --queries 2e6 --steps 5 --warmup 2 --cpu-sample 0 > $O/c2_small_launcher.json 2> $O/c2_small_launcher.err
echo "launcher rc=$?"; tail -2 $O/c2_small_launcher.err | cut -c1-200
VSA_DEEP_PREFIX=15 timeout -k 10 420 python bench.py --gpus 2 --replicas-on-one-gpu --steps 10 --warmup 3 --cpu-sample 0 > $O/c2_full_d15.json 2> $O/c2_full_d15.err
echo "full rc=$?"; tail -2 $O/c2_full_d15.err | cut -c1-200
python3 -c "
import json
for f in ('c2_small','c2_small_launcher','c2_full_d15'):
    try:
        d=json.loads(open('$O/'+f+'.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'no line', e); continue
    r=d['roofline']
    print(f, 'n_gpus', d['n_gpus'], 'step %.3f ms' % d['ms_per_step'], '%.3f G q/s' % (d['value']/1e9), 'kernel %.3f ms' % r['kernel_ms'], 'frac', r.get('frac'), 'bytes/search', r.get('bytes_per_search'), 'traffic', r.get('traffic'))
"
