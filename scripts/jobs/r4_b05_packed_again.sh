#!/bin/bash
# round 4, job 5: packed reads after the plan kernel's fix -- tests, step A/B,
# end-to-end jobs, the multi-device form on one replica (counter reduction
# through page-locked words)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b05
mkdir -p $O
cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py tests/test_gpu_multi.py -x -q -m gpu > $O/tests_packed.log 2>&1
echo "packed tests rc=$?"; tail -12 $O/tests_packed.log | cut -c1-220
for i in 1 2; do
for reads in bytes packed; do
  timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 --reads $reads > $O/step_${reads}_$i.json 2> $O/step_${reads}_$i.err
  python3 -c "
import json
d=json.loads(open('$O/step_${reads}_$i.json').read().strip().splitlines()[-1]); r=d['roofline']
f=[x for x in d['roofline_families'] if 'first' in x['kernel']]
print('$reads step %.3f ms  K2 %.3f ms  first %.3f ms  matches %d searches %d' % (d['ms_per_step'], r['kernel_ms'], f[0]['kernel_ms'] if f else -1, d['matches'], d['query_suffix_searches']))" | tee -a $O/step_ab.txt
done
timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 --gpus 1 --path c > $O/c1_$i.json 2> $O/c1_$i.err
python3 -c "
import json
d=json.loads(open('$O/c1_$i.json').read().strip().splitlines()[-1])
print('c path 1 replica step %.3f ms matches %d' % (d['ms_per_step'], d['matches']))" | tee -a $O/step_ab.txt
done
timeout -k 10 500 python bench.py --no-reference --cpu-sample 0 > $O/bench_full.json 2> $O/bench_full.err
echo "full bench rc=$?"; grep "end to end" $O/bench_full.err | cut -c1-200
python3 -c "
import json
d=json.loads(open('$O/bench_full.json').read().strip().splitlines()[-1]); e=d['end_to_end']
print('packed: mum %.3f G q/s, mumcand %.3f G q/s (%.2f ms/batch); pack %.1f M reads/s/thread' % (e['mum']['end_to_end_queries_per_s']/1e9, e['mumcand']['end_to_end_queries_per_s']/1e9, e['mumcand']['ms_per_batch'], e['pack_reads_per_s_one_host_thread']/1e6))
b=e['bytes']
print('bytes : mum %.3f G q/s, mumcand %.3f G q/s (%.2f ms/batch)' % (b['mum']['end_to_end_queries_per_s']/1e9, b['mumcand']['end_to_end_queries_per_s']/1e9, b['mumcand']['ms_per_batch']))"
