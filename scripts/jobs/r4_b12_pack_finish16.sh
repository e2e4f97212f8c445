#!/bin/bash
# round 4, job 12: the host packer in C (PEXT, threads), the index of flagged
# rows at bits 8.., vsa_pipeline_finish16 -- tests that touch them, then the
# bench's end-to-end section (packing inside the clock, MUM list at 16 bytes)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b12
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_packed.py tests/test_gpu_pipeline.py tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q -m gpu --durations=5 > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -12 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py --no-reference --cpu-sample 0 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('step %.3f ms' % d['ms_per_step'])
e=d['end_to_end']
for k in ('mum','mum16','mumcand','mumcand_incl_packing'):
    x=e.get(k)
    if x: print(k, '%.3f G q/s' % (x['end_to_end_queries_per_s']/1e9), x.get('ms', x.get('ms_per_batch')))
print('pack 1 thread', e['pack_reads_per_s_one_host_thread'], e['pack_reads_per_s'])
print('bytes:', {k: e['bytes'][k]['end_to_end_queries_per_s'] for k in ('mum','mumcand')})"
grep "end to end" $O/bench.err | cut -c1-200
