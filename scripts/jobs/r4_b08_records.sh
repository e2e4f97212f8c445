#!/bin/bash
# round 4, job 8: 64-byte bucket records at deep prefix D - 1 against 16-byte
# slots at D (same bytes): parity under VSA_DEEP_RECORD=1, then the step
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b08
mkdir -p $O
cd $R
VSA_DEEP_RECORD=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py tests/test_gpu_approx.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests_records.log 2>&1
echo "tests (records) rc=$?"; tail -6 $O/tests_records.log | cut -c1-200
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed.py -x -q -m gpu > $O/tests_slots.log 2>&1
echo "tests (slots, new small-bucket scan) rc=$?"; tail -3 $O/tests_slots.log | cut -c1-200
line() { python3 -c "
import json
d=json.loads(open('$1').read().strip().splitlines()[-1]); r=d['roofline']
f=[x for x in d['roofline_families'] if 'first' in x['kernel']]
print('$2 step %.3f ms  K2 %.3f ms  first %.3f ms  index %.1f GB  matches %d' % (d['ms_per_step'], r['kernel_ms'], f[0]['kernel_ms'] if f else -1, d['config']['index_bytes_hbm']/1e9, d['matches']))" | tee -a $O/records_ab.txt; }
for i in 1 2; do
  timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/slots_$i.json 2> $O/slots_$i.err; line $O/slots_$i.json "slot16 D=16      "
  VSA_DEEP_RECORD=1 timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/rec_$i.json 2> $O/rec_$i.err; line $O/rec_$i.json "rec64  D=15      "
done
VSA_DEEP_PREFIX=15 timeout -k 10 300 python bench.py --quick --cpu-sample 0 --steps 20 --warmup 5 > $O/slots_d15.json 2> $O/slots_d15.err; line $O/slots_d15.json "slot16 D=15      "
