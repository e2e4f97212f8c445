#!/bin/bash
# round 4, job 22: the small numbers of a step come to the host through one
# tiny kernel that writes the pinned page -- parity tests, timeline, counter
# passes, default bench
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b22
mkdir -p $O
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu --ignore tests/test_gpu_wide_fullscale.py > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -4 $O/tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --quick --cpu-sample 0 --steps 6 --warmup 2 > $O/trace_line.json 2> $O/trace.err
T=$(ls -S $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 $R/scripts/step_timeline.py $T > $O/step_timeline_r04.txt 2>&1
rm -rf $O/trace
tail -1 $O/step_timeline_r04.txt
cd $R
for i in 1 2; do
  timeout -k 10 200 python bench.py --quick --cpu-sample 0 > $O/quick$i.json 2> $O/quick$i.err
  python3 -c "
import json
d=json.loads(open('$O/quick$i.json').read().strip().splitlines()[-1])
print('quick $i: step %.3f ms  K2 %.3f  first %.3f  bytes form %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_families'][0]['kernel_ms'], d['reads_as_bytes']['ms_per_step']))"
done
mkdir -p $R/gpurun_out/r4_pmc4
cd /tmp
bash $R/scripts/pmc_passes.sh r4_pmc4 --quick > /dev/null
cd $R
python3 scripts/pmc_summary.py gpurun_out/r4_pmc4 gpurun_out/r4_pmc4/bench_pmc_summary.txt --traffic gpurun_out/r4_pmc4/hbm_traffic.json > gpurun_out/r4_pmc4/summary.out 2>&1
echo "pmc summary rc=$?"
rm -rf gpurun_out/r4_pmc4/p*/
cp gpurun_out/r4_pmc4/hbm_traffic.json profiles/hbm_traffic.json
timeout -k 10 420 python bench.py > $O/bench_line_final.json 2> $O/bench_final.err
echo "final bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_line_final.json').read().strip().splitlines()[-1])
print('step %.3f ms  value %.3e' % (d['ms_per_step'], d['value']))
r=d['roofline']
print({k: r[k] for k in ('kernel','kernel_ms','frac','traffic','suftab_scan_frac') if k in r})
e=d['end_to_end']
for k in ('mum','mum16','mumcand','mumcand_incl_packing'):
    print(k, '%.3f G q/s' % (e[k]['end_to_end_queries_per_s']/1e9))"
