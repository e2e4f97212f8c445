#!/bin/bash
# round 4, job 27: the final tree -- suite, timeline of one step, counter
# passes, the default bench (which quotes the traffic of these very sources)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b27
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -x -q -m gpu --durations=5 > $O/gpu_tests_final.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -8 $O/gpu_tests_final.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --quick --cpu-sample 0 --steps 6 --warmup 2 > $O/trace_line.json 2> $O/trace.err
T=$(ls -S $O/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$T" ] && python3 $R/scripts/step_timeline.py $T > $O/step_timeline_r04.txt 2>&1
rm -rf $O/trace
tail -3 $O/step_timeline_r04.txt
mkdir -p $R/gpurun_out/r4_pmc5
bash $R/scripts/pmc_passes.sh r4_pmc5 --quick > /dev/null
cd $R
python3 scripts/pmc_summary.py gpurun_out/r4_pmc5 gpurun_out/r4_pmc5/bench_pmc_summary.txt --traffic gpurun_out/r4_pmc5/hbm_traffic.json > gpurun_out/r4_pmc5/summary.out 2>&1
echo "pmc summary rc=$?"
rm -rf gpurun_out/r4_pmc5/p*/
cp gpurun_out/r4_pmc5/hbm_traffic.json profiles/hbm_traffic.json
timeout -k 10 420 python bench.py > $O/bench_line_final.json 2> $O/bench_final.err
echo "final bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_line_final.json').read().strip().splitlines()[-1])
print('step %.3f ms  value %.3e' % (d['ms_per_step'], d['value']))
r=d['roofline']
print({k: r[k] for k in ('kernel','kernel_ms','frac','traffic','suftab_scan_frac') if k in r})
print(r.get('random_line_ceiling'))
print(r.get('step'))
for f in d['roofline_families']: print('  %-50s %.3f ms frac %.3f  call %.3f ms' % (f['kernel'][:50], f['kernel_ms'], f['frac'], f.get('call_device_ms', -1)))
e=d['end_to_end']
print('150 bp:', d.get('mum_150bp'))
for k in ('mum','mum16','mumcand','mumcand_incl_packing'):
    print(k, '%.3f G q/s' % (e[k]['end_to_end_queries_per_s']/1e9))"
