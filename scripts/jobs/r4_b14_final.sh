#!/bin/bash
# round 4, job 14: the default bench under rocprofv3 --kernel-trace --stats
# (kernel averages next to the HIP-event times of the same run), the default
# bench without the tracer (the line of the round), the stress probes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b14
mkdir -p $O
cd /tmp
timeout -k 10 540 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py > $O/bench_line_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "rocprof bench rc=$?"
S=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$S" ] && cp $S $O/bench_kernel_stats.csv && head -40 $O/bench_kernel_stats.csv | cut -c1-160 | grep -E "k_query_search_planned|k_mum_first|k_mum_plan|k_selfmum_peaks|k_complete_search|k_mem_plan|k_apm_banded|k_mumf|onesweep" | cut -c1-150
rm -rf $O/prof
cd $R
timeout -k 10 420 python bench.py > $O/bench_line_final.json 2> $O/bench_final.err
echo "final bench rc=$?"
python3 -c "
import json
d=json.loads(open('$O/bench_line_final.json').read().strip().splitlines()[-1])
print('step %.3f ms  value %.3e' % (d['ms_per_step'], d['value']))
r=d['roofline']
print({k: r[k] for k in ('kernel','kernel_ms','frac','traffic','suftab_scan_frac') if k in r})
print(r.get('random_line_ceiling'))
print(r.get('step'))"
timeout -k 10 200 python scripts/stress_approx_probe.py 40 404 > $O/stress_bestof.log 2>&1; echo "approx stress rc=$?"; tail -2 $O/stress_bestof.log | cut -c1-200
timeout -k 10 240 python scripts/stress_probe.py 40 4040 > $O/stress_rows.log 2>&1; echo "query stress rc=$?"; tail -2 $O/stress_rows.log | cut -c1-200
