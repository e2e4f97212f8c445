#!/bin/bash
# round 4, job 11: the library cut into translation units per family -- the
# whole -m gpu suite (incl. the driver's launcher around the C path and its
# fallback), then the default bench with the new roofline keys
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b11
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $O/tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -14 $O/tests.log | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 700 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "default bench rc=$?"
tail -3 $O/bench_default.err | cut -c1-300
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print('step %.3f ms  value %.3e' % (d['ms_per_step'], d['value']))
r=d['roofline']
print({k: r[k] for k in ('kernel','kernel_ms','frac','traffic','suftab_scan_frac') if k in r})
print(r.get('random_line_ceiling'))
for f in d['roofline_families']: print('  %-50s %.3f ms frac %.3f  call %.3f ms  matches %s' % (f['kernel'][:50], f['kernel_ms'], f['frac'], f.get('call_device_ms', -1), f.get('matches')))
print(json.dumps(d.get('end_to_end'))[:1500])
print(json.dumps(d.get('cpu_baseline'))[:600])"
