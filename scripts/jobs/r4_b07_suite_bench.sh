#!/bin/bash
# round 4, job 7: the stripped + compressed library on the box (smoke, the
# variant that binds to torch's HIP runtime), the whole GPU suite, the
# default bench line
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4_b07
mkdir -p $O
cd $R
ls -la vstree_amd/*.so | awk '{print $5, $9}'
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 200 python -c "
import torch, sys
sys.path.insert(0, 'tests')
import vstree_amd as V
print('lib', V.LIBPATH)
import numpy as np
g = V.synth_genome(200000); ix = V.Index.build(g, 4, 0, 0)
q = V.synth_queries(g, 1000, 100)
r = V.findquerymatches(ix, V.Queries.from_host_packed(q, 100), 20, mum=True)
print('with torch loaded first:', r.count, 'MUMs')" > $O/nort.log 2>&1; echo "nort rc=$?"; tail -3 $O/nort.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=6 > $O/tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/tests.log | cut -c1-200
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
echo "bench rc=$?"; tail -5 $O/bench.err | cut -c1-200
python3 - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r4_b07"
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("step %.3f ms  %.3f G q/s  K2 %.3f ms frac %.3f  K3 frac %s  bytes form %.3f ms" % (
    d["ms_per_step"], d["value"] / 1e9, r["kernel_ms"], r["frac"], r.get("suftab_scan_frac"),
    d.get("reads_as_bytes", {}).get("ms_per_step", -1)))
e = d.get("end_to_end", {})
if e: print("e2e mum %.3f G q/s mumcand %.3f G q/s" % (e["mum"]["end_to_end_queries_per_s"] / 1e9, e["mumcand"]["end_to_end_queries_per_s"] / 1e9))
c = d.get("cpu_baseline", {})
print("cpu", c.get("value"), c.get("cores"), c.get("gpu_over_reference_all_cores"))
PY
