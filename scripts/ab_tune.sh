#!/bin/bash
# usage: ab_tune.sh OUTDIR NAME="ENV=.. ENV=.." [NAME="..." ...]
# one `bench.py --quick --cpu-sample 0` per variant (the environment switches
# of the library: VSA_TUNE, VSA_SLOT, ...); JSON lines under
# gpurun_out/OUTDIR/NAME.json, one summary line each on stdout.  Stops at the
# first variant that times out (no GPU step behind a killed one).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1
shift
mkdir -p $OUT
STEPS=${AB_STEPS:-20}
EXTRA=${AB_ARGS:-}
for spec in "$@"; do
  name=${spec%%=*}
  envs=${spec#*=}
  env $envs timeout -k 10 ${AB_TIMEOUT:-240} python3 $R/bench.py --quick --cpu-sample 0 \
      --steps $STEPS --warmup 5 $EXTRA > $OUT/$name.json 2> $OUT/$name.err
  rc=$?
  python3 - "$name" "$rc" "$OUT/$name.json" <<'PY'
import json, sys
name, rc, path = sys.argv[1:]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    f = (d.get("roofline_families") or [{}])[0]
    print("%-14s rc=%s step %.3f ms  search %.3f ms  first %.3f ms  matches %d cand %d searches %d"
          % (name, rc, d["ms_per_step"], r.get("kernel_ms", -1),
             f.get("kernel_ms", -1), d["matches"], d["candidates"],
             d["query_suffix_searches"]))
except Exception as e:
    print("%-14s rc=%s no line (%r)" % (name, rc, e))
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "variant $name timed out: stopping"
    exit 1
  fi
done
