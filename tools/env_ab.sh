#!/bin/bash
# A/B of a run-time switch of the library on the GPU box: bash tools/env_ab.sh FS_PLAN_OVERLAP 1 0 1 0
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps ${AB_STEPS:-400} --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$var=$v failed"; tail -3 /tmp/o.err; continue; }
  python - "$var=$v" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
done
