#!/bin/bash
# A/B of experimental builds: bash tools/ab_builds.sh b8 b16 b24
set -o pipefail
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null > /tmp/o.json || { echo "$v failed"; continue; }
  python - "$v" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), j['kernel_ms'], 'pipelined', round(j['pipelined']['value']/1e6,1))
PY
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
