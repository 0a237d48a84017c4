#!/bin/bash
# A/B of experimental builds (tools/build_variant.sh) on the GPU box: bash tools/ab_builds.sh v1 v2 ...
# AB_TEST=1: also run the closest-hit / energy parity subset of the GPU suite with every variant.
set -o pipefail
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps ${AB_STEPS:-200} --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$v failed"; tail -3 /tmp/o.err; continue; }
  python - "$v" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
  if [ "${AB_TEST:-0}" = 1 ]; then
    timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "brute_force or soups or baseline_configs or golden" 2>&1 | tail -2
  fi
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
