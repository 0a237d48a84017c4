#!/bin/bash
# reconstruct kernel shapes against the fused frame kernel they share the chip with (default bench, 3 frames per launch)
set -o pipefail
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 30 2>/tmp/o.err > /tmp/o.json || { echo "$v failed"; tail -3 /tmp/o.err; continue; }
  python - "$v" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "impulse_response or golden" 2>&1 | tail -1
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
