#!/bin/bash
# Occupancy experiment: the narrow frame kernel limited to 96 VGPRs (tools/tmp/w5, -DFS_FRAME_MIN_WAVES=5) with LDS stack
# caps that let 4 or 5 workgroups share a CU (safe: the deep store takes the overflow)
set -o pipefail
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
run() {  # name env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$name failed"; tail -3 /tmp/o.err; return 1; }
  python - "$name" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
}
run base FS_X=1 &&
run base_cap15 FS_STACK_ROWS_CAP=15 &&
cp tools/tmp/w5/libfrequensee.so audio-pathtracer_amd/libfrequensee.so &&
run w5_cap21 FS_X=1 &&
run w5_cap15 FS_STACK_ROWS_CAP=15 &&
run w5_cap13 FS_STACK_ROWS_CAP=13
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
