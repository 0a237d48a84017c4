#!/bin/bash
# Occupancy experiment (timing only: fewer stack rows than the tree's worst case are UNSAFE): frame kernel limited to
# 128 VGPRs (tools/tmp/w4, -DFS_FRAME_MIN_WAVES=4 -DFS_EXPERIMENTS) with LDS for 3 or 4 workgroups per CU
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
cp tools/tmp/w4/libfrequensee.so audio-pathtracer_amd/libfrequensee.so &&
run w4_rows_default FS_X=1 &&
run w4_rows24 FS_UNSAFE_STACK_ROWS=24 &&
run w4_rows22 FS_UNSAFE_STACK_ROWS=22 &&
run w4_rows20 FS_UNSAFE_STACK_ROWS=20
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
