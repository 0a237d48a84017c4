#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cp tools/tmp/x0/libfrequensee.so audio-pathtracer_amd/libfrequensee.so
for rows in 33 32 31 30 29 28 27 25 23 21 19; do
  FS_UNSAFE_STACK_ROWS=$rows timeout -k 10 200 python bench.py --no-cpu-baseline --no-pipelined --steps 200 --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$rows failed"; tail -3 /tmp/o.err; continue; }
  python - "$rows" <<'PY' | tee -a $out/r02_rows_sweep.log
import json,sys
j=json.load(open('/tmp/o.json')); print('rows', sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
done
