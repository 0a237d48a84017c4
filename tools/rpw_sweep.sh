#!/bin/bash
# Subpaths per wave of the walk kernel at the headline frame (FS_WALK_RAYS_PER_WAVE; 0 = the library's choice):
#   bash tools/rpw_sweep.sh 0 56 48 40 32
for r in "$@"; do
  FS_WALK_RAYS_PER_WAVE=$r timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$r failed"; tail -3 /tmp/o.err; continue; }
  python - "$r" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print('rays_per_wave', sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()}, 'parity', j.get('parity_rel_rms_max'))
PY
done
