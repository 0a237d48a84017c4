#!/bin/bash
# A/B of run-time knobs on the default bench: bash tools/ab_env.sh "A=1" "B=2 C=3" ...   (each setting once; FS_X=1 = defaults)
for v in "$@"; do
  echo -n "== $v : "
  env $v python bench.py --steps 300 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
