#!/bin/bash
# small-frame configurations against the frame kernel's register budget and the LDS stack cap
set -o pipefail
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base mw1; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  for cap in 21 64; do
    for w in cfg2_starter_room cfg4_old_mine_d12 cfg3_old_mine; do
      echo -n "$v cap=$cap $w : "
      FS_STACK_ROWS_CAP=$cap python bench.py --workload $w --steps 200 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
    done
  done
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
