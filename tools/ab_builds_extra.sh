#!/bin/bash
# A/B of experimental builds (tools/build_variant.sh) on the latency-bound lines of the bench: the headline, the waited-for
# uncapped frame and the reference's tick.  bash tools/ab_builds_extra.sh v1 v2 ...   (on the GPU box; one line per build)
set -o pipefail
mkdir -p gpurun_out
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base "$@" base; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps ${AB_STEPS:-100} --warmup 10 2>/tmp/o.err > /tmp/o.json || { echo "$v failed"; tail -3 /tmp/o.err; continue; }
  cp /tmp/o.json gpurun_out/ab_extra_$v.json
  python3 - "$v" <<'PY' | tee -a gpurun_out/ab_builds_extra.jsonl
import json,sys
j=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); e=j['extra']
tick=lambda s: {k: round(x['ms_per_tick_median'],4) for k,x in e['reference_tick'][s].items() if isinstance(x,dict) and 'ms_per_tick_median' in x}
print(json.dumps({"build": sys.argv[1], "Mrays": round(j['value']/1e6,1), "uncapped_waited_ms": round(e['unbounded']['unpipelined']['ms_per_frame'],4),
                  "uncapped_pipelined_M": round(e['unbounded']['pipelined']['rays_per_s']/1e6,1),
                  "tick_room": tick('starter_room'), "tick_mine": tick('old_mine'), "cfg2_ms": round(e['cfg2_starter_room']['ms_per_frame'],4),
                  "parity": (j.get('parity') or {}).get('max_rel_rms_per_band')}))
PY
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
