#!/bin/bash
# cfg4's per-GPU share (131 072 rays, depth 12) in N separate processes — round 4 saw one slow process in five
# (458 vs 580 M rays/s: 0.12 ms between launches of a stream whose launches crossed two hardware queues).
n=${1:-10}
for i in $(seq 1 $n); do
  timeout -k 10 200 python3 bench.py --workload cfg4_old_mine_d12 --steps 100 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timeline']; print(json.dumps({'process': $i, 'Mrays_per_s': round(d['value']/1e6,1), 'ms_per_step': round(d['ms_per_step'],4), 'frame_kernel_ms': round(d['kernel_ms']['frame'],4), 'hip_event_region_ms': t['hip_event_region_ms'], 'tail_stream_ops': t['library']['tail_stream_ops']}))"
done
