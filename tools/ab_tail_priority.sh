#!/bin/bash
# the default bench line (no CPU leg) with ordinary (0) and highest-priority (1) tail / reverb streams, three runs each
for i in 1 2 3; do for m in 0 1; do
  FS_TAIL_STREAM_PRIORITY=$m python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); e=d['extra']['cfg2_starter_room']; r=d['extra']['reference_defaults']; u=d['extra']['unbounded']
print('tail priority $m:', round(d['value']/1e6,1), 'cfg2', round(e['ms_per_frame'],4), round(e['ms_per_frame_unpipelined'],4), round(e['ms_per_frame_at_frames_per_launch']['2'],4), 'reference-sized update', round(r['starter_room']['gpu_ms_per_update_median'],3), 'uncapped', round(u['pipelined']['rays_per_s']/1e6,1), round(u['pipelined_4_frames_per_launch']['rays_per_s']/1e6,1), 'one frame per launch', round(d['extra']['one_frame_per_launch']['value']/1e6,1))"
done; done
