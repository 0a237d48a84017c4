#!/bin/bash
# connect workgroups behind the walks (FS_FRAME_CONNECT_FIRST=-1) against the launch rule (unset) on the other configurations
set -o pipefail
run() { timeout -k 10 300 python3 bench.py "$@" --no-extra --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), round(d['ms_per_step'],4))"; }
for w in "--workload cfg4_old_mine_d12 --steps 100" "--workload cfg4_old_mine_1m_d12 --steps 50" "--workload cfg5_multi_source --steps 50" \
         "--workload cfg3_old_mine --fixed-depth --steps 100" "--workload cfg3_old_mine --deterministic --steps 100" \
         "--workload cfg3_old_mine --depth 0 --steps 100" "--workload cfg3_old_mine --depth 0 --frames-per-launch 4 --steps 100" \
         "--workload cfg3_old_mine --frames-per-launch 3 --steps 150" "--workload cfg3_old_mine --frames-per-launch 4 --steps 200" "--workload cfg3_old_mine --steps 200"; do
  for c in -1 0; do
    echo -n "$w  FS_FRAME_CONNECT_FIRST=$c : "
    if [ $c = 0 ]; then run $w || exit 1; else FS_FRAME_CONNECT_FIRST=$c run $w || exit 1; fi
  done
done
