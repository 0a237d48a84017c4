# A/B of host-builder variants on the default bench (rays/s in M, ms per frame, ms per depth-0 frame); each twice
set -e
for v in "FS_X=1" "FS_BVH_DP_COLLAPSE=1" "FS_STACK_ROWS_CAP=20" "FS_STACK_ROWS_CAP=22" "FS_X=1" "FS_BVH_DP_COLLAPSE=1" "FS_STACK_ROWS_CAP=20" "FS_STACK_ROWS_CAP=22"; do
  echo "== $v"
  env $v python bench.py --steps 300 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), d['ms_per_step'], d['extra']['unbounded']['pipelined']['ms_per_frame'])"
done
