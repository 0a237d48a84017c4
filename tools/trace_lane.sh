# kernel traces of waited-for uncapped frames / ticks under environment settings: bash tools/trace_lane.sh  (on the GPU box)
set -e
mkdir -p gpurun_out/r05/lane
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name "ENV=V ENV=V" script args...
  name=$1; envs=$2; shift 2
  for kv in $envs; do export "$kv"; done
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05/lane/$name -- python3 "$@" > /dev/null 2>&1
  python3 $R/tools/tick_trace_summary.py $R/gpurun_out/r05/lane/$name > $R/gpurun_out/r05/lane/$name.json
  find $R/gpurun_out/r05/lane/$name -name "*.csv" -delete
  echo "== $name $envs"; python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/r05/lane/$name.json'))
print(d['first_kernel_start_to_last_kernel_end_us_median'], {k:v['us_median'] for k,v in d['kernels'].items()})"
  for kv in $envs; do unset "${kv%%=*}"; done
}
run d_t32_room "" $R/tools/tick_trace.py starter_room 32
run d_t128_room "" $R/tools/tick_trace.py starter_room 128
run d_t32_mine "" $R/tools/tick_trace.py old_mine 32
run d_t128_mine "" $R/tools/tick_trace.py old_mine 128
run d_f_mine "" $R/tools/uncapped_trace.py old_mine 262144
run d_f_room "" $R/tools/uncapped_trace.py starter_room 262144
