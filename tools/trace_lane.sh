set -e
mkdir -p gpurun_out/r05/lane
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name lane script args...
  name=$1; lane=$2; shift 2
  export FS_SYNC_LANE=$lane
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05/lane/$name -- python3 "$@" > /dev/null 2>&1
  python3 $R/tools/tick_trace_summary.py $R/gpurun_out/r05/lane/$name > $R/gpurun_out/r05/lane/$name.json
  find $R/gpurun_out/r05/lane/$name -name "*.csv" -delete
  echo "== $name lane=$lane"; python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/r05/lane/$name.json'))
print(d['first_kernel_start_to_last_kernel_end_us_median'], {k:v['us_median'] for k,v in d['kernels'].items()})"
}
run t128_off 0 $R/tools/tick_trace.py starter_room 128
run t128_48 48,0 $R/tools/tick_trace.py starter_room 128
run t128_64 64,0 $R/tools/tick_trace.py starter_room 128
run t32_40 40,78 $R/tools/tick_trace.py starter_room 32
run f_mine_off 0 $R/tools/uncapped_trace.py old_mine 262144
run f_mine_64 64,0 $R/tools/uncapped_trace.py old_mine 262144
run f_mine_56 56,0 $R/tools/uncapped_trace.py old_mine 262144
