#!/bin/bash
# longer runs of tools/stress.py on the default streamed configuration and its neighbours, several seeds each
it=${1:-2500}
off=${2:-0}
mkdir -p gpurun_out/r05
log=gpurun_out/r05/stress2.log; : > $log
for seed in $((101+off)) $((102+off)) $((103+off)) $((104+off)) $((105+off)) $((106+off)); do
  for cfg in "FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2" "FS_STRESS_PIPELINE=2 FS_STRESS_FPL=4" "FS_STRESS_PIPELINE=1"; do
    echo "=== seed $seed $cfg" >> $log
    env $cfg timeout -k 10 300 python3 tools/stress.py $it $seed >> $log 2>&1; echo "rc=$?" >> $log
  done
done
grep -c "^rc=0" $log; grep "^rc=\|^===" $log | paste - - | grep -v "rc=0"; grep "stress ok" $log | tail -2
