#!/bin/bash
# tools/stress.py in the configurations the publish / pipeline machinery distinguishes (run on the GPU box):
#   bash tools/stress_campaign.sh [iterations=500]   ->  gpurun_out/r05/stress.log
it=${1:-500}
mkdir -p gpurun_out/r05
log=gpurun_out/r05/stress.log; : > $log
run() { echo "=== $*" >> $log; env "$@" timeout -k 10 400 python3 tools/stress.py $it $seed >> $log 2>&1; echo "rc=$?" >> $log; }
seed=11; run FS_STRESS_PIPELINE=0
seed=12; run FS_STRESS_PIPELINE=1
seed=13; run FS_STRESS_PIPELINE=2
seed=14; run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2
seed=15; run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=4
seed=16; run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=3 FS_STACK_ROWS_CAP=12
seed=17; run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2 FS_FUSED_DRAIN=0
seed=18; run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2 FS_FUSED_RECON=0
seed=19; run FS_STRESS_PIPELINE=1 FS_STRESS_FPL=2 FS_FRAME_CONNECT_FIRST=64
grep -c "^rc=0" $log; grep "^rc=\|^===" $log | paste - - | grep -v "rc=0" ; tail -2 $log
