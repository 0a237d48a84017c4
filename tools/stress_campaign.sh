#!/bin/bash
# tools/stress.py in the configurations the publish / pipeline machinery distinguishes (run on the GPU box):
#   bash tools/stress_campaign.sh [iterations=500] [seed offset=0]   ->  gpurun_out/r05/stress.log
it=${1:-500}
off=${2:-0}
mkdir -p gpurun_out/r05
log=gpurun_out/r05/stress.log; : > $log
run() { echo "=== $*" >> $log; env "$@" timeout -k 10 400 python3 tools/stress.py $it $seed >> $log 2>&1; echo "rc=$?" >> $log; }
seed=$((11+off)); run FS_STRESS_PIPELINE=0
seed=$((12+off)); run FS_STRESS_PIPELINE=1
seed=$((13+off)); run FS_STRESS_PIPELINE=2
seed=$((14+off)); run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2
seed=$((15+off)); run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=4
seed=$((16+off)); run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=3 FS_STACK_ROWS_CAP=12
seed=$((17+off)); run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2 FS_FUSED_DRAIN=0
seed=$((18+off)); run FS_STRESS_PIPELINE=2 FS_STRESS_FPL=2 FS_FUSED_RECON=0
seed=$((19+off)); run FS_STRESS_PIPELINE=1 FS_STRESS_FPL=2 FS_FRAME_CONNECT_FIRST=64
grep -c "^rc=0" $log; grep "^rc=\|^===" $log | paste - - | grep -v "rc=0" ; tail -2 $log
