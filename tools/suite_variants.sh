mkdir -p gpurun_out/r05
log=gpurun_out/r05/suite_variants.log; : > $log
run() { echo "=== $*" >> $log; env "$@" timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 >> $log; }
run FS_STACK_ROWS_CAP=12
run FS_FRAME_CONNECT_FIRST=64
run FS_FUSED_DRAIN=0
run FS_FUSED_RECON=0
run FS_SYNC_LANE=0
run FS_CONNECT_AHEAD=1 FS_FLUSH_RECON_ON_COMPUTE=0
run FS_SYNC_LANE=40,60 FS_SYNC_STAGE_FROM=1
cat $log
