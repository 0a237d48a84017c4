#!/bin/bash
# SQ / TA / TD counters of the stand-alone walk kernel for several builds (tools/build_variant.sh), one counter group per
# run as the pool requires:  bash tools/pmc_ab.sh base v1 v2 ...   ->  gpurun_out/pmc_ab_<variant>.txt
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out
cp audio-pathtracer_amd/libfrequensee.so /tmp/pmc_base.so
for v in "$@"; do
  if [ $v = base ]; then cp /tmp/pmc_base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  rm -rf $out/pmc_ab_$v
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
    n=$(echo $c | tr " " "_" | cut -c1-30)
    timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/pmc_ab_$v/$n -- python3 bench.py --no-pipeline --steps 4 --warmup 1 --prewarm 5 --no-cpu-baseline --no-extra > $out/pmc_ab_${v}_$n.log 2>&1 || { echo "pmc pass $v $n failed"; tail -3 $out/pmc_ab_${v}_$n.log; }
  done
  python3 profiles/summarize_pmc.py $out/pmc_ab_$v > $out/pmc_ab_$v.txt
  echo "=== $v"; grep -A22 "== walk_kernel_shared<0, false>" $out/pmc_ab_$v.txt
  rm -rf $out/pmc_ab_$v
done
cp /tmp/pmc_base.so audio-pathtracer_amd/libfrequensee.so
