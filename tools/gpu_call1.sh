#!/bin/bash
# round-2 call 1: VALU issue microbench, TA/TCP/SQ counters of the current kernels, A/B of child-ordering variants
set -o pipefail
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip 2>/dev/null && timeout -k 5 60 /tmp/valu_issue_bench > $out/r02_valu_issue.jsonl 2>&1
cat $out/r02_valu_issue.jsonl
echo "== A/B"
bash tools/ab_builds.sh n1 a2 a1 n1a2 2>&1 | tee $out/r02_ab1.log
echo "== PMC"
rm -rf $out/r02_pmc1
for c in "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
         "TD_TD_BUSY_sum TD_TC_STALL_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH SQ_INSTS_SMEM"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $out/r02_pmc1/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined > $out/r02_pmc1_$n.log 2>&1 || { echo "pmc pass $n failed"; tail -3 $out/r02_pmc1_$n.log; }
done
python3 profiles/summarize_pmc.py $out/r02_pmc1 > $out/r02_pmc1_summary.txt
grep -A30 "== walk_kernel" $out/r02_pmc1_summary.txt
grep -A30 "== connect_kernel" $out/r02_pmc1_summary.txt
