#!/bin/bash
# Round-end measurement recipe (run on the GPU box from the repo root):
#   bash tools/profile_round.sh r02
# 1. rocprofv3 kernel trace of the bench command  -> profiles/<tag>_kernel_stats.csv
# 2. PMC passes, one counter group per run        -> profiles/<tag>_pmc_summary.txt + profiles/pmc.json (bench.py reads it)
# 3. the default bench line                       -> profiles/<tag>_bench_default.json
# Counters are collected in their own runs (never together with a trace), as the pool requires.  Everything is also
# left under gpurun_out/ (scratch).
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out profiles
export TMPDIR=/tmp
rm -rf $out/${tag}_trace $out/${tag}_pmc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err || { echo "trace failed"; tail -5 $out/${tag}_trace.err; exit 1; }
f=$(find $out/${tag}_trace -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" \
         "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $out/${tag}_pmc_$n.log 2>&1 || { echo "pmc pass $n failed"; tail -3 $out/${tag}_pmc_$n.log; exit 1; }
done
python3 profiles/summarize_pmc.py $out/${tag}_pmc > $out/${tag}_pmc_summary.txt
python3 profiles/summarize_pmc.py $out/${tag}_pmc --json cfg3_old_mine $tag
# the small configuration (bench.py reports it under "extra", a launch per frame): the counters its fractions need
rm -rf $out/${tag}_pmc_cfg2
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$(echo $c | tr " " "_" | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/${tag}_pmc_cfg2/$n -- python3 bench.py --workload cfg2_starter_room --frames-per-launch 1 --steps 3 --warmup 1 --no-cpu-baseline > $out/${tag}_pmc_cfg2_$n.log 2>&1 || { echo "cfg2 pmc pass $n failed"; tail -3 $out/${tag}_pmc_cfg2_$n.log; exit 1; }
done
python3 profiles/summarize_pmc.py $out/${tag}_pmc_cfg2 --json cfg2_starter_room $tag
timeout -k 10 600 python3 bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err || { echo "bench failed"; tail -5 $out/${tag}_bench_default.err; exit 1; }
cp $out/${tag}_kernel_stats.csv $out/${tag}_pmc_summary.txt $out/${tag}_bench_default.json profiles/ 2>/dev/null
cp profiles/pmc.json $out/pmc.json
cut -c1-160 $out/${tag}_kernel_stats.csv
grep -A28 "== walk_kernel_shared<0, false>" $out/${tag}_pmc_summary.txt
cat $out/${tag}_bench_default.json
