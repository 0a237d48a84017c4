#!/bin/bash
# The BASELINE.json configurations other than the headline one, plus the mode variants, on one GPU:
#   bash tools/bench_other.sh > gpurun_out/bench_other.jsonl
# One JSON line per run (bench.py's own line); copy into profiles/rNN_bench_other.jsonl.
set -o pipefail
run() { timeout -k 10 400 python3 bench.py "$@" --no-extra 2>/dev/null | tail -1; }
run --workload cfg2_starter_room --steps 200 --warmup 20
run --workload cfg4_old_mine_d12 --steps 100 --warmup 10
run --workload cfg4_old_mine_1m_d12 --steps 50 --warmup 5
run --workload cfg5_multi_source --steps 50 --warmup 5
run --workload cfg3_old_mine --fixed-depth --steps 100 --warmup 10
run --workload cfg3_old_mine --deterministic --steps 100 --warmup 10 --no-cpu-baseline
run --workload cfg3_old_mine --all-connections --steps 30 --warmup 3
run --workload cfg3_old_mine --mis-balance --steps 30 --warmup 3
run --workload cfg3_old_mine --double-positions --steps 100 --warmup 10
run --workload cfg3_old_mine --depth 0 --steps 100 --warmup 10
