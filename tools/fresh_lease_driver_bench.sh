#!/bin/bash
# The driver's command — python3 bench.py --gpus 1 --steps 20 --warmup 5 — as the FIRST GPU work of N separate gpurun leases
# (every gpurun call is a fresh box).  Every line is kept unfiltered: gpurun_out/r05/fresh_<tag>_<i>.json; summarise with
# tools/fresh_lease_summary.py.  usage: tools/fresh_lease_driver_bench.sh <tag> <N>
tag=${1:-a}; n=${2:-8}
cd "$(dirname "$0")/.."
for i in $(seq 1 $n); do
  for attempt in 1 2 3 4 5 6; do
    /usr/local/graft/bin/gpurun --timeout 300 -- "mkdir -p gpurun_out/r05 && python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/fresh_${tag}_$i.json 2> gpurun_out/r05/fresh_${tag}_$i.err" > gpurun_out/fresh_${tag}_$i.log 2>&1
    rc=$?
    [ $rc -ne 3 ] && break      # 3 = no box free right now: nothing ran, try again
    sleep 60
  done
done
echo done > gpurun_out/fresh_${tag}.done
