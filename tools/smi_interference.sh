mkdir -p gpurun_out/r05
( for i in $(seq 1 400); do rocm-smi --showuse --showmemuse --showpower --json > gpurun_out/r05/smi_last.json 2>/dev/null; sleep 0.05; done ) &
SMI=$!
for i in 1 2 3 4 5 6; do
  FS_BENCH_DEBUG_STEPS=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/r05/smi$i.json 2> gpurun_out/r05/smi$i.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/r05/smi$i.json').read().strip().splitlines()[-1]); print($i, round(d['value']/1e6,1), round(d['ms_per_step'],4), round(d['roofline']['launch_period_ms'],3), round(d['roofline']['avg_launch_ms'],3))"
  grep -a "step deltas" gpurun_out/r05/smi$i.err
done
kill $SMI
