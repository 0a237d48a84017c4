timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log >> gpurun_out/sweep.log
timeout -k 10 120 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e6, d['kernel_ms'])" >> gpurun_out/sweep.log || exit 1
