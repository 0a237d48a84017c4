#!/usr/bin/env python3
"""The driver's bench command several times in a row INSIDE ONE LEASE (fresh process each, with --no-cpu-baseline --no-extra): value,
ms per step and the launch period — a 20-step timed region is 6 ms, so anything that stalls the producer once shows.
NOT the driver's situation: the driver's process is the first GPU work of a fresh lease and runs the command without
the two switches — tools/fresh_lease_driver_bench.sh (+ tools/fresh_lease_summary.py) reproduces that, one gpurun lease per run.
usage: python tools/repeat_driver_bench.py [runs [KEY=VALUE ...]]   (environment of the bench processes)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 6
env = dict(os.environ)
env.update(dict(a.split("=", 1) for a in sys.argv[2:]))
vals = []
for i in range(runs):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-extra"],
                       capture_output=True, text=True, timeout=300, env=env)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("run", i, "failed", r.stderr[-300:], flush=True)
        continue
    d = json.loads(line[-1])
    for l in r.stderr.splitlines():
        if "stalls:" in l or "step deltas" in l or "flush " in l:
            print("   ", l.strip(), flush=True)
    vals.append(round(d["value"] / 1e6, 1))
    print(json.dumps({"run": i, "M_rays_per_s": round(d["value"] / 1e6, 1), "ms_per_step": round(d["ms_per_step"], 4), "kernel_ms": d.get("kernel_ms"),
                      "launch_period_ms": (d.get("roofline") or {}).get("launch_period_ms")}), flush=True)
print(json.dumps({"env": sys.argv[2:], "M_rays_per_s": vals}), flush=True)
