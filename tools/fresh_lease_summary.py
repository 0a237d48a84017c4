#!/usr/bin/env python3
"""Summarise tools/fresh_lease_driver_bench.sh: every fresh-lease run of the driver's command, unfiltered.
usage: tools/fresh_lease_summary.py <tag> [out.json]   (reads gpurun_out/r05/fresh_<tag>_*.json + gpurun_out/fresh_<tag>_*.log)"""
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
runs = []
for path in sorted(glob.glob(os.path.join(root, "gpurun_out", "r05", f"fresh_{tag}_*.json")), key=lambda p: int(re.findall(r"_(\d+)\.json$", p)[0])):
    i = int(re.findall(r"_(\d+)\.json$", path)[0])
    rec = {"lease": i}
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
        t = dict(d.get("timeline", {}))
        t.pop("note", None)
        rec.update({"value_rays_per_s": d["value"], "ms_per_step": d["ms_per_step"],
                    "launch_period_ms": d.get("roofline", {}).get("launch_period_ms"), "avg_launch_ms": d.get("roofline", {}).get("avg_launch_ms"),
                    "parity_rel_rms": (d.get("parity") or {}).get("rel_rms_per_band_max") if isinstance(d.get("parity"), dict) else None,
                    "timeline": t})
    except Exception as exc:   # a run that produced no line is a result too
        rec["error"] = repr(exc)
    log = os.path.join(root, "gpurun_out", f"fresh_{tag}_{i}.log")
    if os.path.exists(log):
        m = re.search(r"status=(\w+) rc=(\S+) charged=([\d.]+)s", open(log, errors="ignore").read())
        if m:
            rec["gpurun"] = {"status": m.group(1), "rc": m.group(2), "charged_s": float(m.group(3))}
    runs.append(rec)
vals = [r["value_rays_per_s"] for r in runs if "value_rays_per_s" in r]
out = {"command": "python3 bench.py --gpus 1 --steps 20 --warmup 5 (first GPU work of a fresh gpurun lease, one lease per run)",
       "tag": tag, "runs": runs,
       "summary": {"n": len(runs), "min_M": min(vals) / 1e6 if vals else None, "max_M": max(vals) / 1e6 if vals else None,
                   "below_850M": sum(1 for v in vals if v < 850e6)}}
txt = json.dumps(out, indent=1)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
print(json.dumps(out["summary"]), [round(v / 1e6, 1) for v in vals])
