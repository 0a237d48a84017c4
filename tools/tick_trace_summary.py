#!/usr/bin/env python3
"""Summarise the kernel trace of tools/tick_trace.py: per tick the kernels' durations and the gaps between them.
usage: python tools/tick_trace_summary.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import json
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
KNOWN = ["plan_kernel", "walk_kernel_lane", "walk_kernel_coop_big", "walk_kernel_coop", "walk_kernel_sparse", "walk_kernel_shared", "connect_kernel",
         "reconstruct_batch_kernel", "reconstruct_kernel", "fixed_to_energy", "coop16_kernel", "coop_nodes_kernel"]
rows = []
for r in csv.DictReader(open(f)):
    full = r["Kernel_Name"]
    name = next((k for k in KNOWN if k in full), full[:40])
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name))
rows.sort()
ticks, cur = [], []
for st, en, name in rows:
    if name.startswith("plan_kernel") and cur:
        ticks.append(cur); cur = []
    cur.append((st, en, name))
ticks.append(cur)
ticks = [t for t in ticks if any(n.startswith("plan_kernel") for _, _, n in t)][-40:]
agg = {}
spans = []
for t in ticks:
    spans.append((t[-1][1] - t[0][0]) / 1e3)
    prev_end = None
    seen = {}
    for st, en, name in t:
        seen[name] = seen.get(name, 0) + 1
        if seen[name] > 1: name = "%s #%d" % (name, seen[name])   # (the stages of a staged walk)
        a = agg.setdefault(name, {"us": [], "gap_before_us": []})
        a["us"].append((en - st) / 1e3)
        if prev_end is not None:
            a["gap_before_us"].append((st - prev_end) / 1e3)
        prev_end = en
med = lambda v: sorted(v)[len(v) // 2] if v else None
print(json.dumps({"ticks": len(ticks), "first_kernel_start_to_last_kernel_end_us_median": med(spans),
                  "kernels": {k: {"us_median": med(v["us"]), "gap_before_us_median": med(v["gap_before_us"])} for k, v in agg.items()}}, indent=1))
