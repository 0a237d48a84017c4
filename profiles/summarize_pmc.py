#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per launch.

usage: python profiles/summarize_pmc.py gpurun_out/<tag>_pmc [> profiles/<tag>_pmc_summary.txt]
       python profiles/summarize_pmc.py gpurun_out/<tag>_pmc --json <workload> <tag>   (merged into profiles/pmc.json,
       the per-launch counters bench.py prices its roofline objects with)

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM section) — both raw and x2 figures are printed.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(\w+_kernel\w*)\s*(<[^>]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    for k in ("copyBuffer", "fillBuffer"):
        if k in name:
            return k
    return name[:48]


def collect(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main(argv):
    root = argv[1] if len(argv) > 1 else "gpurun_out"
    acc = collect(root)
    if len(argv) > 2 and argv[2] == "--json":
        workload, tag = argv[3], argv[4]
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc.json")
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc["_comment"] = ("mean per-launch hardware counters of the bench workloads' kernels (rocprofv3 --pmc, one counter "
                           "group per run, tools/profile_round.sh); FETCH_SIZE / WRITE_SIZE in KiB; the timed instantiations "
                           "only (the COUNT instantiations of profiling level 3 are left out)")
        doc["_tag"] = tag
        entry = {}
        for k in acc:
            # the counting instantiations (fs_set_profiling level 3: one untimed frame of the bench) are left out:
            # walk_kernel_*<LOBES, COUNT, EXT>, connect_kernel<B, LOBES, BATCH, COUNT, EXT>
            targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k and ">" in k else []
            if k.startswith("walk_kernel") and len(targs) > 1 and targs[1] == "true":
                continue
            if k.startswith("connect_kernel") and len(targs) > 3 and targs[3] == "true":
                continue
            base = "walk_kernel" if k.startswith("walk_kernel") else ("connect_kernel" if k.startswith("connect_kernel")
                                                                       else (k.split("<")[0] if k.endswith(">") else None))
            if base is None:
                continue
            # several instantiations of one family may have run (frame_kernel<8, true> for the launches of grouped frames,
            # <8, false> for a pipeline-fill launch or two): the one with the most launches is the workload's
            n = max(len(v) for v in acc[k].values())
            if base in entry and entry[base]["_launches"] >= n:
                continue
            entry[base] = {"instantiation": k, "_launches": n}
            for c, v in acc[k].items():
                entry[base][c] = sum(v) / len(v)
        doc[workload] = entry
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
        print(f"wrote {path}: {workload} <- {sorted(entry)}")
        return
    for k in sorted(acc):
        print(f"== {k}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            mean = sum(v) / len(v)
            extra = ""
            if c == "FETCH_SIZE":
                extra = f"  = {mean * 1024 / 1e6:.2f} MB/launch raw, {2 * mean * 1024 / 1e6:.2f} MB x2-corrected"
            if c == "WRITE_SIZE":
                extra = f"  = {mean * 1024 / 1e6:.2f} MB/launch"
            print(f"   {c:34s} n={len(v):3d} mean={mean:16.1f}{extra}")


if __name__ == "__main__":
    main(sys.argv)
