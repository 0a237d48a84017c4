#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel per launch.

usage: python profiles/summarize_pmc.py gpurun_out/pmc_v1 [> profiles/rNN_pmc_summary.txt]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads (MI355X_MICROARCH.md §HBM) — both raw and x2 figures are printed.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    for k in ("walk_kernel", "connect_kernel", "reconstruct_kernel", "trace_rays_kernel", "subpath", "copyBuffer",
              "fillBuffer"):
        if k in name:
            return k + (name[name.index("<"):name.index(">") + 1] if "<" in name and k.endswith("kernel") else "")
    return name[:48]


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
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
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
