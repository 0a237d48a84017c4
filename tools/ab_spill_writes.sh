#!/bin/bash
# are the frame kernel's HBM writes (WRITE_SIZE 120 MB per two-frame launch against ~45 MB of payload) its 23
# spilled registers?  The same bench under rocprofv3 --pmc WRITE_SIZE with the product library (128-VGPR cap, 23 spills) and with
# a build whose frame kernel may use 256 registers (tools/build_variant.sh nospill "-DFS_FRAME_MIN_WAVES=2": no spills).
#   bash tools/ab_spill_writes.sh      ->  gpurun_out/r05/spill_writes.txt
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/r05
mkdir -p $out
cp audio-pathtracer_amd/libfrequensee.so /tmp/base.so
for v in base nospill; do
  if [ $v = base ]; then cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so; else cp tools/tmp/$v/libfrequensee.so audio-pathtracer_amd/libfrequensee.so; fi
  for c in WRITE_SIZE FETCH_SIZE SQ_INSTS_VMEM_WR; do
    rm -rf $out/sw_${v}_$c
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $out/sw_${v}_$c -- python3 bench.py --steps 6 --warmup 2 --prewarm 10 --no-cpu-baseline --no-extra > $out/sw_${v}_$c.log 2>&1 || { echo "pmc pass $v $c failed"; tail -3 $out/sw_${v}_$c.log; }
  done
done
cp /tmp/base.so audio-pathtracer_amd/libfrequensee.so
python3 - <<'PY' | tee gpurun_out/r05/spill_writes.txt
import csv, glob, collections
for v in ("base", "nospill"):
    for c in ("WRITE_SIZE", "FETCH_SIZE", "SQ_INSTS_VMEM_WR"):
        acc = collections.defaultdict(lambda: [0.0, set()])
        for f in glob.glob(f"gpurun_out/r05/sw_{v}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != c: continue
                k = ("frame_kernel" if "frame_kernel" in r["Kernel_Name"] else r["Kernel_Name"].split("(")[0])[:60]
                acc[k][0] += float(r["Counter_Value"]); acc[k][1].add(r["Dispatch_Id"])
        for k, (s, d) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:3]:
            print(v, c, k, "per launch %.1f" % (s / max(len(d), 1)), "launches", len(d))
PY
