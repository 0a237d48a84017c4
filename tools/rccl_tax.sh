#!/bin/bash
# The single-rank cost of the library's collective (794 -> 750 M rays/s with a communicator attached):
# bench with and without FS_BENCH_FORCE_REDUCE=1, then a kernel trace of the reduced run: which kernels run on the tail
# stream per frame, how long they take and what they do to the frame kernel.
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out
for mode in plain reduce plain reduce; do
  if [ $mode = reduce ]; then export FS_BENCH_FORCE_REDUCE=1; else unset FS_BENCH_FORCE_REDUCE; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extra --steps 300 --warmup 20 2>/tmp/o.err > /tmp/o.json || { echo "$mode failed"; tail -3 /tmp/o.err; continue; }
  python3 - "$mode" <<'PY'
import json,sys
j=json.load(open('/tmp/o.json')); print(sys.argv[1], 'ms', round(j['ms_per_step'],4), 'Mrays/s', round(j['value']/1e6,1), {k: round(v,4) for k,v in j['kernel_ms'].items()})
PY
done
export FS_BENCH_FORCE_REDUCE=1
rm -rf $out/rccl_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/rccl_trace -- python3 bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 > $out/rccl_trace.json 2> $out/rccl_trace.err || { echo trace failed; tail -3 $out/rccl_trace.err; }
f=$(find $out/rccl_trace -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -8; cp "$f" $out/rccl_kernel_stats.csv
g=$(find $out/rccl_trace -name "*kernel_trace.csv" | head -1)
python3 - "$g" <<'PY'
import csv,sys,statistics as st
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
fr=[r for r in rows if "frame_kernel" in r["Kernel_Name"]]
nc=[r for r in rows if "ccl" in r["Kernel_Name"].lower() or "AllReduce" in r["Kernel_Name"]]
print("frame launches", len(fr), "rccl kernels", len(nc))
if nc:
    d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in nc]
    print("rccl kernel us: median", st.median(d), "p90", sorted(d)[9*len(d)//10], "name", nc[0]["Kernel_Name"][:80], "grid", nc[0].get("Grid_Size_X", nc[0].get("Grid_Size")), "wg", nc[0].get("Workgroup_Size_X", nc[0].get("Workgroup_Size")))
    # overlap: frame kernels that overlap an rccl kernel vs not
    import bisect
    ov=[];no=[]
    for r in fr[5:]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        hit=any(int(n["Start_Timestamp"])<e and int(n["End_Timestamp"])>s for n in nc)
        (ov if hit else no).append((e-s)/1e3)
    print("frame us overlapping an rccl kernel: n", len(ov), "median", st.median(ov) if ov else None, "| not overlapping: n", len(no), "median", st.median(no) if no else None)
gaps=[(int(fr[i+1]["Start_Timestamp"])-int(fr[i]["End_Timestamp"]))/1e3 for i in range(5,len(fr)-1)]
print("gap between frame kernels us: median", st.median(gaps), "p90", sorted(gaps)[9*len(gaps)//10])
# what ran between two launches in the steady state (kernels that START after launch k has ended, and the ones of the launch's last 60 us)
idx=[i for i,r in enumerate(rows) if "frame_kernel<8, true>" in r["Kernel_Name"]]
for k in (40, 41):
    a, b = rows[idx[k]], rows[idx[k+1]]
    ea, sb = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    print("gap", round((sb-ea)/1e3,1), "us; queue of the launches", a.get("Queue_Id"), b.get("Queue_Id"))
    for r in rows:
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if r is a or r is b or e0 < ea - 60000 or s0 > sb: continue
        print("    queue", r.get("Queue_Id"), r["Kernel_Name"][:70], "start %+.1f us" % ((s0-ea)/1e3), "dur %.1f us" % ((e0-s0)/1e3))
PY
rm -rf $out/rccl_trace
