#!/usr/bin/env python3
"""VERDICT r3 item 4, second half: the re-binning itself.  A waited-for frame (262 144 rays, old_mine, 8 bands; cfg3's depth 8
and uncapped walks) walks in stages of ONE bounce on dense waves (FS_SYNC_WALK_STAGES=1,2,3,4,5,6,7); between two stages the
walks still alive are counting-sorted by (Morton code of the 16^3 cell of their position, octant of the surface normal they
leave from) and the next stage's lane i walks the i-th slot of that order (FS_DEBUG_REBIN=1).  Results do not depend on which
lane walks a slot (checked here in deterministic mode, bit for bit).  Reported per variant: ms per frame, the walk launches'
time with and without the three sort kernels per stage, and — through the counting instantiation — lanes and distinct 64-B
records per node-request instruction, i.e. the coherence the order buys.
usage (GPU box): python tools/rebin_experiment.py > gpurun_out/r04_rebin_experiment.json"""
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
sc = pkg.scenes.by_name("old_mine", 8)
DET = pkg._capi.FLAG_DETERMINISTIC
STAGED = {"FS_SYNC_WALK_STAGES": "1,2,3,4,5,6,7", "FS_SYNC_FIRST_RPW": "64", "FS_SYNC_LATE_RPW": "64"}
UNCAPPED = {"FS_SYNC_WALK_STAGES": "1,2,3,4,5,6,8", "FS_SYNC_FIRST_RPW": "64", "FS_SYNC_LATE_RPW": "64"}
# (depth, label, environment)
VARIANTS = [(8, "cfg3_depth_8_one_piece", {}),
            (8, "cfg3_depth_8_stages_of_one_bounce", dict(STAGED, FS_DEBUG_REBIN="3")),
            (8, "cfg3_depth_8_stages_of_one_bounce_order_computed_not_used", dict(STAGED, FS_DEBUG_REBIN="2")),
            (8, "cfg3_depth_8_stages_of_one_bounce_rebinned", dict(STAGED, FS_DEBUG_REBIN="1")),
            (0, "uncapped_default_bound_24_cooperative_late_stage", {}),
            (0, "uncapped_stages_1_2_3_4_5_6_8", dict(UNCAPPED, FS_DEBUG_REBIN="3")),
            (0, "uncapped_stages_1_2_3_4_5_6_8_rebinned", dict(UNCAPPED, FS_DEBUG_REBIN="1"))]
KEYS = ("FS_SYNC_WALK_STAGES", "FS_SYNC_FIRST_RPW", "FS_SYNC_LATE_RPW", "FS_DEBUG_REBIN")
out = {"workload": "old_mine: 100000 tris, 262144 rays/frame, 8 bands, one frame at a time (waited for, nothing pipelined); depth 8 = cfg3's frame, depth 0 = uncapped walks",
       "variants": {}}
for depth, label, env in VARIANTS:
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    c.set_listener(sc.listener)
    s = c.create_source(sc.source)
    r = {}
    p = pkg.default_params(num_rays=262144, depth=depth, seed=0x5EED, flags=DET)
    e = np.asarray(c.compute_energy_response(s, p))
    r["energy_sha1_deterministic_mode"] = hashlib.sha1(np.ascontiguousarray(e).tobytes()).hexdigest()[:16]
    p = pkg.default_params(num_rays=262144, depth=depth, seed=1)
    for i in range(10):
        p.seed = 100 + i
        c.compute_energy_response_async(s, p); c.synchronize()
    times = []
    for i in range(30):
        p.seed = 1000 + i
        t1 = time.perf_counter()
        c.compute_energy_response_async(s, p); c.synchronize()
        times.append(time.perf_counter() - t1)
    times.sort()
    r["ms_per_frame_median"] = round(1e3 * times[len(times) // 2], 4)
    c.reset_stats(); c.set_profiling(2)
    for i in range(8):
        p.seed = 5000 + i
        c.compute_energy_response_async(s, p); c.synchronize()
    st = c.stats(); c.set_profiling(0)
    r["walk_launches_ms"] = round(st["walk_kernel_ms_sum"] / max(st["timed_frames"], 1), 4)
    if env.get("FS_DEBUG_REBIN") or depth == 8:
        c.reset_stats(); c.set_profiling(3)
        p.seed = 0x5EED
        c.compute_energy_response_async(s, p); c.synchronize()
        st = c.stats(); c.set_profiling(0)
        r["node_requests"] = {"instructions": st["node_request_insts"], "lanes_per_instruction": round(st["node_request_lanes"] / max(st["node_request_insts"], 1), 2),
                              "distinct_records_per_lane": round(st["node_request_distinct"] / max(st["node_request_lanes"], 1), 4),
                              "distinct_records_per_instruction": round(st["node_request_distinct"] / max(st["node_request_insts"], 1), 2)}
    out["variants"][label] = r
    c.close()
    print(label, r, file=sys.stderr, flush=True)
v = out["variants"]
out["bit_identical"] = {"depth_8": len({r["energy_sha1_deterministic_mode"] for k, r in v.items() if k.startswith("cfg3")}) == 1,
                        "uncapped": len({r["energy_sha1_deterministic_mode"] for k, r in v.items() if k.startswith("uncapped")}) == 1}
a, b, o = (v["cfg3_depth_8_stages_of_one_bounce" + x] for x in ("", "_rebinned", "_order_computed_not_used"))
out["summary_depth_8"] = {"one_piece_walk_ms": v["cfg3_depth_8_one_piece"]["walk_launches_ms"],
                          "staged_walk_launches_ms": a["walk_launches_ms"],
                          "sort_kernels_ms_per_frame": round(o["walk_launches_ms"] - a["walk_launches_ms"], 4),
                          "walk_kernels_gain_from_the_order_ms": round(o["walk_launches_ms"] - b["walk_launches_ms"], 4),
                          "rebinned_walk_launches_ms_sorts_included": b["walk_launches_ms"],
                          "rebinned_walk_kernels_alone_ms": round(b["walk_launches_ms"] - (o["walk_launches_ms"] - a["walk_launches_ms"]), 4)}
print(json.dumps(out))
