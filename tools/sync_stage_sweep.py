#!/usr/bin/env python3
"""Stage bounds of depth = 0 frames that are waited for (FS_SYNC_WALK_STAGES): the reference's tick at S sources and the
headline-size frame, per bounds set.  usage (GPU box): python tools/sync_stage_sweep.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()

SETS = sys.argv[1:] or ["", "16", "8,24", "16,40", "12,32,64", "6,16,40", "24", "10,24,48,96"]   # "bounds[:late rays per wave[:first-stage rays per wave]]"
for name, bands in (("starter_room", 1), ("old_mine", 1)):
    sc = pkg.scenes.by_name(name, bands)
    for bounds in SETS:
        bounds, _, late = bounds.partition(":")
        late, _, first = late.partition(":")
        os.environ["FS_SYNC_WALK_STAGES"] = bounds
        os.environ["FS_SYNC_LATE_RPW"] = late or "0"
        os.environ["FS_SYNC_FIRST_RPW"] = first or "0"
        ctx = pkg.Context(num_bands=bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        ctx.set_listener(sc.listener)
        rng = np.random.default_rng(9)
        lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
        srcs = [ctx.create_source((np.asarray(sc.source, np.float32) + rng.uniform(-0.03, 0.03, 3).astype(np.float32) * (hi - lo)).astype(np.float32))
                for _ in range(128)]
        p = pkg.default_params(num_rays=2000, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
        out = {"scene": name, "bounds": bounds, "late_rpw": late or "auto", "first_rpw": first or "auto"}
        for S in (8, 32, 128):
            times = []
            for i in range(24):
                p.seed = 1000 + i
                t1 = time.perf_counter()
                ctx.update_sources(srcs[:S], p)
                times.append(time.perf_counter() - t1)
            times = sorted(times[6:])
            out[f"tick_{S}_ms"] = round(1e3 * times[len(times) // 2], 4)
        pb = pkg.default_params(num_rays=262144, depth=0, seed=1)
        times = []
        for i in range(12):
            pb.seed = 77 + i
            t1 = time.perf_counter()
            ctx.compute_energy_response_async(srcs[0], pb)
            ctx.reconstruct_impulse_response_async(srcs[0], pb)
            ctx.synchronize()
            times.append(time.perf_counter() - t1)
        times = sorted(times[3:])
        out["frame_262144_ms"] = round(1e3 * times[len(times) // 2], 4)
        print(json.dumps(out), flush=True)
        ctx.close()
