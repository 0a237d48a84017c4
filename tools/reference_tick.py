#!/usr/bin/env python3
"""The reference's tick (UAudioRayTracingSubsystem::UpdateSources, ARTS.cpp:100-126): every active source gets one
UpdateSource — 1000 pairs (USED_RAY_COUNT, ARTS.h:176), uncapped walks (ARTS.cpp:294), one band, normaliser 1/1000 —
and the game thread has the IRs when the tick ends.  Here: S sources as ONE batched frame
(fs_compute_energy_response_batch_async) + S reconstructs + fs_synchronize, timed per tick.
usage (GPU box): python tools/reference_tick.py [scene ...]   (env FS_WALK_COOP / FS_WALK_RAYS_PER_WAVE to compare)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()

for name in (sys.argv[1:] or ["starter_room", "old_mine"]):
    sc = pkg.scenes.by_name(name, 1)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    rng = np.random.default_rng(9)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    srcs = [ctx.create_source((np.asarray(sc.source, np.float32) + rng.uniform(-0.03, 0.03, 3).astype(np.float32) * (hi - lo)).astype(np.float32))
            for _ in range(128)]
    p = pkg.default_params(num_rays=2000, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
    out = {"scene": name, "triangles": int(sc.num_triangles)}
    for S in (1, 8, 32, 128):
        times = []
        for i in range(40):
            p.seed = 1000 + i
            t1 = time.perf_counter()
            if os.environ.get("FS_TICK_SEPARATE_CALLS") == "1":
                ctx.compute_energy_response_batch_async(srcs[:S], p)
                ctx.reconstruct_impulse_response_batch_async(srcs[:S], p)
                ctx.synchronize()
            else:
                ctx.update_sources(srcs[:S], p)
            times.append(time.perf_counter() - t1)
        times = sorted(times[8:])
        out[str(S)] = {"ms_per_tick_median": round(1e3 * times[len(times) // 2], 4), "ms_per_tick_min": round(1e3 * times[0], 4),
                       "ms_per_source": round(1e3 * times[len(times) // 2] / S, 4)}
        ctx.reset_stats(); ctx.set_profiling(2)      # a few ticks with events around every kernel (reconstructs go one by one then)
        for i in range(6):
            p.seed = 5000 + i
            if S == 1:
                ctx.compute_energy_response_async(srcs[0], p)
            else:
                ctx.compute_energy_response_batch_async(srcs[:S], p)
            ctx.synchronize()
        st = ctx.stats(); ctx.set_profiling(0)
        out[str(S)]["kernel_ms"] = {"walk": round(st["walk_kernel_ms_sum"] / max(st["timed_frames"], 1), 4),
                                    "connect": round(st["connect_kernel_ms_sum"] / max(st["timed_connects"], 1), 4)}
    print(json.dumps(out), flush=True)
    ctx.close()
