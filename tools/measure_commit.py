#!/usr/bin/env python3
"""fs_scene_commit (host binned-SAH build) against fs_scene_commit_fast (device build: Morton order + PLOC) at the cfg3
scene, and what each tree costs per traced frame.  usage (GPU box): python tools/measure_commit.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
res = {}
for name, bands, rays in (("old_mine", 8, 262144), ("starter_room", 4, 16384)):
    sc = pkg.scenes.by_name(name, bands)
    ctx = pkg.Context(num_bands=bands)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    p = pkg.default_params(num_rays=rays, depth=8)
    row = {"triangles": sc.num_triangles}
    for kind, fast in (("host_sah", False), ("device_morton", True)):
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast=fast)     # warm (allocations, code objects)
        best = best_all = 1e9
        commit = ctx.lib.fs_scene_commit_fast if fast else ctx.lib.fs_scene_commit
        for _ in range(5):
            t0 = time.perf_counter()
            ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast=fast)   # copies the inputs, then commits
            ctx.synchronize()
            best_all = min(best_all, time.perf_counter() - t0)
            t0 = time.perf_counter()
            ctx.check(commit(ctx.h))                                                 # the commit alone
            ctx.synchronize()
            best = min(best, time.perf_counter() - t0)
        st = ctx.stats()
        for i in range(20):
            p.seed = 100 + i
            ctx.compute_energy_response_async(src, p)
        ctx.synchronize()
        ctx.reset_stats()
        ctx.set_profiling(2)
        t0 = time.perf_counter()
        for i in range(100):
            p.seed = 1000 + i
            ctx.compute_energy_response_async(src, p)
        ctx.synchronize()
        el = time.perf_counter() - t0
        s2 = ctx.stats()
        ctx.set_profiling(0)
        row[kind] = {"commit_ms": 1e3 * best, "set_triangles_and_commit_ms": 1e3 * best_all, "bvh_nodes": st["bvh_nodes"], "stack_need": st["bvh_stack_need"],
                     "frame_ms": 1e3 * el / 100, "walk_ms": s2["walk_kernel_ms_sum"] / max(s2["timed_frames"], 1),
                     "connect_ms": s2["connect_kernel_ms_sum"] / max(s2["timed_connects"], 1)}
    # fs_scene_commit_progressive: the device tree at once, the SAH tree swapped in when the background build is done
    t0 = time.perf_counter()
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast="progressive")
    ctx.synchronize()
    t_call = time.perf_counter() - t0
    frames_before = 0
    while ctx.refine_pending():                       # frames keep going through the Morton tree meanwhile
        p.seed = 5000 + frames_before
        t1 = time.perf_counter()
        ctx.compute_energy_response_async(src, p)
        ctx.synchronize()
        t_last = time.perf_counter() - t1             # the last one of these includes the swap
        frames_before += 1
    t_ready = time.perf_counter() - t0
    t1 = time.perf_counter()
    for i in range(50):
        p.seed = 6000 + i
        ctx.compute_energy_response_async(src, p)
    ctx.synchronize()
    row["progressive"] = {"set_triangles_and_commit_ms": 1e3 * t_call, "frames_traced_before_the_swap": frames_before,
                          "swapped_after_ms": 1e3 * t_ready, "frame_with_the_swap_ms": 1e3 * t_last,
                          "frame_ms_afterwards": 1e3 * (time.perf_counter() - t1) / 50, "bvh_nodes": ctx.stats()["bvh_nodes"]}
    res[name] = row
    ctx.close()
print(json.dumps(res))
