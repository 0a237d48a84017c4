#!/usr/bin/env python3
"""Row f4 timing: a moved prop (2 000 of cfg3's 100 000 triangles) through fs_scene_update_triangles +
fs_scene_refit against a full fs_scene_commit (host SAH build + upload).  Product path only (no oracle).
usage: python tools/measure_refit.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
t0 = time.perf_counter()
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.synchronize()
commit_ms = 1e3 * (time.perf_counter() - t0)
ctx.set_listener(sc.listener)
src = ctx.create_source(sc.source)
p = pkg.default_params(num_rays=262144, depth=8)
ctx.compute_energy_response(src, p)
T = sc.triangles.shape[0]
a, n = T - 30000, 2000
base = sc.triangles[a:a + n].copy()
res = {}
for label, count in (("2000 triangles", n), ("all 100000 triangles", T)):
    first = a if count == n else 0
    tri = base if count == n else sc.triangles
    for i in range(3):
        ctx.update_triangles(first, tri + np.float32(i))
        ctx.refit()
    ctx.synchronize()
    reps = 30
    t0 = time.perf_counter()
    for i in range(reps):
        ctx.update_triangles(first, tri + np.float32(0.5 * (i % 5)))
        ctx.refit()
    ctx.synchronize()
    res[label] = 1e3 * (time.perf_counter() - t0) / reps
t0 = time.perf_counter()
for i in range(20):
    ctx.refit()
ctx.synchronize()
refit_only = 1e3 * (time.perf_counter() - t0) / 20
ctx.update_triangles(0, sc.triangles)
walk0 = ctx.stats()
ctx.set_profiling(1)
ctx.reset_stats()
for _ in range(20):
    ctx.compute_energy_response_async(src, p)
ctx.synchronize()
st = ctx.stats()
print(json.dumps({"scene": "old_mine 100000 triangles", "full_commit_ms": commit_ms,
                  "update_plus_refit_ms": res, "refit_only_ms": refit_only,
                  "walk_ms_after_refit_to_original": st["walk_kernel_ms_sum"] / max(st["timed_frames"], 1)}))
