#!/usr/bin/env python3
"""Throughput against frame size on the cfg3 scene (old_mine, depth 8, 8 bands): where the frame stops being the
latency of its longest walk and starts filling the chip.  Product path only.  usage: python tools/frame_size_sweep.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
src = ctx.create_source(sc.source)
rows = []
for rays in (2048, 16384, 65536, 262144, 1048576, 4194304):
    p = pkg.default_params(num_rays=rays, depth=8)
    for i in range(5):
        p.seed = 10 + i
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    n = max(10, min(400, int(2.0e8 / rays)))
    t = time.perf_counter()
    for i in range(n):
        p.seed = 100 + i
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    dt = (time.perf_counter() - t) / n
    rows.append({"rays_per_frame": rays, "ms_per_frame": 1e3 * dt, "rays_per_s": rays / dt})
print(json.dumps({"scene": "old_mine 100000 triangles, depth 8, 8 bands, frames traced one after the other", "sweep": rows}))
