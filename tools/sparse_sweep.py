#!/usr/bin/env python3
"""Frame latency against subpaths-per-wave for small frames (walk_kernel_sparse); sets FS_WALK_RAYS_PER_WAVE per
context.  usage: python tools/sparse_sweep.py  (product path only; prints one JSON object)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
rows = []
for scene, bands, depth in (("old_mine", 8, 8), ("starter_room", 4, 0)):
    sc = pkg.scenes.by_name(scene, bands)
    for rpw in (64, 32, 16, 8, 4):
        os.environ["FS_WALK_RAYS_PER_WAVE"] = str(rpw)
        ctx = pkg.Context(num_bands=bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        ctx.set_listener(sc.listener)
        src = ctx.create_source(sc.source)
        for rays in (2000, 8192, 16384, 32768, 65536, 131072, 262144):
            if rays * 64 // rpw > 4_200_000:
                continue
            p = pkg.default_params(num_rays=rays, depth=depth)
            for i in range(5):
                p.seed = 10 + i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            n = 100
            t = time.perf_counter()
            for i in range(n):
                p.seed = 100 + i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            rows.append({"scene": scene, "depth": depth, "rays_per_wave": rpw, "rays": rays,
                         "ms_per_frame": 1e3 * (time.perf_counter() - t) / n})
        ctx.close()
print(json.dumps({"rows": rows}))
