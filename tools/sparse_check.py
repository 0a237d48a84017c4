#!/usr/bin/env python3
"""Spot check of the automatic subpaths-per-wave rule against fixed settings on large unbounded-depth frames."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
sc = pkg.scenes.by_name("starter_room", 4)
rows = []
for rpw in (0, 64, 32, 16):
    os.environ["FS_WALK_RAYS_PER_WAVE"] = str(rpw)
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    for rays in (524288, 1048576, 2097152):
        p = pkg.default_params(num_rays=rays, depth=0)
        for i in range(3):
            p.seed = 10 + i
            ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
        ctx.synchronize()
        n = 20
        t = time.perf_counter()
        for i in range(n):
            p.seed = 100 + i
            ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
        ctx.synchronize()
        rows.append({"rays_per_wave": rpw, "rays": rays, "ms_per_frame": 1e3 * (time.perf_counter() - t) / n})
    ctx.close()
print(json.dumps(rows))
