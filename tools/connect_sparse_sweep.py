#!/usr/bin/env python3
"""Frame latency against pairs-per-wave of the connect kernel for small frames (FS_CONNECT_PAIRS_PER_WAVE)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
rows = []
for scene, bands, depth in (("old_mine", 8, 8), ("starter_room", 4, 8)):
    sc = pkg.scenes.by_name(scene, bands)
    for ppw in (64, 32, 16, 8, 4):
        os.environ["FS_CONNECT_PAIRS_PER_WAVE"] = str(ppw)
        ctx = pkg.Context(num_bands=bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        ctx.set_listener(sc.listener)
        src = ctx.create_source(sc.source)
        ctx.set_profiling(2)
        for rays in (2000, 16384, 65536, 131072, 262144):
            p = pkg.default_params(num_rays=rays, depth=depth)
            for i in range(5):
                p.seed = 10 + i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            ctx.reset_stats()
            n = 60
            t = time.perf_counter()
            for i in range(n):
                p.seed = 100 + i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            st = ctx.stats()
            rows.append({"scene": scene, "pairs_per_wave": ppw, "rays": rays,
                         "ms_per_frame": 1e3 * (time.perf_counter() - t) / n,
                         "connect_ms": st["connect_kernel_ms_sum"] / max(1, st["timed_connects"])})
        ctx.close()
print(json.dumps({"rows": rows}))
