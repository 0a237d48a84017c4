#!/usr/bin/env python3
"""Subpaths per walk wave against frame size with PIPELINED frames (fs_set_pipelining 2) on the cfg3 scene: the table
behind auto_rays_per_wave (csrc/fs_capi.cpp).  Product path only.  usage: python tools/pipelined_rpw_sweep.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
rows = []
for depth in (8, 12):
    for rays in (16384, 32768, 65536, 131072, 262144):
        best = None
        for rpw in (0, 4, 8, 16, 32, 48, 64):
            os.environ["FS_WALK_RAYS_PER_WAVE"] = str(rpw)
            ctx = pkg.Context(num_bands=8)
            ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
            ctx.set_listener(sc.listener)
            src = ctx.create_source(sc.source)
            ctx.set_pipelining(2)
            p = pkg.default_params(num_rays=rays, depth=depth)
            for i in range(30):
                p.seed = i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            n = 300
            t0 = time.perf_counter()
            for i in range(n):
                p.seed = 100 + i
                ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / n
            ctx.close()
            rows.append({"depth": depth, "subpaths": rays, "rays_per_wave": rpw, "ms_per_frame": round(ms, 4)})
            if rpw and (best is None or ms < best[1]):
                best = (rpw, ms)
        print(f"depth {depth} subpaths {rays}: " + ", ".join(f"{r['rays_per_wave']}: {r['ms_per_frame']}" for r in rows[-7:]) + f"  best {best[0]}", file=sys.stderr)
print(json.dumps(rows))
