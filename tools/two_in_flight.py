#!/usr/bin/env python3
"""Two frames in flight: N contexts on the same scene, frames issued round-robin without waiting (product path only).
Does the walk of one frame fill the thin tail of the previous one?  usage: python tools/two_in_flight.py [contexts] [rays] [frames] [depth]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 400
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 8      # 0 = unbounded walks (the reference's while (true))
sc = pkg.scenes.old_mine(8)
ctxs, srcs = [], []
for _ in range(nctx):
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    c.set_listener(sc.listener)
    ctxs.append(c)
    srcs.append(c.create_source(sc.source))
p = pkg.default_params(num_rays=rays, depth=depth)


def run(n, seed0):
    for i in range(n):
        p.seed = seed0 + i
        c, s = ctxs[i % nctx], srcs[i % nctx]
        c.compute_energy_response_async(s, p)
        c.reconstruct_impulse_response_async(s, p)
    for c in ctxs:
        c.synchronize()


run(40, 10)
t = time.perf_counter()
run(frames, 100)
dt = (time.perf_counter() - t) / frames
print(json.dumps({"contexts": nctx, "depth": depth, "rays_per_frame": rays, "ms_per_frame": 1e3 * dt, "rays_per_s": rays / dt}))
