#!/usr/bin/env python3
"""Pipelined depth-8 frames of several sizes on the cfg3 scene: rays/s against rays per frame (what bigger launches would buy).
usage (GPU box): python tools/frame_size_pipelined.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
for rays in (131072, 262144, 393216, 524288, 1048576, 2097152):
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
    s = c.create_source(sc.source); c.set_pipelining(2)
    p = pkg.default_params(num_rays=rays, depth=8)
    def run(n, seed0):
        for i in range(n):
            p.seed = seed0 + i
            c.compute_energy_response_async(s, p); c.reconstruct_impulse_response_async(s, p)
        c.synchronize()
    frames = max(30, min(300, int(80e6 / rays)))
    run(30, 10); t = time.perf_counter(); run(frames, 100); dt = (time.perf_counter() - t) / frames
    print(json.dumps({"rays": rays, "ms_per_frame": round(1e3 * dt, 4), "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)
    c.close()
