#!/usr/bin/env python3
"""fs_set_frames_per_launch: results against one frame per launch (deterministic mode, bit for bit) and the rate of a stream of cfg3
frames for 1 .. 4 frames per launch.  usage (GPU box): python tools/frames_per_launch.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
DET = 8
# 1. correctness: grouped == ungrouped, bit for bit (deterministic mode), IRs too
out = {}
for n in (1, 2, 3, 4):
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
    s = c.create_source(sc.source); s2 = c.create_source(np.asarray(sc.source, np.float32) + np.float32(30.0))
    c.set_pipelining(2); c.set_frames_per_launch(n)
    p = pkg.default_params(num_rays=32768, depth=8, flags=DET)
    es, irs = [], []
    for i in range(7):
        p.seed = 500 + i
        src = s if i % 3 else s2
        c.compute_energy_response_async(src, p); c.reconstruct_impulse_response_async(src, p)
        if i in (2, 5):
            es.append(c.energy_buffer(src).copy()); c.synchronize(); irs.append(c.impulse_response(src, 0).copy())   # (the accessor returns the newest PUBLISHED IR: wait for it)
    c.synchronize()
    es.append(c.energy_buffer(s).copy()); es.append(c.energy_buffer(s2).copy())
    irs.append(c.impulse_response(s, 0).copy()); irs.append(c.impulse_response(s2, 0).copy())
    out[n] = (es, irs, c.stats()["frames"])
    c.close()
for n in (2, 3, 4):
    ok = all(np.array_equal(a, b) for a, b in zip(out[1][0], out[n][0])) and all(np.array_equal(a, b) for a, b in zip(out[1][1], out[n][1]))
    print("frames_per_launch", n, "identical to 1:", ok, "frames", out[n][2], out[1][2], flush=True)
# 2. speed
for n in (1, 2, 3, 4):
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
    s = c.create_source(sc.source); c.set_pipelining(2); c.set_frames_per_launch(n)
    p = pkg.default_params(num_rays=262144, depth=8)
    def run(k, seed0):
        for i in range(k):
            p.seed = seed0 + i
            c.compute_energy_response_async(s, p); c.reconstruct_impulse_response_async(s, p)
        c.synchronize()
    run(48, 10); t = time.perf_counter(); run(240, 100); dt = (time.perf_counter() - t) / 240
    print(json.dumps({"frames_per_launch": n, "ms_per_frame": round(1e3 * dt, 4), "Mrays_per_s": round(262144 / dt / 1e6, 1)}), flush=True)
    c.close()
