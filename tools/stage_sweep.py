#!/usr/bin/env python3
"""Staged walks: a stream of depth = 0 frames (the reference's uncapped walks, ARTS.cpp:294) on ONE context, unpipelined
and pipelined with several stage-bound sets.  usage: python tools/stage_sweep.py [rays] [frames] [scene]
Prints one JSON line per setting: ms per frame, rays/s."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scene = sys.argv[3] if len(sys.argv) > 3 else "old_mine"
bands = 8 if scene == "old_mine" else 4
sc = getattr(pkg.scenes, scene)(bands)
SETTINGS = [
    ("unpipelined", 0, None),
    ("12,24,36,48,64,80,104", 2, [12, 24, 36, 48, 64, 80, 104]),
    ("12,24,36,48,60,72,84", 2, [12, 24, 36, 48, 60, 72, 84]),
    ("10,20,30,40,52,64,80", 2, [10, 20, 30, 40, 52, 64, 80]),
    ("8,16,24,32,44,56,72", 2, [8, 16, 24, 32, 44, 56, 72]),
    ("12,24,36,48,64,88", 2, [12, 24, 36, 48, 64, 88]),
    ("14,28,42,56,72,96", 2, [14, 28, 42, 56, 72, 96]),
    ("10,20,32,48,64,96", 2, [10, 20, 32, 48, 64, 96]),
]
if len(sys.argv) > 4:
    SETTINGS = [(sys.argv[4], 2, [int(x) for x in sys.argv[4].split(",")])]
for name, depth, bounds in SETTINGS:
    c = pkg.Context(num_bands=bands)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    c.set_listener(sc.listener)
    s = c.create_source(sc.source)
    c.set_pipelining(depth)
    if depth and os.environ.get("FS_SWEEP_FPL"):
        c.set_frames_per_launch(int(os.environ["FS_SWEEP_FPL"]))   # frames that share a launch (fs_set_frames_per_launch)
    if bounds is not None:
        c.set_walk_stages(bounds)
    p = pkg.default_params(num_rays=rays, depth=0)

    def run(n, seed0):
        for i in range(n):
            p.seed = seed0 + i
            c.compute_energy_response_async(s, p)
            c.reconstruct_impulse_response_async(s, p)
        c.synchronize()

    run(30, 10)
    t = time.perf_counter()
    run(frames, 100)
    dt = (time.perf_counter() - t) / frames
    print(json.dumps({"setting": name, "rays_per_frame": rays, "scene": scene, "ms_per_frame": round(1e3 * dt, 4),
                      "Mrays_per_s": round(rays / dt / 1e6, 1)}), flush=True)
    c.close()
