#!/usr/bin/env python3
"""fs_scene_commit time (host SAH build + upload) for cfg3's 100 000 triangles and for a million-triangle slab,
with the builder's worker threads off and on.  usage: python tools/measure_commit.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, json
import numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as graft
pkg = graft.load_package()
out = {}
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)        # warm: first HIP allocations
for name, tri, mat, ab in (("old_mine_100k", sc.triangles, sc.material_ids, sc.absorption),):
    ts = []
    for _ in range(3):
        t = time.perf_counter(); ctx.set_scene(tri, mat, ab); ctx.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    out[name] = min(ts)
rng = np.random.default_rng(1)
T = 1000000
c = rng.uniform(0, 10000, (T, 1, 3)); c[:, :, 2] *= 0.05
tri = (c + rng.uniform(-30, 30, (T, 3, 3))).astype(np.float32)
t = time.perf_counter(); ctx.set_scene(tri, np.zeros(T, np.uint16), sc.absorption); ctx.synchronize()
out["slab_1m"] = 1e3 * (time.perf_counter() - t)
print(json.dumps(out))
''' % ROOT
res = {}
for threads in (1, 8, 16):
    env = dict(os.environ, FS_BVH_THREADS=str(threads))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res[f"threads_{threads}"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else r.stderr[-300:]
print(json.dumps({"fs_scene_commit_ms": res}))
