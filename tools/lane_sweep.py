#!/usr/bin/env python3
"""The long-walk lane of waited-for uncapped frames (FS_SYNC_LANE = "len,end" and friends, read at context creation: one subprocess per setting):
the 262 144-ray frame and ticks of 32 / 128 reference sources.  usage: python tools/lane_sweep.py [scene ["settings"]] > profiles/r05_lane_sweep.jsonl"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
scene = sys.argv[1] if len(sys.argv) > 1 else "old_mine"
CHILD = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
if os.environ.get("FS_LIB_PATH"):   # (an experimental build: tools/build_variant.sh)
    pkg._capi.LIB_PATH = os.path.abspath(os.environ["FS_LIB_PATH"])
    pkg._capi._lib = None
scene = sys.argv[1]
res = {}
# the headline-sized frame
bands = 8 if scene == "old_mine" else 4
sc = getattr(pkg.scenes, scene)(bands)
c = pkg.Context(num_bands=bands)
c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
c.set_listener(sc.listener)
s = c.create_source(sc.source)
p = pkg.default_params(num_rays=262144, depth=0)
def run(n, seed0):
    for i in range(n):
        p.seed = seed0 + i
        c.compute_energy_response_async(s, p)
        c.reconstruct_impulse_response_async(s, p)
        c.synchronize()
if os.environ.get("LANE_SWEEP_FRAME", "1") != "0":
    run(10, 10)
    t = time.perf_counter(); run(40, 100); res["frame_ms"] = round(1e3 * (time.perf_counter() - t) / 40, 4)
c.close()
# reference ticks
sc = pkg.scenes.by_name(scene, 1)
for S in [int(x) for x in os.environ.get("LANE_SWEEP_TICKS", "32,128").split(",") if x]:
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    rng = np.random.default_rng(9)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    srcs = [ctx.create_source((np.asarray(sc.source, np.float32) + rng.uniform(-0.03, 0.03, 3).astype(np.float32) * (hi - lo)).astype(np.float32)) for _ in range(S)]
    pt = pkg.default_params(num_rays=2000, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
    tt = []
    for i in range(60):
        pt.seed = 100 + i
        t1 = time.perf_counter(); ctx.update_sources(srcs, pt); tt.append(time.perf_counter() - t1)
    tt = sorted(tt[10:])
    res["tick%d_ms" % S] = round(1e3 * tt[len(tt) // 2], 4)
    ctx.close()
print(json.dumps(res))
'''
# a setting = environment assignments separated by blanks ("FS_SYNC_LANE=48,0 FS_COOP_BIG=0"); settings separated by ";"
SETTINGS = sys.argv[2].split(";") if len(sys.argv) > 2 else ["FS_SYNC_LANE=0", "FS_SYNC_LANE=48,0", "FS_SYNC_LANE=56,0", "FS_SYNC_LANE=64,0", "FS_SYNC_LANE=48,78",
                                                              "FS_SYNC_LANE=40,78", "FS_SYNC_LANE=56,100", "FS_SYNC_LANE=40,60"]
for setting in SETTINGS:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run([sys.executable, "-c", CHILD, scene], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE), timeout=400)
    try:
        out = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        out = {"error": (r.stderr or r.stdout)[-300:]}
    print(json.dumps({"scene": scene, "setting": setting, **out}), flush=True)
