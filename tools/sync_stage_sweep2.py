#!/usr/bin/env python3
"""Waited-for uncapped frames (depth = 0, the reference's while (true), ARTS.cpp:294) at the headline size: stage bounds and
subpaths per wave per stage (FS_SYNC_WALK_STAGES / FS_SYNC_STAGE_RPW, read at context creation: one subprocess per setting).
usage: python tools/sync_stage_sweep2.py [scene] > profiles/r05_sync_stage_sweep.jsonl"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
scene = sys.argv[1] if len(sys.argv) > 1 else "old_mine"
CHILD = r'''
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
scene = sys.argv[1]
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
run(10, 10)
t = time.perf_counter(); run(40, 100); dt = (time.perf_counter() - t) / 40
print(json.dumps({"ms_per_frame": round(1e3 * dt, 4)}))
'''
SETTINGS = [("24", ""), ("24", "64"), ("10,24", ""), ("10,24", "32,64,0"), ("10,24", "64,64,0"), ("8,24", "32,64,0"), ("12,28", "32,64,0"),
            ("6,14,28", "32,64,64,0"), ("8,18,32", "32,64,32,0"), ("8,20", "32,64,0"), ("10,20,32", "32,64,16,0"), ("12,24,40", "32,64,0,0")]
for bounds, rpw in SETTINGS:
    env = dict(os.environ, FS_SYNC_WALK_STAGES=bounds)
    if rpw:
        env["FS_SYNC_STAGE_RPW"] = rpw
    r = subprocess.run([sys.executable, "-c", CHILD, scene], env=env, capture_output=True, text=True, cwd=os.path.dirname(HERE), timeout=300)
    try:
        out = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        out = {"error": (r.stderr or r.stdout)[-300:]}
    print(json.dumps({"scene": scene, "bounds": bounds, "stage_rpw": rpw or "default", **out}), flush=True)
