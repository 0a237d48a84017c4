#!/usr/bin/env python3
"""Diagnostic (not a test, not the product build): SIMD occupancy of the traversal loop's two step kinds.
Builds a -DFS_TRAV_STATS copy of libfrequensee.so into gpurun_out/ and runs one cfg3 frame through it.
usage (GPU box): python tests/trav_stats.py"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "stats_build")
os.makedirs(out, exist_ok=True)
src = os.path.join(ROOT, "audio-pathtracer_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                "-munsafe-fp-atomics", "--offload-arch=gfx950", "-DFS_TRAV_STATS", "-shared", "-o",
                os.path.join(out, "libfrequensee.so"), "-x", "hip", os.path.join(src, "fs_capi.cpp"),
                os.path.join(src, "fs_bvh.cpp"), os.path.join(src, "fs_kernels.hip")], check=True)
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
pkg._capi.LIB_PATH = os.path.join(out, "libfrequensee.so")
pkg._capi._lib = None
lib = pkg._capi.load()
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
s = ctx.create_source(sc.source)
p = pkg.default_params(num_rays=262144, depth=8)
buf = (C.c_ulonglong * 8)()
res = {}
for plan in (1,):
    lib.fs_debug_trav_stats(buf, 1)
    ctx.compute_energy_response(s, p)
    lib.fs_debug_trav_stats(buf, 1)
    v = list(buf)
    res = {"wave_step_calls": v[0], "node_iterations": v[1], "node_lane_steps": v[2], "tri_iterations": v[3],
           "tri_lane_steps": v[4], "node_lanes_per_iteration": v[2] / max(v[1], 1), "tri_lanes_per_iteration": v[4] / max(v[3], 1)}
print(json.dumps(res))
