#!/usr/bin/env python3
"""Diagnostic (timeline build: bash tools/build_variant.sh timeline -DFS_WAVE_TIMELINE): where the waves of the stand-alone connect pass
of ONE waited-for uncapped frame spend their time - set-up and end-state loads | visibility queries | path evaluation + deposits |
flush - as percentiles over the waves.   usage (GPU box): python tools/connect_phases.py [scene [rays]]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
pkg._capi.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "timeline", "libfrequensee.so")
pkg._capi._lib = None
lib = pkg._capi.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
lib.fs_debug_wave_buffer.argtypes = [C.c_void_p]
lib.fs_debug_connect_buffer.argtypes = [C.c_void_p]
scene = sys.argv[1] if len(sys.argv) > 1 else "old_mine"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
bands = 8 if scene == "old_mine" else 4
sc = getattr(pkg.scenes, scene)(bands)
ctx = pkg.Context(num_bands=bands)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
s = ctx.create_source(sc.source)
p = pkg.default_params(num_rays=rays, depth=0)
NW = 65536
cptr = C.c_void_p()
assert hip.hipMalloc(C.byref(cptr), 64 * NW) == 0
hip.hipMemset(cptr, 0, 64 * NW)
lib.fs_debug_wave_buffer(None); lib.fs_debug_connect_buffer(None)
for i in range(8):
    p.seed = 10 + i
    ctx.compute_energy_response_async(s, p); ctx.synchronize()
lib.fs_debug_connect_buffer(cptr)
p.seed = 0x5EED
ctx.compute_energy_response_async(s, p); ctx.synchronize()
lib.fs_debug_connect_buffer(None)
cbuf = np.zeros((NW, 8), np.uint64)
assert hip.hipMemcpy(cbuf.ctypes.data, cptr, cbuf.nbytes, 2) == 0
cl = cbuf[cbuf[:, 4] > 0].astype(np.float64)
t0 = cl[:, 0].min()
ph = {"start": (cl[:, 0] - t0) / 100.0, "setup": (cl[:, 1] - cl[:, 0]) / 100.0, "visibility": (cl[:, 2] - cl[:, 1]) / 100.0,
      "evaluate_deposit": (cl[:, 3] - cl[:, 2]) / 100.0, "flush": (cl[:, 4] - cl[:, 3]) / 100.0, "end": (cl[:, 4] - t0) / 100.0}
out = {"scene": scene, "rays": rays, "connect_waves": int(len(cl)),
       "us": {k: {q: round(float(np.percentile(v, x)), 1) for q, x in (("p10", 10), ("p50", 50), ("p90", 90), ("max", 100))} for k, v in ph.items()}}
print(json.dumps(out))
ctx.close()
