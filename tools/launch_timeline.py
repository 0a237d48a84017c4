#!/usr/bin/env python3
"""Diagnostic (not the product build): which waves of ONE fused cfg3 launch (plans | walks | connects | reconstructs of
FPL frames each) are resident when.  Needs the timeline build: bash tools/build_variant.sh timeline -DFS_WAVE_TIMELINE
usage (GPU box): [FPL=2] python tools/launch_timeline.py > gpurun_out/r03_launch_timeline.json"""
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
fpl = int(os.environ.get("FPL", "2"))
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
s = ctx.create_source(sc.source)
R = 262144
p = pkg.default_params(num_rays=R, depth=8)
NW = 65536
dptr, cptr = C.c_void_p(), C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), 64 * NW) == 0 and hip.hipMalloc(C.byref(cptr), 64 * NW) == 0
hip.hipMemset(dptr, 0, 64 * NW); hip.hipMemset(cptr, 0, 64 * NW)
lib.fs_debug_wave_buffer(None); lib.fs_debug_connect_buffer(None)
ctx.set_pipelining(2)
ctx.set_frames_per_launch(fpl)


def frames(n, seed):
    for i in range(n):
        p.seed = seed + i
        ctx.compute_energy_response_async(s, p)
        ctx.reconstruct_impulse_response_async(s, p)


frames(12 * fpl, 200)
hip.hipDeviceSynchronize()
lib.fs_debug_wave_buffer(dptr); lib.fs_debug_connect_buffer(cptr)
frames(fpl, 0x5EED)          # exactly one launch
hip.hipDeviceSynchronize()
lib.fs_debug_wave_buffer(None); lib.fs_debug_connect_buffer(None)
frames(4 * fpl, 900)
ctx.synchronize()
buf = np.zeros((NW, 8), np.uint64)
assert hip.hipMemcpy(buf.ctypes.data, dptr, buf.nbytes, 2) == 0
cbuf = np.zeros((NW, 8), np.uint64)
assert hip.hipMemcpy(cbuf.ctypes.data, cptr, cbuf.nbytes, 2) == 0
b = buf[buf[:, 1] > 0].astype(np.float64)
cl = cbuf[cbuf[:, 4] > 0].astype(np.float64)
t0 = min(b[:, 0].min(), cl[:, 0].min())
ws, we = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0
cs, ce = (cl[:, 0] - t0) / 100.0, (cl[:, 4] - t0) / 100.0
span = float(max(we.max(), ce.max()))
segs = b[:, 5]
res = {"frames_per_launch": fpl, "walk_waves": int(len(b)), "connect_waves": int(len(cl)), "span_us_walks_and_connects": span,
       "walk_end_us_max": float(we.max()), "connect_end_us_max": float(ce.max()), "connect_start_us_min": float(cs.min()),
       "connect_start_us_p50": float(np.median(cs))}
by = {}
for L in range(1, 9):
    m = segs == L
    if m.sum():
        by[L] = {"waves": int(m.sum()), "start_us_p50": float(np.median(ws[m])), "start_us_max": float(ws[m].max()),
                 "duration_us_p50": float(np.median(we[m] - ws[m])), "duration_us_max": float((we[m] - ws[m]).max()),
                 "end_us_p50": float(np.median(we[m])), "end_us_max": float(we[m].max())}
res["walk_waves_by_length"] = by
grid = np.linspace(0, span, 45)
res["resident_over_time"] = [[round(float(t), 1), int(((ws <= t) & (we > t)).sum()), int(((cs <= t) & (ce > t)).sum())] for t in grid]
print(json.dumps(res))
