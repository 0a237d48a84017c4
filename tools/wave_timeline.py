#!/usr/bin/env python3
"""Diagnostic (not the product build): when every wave of the cfg3 walk kernel ran and what it spent its cycles on.
Builds a -DFS_WAVE_TIMELINE copy of libfrequensee.so into gpurun_out/ and traces a few cfg3 frames through it.
usage (GPU box): [FS_TIMELINE_PIPELINED=1] python tools/wave_timeline.py [extra -D flags]  > gpurun_out/r02_wave_timeline.json"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "timeline_build")
os.makedirs(out, exist_ok=True)
src = os.path.join(ROOT, "audio-pathtracer_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                "-munsafe-fp-atomics", "--offload-arch=gfx950", "-DFS_WAVE_TIMELINE", *sys.argv[1:], "-shared", "-o",
                os.path.join(out, "libfrequensee.so"), "-x", "hip", os.path.join(src, "fs_capi.cpp"),
                os.path.join(src, "fs_bvh.cpp"), os.path.join(src, "fs_kernels.hip"), os.path.join(src, "fs_fft.hip"),
                os.path.join(src, "fs_refit.hip"), os.path.join(src, "fs_build.hip")], check=True, stderr=subprocess.DEVNULL)
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
pkg._capi.LIB_PATH = os.path.join(out, "libfrequensee.so")
pkg._capi._lib = None
lib = pkg._capi.load()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
s = ctx.create_source(sc.source)
R = 262144
W = R // 64
p = pkg.default_params(num_rays=R, depth=8)
dptr = C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), 8 * 8 * W) == 0
lib.fs_debug_wave_buffer.argtypes = [C.c_void_p]
for i in range(5):   # warm
    p.seed = 100 + i
    ctx.compute_energy_response(s, p)
hip.hipMemset(dptr, 0, 8 * 8 * W)
lib.fs_debug_wave_buffer(dptr)
cptr = C.c_void_p()
CW = 4096   # connect waves (grid is capped at 1024 workgroups)
assert hip.hipMalloc(C.byref(cptr), 8 * 8 * CW) == 0
hip.hipMemset(cptr, 0, 8 * 8 * CW)
lib.fs_debug_connect_buffer.argtypes = [C.c_void_p]
lib.fs_debug_connect_buffer(cptr)
p.seed = 0x5EED
if os.environ.get("FS_TIMELINE_PIPELINED") == "1":
    # the pipelined launch instead: {plan of frame f, walk of frame f-1, connect of frame f-2} — arm the buffers for
    # exactly one such launch (the stream is drained through HIP, not through the library, which would flush)
    lib.fs_debug_wave_buffer(None); lib.fs_debug_connect_buffer(None)
    ctx.set_pipelining(2)
    for i in range(6):
        p.seed = 200 + i
        ctx.compute_energy_response_async(s, p)
    hip.hipDeviceSynchronize()
    lib.fs_debug_wave_buffer(dptr); lib.fs_debug_connect_buffer(cptr)
    p.seed = 0x5EED
    ctx.compute_energy_response_async(s, p)
    hip.hipDeviceSynchronize()
    lib.fs_debug_wave_buffer(None); lib.fs_debug_connect_buffer(None)
    ctx.synchronize()
else:
    ctx.compute_energy_response(s, p)
buf = np.zeros((W, 8), np.uint64)
assert hip.hipMemcpy(buf.ctypes.data, dptr, buf.nbytes, 2) == 0
lib.fs_debug_wave_buffer(None)
cbuf = np.zeros((CW, 8), np.uint64)
assert hip.hipMemcpy(cbuf.ctypes.data, cptr, cbuf.nbytes, 2) == 0
lib.fs_debug_connect_buffer(None)
live = buf[:, 1] > 0
b = buf[live].astype(np.float64)
t0 = b[:, 0].min()
start, end = (b[:, 0] - t0) / 100.0, (b[:, 1] - t0) / 100.0      # microseconds
segs = b[:, 5]
res = {"waves_with_work": int(live.sum()), "kernel_span_us": float(end.max()),
       "clock_mhz": float(np.median(b[:, 3] / np.maximum(b[:, 1] - b[:, 0], 1) * 100.0))}
by_len = {}
for L in range(1, 9):
    m = segs == L
    if m.sum():
        by_len[L] = {"waves": int(m.sum()), "start_us_p50": float(np.median(start[m])), "start_us_max": float(start[m].max()),
                     "duration_us_p50": float(np.median(end[m] - start[m])), "duration_us_p90": float(np.percentile(end[m] - start[m], 90)),
                     "duration_us_max": float((end[m] - start[m]).max()), "end_us_max": float(end[m].max()),
                     "cycles_per_segment_p50": float(np.median(b[m, 3] / L)),
                     "traversal_share": float(np.median(b[m, 2] / np.maximum(b[m, 3], 1)))}
res["by_walk_length"] = by_len
grid = np.linspace(0, end.max(), 41)
res["active_waves_over_time"] = [[float(t), int(((start <= t) & (end > t)).sum())] for t in grid]
cu = (b[:, 6].astype(np.uint64) & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (b[:, 6].astype(np.uint64) >> np.uint64(32)).astype(np.int64)
res["xcc_ids_seen"] = sorted(set(int(x) for x in xcc))
m8 = segs == 8
if m8.sum():
    key = xcc[m8] * 100000 + cu[m8]
    _, counts = np.unique(key, return_counts=True)
    res["len8_waves_per_hw_slot_hist"] = {int(k): int(v) for k, v in zip(*np.unique(counts, return_counts=True))}
cl = cbuf[cbuf[:, 4] > 0].astype(np.float64)
if len(cl):
    c0 = cl[:, 0].min()
    ph = lambda a, b: {"p50": float(np.median((cl[:, b] - cl[:, a]) / 100.0)), "p90": float(np.percentile((cl[:, b] - cl[:, a]) / 100.0, 90)),
                       "max": float(((cl[:, b] - cl[:, a]) / 100.0).max())}
    cs, ce = (cl[:, 0] - t0) / 100.0, (cl[:, 4] - t0) / 100.0
    res["connect_waves_over_time"] = [[float(t), int(((cs <= t) & (ce > t)).sum())] for t in np.linspace(0, max(end.max(), ce.max()), 41)]
    res["launch_span_us"] = float(max(end.max(), ce.max()))
    res["connect"] = {"waves": int(len(cl)), "span_us": float((cl[:, 4].max() - c0) / 100.0),
                      "first_start_us": float(cs.min()), "start_us_p50": float(np.median(cs)), "end_us_max": float(ce.max()),
                      "start_us_max": float((cl[:, 0].max() - c0) / 100.0),
                      "gap_after_walk_us": float((c0 - t0) / 100.0 - end.max()),
                      "setup_us": ph(0, 1), "visibility_us": ph(1, 2), "evaluate_us": ph(2, 3), "flush_us": ph(3, 4), "wave_us": ph(0, 4)}
print(json.dumps(res))
