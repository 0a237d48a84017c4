#!/usr/bin/env python3
"""Diagnostic (not the product build): where a cooperative walk wave of the reference-sized update (1000 pairs, uncapped
walks, 1 band) spends its cycles.  Needs the timeline build: bash tools/build_variant.sh timeline -DFS_WAVE_TIMELINE
usage (GPU box): python tools/coop_timing.py [scene] > gpurun_out/r04_coop_timing.json"""
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
for name in (sys.argv[1:] or ["starter_room", "old_mine"]):
    sc = pkg.scenes.by_name(name, 1)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    s = ctx.create_source(sc.source)
    p = pkg.default_params(num_rays=2000, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
    NW = 8192
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), 64 * NW) == 0
    hip.hipMemset(dptr, 0, 64 * NW)
    lib.fs_debug_wave_buffer(None)
    for i in range(20):
        p.seed = 100 + i
        ctx.compute_energy_response_async(s, p)
        ctx.synchronize()
    lib.fs_debug_wave_buffer(dptr)
    p.seed = 0x5EED
    ctx.compute_energy_response_async(s, p)
    ctx.synchronize()
    lib.fs_debug_wave_buffer(None)
    buf = np.zeros((NW, 8), np.uint64)
    assert hip.hipMemcpy(buf.ctypes.data, dptr, buf.nbytes, 2) == 0
    buf_i = buf[buf[:, 1] > 0]
    hw = (buf_i[:, 5] >> np.uint64(32)).astype(np.int64)          # HW_ID bits [15:0] | XCC_ID << 16
    buf_i[:, 5] &= np.uint64(0xFFFFFFFF)
    b = buf_i.astype(np.float64)
    t0 = b[:, 0].min()
    dur = (b[:, 1] - b[:, 0]) / 100.0
    order = np.argsort(-b[:, 5])
    rows = []
    for i in order[:10]:
        q = max(b[i, 5], 1)
        rows.append({"queries": int(b[i, 5]), "duration_us": round(float(dur[i]), 1), "us_per_query": round(float(dur[i] / q), 2),
                     "cycles_total": int(b[i, 3]), "cycles_per_query": int(b[i, 3] / q), "traversal_cycles_per_query": int(b[i, 2] / q),
                     "loop_head_cycles_per_query": int(b[i, 6] / q), "steps_per_query": round(float(b[i, 4] / q), 2),
                     "cycles_per_step": int(b[i, 2] / max(b[i, 4], 1)), "start_us": round(float((b[i, 0] - t0) / 100.0), 1),
                     "where": {"xcc": int(hw[i] >> 16) & 15, "se": int(hw[i] >> 13) & 7, "sh": int(hw[i] >> 12) & 1, "cu": int(hw[i] >> 8) & 15, "simd": int(hw[i] >> 4) & 3, "wave_slot": int(hw[i]) & 15},
                     "same_simd_waves": sorted(int(x) for x in b[(hw >> 4) == (hw[i] >> 4), 5])[::-1][:6],
                     "per_query_cycles": {"before_loop": int(16 * (int(buf_i[i, 7]) & 0xFFFF) / q), "triangle_sections": int(16 * ((int(buf_i[i, 7]) >> 16) & 0xFFFF) / q),
                                          "waiting_for_records": int(16 * ((int(buf_i[i, 7]) >> 32) & 0xFFFF) / q), "behind_loop": int(16 * ((int(buf_i[i, 7]) >> 48) & 0xFFFF) / q)}})
    # what a query costs a walk by the walk's length: the short walks only see the first bounces, when every wave of the frame is alive
    by_len = []
    for lo, hi in ((1, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 1000)):
        m = (b[:, 5] >= lo) & (b[:, 5] < hi)
        if m.any():
            by_len.append({"queries": "%d..%d" % (lo, hi - 1), "waves": int(m.sum()), "us_per_query": round(float(dur[m].sum() / b[m, 5].sum()), 2),
                           "cycles_per_query": int(b[m, 3].sum() / b[m, 5].sum())})
    tot_q = b[:, 5].sum()
    print(json.dumps({"scene": name, "waves": int(len(b)), "span_us": float((b[:, 1].max() - t0) / 100.0),
                      "all_waves": {"queries": int(tot_q), "steps_per_query": float(b[:, 4].sum() / tot_q),
                                    "traversal_cycles_per_query": float(b[:, 2].sum() / tot_q), "cycles_per_query": float(b[:, 3].sum() / tot_q)},
                      "by_walk_length": by_len, "longest_walks": rows}), flush=True)
    ctx.close()
