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
                os.path.join(src, "fs_bvh.cpp"), os.path.join(src, "fs_kernels.hip"), os.path.join(src, "fs_fft.hip"),
                os.path.join(src, "fs_refit.hip"), os.path.join(src, "fs_build.hip")], check=True)
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
buf = (C.c_ulonglong * 32)()
res = {}
for plan in (1,):
    lib.fs_debug_trav_stats(buf, 1)
    ctx.compute_energy_response(s, p)
    lib.fs_debug_trav_stats(buf, 1)
    v = list(buf)
    res = {"wave_step_calls": v[0], "node_iterations": v[1], "node_lane_steps": v[2], "tri_iterations": v[3],
           "tri_lane_steps": v[4], "node_lanes_per_iteration": v[2] / max(v[1], 1), "tri_lanes_per_iteration": v[4] / max(v[3], 1),
           "node_visits_by_children_hit_0_1_2plus": [v[5], v[6], v[7]],
           "lanes_with_both_kinds": v[11], "sharing_loop_iterations": v[10], "busy_lanes_per_iteration": v[8] / max(v[10], 1),
           "lanes_on_taken_work_per_iteration": v[9] / max(v[10], 1),
           "visibility": {"wave_step_calls": v[16], "node_iterations": v[17], "node_lane_steps": v[18], "tri_iterations": v[19],
                          "tri_lane_steps": v[20], "sharing_loop_iterations": v[26], "busy_lanes_per_iteration": v[24] / max(v[26], 1),
                          "lanes_on_taken_work_per_iteration": v[25] / max(v[26], 1)}}
if os.environ.get("FS_WALK_VARIANT", "2") != "0":   # the per-segment counts below are recorded by the one-subpath-per-lane
    print(json.dumps(res))                           # kernel only: FS_WALK_VARIANT=0 python tests/trav_stats.py
    sys.exit(0)
# per-segment iteration counts: how much of a wave's time is waiting for its slowest ray, and how well the
# cost of a subpath's next segment could be predicted from its previous one
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
import numpy as np  # noqa: E402
D, R = 8, 262144
dptr = C.c_void_p()
assert hip.hipMalloc(C.byref(dptr), 2 * D * R) == 0
hip.hipMemset(dptr, 0, 2 * D * R)
lib.fs_debug_step_buffer.argtypes = [C.c_void_p]
lib.fs_debug_step_buffer(dptr)
ctx.compute_energy_response(s, p)
steps = np.zeros((D, R), np.uint16)
assert hip.hipMemcpy(steps.ctypes.data, dptr, steps.nbytes, 2) == 0
lib.fs_debug_step_buffer(None)
length = (steps > 0).sum(axis=0)
res["segments_recorded"] = int((steps > 0).sum())
res["mean_steps_per_segment"] = float(steps[steps > 0].mean())
res["p50_p90_p99_max"] = [float(np.percentile(steps[steps > 0], q)) for q in (50, 90, 99, 100)]
full = steps[:, length == D].astype(np.float64)            # the walks that take all 8 segments
res["corr_consecutive_segments"] = float(np.mean([np.corrcoef(full[k], full[k + 1])[0, 1] for k in range(D - 1)]))
rng = np.random.default_rng(0)
n = full.shape[1] // 64 * 64
def wave_steps(order):   # per bounce the wave waits for its slowest lane (walks of equal length grouped, as the plan does)
    return float(sum(full[k, order[k]][:n].reshape(-1, 64).max(axis=1).sum() for k in range(D)))
ident = [np.arange(full.shape[1])] * D
res["ideal_over_actual_wave_steps"] = float(full[:, :n].sum() / 64.0 / wave_steps(ident))
# regroup every bounce by the previous segment's cost (what an in-block re-sort between bounces could do at best)
by_prev = [np.arange(full.shape[1])] + [np.argsort(full[k - 1], kind="stable") for k in range(1, D)]
res["wave_steps_sorted_by_previous_cost_over_actual"] = wave_steps(by_prev) / wave_steps(ident)
# two walks per lane, traced back to back in each bounce
pair_cost = full[:, :n // 128 * 128].reshape(D, -1, 2).sum(axis=2)
res["two_walks_per_lane_over_actual"] = float(sum(pair_cost[k].reshape(-1, 64).max(axis=1).sum() for k in range(D)) /
                                              wave_steps([np.arange(full.shape[1])[:n // 128 * 128]] * D)) if n >= 128 else None
# bounce 0: all rays of a side leave one point — would grouping them by direction shorten the waves' waits?
import oracle  # noqa: E402  (diagnostic script under tests/: may use the oracle)
olib = oracle.load()
P = R // 2
dirs = np.zeros((R, 3), np.float32)
ctr = (C.c_uint32 * 4)(); key = (C.c_uint32 * 2)(p.seed & 0xFFFFFFFF, p.seed >> 32); out4 = (C.c_uint32 * 4)(); d3 = (C.c_float * 3)()
sub = np.arange(0, R, 8)                                   # every 8th subpath is plenty for the estimate
for g in sub:
    side = 1 if g >= P else 0
    pair = g - side * P
    ctr[0], ctr[1], ctr[2], ctr[3] = pair, side, 0, 0x46533031
    olib.fso_philox4x32_10(ctr, key, out4)
    olib.fso_sample_sphere(p.seed, pair, side, 0, out4, d3)
    dirs[g] = (d3[0], d3[1], d3[2])
s0 = steps[0, sub].astype(np.float64)
alive = s0 > 0
octant = ((dirs[sub, 0] > 0).astype(int) | ((dirs[sub, 1] > 0).astype(int) << 1) | ((dirs[sub, 2] > 0).astype(int) << 2)) + 8 * (sub >= P)
def waves(order):
    v = s0[alive][order]
    m = v.size // 64 * 64
    return float(v[:m].reshape(-1, 64).max(axis=1).sum())
base = waves(np.arange(alive.sum()))
az = np.arctan2(dirs[sub, 1], dirs[sub, 0])[alive]
res["bounce0_wave_steps_sorted_by_octant_over_unsorted"] = waves(np.argsort(octant[alive], kind="stable")) / base
res["bounce0_wave_steps_sorted_by_side_azimuth_over_unsorted"] = waves(np.lexsort((az, (sub >= P)[alive]))) / base
res["bounce0_wave_steps_sorted_by_cost_over_unsorted"] = waves(np.argsort(s0[alive])) / base
res["bounce0_share_of_all_segments"] = float((steps[0] > 0).sum() / (steps > 0).sum())
print(json.dumps(res))
