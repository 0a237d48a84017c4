#!/usr/bin/env python3
"""Soak of the publish path (round 5: ring slot written by the launch + pinned host word): a producer streams pipelined frames that
alternate between two deterministic parameter sets (each with ONE bit-exact IR) while reader threads keep copying whatever
fs_get_impulse_response points at, bracketed by the sequence number; a copy taken while the number stood still must equal one of the
two IRs.  A word that overtook its samples, a slot recycled too early or a zero block skipped wrongly shows as a mismatch.
usage (GPU box): python tools/publish_soak.py [seconds=60] [frames_per_launch=2] [readers=2]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
fpl = int(sys.argv[2]) if len(sys.argv) > 2 else 2
readers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
sc = pkg.scenes.starter_room(4)
ctx = pkg.Context(num_bands=4)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
src = ctx.create_source(sc.source)
DET = pkg._capi.FLAG_DETERMINISTIC
# two parameter sets whose IRs differ in WHICH blocks are non-zero (the second: delays x 100 and a gain that lifts them over the
# amplitude threshold: blocks 4 - 6 only), so that the zero-block rule flips bits of the slot masks all the time
params = [pkg.default_params(num_rays=4096, depth=8, seed=5, flags=DET),
          pkg.default_params(num_rays=4096, depth=8, seed=6, flags=DET, dist_divisor=10.0, energy_gain=1e12)]
irs = []
for p in params:
    ctx.compute_energy_response(src, p)
    ctx.reconstruct_impulse_response(src, p)
    irs.append(ctx.impulse_response(src, 0).copy())
assert irs[0].any() and irs[1].any() and not np.array_equal(irs[0], irs[1])
ctx.set_pipelining(2)
ctx.set_frames_per_launch(fpl)
stop = threading.Event()
stat = [dict(reads=0, stable=0, bad=0, a=0, b=0) for _ in range(readers)]


def reader(k):
    st = stat[k]
    while not stop.is_set():
        s0 = ctx.impulse_response_sequence(src)
        v = ctx.impulse_response_view(src, 0).copy()
        s1 = ctx.impulse_response_sequence(src)
        st["reads"] += 1
        if s0 != s1 or s0 == 0:
            continue                      # a publish arrived meanwhile: the front may have moved under the copy
        st["stable"] += 1
        if np.array_equal(v, irs[0]):
            st["a"] += 1
        elif np.array_equal(v, irs[1]):
            st["b"] += 1
        else:
            st["bad"] += 1
            bad_at = int(np.flatnonzero((v != irs[0]) & (v != irs[1]))[0]) if ((v != irs[0]) & (v != irs[1])).any() else -1
            print(f"reader {k}: publish {s0} matches neither IR (first foreign sample {bad_at})", flush=True)


ths = [threading.Thread(target=reader, args=(k,)) for k in range(readers)]
for t in ths:
    t.start()
t0 = time.time()
frames = 0
rng = np.random.default_rng(1)
try:
    while time.time() - t0 < seconds:
        run = int(rng.integers(1, 14))                      # runs shorter and longer than the ring of 8
        which = int(rng.integers(0, 2))
        for _ in range(run):
            ctx.compute_energy_response_async(src, params[which])
            ctx.reconstruct_impulse_response_async(src, params[which])
            frames += 1
        if rng.random() < 0.05:
            ctx.submit()
        if rng.random() < 0.01:
            ctx.synchronize()
    ctx.synchronize()
finally:
    stop.set()
    for t in ths:
        t.join()
tot = {k: sum(s[k] for s in stat) for k in stat[0]}
print({"seconds": round(time.time() - t0, 1), "frames": frames, "frames_per_launch": fpl, **tot, "counters": ctx.pipeline_counters()})
assert tot["bad"] == 0 and tot["a"] > 0 and tot["b"] > 0
print("soak ok")
ctx.close()
