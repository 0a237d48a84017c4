#!/usr/bin/env python3
"""How far is the chunked reconstruct (each chunk restarts the one-pole filter 96 samples early, fs_device.hpp: reconstruct_body*)
from the serial recurrence of FSAC.cpp:366-375 in BITS?  Counts the samples of the band IRs and of the channel view that differ
from the oracle's serial loop, and the largest difference in ulps.  (loads the oracle: a measurement script, GPU box)
usage: python tests/measure_recon_bits.py > profiles/r05_recon_bits.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

pkg = graft.load_package()
oracle.load()


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia); ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return np.abs(ia - ib)


out = []
for scene, bands, rays, depth, kw in (("shoebox", 1, 1024, 4, {}), ("starter_room", 4, 16384, 8, {}), ("old_mine", 8, 262144, 8, {}),
                                      ("starter_room", 4, 16384, 8, {"dist_divisor": 100.0}), ("old_mine", 8, 262144, 0, {}),
                                      ("starter_room", 4, 65536, 8, {"dist_divisor": 10.0, "energy_gain": 1e6})):
    sc = pkg.scenes.by_name(scene, bands)
    ctx = pkg.Context(num_bands=bands)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    for seed in (1, 2, 3):
        p = pkg.default_params(num_rays=rays, depth=depth, seed=seed, **kw)
        e = np.array(ctx.compute_energy_response(src, p), copy=True)
        ctx.reconstruct_impulse_response(src, p)
        # "significant": at least 1e-6 of the row's largest sample (below that lies the filter's 0.75^n tail of the last occupied bin,
        # which the serial loop follows down through the denormals and the chunked kernel ends with exact zeros after its run-in)
        diff = 0; worst = 0; sig = 0; tail_abs = 0.0; big = 0.0
        for b in range(bands):
            ref = oracle.reconstruct(e[b]); got = np.array(ctx.band_impulse_response(src, b), copy=True)
            m = np.abs(ref) >= 1e-6 * np.abs(ref).max()
            u = ulps(got, ref); diff += int((u[m] != 0).sum()); worst = max(worst, int(u[m].max())); sig += int(m.sum())
            if (u[m] != 0).any(): big = max(big, float(np.abs(ref[m][u[m] != 0]).max() / np.abs(ref).max()))
            tail_abs = max(tail_abs, float(np.abs(got[~m] - ref[~m]).max() / np.abs(ref).max()) if (~m).any() else 0.0)
        mean_e = (e.sum(axis=0, dtype=np.float32) / np.float32(bands)).astype(np.float32)
        ref = oracle.reconstruct(mean_e); got = np.array(ctx.impulse_response(src, 0), copy=True)
        m = np.abs(ref) >= 1e-6 * np.abs(ref).max()
        u = ulps(got, ref)
        out.append({"scene": scene, "rays": rays, "depth": depth, "seed": seed, **kw, "band_samples_significant": sig, "band_samples_differing": diff,
                    "band_worst_ulps": worst, "channel_samples_significant": int(m.sum()), "channel_samples_differing": int((u[m] != 0).sum()),
                    "channel_worst_ulps": int(u[m].max()), "insignificant_max_abs_diff_over_peak": tail_abs,
                    "largest_differing_band_sample_over_peak": big})
    ctx.close()
print(json.dumps(out, indent=1))
