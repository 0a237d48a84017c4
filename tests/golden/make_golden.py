#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run in the build container):

    python tests/golden/make_golden.py [case names ...]

The reference has no golden vectors of its own and cannot run here, so these pin the
ORACLE (a later change to oracle/fs_oracle.c that alters any output fails tests/test_golden.py) and give
the GPU tests an input/expected-output pair that does not need the oracle at run time.
Fixtures are data only: seeds/params (inputs) and energy histograms, IR checksums, ray hits (outputs).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

CASES = [
    # name, scene, bands, pairs, depth, seed, extra oracle params
    ("cfg1_shoebox", "shoebox", 1, 512, 4, 0x5EED, {}),
    ("cfg1_shoebox_unbounded", "shoebox", 1, 1000, 0, 0x5EED, {}),
    ("cfg2_starter_room", "starter_room", 4, 8192, 8, 0x5EED, {}),
    ("cfg2_starter_room_metres", "starter_room", 4, 4096, 8, 77, {"dist_divisor": 100.0}),
    ("cfg2_starter_room_fixed_depth", "starter_room", 4, 2048, 6, 5, {"russian_roulette": 0}),
    ("cfg3_old_mine", "old_mine", 8, 131072, 8, 0x5EED, {}),
    # row f3 (FS_FLAG_ALL_CONNECTIONS == FSO_FLAG_ALL_CONNECTIONS == 16)
    ("cfg1_shoebox_all_connections", "shoebox", 1, 512, 4, 0x5EED, {"flags": 16}),
    ("cfg2_starter_room_all_connections", "starter_room", 4, 2048, 8, 9, {"flags": 16}),
    # row f3 with balance-heuristic weights (FS_FLAG_MIS_BALANCE == FSO_FLAG_MIS_BALANCE == 32)
    ("cfg1_shoebox_mis_balance", "shoebox", 1, 512, 4, 0x5EED, {"flags": 32}),
    ("cfg2_starter_room_mis_balance", "starter_room", 4, 2048, 8, 9, {"flags": 32}),
    # row f4: lobes in the walk (FS_FLAG_MATERIAL_LOBES == FSO_FLAG_MATERIAL_LOBES == 64), Transmission / Scattering
    # arrays from scenes.material_lobes(scene) ("lobes": 1 is a fixture key, not a parameter)
    ("cfg1_shoebox_material_lobes", "shoebox", 1, 512, 6, 0x5EED, {"flags": 64, "lobes": 1}),
    ("cfg2_starter_room_material_lobes", "starter_room", 4, 2048, 8, 9, {"flags": 64, "lobes": 1}),
    # BASELINE.json configs[4] (8 sources x 131 072 rays in old_mine, one listener): source 4 of the eight — what a batched frame
    # must give that source ("source": k is a fixture key: the source sits at scene.extra_sources[k])
    ("cfg5_multi_source", "old_mine", 8, 65536, 8, 0x5EED, {"source": 4}),
    # BASELINE.json configs[3] at FULL size: 1 048 576 rays (524 288 pairs), depth 12 — the frame the 8 ranks' shards sum to
    ("cfg4_old_mine_1m_d12", "old_mine", 8, 524288, 12, 0x5EED, {}),
]
FIXTURE_KEYS = ("lobes", "source")   # keys of `extra` that describe the fixture, not oracle parameters


def main():
    pkg = graft.load_package()
    only = set(sys.argv[1:])   # optional: regenerate just the named cases
    for name, scene, bands, pairs, depth, seed, extra in CASES:
        if only and name not in only:
            continue
        sc = pkg.scenes.by_name(scene, bands)
        tau, sigma = pkg.scenes.material_lobes(sc) if extra.get("lobes") else (None, None)
        osc = oracle.Scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
        p = oracle.default_params(num_pairs=pairs, depth=depth, seed=seed, **{k: v for k, v in extra.items() if k not in FIXTURE_KEYS})
        src_pos = sc.extra_sources[int(extra["source"])] if "source" in extra else sc.source
        e32, e64, cnt = osc.compute_energy_mt(p, src_pos, sc.listener, threads=8) if pairs > 20000 else \
            osc.compute_energy(p, src_pos, sc.listener)
        if pairs > 20000:  # the literal sequential-f32 histogram needs the single-threaded order
            e32 = osc.compute_energy(p, src_pos, sc.listener)[0]
        ir = np.stack([oracle.reconstruct(e32[b]) for b in range(bands)])
        mean_e = (e32.astype(np.float32).sum(axis=0, dtype=np.float32) / np.float32(bands)).astype(np.float32) \
            if bands > 1 else e32[0]
        rng = np.random.default_rng(seed)
        lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
        n_rays = 256
        o = np.where(rng.random((n_rays, 1)) < 0.5, src_pos + rng.normal(0, 50, (n_rays, 3)),
                     rng.uniform(lo, hi, (n_rays, 3))).astype(np.float32)
        d = rng.normal(size=(n_rays, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
        hit = np.zeros(n_rays, np.int32)
        t = np.zeros(n_rays, np.float32)
        tri = np.full(n_rays, -1, np.int32)
        nrm = np.zeros((n_rays, 3), np.float32)
        for i in range(n_rays):
            h, tt, ti, nn = osc.trace_closest(o[i], d[i], 1e6, brute=True)
            hit[i], t[i], tri[i], nrm[i] = h, tt if h else 1e6, ti if h else -1, nn if h else 0
        out = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(
            out, scene=scene, bands=bands, pairs=pairs, depth=depth, seed=seed,
            extra_keys=np.array(list(extra.keys())), extra_vals=np.array([float(v) for v in extra.values()]),
            energy_f32=e32, energy_f64=e64, connected=cnt.connected, closest_rays=cnt.closest_rays,
            ir_sum=ir.astype(np.float64).sum(axis=1), ir_abs_sum=np.abs(ir).astype(np.float64).sum(axis=1),
            ir_max=ir.max(axis=1), ir_decimated=ir[:, ::16].astype(np.float32),
            mean_energy=mean_e, ray_o=o, ray_d=d, ray_hit=hit, ray_t=t, ray_tri=tri, ray_n=nrm,
            tri_checksum=np.float64(sc.triangles.astype(np.float64).sum()), num_tris=sc.num_triangles)
        print(f"{name}: connected {cnt.connected}/{pairs}, nonzero bins {int((e32 != 0).sum())}, "
              f"{os.path.getsize(out)} bytes")


if __name__ == "__main__":
    main()
