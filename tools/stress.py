#!/usr/bin/env python3
"""API state-machine stress (product path only): a random sequence of frames in all modes (also batched over three
sources), moving geometry, energy
helpers, installed IRs, reverb callbacks and stats on one context, with checkpoints where the energy of a frame is
compared with the same frame computed synchronously on a second, freshly synchronised context that saw the same
geometry.  usage: [FS_STRESS_PIPELINE=1|2] [FS_STRESS_FPL=2..4] [FS_STACK_ROWS_CAP=12] python tools/stress.py [iterations=400] [seed=1]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-300))


def main(iters=400, seed=1):
    pkg = graft.load_package()
    rng = np.random.default_rng(seed)
    sc = pkg.scenes.starter_room(4)
    tri = sc.triangles.copy()
    T = tri.shape[0]
    ctxs = []
    tau, sigma = pkg.scenes.material_lobes(sc)
    for _ in range(2):
        c = pkg.Context(num_bands=4)
        c.set_scene(tri, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma, object_ids=sc.object_ids)
        c.set_listener(sc.listener)
        ctxs.append((c, c.create_source(sc.source)))
    (a, sa), (b, sb) = ctxs
    # two more sources on both contexts: batched frames on `a` against one-by-one frames on `b`
    extra_pos = [sc.source + np.array([150, -80, 20], np.float32), sc.source + np.array([-200, 120, -30], np.float32)]
    extra_a = [a.create_source(q) for q in extra_pos]
    extra_b = [b.create_source(q) for q in extra_pos]
    a.reverb_init(sa, 1024)
    if os.environ.get("FS_STRESS_PIPELINE") in ("1", "2"):   # context a holds connect passes back; b (the reference) never does
        a.set_pipelining(int(os.environ["FS_STRESS_PIPELINE"]))
        if os.environ.get("FS_STRESS_FPL"):                        # ... and lets same-kind frames share launches
            a.set_frames_per_launch(int(os.environ["FS_STRESS_FPL"]))
    F = pkg._capi
    flags_pool = [0, 0, 0, F.FLAG_DETERMINISTIC, F.FLAG_ALL_CONNECTIONS, F.FLAG_ALL_CONNECTIONS | F.FLAG_DETERMINISTIC,
                  F.FLAG_COSINE_SAMPLING, F.FLAG_MIS_BALANCE, F.FLAG_MATERIAL_LOBES, F.FLAG_MATERIAL_LOBES | F.FLAG_ALL_CONNECTIONS,
                  F.FLAG_MATERIAL_LOBES | F.FLAG_DETERMINISTIC]
    checks = 0
    last = None
    for it in range(iters):
        op = rng.integers(0, 14)
        if op == 10:                                            # a batched frame over three sources, checked at once
            p = pkg.default_params(num_rays=int(rng.choice([2, 1554, 4096, 32768])), depth=int(rng.choice([1, 4, 8])),
                                   seed=int(rng.integers(1, 1 << 40)), dist_divisor=100.0, flags=int(rng.choice(flags_pool)),
                                   russian_roulette=int(rng.random() < 0.85))
            group_a, group_b = [sa] + extra_a, [sb] + extra_b
            a.compute_energy_response_batch_async(group_a, p)
            for h in group_a:
                a.reconstruct_impulse_response_async(h, p)
            for ha, hb in zip(group_a, group_b):
                got = a.energy_buffer(ha)
                want = b.compute_energy_response(hb, p)
                if os.environ.get("FS_STRESS_DIAG") == "1" and any(want[k].any() and rel_rms(got[k], want[k]) > 2e-5 for k in range(4)):
                    again_a = a.compute_energy_response(ha, p)
                    again_b = b.compute_energy_response(hb, p)
                    print("DIAG it", it, "a_async vs b", [rel_rms(got[k], want[k]) for k in range(4)], "a_sync vs b", [rel_rms(again_a[k], want[k]) for k in range(4)],
                          "b again vs b", [rel_rms(again_b[k], want[k]) for k in range(4)], "a_sync vs a_async", [rel_rms(again_a[k], got[k]) for k in range(4)],
                          "bins differing", [int(np.count_nonzero(np.abs(got[k] - want[k]) > 1e-6 * np.abs(want[k]).max())) for k in range(4)], flush=True)
                for k in range(4):
                    if want[k].any():
                        assert rel_rms(got[k], want[k]) <= 2e-5, (it, "batch", k, rel_rms(got[k], want[k]), p.num_rays, p.depth, hex(p.flags), p.russian_roulette, p.seed, int(np.count_nonzero(want[k])), float(np.abs(got[k] - want[k]).max()), float(np.abs(want[k]).max()))
            checks += 1
            last = None
        elif op == 11:                                          # the tick as one call (round 4): three sources, every IR published on return
            p = pkg.default_params(num_rays=int(rng.choice([2, 2000, 4096, 40000])), depth=int(rng.choice([0, 0, 4, 8])),
                                   seed=int(rng.integers(1, 1 << 40)), dist_divisor=100.0,
                                   flags=int(rng.choice([0, 0, F.FLAG_DETERMINISTIC, F.FLAG_COSINE_SAMPLING, F.FLAG_FIXED_NORM_1000])),
                                   russian_roulette=1)
            group_a, group_b = [sa] + extra_a, [sb] + extra_b
            before = [a.impulse_response_sequence(h) for h in group_a]
            a.update_sources(group_a, p)
            for ha, hb, n0 in zip(group_a, group_b, before):
                assert a.impulse_response_sequence(ha) > n0, (it, "update_sources published nothing")
                got = a.energy_buffer(ha)
                want = b.compute_energy_response(hb, p)
                for k in range(4):
                    if want[k].any():
                        assert rel_rms(got[k], want[k]) <= 2e-5, (it, "tick", k, rel_rms(got[k], want[k]))
                assert np.all(np.isfinite(a.impulse_response_view(ha, 0)))
            checks += 1
            last = None
        elif op == 12:                                          # the end of a tick: nothing stays held back; or the walks' own actors change
            if rng.random() < 0.5:
                a.submit()
            else:
                oid = int(rng.choice([F.NO_OBJECT, int(sc.object_ids[0]), int(sc.object_ids[-1])]))
                a.set_source_object(sa, oid); b.set_source_object(sb, oid)
                lid = int(rng.choice([F.NO_OBJECT, F.NO_OBJECT, int(sc.object_ids[len(sc.object_ids) // 2])]))
                a.set_listener_object(lid); b.set_listener_object(lid)
            last = None
        elif op == 13:                                          # frames of all three sources, their reconstructs as one launch
            p = pkg.default_params(num_rays=int(rng.choice([512, 4096])), depth=int(rng.choice([0, 4, 8])), seed=int(rng.integers(1, 1 << 40)),
                                   dist_divisor=100.0, flags=int(rng.choice([0, F.FLAG_DETERMINISTIC])), russian_roulette=1)
            group_a = [sa] + extra_a
            a.compute_energy_response_batch_async(group_a, p)
            a.reconstruct_impulse_response_batch_async(group_a, p)
            last = p
        elif op <= 4:                                           # a frame, asynchronously
            p = pkg.default_params(num_rays=int(rng.choice([2, 512, 4096, 16384])), depth=int(rng.choice([1, 4, 8, 0])),
                                   seed=int(rng.integers(1, 1 << 40)), dist_divisor=100.0, flags=int(rng.choice(flags_pool)),
                                   russian_roulette=int(rng.random() < 0.85))
            a.compute_energy_response_async(sa, p)
            if rng.random() < 0.8:
                a.reconstruct_impulse_response_async(sa, p)
            last = p
        elif op == 5 and last is not None:                      # checkpoint: the newest frame against a synchronous one
            got = a.energy_buffer(sa)
            want = b.compute_energy_response(sb, last)
            assert np.array_equal(got != 0, want != 0) or (last.flags & pkg._capi.FLAG_DETERMINISTIC), (it, "bins")
            for k in range(4):
                if want[k].any():
                    assert rel_rms(got[k], want[k]) <= 2e-5, (it, k, rel_rms(got[k], want[k]))
            checks += 1
            last = None
        elif op == 6:                                           # a prop moves (both contexts see it)
            first = int(rng.integers(0, T - 300))
            n = int(rng.integers(1, 300))
            tri[first:first + n] += rng.normal(0, 15, 3).astype(np.float32)
            a.update_triangles(first, tri[first:first + n])
            b.update_triangles(first, tri[first:first + n])
            if rng.random() < 0.5:
                a.refit()
            last = None
        elif op == 7:                                           # energy helpers + reconstruct
            a.check(a.lib.fs_flush_energy_buffer(a.h, sa))
            a.check(a.lib.fs_add_energy_at_delay(a.h, sa, int(rng.integers(0, 4)), float(rng.uniform(0, 1.2)), 1.0))
            a.reconstruct_impulse_response_async(sa)
            last = None
        elif op == 8:                                           # the audio thread's side
            a.reverb_process(sa, np.clip(rng.normal(0, 0.2, 2048), -1, 1).astype(np.float32))
            v = a.impulse_response_view(sa, 0)
            assert np.all(np.isfinite(v))
            if rng.random() < 0.2:
                a.set_impulse_response(sa, (rng.normal(0, 0.01, 48000)).astype(np.float32))
        else:
            st = a.stats()
            assert st["frames"] >= 0 and st["bvh_stack_need"] <= 64
            if rng.random() < 0.3:
                a.reset_stats()
    a.synchronize()
    a.close(); b.close()
    print(f"stress ok: {iters} operations, {checks} checkpoints, seed {seed}")
    return 0


if __name__ == "__main__":
    sys.exit(main(*(int(x) for x in sys.argv[1:3])))
