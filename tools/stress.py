#!/usr/bin/env python3
"""API state-machine stress (product path only): a random sequence of frames in all modes, moving geometry, energy
helpers, installed IRs, reverb callbacks and stats on one context, with checkpoints where the energy of a frame is
compared with the same frame computed synchronously on a second, freshly synchronised context that saw the same
geometry.  usage: python tools/stress.py [iterations=400] [seed=1]"""
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
    for _ in range(2):
        c = pkg.Context(num_bands=4)
        c.set_scene(tri, sc.material_ids, sc.absorption, object_ids=sc.object_ids)
        c.set_listener(sc.listener)
        ctxs.append((c, c.create_source(sc.source)))
    (a, sa), (b, sb) = ctxs
    a.reverb_init(sa, 1024)
    flags_pool = [0, 0, 0, pkg._capi.FLAG_DETERMINISTIC, pkg._capi.FLAG_ALL_CONNECTIONS,
                  pkg._capi.FLAG_ALL_CONNECTIONS | pkg._capi.FLAG_DETERMINISTIC, pkg._capi.FLAG_COSINE_SAMPLING]
    checks = 0
    last = None
    for it in range(iters):
        op = rng.integers(0, 10)
        if op <= 4:                                             # a frame, asynchronously
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
