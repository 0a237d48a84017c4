"""Where one reference-sized update (1000 pairs, uncapped walks, 1 band) spends its time: per-kernel event times
(fs_set_profiling(2)) beside the wall time of compute + reconstruct + synchronize.  Usage: python tools/ref_defaults_breakdown.py"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("audio-pathtracer_amd")

for rname in ("starter_room", "old_mine"):
    for depth in (0, 8):
        sc = pkg.scenes.by_name(rname, 1)
        c = pkg.Context(num_bands=1)
        c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        c.set_listener(sc.listener)
        s = c.create_source(sc.source)
        p = pkg.default_params(num_rays=2000, depth=depth, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
        out = {"scene": rname, "depth": depth}
        for prof in (0, 2):
            c.set_profiling(prof, 1)
            tt = []
            c.reset_stats() if hasattr(c, "reset_stats") else None
            st0 = c.stats()
            for i in range(60):
                p.seed = 100 + i
                t1 = time.perf_counter()
                c.compute_energy_response_async(s, p)
                t2 = time.perf_counter()
                c.reconstruct_impulse_response_async(s, p)
                t3 = time.perf_counter()
                c.synchronize()
                tt.append((time.perf_counter() - t1, t2 - t1, t3 - t2))
            tt = sorted(tt[10:])
            m = tt[len(tt) // 2]
            out[f"prof{prof}"] = {"wall_ms": round(1e3 * m[0], 4), "issue_compute_ms": round(1e3 * m[1], 4), "issue_recon_ms": round(1e3 * m[2], 4)}
            if prof:
                st = c.stats()
                d = {k: st[k] - st0.get(k, 0) for k in st if isinstance(st[k], (int, float))}
                out["kernel_ms"] = {k: round(d[k] / max(d.get("timed_frames", 1), 1), 4) for k in d if k.endswith("_ms_sum")}
                out["segments_per_subpath"] = d.get("segments", 0) / max(d.get("rays", 1), 1)
        print(json.dumps(out), flush=True)
        c.close()
