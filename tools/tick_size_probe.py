#!/usr/bin/env python3
"""What a bounce of the longest walk costs a one-source tick of N subpaths (uncapped walks, one band): tick time, the expected longest
walk ((ln N + 0.577) / ln(1 / 0.9)) and their quotient (45 us of plan / connect / reconstruct taken off).  Found with it: the per-bounce
cost is the same 3.3 - 3.7 us (starter_room) / 4.3 - 4.8 us (old_mine) from 128 to 16 000 subpaths - a lone chain is not slowed by
its neighbours - and a tick of S sources has the walk LENGTHS of one source (every source draws the same RNG pairs): its chain is
78 bounces at any S, not ln(2000 S) / 0.105.   usage (GPU box): python tools/tick_size_probe.py"""
import json, os, sys, time, math
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
for scene in ("starter_room", "old_mine"):
    sc = pkg.scenes.by_name(scene, 1)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    s = ctx.create_source(sc.source)
    for rays in (128, 250, 500, 1000, 2000, 4000, 8000, 16000):
        p = pkg.default_params(num_rays=rays, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
        tt = []
        for i in range(60):
            p.seed = 100 + i
            t1 = time.perf_counter(); ctx.update_sources([s], p); tt.append(time.perf_counter() - t1)
        tt = sorted(tt[10:]); med = 1e3 * tt[len(tt) // 2]
        emax = (math.log(rays) + 0.577) / 0.10536
        print(json.dumps({"scene": scene, "rays": rays, "tick_ms": round(med, 4), "expected_longest_walk": round(emax, 1), "us_per_bounce_of_the_longest": round(1e3 * (med - 0.045) / emax, 2)}), flush=True)
    ctx.close()
