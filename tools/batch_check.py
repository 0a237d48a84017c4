#!/usr/bin/env python3
"""Eight sources x 131 072 rays on one context: separate frames vs one batched frame (fs_compute_energy_response_batch_async),
with and without the per-source tail (reconstruct + publish), and the host time spent enqueueing.  BC_TORCH=1 imports
torch first: the process then runs on torch's bundled HIP runtime, whose API calls cost several times more host time
(the ~70 calls of eight per-source tails become the bottleneck).  BC_PROF=n sets the profiling level."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
if os.environ.get("BC_TORCH"):
    import torch  # noqa: F401
    torch.cuda.init()
pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
ctx = pkg.Context(num_bands=8)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
srcs = [ctx.create_source(pos) for pos in sc.extra_sources[:8]]
if os.environ.get("BC_PROF"):
    ctx.set_profiling(int(os.environ["BC_PROF"]))
p = pkg.default_params(num_rays=131072, depth=8)
rows = {}
for mode in ("batch", "batch+tail", "separate+tail"):
    for rep in range(2):
        n = 30
        ctx.synchronize()
        t = time.perf_counter()
        for i in range(n):
            p.seed = 100 + i
            if mode.startswith("batch"):
                ctx.compute_energy_response_batch_async(srcs, p)
            else:
                for s in srcs:
                    ctx.compute_energy_response_async(s, p)
            if mode.endswith("tail"):
                for s in srcs:
                    ctx.reconstruct_impulse_response_async(s, p)
        rows[mode + " (host enqueue)"] = 1e3 * (time.perf_counter() - t) / n
        ctx.synchronize()
        rows[mode] = 1e3 * (time.perf_counter() - t) / n
print(json.dumps({"ms_per_step_8_sources": rows, "rays_per_step": 8 * 131072}))
