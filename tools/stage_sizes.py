import json, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
sc = pkg.scenes.old_mine(8)
for rays in (16384, 65536, 131072, 524288, 1048576):
    row = {"rays": rays}
    for name, depth in (("unpipelined", 0), ("staged", 2)):
        c = pkg.Context(num_bands=8)
        c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
        s = c.create_source(sc.source); c.set_pipelining(depth)
        p = pkg.default_params(num_rays=rays, depth=0)
        def run(n, seed0):
            for i in range(n):
                p.seed = seed0 + i
                c.compute_energy_response_async(s, p); c.reconstruct_impulse_response_async(s, p)
            c.synchronize()
        frames = max(20, min(150, int(40e6 / rays)))
        run(20, 10); t = time.perf_counter(); run(frames, 100); dt = (time.perf_counter() - t) / frames
        row[name + "_ms"] = round(1e3 * dt, 4); row[name + "_Mrays"] = round(rays / dt / 1e6, 1)
        c.close()
    print(json.dumps(row), flush=True)
