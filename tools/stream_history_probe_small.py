"""Small frames (cfg2) against the streams the process created before the context: see profiles/r03_stream_queues.log.
usage (GPU box): PROBE_K=<earlier streams> [FS_TAIL_STREAM_PRIORITY=0] python tools/stream_history_probe_small.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
hip = C.CDLL("libamdhip64.so")
sc = pkg.scenes.starter_room(4)
K = int(os.environ.get("PROBE_K", "0"))
dummies = []
for i in range(K):
    h = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0; dummies.append(h)
res = {"K": K}
for mode in ("unpipelined", "fpl1", "fpl2"):
    c = pkg.Context(num_bands=4)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
    s = c.create_source(sc.source)
    if mode != "unpipelined":
        c.set_pipelining(2); c.set_frames_per_launch(int(mode[3:]))
    p = pkg.default_params(num_rays=16384, depth=8)
    def run(k, seed0):
        for i in range(k):
            p.seed = seed0 + i
            c.compute_energy_response_async(s, p); c.reconstruct_impulse_response_async(s, p)
            if mode == "unpipelined": c.synchronize()
        c.submit(); c.synchronize()
    run(100, 10)
    t = time.perf_counter(); run(400, 1000); dt = (time.perf_counter() - t) / 400
    res[mode] = round(1e3 * dt, 4)
    c.close()
print(json.dumps(res), flush=True)
