"""Does the rate of a pipelined stream of frames depend on the streams the process created BEFORE the context?  (It did: profiles/r03_stream_queues.log.)
usage (GPU box): PROBE_K=<earlier streams> [PROBE_RECON=0] [FS_TAIL_STREAM_PRIORITY=0|1|2] [FS_FUSED_RECON=0] python tools/stream_history_probe.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as graft
pkg = graft.load_package()
hip = C.CDLL("libamdhip64.so")
sc = pkg.scenes.old_mine(8)
K = int(os.environ.get("PROBE_K", "0"))
RECON = os.environ.get("PROBE_RECON", "1") == "1"
dummies = []
for i in range(K):
    h = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(h), 1) == 0; dummies.append(h)
for fpl in (2, 4):
    c = pkg.Context(num_bands=8)
    c.set_scene(sc.triangles, sc.material_ids, sc.absorption); c.set_listener(sc.listener)
    s = c.create_source(sc.source); c.set_pipelining(2); c.set_frames_per_launch(fpl)
    p = pkg.default_params(num_rays=262144, depth=8)
    def run(k, seed0):
        for i in range(k):
            p.seed = seed0 + i
            c.compute_energy_response_async(s, p)
            if RECON: c.reconstruct_impulse_response_async(s, p)
        c.submit(); c.synchronize()
    run(48, 10)
    t = time.perf_counter(); run(240, 100); dt = (time.perf_counter() - t) / 240
    print(json.dumps({"K": K, "recon": RECON, "fused": os.environ.get("FS_FUSED_RECON", "1"), "fpl": fpl, "Mrays_per_s": round(262144 / dt / 1e6, 1)}), flush=True)
    c.close()
