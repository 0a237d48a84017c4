import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
hip = C.CDLL("libamdhip64.so")
def free_mem():
    a, b = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(a), C.byref(b))
    return a.value
sc = pkg.scenes.starter_room(4)
vals = []
for it in range(60):
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    srcs = [ctx.create_source(sc.source) for _ in range(4)]
    ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
    for i in range(12):
        p = pkg.default_params(num_rays=4096, depth=8 if i % 3 else 0, seed=i + 1)
        ctx.compute_energy_response_async(srcs[i % 4], p); ctx.reconstruct_impulse_response_async(srcs[i % 4], p)
    ctx.update_sources(srcs, pkg.default_params(num_rays=2000, depth=0, seed=7))
    ctx.destroy_source(srcs[1]); srcs[1] = ctx.create_source(sc.source)
    ctx.compute_energy_response(srcs[1], pkg.default_params(num_rays=4096, depth=8, seed=3))
    ctx.reconstruct_impulse_response(srcs[1])
    ctx.close()
    if it % 10 == 9:
        vals.append(free_mem())
print("free device memory every 10 contexts (MB):", [round(v / 2**20, 1) for v in vals])
assert vals[0] - vals[-1] < 64 * 2**20, "device memory leaks"
print("no leak")
