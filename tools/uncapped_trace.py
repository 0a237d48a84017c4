#!/usr/bin/env python3
"""Waited-for uncapped frames (depth = 0, ARTS.cpp:294) at the headline size for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/uncapped_trace.py [scene [rays]]; then tools/tick_trace_summary.py DIR"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
if os.environ.get("FS_LIB_PATH"):   # (an experimental build: tools/build_variant.sh)
    pkg._capi.LIB_PATH = os.environ["FS_LIB_PATH"]
    pkg._capi._lib = None
scene = sys.argv[1] if len(sys.argv) > 1 else "old_mine"
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
bands = 8 if scene == "old_mine" else 4
sc = getattr(pkg.scenes, scene)(bands)
c = pkg.Context(num_bands=bands)
c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
c.set_listener(sc.listener)
s = c.create_source(sc.source)
p = pkg.default_params(num_rays=rays, depth=0)


def run(n, seed0):
    for i in range(n):
        p.seed = seed0 + i
        c.compute_energy_response_async(s, p)
        c.reconstruct_impulse_response_async(s, p)
        c.synchronize()


run(10, 10)
t = time.perf_counter()
run(40, 100)
print("ms per frame", 1e3 * (time.perf_counter() - t) / 40, file=sys.stderr)
c.close()
