#!/usr/bin/env python3
"""Reference ticks (1000 pairs per source, uncapped walks, one band) for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/tick_trace.py [scene [sources]]; then tools/tick_trace_summary.py DIR"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
pkg = graft.load_package()
if os.environ.get("FS_LIB_PATH"):   # (an experimental build: tools/build_variant.sh)
    pkg._capi.LIB_PATH = os.environ["FS_LIB_PATH"]
    pkg._capi._lib = None
name = sys.argv[1] if len(sys.argv) > 1 else "starter_room"
sc = pkg.scenes.by_name(name, 1)
ctx = pkg.Context(num_bands=1)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(9)
lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
srcs = [ctx.create_source(sc.source if S == 1 else (np.asarray(sc.source, np.float32) + rng.uniform(-0.03, 0.03, 3).astype(np.float32) * (hi - lo)).astype(np.float32))
        for _ in range(S)]
p = pkg.default_params(num_rays=2000, depth=0, seed=1, flags=pkg._capi.FLAG_FIXED_NORM_1000)
tt = []
for i in range(60):
    p.seed = 100 + i
    t1 = time.perf_counter()
    ctx.update_sources(srcs, p)
    tt.append(time.perf_counter() - t1)
tt = sorted(tt[10:])
print("median tick ms", 1e3 * tt[len(tt) // 2], file=sys.stderr)
ctx.close()
