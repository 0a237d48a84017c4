#!/usr/bin/env python3
"""Row f4 timing: ApplyMaterialFD on one block through fs_apply_material_fd (HIP radix-2 FFT pair; the time
includes the host<->device copies of the block, the three curves and the three outputs, and the sync) against
the reference's own KissFFT path (oracle/_ref) on one host core.
usage: python tests/measure_material_fd.py   (lives under tests/: it loads the oracle)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

pkg = graft.load_package()
sub = pkg.AudioRayTracingSubsystem(num_bands=1)
rng = np.random.default_rng(0)
rows = []
for L in (1024, 48000, 1 << 20):
    n = 1
    while n < L:
        n <<= 1
    bins = n // 2 + 1
    x = rng.standard_normal(L).astype(np.float32)
    a, t, s = [rng.uniform(0, 1, bins).astype(np.float32) for _ in range(3)]
    for _ in range(5):
        sub.ctx.apply_material_fd(x, a, t, s)
    reps = 200 if L < 100000 else 30
    t0 = time.perf_counter()
    for _ in range(reps):
        sub.ctx.apply_material_fd(x, a, t, s)
    gpu_ms = 1e3 * (time.perf_counter() - t0) / reps
    oracle.apply_material_fd(x, a, t, s)
    creps = 20 if L < 100000 else 3
    t0 = time.perf_counter()
    for _ in range(creps):
        oracle.apply_material_fd(x, a, t, s)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / creps
    rows.append({"block": L, "fft": n, "gpu_ms": gpu_ms, "reference_kissfft_ms_1core": cpu_ms})
print(json.dumps({"apply_material_fd": rows}))
