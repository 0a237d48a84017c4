#!/usr/bin/env python3
"""Row f2 timing: one audio callback (1024 stereo frames, 48 000-tap IR) through fs_reverb_process (HIP direct
convolution, includes the 8 KB H2D/D2H round trip and the stream sync the audio thread needs) against the
reference's own KissFFT path (oracle/_ref) on one host core.  usage: python tests/measure_reverb.py   (lives under tests/: it loads the oracle)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402
import oracle  # noqa: E402

pkg = graft.load_package()
sc = pkg.scenes.starter_room(4)
sub = pkg.AudioRayTracingSubsystem(num_bands=4)
sub.RegisterGeometry(sc.triangles, sc.material_ids)
sub.SetMaterials(sc.absorption)
comp = pkg.FrequenSeeAudioComponent(sc.source)
comp.OnRegister(sub)
sub.SetListenerLocation(sc.listener)
sub.UpdateSource(comp, pkg.default_params(num_rays=16384, depth=8, dist_divisor=100.0))
ir = comp.GetImpulseResponse()[0].copy()
plug = pkg.FrequenSeeAudioReverbPlugin(sub)
plug.OnInitSource(comp)
rng = np.random.default_rng(0)
blk = np.clip(rng.normal(0, 0.3, 2048), -1, 1).astype(np.float32)
for _ in range(20):
    plug.ProcessSourceAudio(comp, blk)
n = 300
t = time.perf_counter()
for _ in range(n):
    plug.ProcessSourceAudio(comp, blk)
gpu_ms = 1e3 * (time.perf_counter() - t) / n
ref = oracle.ReverbRef()
for _ in range(3):
    ref.process(ir, ir, blk)
t = time.perf_counter()
for _ in range(30):
    ref.process(ir, ir, blk)
cpu_ms = 1e3 * (time.perf_counter() - t) / 30
print(json.dumps({"callback": "1024 stereo frames, 48000-tap IR", "gpu_ms_per_callback": gpu_ms,
                  "reference_kissfft_ms_per_callback_1core": cpu_ms, "realtime_budget_ms": 1024 / 48.0}))
