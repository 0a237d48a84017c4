#!/usr/bin/env python3
"""Offline audition, the loop the reference runs around its text files (FrequenSeeAudioComponent.cpp:300-305,
507-539: write saved_ir.txt, launch an external script, read output_audio.txt — the script itself is not in the
reference tree).  Convolves a mono signal with an impulse response through the library's reverb callback
(fs_reverb_process, 1024-frame blocks, the path the audio thread uses) and writes the result in the same
one-float-per-line format.

    python tools/convolve_ir.py saved_ir.txt input_audio.txt output_audio.txt

Both inputs are one float per line (SaveArrayToFile / LoadFloatArray); the IR is padded or cut to the
component's 48 000 samples.  Product path only (no oracle); needs an MI355X.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft  # noqa: E402

FRAME = 1024


def convolve(pkg, ir, audio):
    sub = pkg.AudioRayTracingSubsystem(num_bands=1)
    comp = pkg.FrequenSeeAudioComponent((0.0, 0.0, 0.0))
    comp.OnRegister(sub)
    n = comp.NumSamples
    fixed = np.zeros(n, np.float32)
    fixed[: min(n, ir.size)] = ir[:n]
    comp.SetImpulseResponse(fixed)
    plug = pkg.FrequenSeeAudioReverbPlugin(sub)
    plug.Initialize(BufferLength=FRAME)
    plug.OnInitSource(comp)
    total = audio.size + n - 1                                   # the full tail of the last sample
    blocks = (total + FRAME - 1) // FRAME
    padded = np.zeros(blocks * FRAME, np.float32)
    padded[: audio.size] = audio
    out = np.empty(blocks * FRAME, np.float32)
    inter = np.empty(2 * FRAME, np.float32)
    for b in range(blocks):
        chunk = padded[b * FRAME:(b + 1) * FRAME]
        inter[0::2] = chunk
        inter[1::2] = chunk
        out[b * FRAME:(b + 1) * FRAME] = plug.ProcessSourceAudio(comp, inter)[0::2]
    sub.Deinitialize()
    return out[:total]


def main(argv):
    if len(argv) != 4:
        print(__doc__)
        return 2
    pkg = graft.load_package()
    ir = pkg._capi.load_float_array(argv[1])
    audio = pkg._capi.load_float_array(argv[2])
    out = convolve(pkg, ir, audio)
    pkg._capi.save_array_to_file(out, argv[3])
    print(f"{argv[3]}: {out.size} samples, peak {np.abs(out).max():.6f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
