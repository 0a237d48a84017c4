#!/usr/bin/env python3
"""Generates tests/golden/material_fd.npz: inputs and expected outputs of
UMaterialAcousticProcessor::ApplyMaterialFD (MaterialAcousticProcessor.cpp:8-107) computed HERE by the
restatement linked against the reference's own KissFFT (oracle/_ref, needs /root/reference):

    python tests/golden/make_golden_material_fd.py

Data only: a 1500-sample block (N = 2048, 1025 bins), three response curves, three output blocks.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402


def main():
    rng = np.random.default_rng(0xF4)
    L, bins = 1500, 1025
    t = np.arange(L) / 48000.0
    x = (0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.1 * rng.standard_normal(L)).astype(np.float32)
    f = np.linspace(0.0, 1.0, bins)
    absorption = (0.05 + 0.6 * f ** 0.5).astype(np.float32)              # more absorption at high frequency
    transmission = (0.5 * np.exp(-4.0 * f)).astype(np.float32)             # > 1 - Refl at some bins: clamp path taken
    scattering = (0.1 + 0.8 * f).astype(np.float32)
    spec, diff, trans = oracle.apply_material_fd(x, absorption, transmission, scattering)
    np.savez_compressed(os.path.join(HERE, "material_fd.npz"), in_buffer=x, absorption=absorption,
                        transmission=transmission, scattering=scattering, specular=spec, diffuse=diff,
                        transmitted=trans)
    refl = 1.0 - absorption
    print("clamped bins:", int(((refl + transmission) > 1.0).sum()), "of", bins)


if __name__ == "__main__":
    main()
