"""Row f4: UMaterialAcousticProcessor::ApplyMaterialFD (MaterialAcousticProcessor.cpp:8-107).
Oracle = the restatement linked against the REFERENCE'S OWN KissFFT (oracle/_ref); second opinion = float64
numpy; golden = tests/golden/material_fd.npz (made by tests/golden/make_golden_material_fd.py).
Tolerance: two fp32 FFT implementations of length N differ by O(eps * log2 N) relative to the block's peak:
2e-6 relative RMS, 1e-5 of the peak per sample."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "material_fd.npz")


def curves(rng, bins):
    return [rng.uniform(0.0, 1.0, bins).astype(np.float32) for _ in range(3)]


def nbins(L):
    n = 1
    while n < L:
        n <<= 1
    return n // 2 + 1


def close(got, want, peak):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    rms = np.sqrt(np.mean((got - want) ** 2))
    ref = max(np.sqrt(np.mean(want ** 2)), 1e-3 * peak, 1e-30)
    return rms <= 2e-6 * ref + 1e-12 and np.abs(got - want).max() <= 1e-5 * max(peak, 1e-30)


def have_ref(oracle_mod):
    return os.path.exists(oracle_mod._REF_SO) or os.path.isdir(oracle_mod._KISS_DIR)


def test_oracle_matches_float64_and_golden(oracle_mod):
    if not have_ref(oracle_mod):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    g = np.load(GOLD)
    got = oracle_mod.apply_material_fd(g["in_buffer"], g["absorption"], g["transmission"], g["scattering"])
    for name, v in zip(("specular", "diffuse", "transmitted"), got):
        assert np.array_equal(v, g[name]), name                      # the fixture pins the oracle bit for bit
    rng = np.random.default_rng(3)
    for L in (1, 2, 3, 17, 1024, 1500, 4097):
        x = rng.standard_normal(L).astype(np.float32)
        a, t, s = curves(rng, nbins(L))
        r = oracle_mod.apply_material_fd(x, a, t, s)
        q = oracle_mod.apply_material_fd_numpy(x, a, t, s)
        for i in range(3):
            assert close(r[i], q[i], np.abs(x).max()), (L, i)
    # MAP.cpp:20-26: wrong curve length -> error, empty outputs
    assert oracle_mod.apply_material_fd(np.zeros(8, np.float32), *curves(rng, 4)) is None


def test_golden_properties():
    """what the fixture must satisfy whatever produced it: the three gains add up to Refl + tau', so
    specular + diffuse + transmitted is the block filtered by (1 - alpha + tau')"""
    g = np.load(GOLD)
    x = g["in_buffer"].astype(np.float64)
    refl = 1.0 - g["absorption"].astype(np.float64)
    tau = np.minimum(g["transmission"].astype(np.float64), 1.0 - refl)
    want = np.fft.irfft(np.fft.rfft(x, 2048) * (refl + tau), 2048)[:1500]
    got = g["specular"].astype(np.float64) + g["diffuse"] + g["transmitted"]
    assert np.abs(got - want).max() < 2e-6


@pytest.mark.gpu
def test_gpu_apply_material_fd(pkg, oracle_mod):
    sub = pkg.AudioRayTracingSubsystem(num_bands=1)
    proc = pkg.MaterialAcousticProcessor(sub)
    g = np.load(GOLD)
    out = proc.ApplyMaterialFD(g["in_buffer"], (g["absorption"], g["transmission"], g["scattering"]))
    peak = np.abs(g["in_buffer"]).max()
    for name, key in (("Specular", "specular"), ("Diffuse", "diffuse"), ("Transmitted", "transmitted")):
        assert close(out[name], g[key], peak), name
    rng = np.random.default_rng(4)
    # block sizes on both sides of the LDS-chunk boundary (2048), non powers of two, the plugin's sizes
    for L in (1, 2, 3, 5, 64, 1000, 1024, 1500, 2048, 2049, 4096, 10000, 48000, 49023, 65536, 100000):
        x = rng.standard_normal(L).astype(np.float32)
        a, t, s = curves(rng, nbins(L))
        got = sub.ctx.apply_material_fd(x, a, t, s)
        want = oracle_mod.apply_material_fd(x, a, t, s) if have_ref(oracle_mod) else \
            oracle_mod.apply_material_fd_numpy(x, a, t, s)
        for i in range(3):
            assert close(got[i], want[i], np.abs(x).max()), (L, i)
    # error behaviour (MAP.cpp:20-26): wrong curve length
    with pytest.raises(pkg.FrequenSeeError) as e:
        sub.ctx.apply_material_fd(np.zeros(8, np.float32), *curves(rng, 4))
    assert e.value.code == pkg._capi.ERR_SIZE_MISMATCH
    # empty block: N = 1, one bin
    got = sub.ctx.apply_material_fd(np.zeros(0, np.float32), *curves(rng, 1))
    assert all(v.size == 0 for v in got)
    sub.Deinitialize()


@pytest.mark.gpu
def test_gpu_material_fd_properties_full_size(pkg):
    """size-independent properties at 2^20 samples: identity for a fully reflective, fully specular
    surface; linearity; the clamp keeps Refl + tau <= 1 (total output never exceeds the input's energy)."""
    sub = pkg.AudioRayTracingSubsystem(num_bands=1)
    rng = np.random.default_rng(5)
    L = 1 << 20
    bins = L // 2 + 1
    x = rng.standard_normal(L).astype(np.float32)
    zero, one = np.zeros(bins, np.float32), np.ones(bins, np.float32)
    spec, diff, trans = sub.ctx.apply_material_fd(x, zero, zero, zero)      # alpha = 0, sigma = 0, tau = 0
    assert np.abs(spec - x).max() < 2e-5 and not diff.any() and not trans.any()
    a, t, s = curves(rng, bins)
    y = rng.standard_normal(L).astype(np.float32)
    fx = sub.ctx.apply_material_fd(x, a, t, s)
    fy = sub.ctx.apply_material_fd(y, a, t, s)
    fxy = sub.ctx.apply_material_fd(x + np.float32(2.0) * y, a, t, s)
    for i in range(3):
        assert np.abs(fxy[i] - (fx[i] + np.float32(2.0) * fy[i])).max() < 1e-4
    _, _, tr = sub.ctx.apply_material_fd(x, zero, np.full(bins, 5.0, np.float32), zero)   # tau clamped to 1 - Refl = 0
    assert np.abs(tr).max() < 1e-6
    total = sum(v.astype(np.float64) for v in fx)
    assert np.sum(total ** 2) <= np.sum(x.astype(np.float64) ** 2) * (1 + 1e-5)
    sub.Deinitialize()
