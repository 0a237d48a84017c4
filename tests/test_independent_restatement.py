"""An INDEPENDENT float64 restatement of the three closed-form pieces of the path — written from the reference's
lines, not from oracle/fs_oracle.c — checked against the C oracle on thousands of random inputs:

  EvaluatePath                 Private/AudioRayTracingSubsystem.cpp:358-420
  AddEnergyAtDelay (bin rule)  Public/FrequenSeeAudioComponent.h:87-91
  ReconstructImpulseResponse   Private/FrequenSeeAudioComponent.cpp:320-380 (+ NormalizeImpulseResponse :382-406)

The reference ships no golden vectors for this path ("parity unpinned"): what pins the oracle is hand-derived KATs
(tests/test_oracle_kat.py) and this second, differently written implementation — two restatements that agree on
random data are much less likely to share a misreading than one is to contain it.  The oracle computes in fp32 like
the reference, this file in float64: agreement is asserted to fp32 accuracy, bins exactly (away from bin edges).
"""
import math

import numpy as np
import pytest

SOUND_SPEED = 343.0        # ARTS.cpp:362
AIR = 0.05                 # ARTS.cpp:395
RAY_COUNT_GAIN = 10.0      # ARTS.cpp:413
NUM_BINS = 1000            # FSAC.h:137
SAMPLE_RATE = 48000        # FSAC.h:133


def evaluate_path_f64(positions, reflectivity, has_material, probability):
    """ARTS.cpp:358-420 for one path: positions [n][3] (cm), per node: Absorption[2] value, whether the node has a
    geometry component with a material, and its Probability.  Returns (DelaySeconds, Gain)."""
    energy, scaled = 1.0, 0.0
    for i in range(len(positions) - 1):                                     # :368
        node_distance = math.dist(positions[i], positions[i + 1]) / 1000.0  # :373
        scaled += node_distance                                             # :374
        if node_distance < 1.0:                                             # :375-378
            continue
        bsdf = reflectivity[i] / math.pi if has_material[i] else 1.0        # :382-386
        geometry = 1.0 / (4.0 * math.pi * node_distance * node_distance)    # :391
        energy *= bsdf                                                      # :392
        energy *= geometry                                                  # :393
        energy *= math.exp(-AIR * node_distance)                            # :395-397
        energy /= probability[i] ** 0.1                                     # :398
    energy = min(energy, 1.0)                                               # :410
    energy *= RAY_COUNT_GAIN                                                # :413
    return scaled / SOUND_SPEED, energy                                     # :419


def bin_of(delay_seconds, bin_size_ms=1, num_bins=NUM_BINS):
    """FSAC.h:89: FMath::Clamp(FMath::FloorToInt((DelaySeconds * 1000.f) / BinSizeMs), 0, EnergyBuffer.Num() - 1)"""
    return int(min(max(math.floor(delay_seconds * 1000.0 / bin_size_ms), 0), num_bins - 1))


def reconstruct_f64(energy, num_samples=SAMPLE_RATE, samples_per_bin=49):
    """FSAC.cpp:320-380.  NumSamplesPerBin = CeilToInt(0.001f * 48000) = 49 in float32 arithmetic (0.001f * 48000 =
    48.000004f); EnergyResponse and EnergyNorms alias the same buffer (:325, :329); NormalizeImpulseResponse zeroes the
    UNFILTERED array, which is then replaced by the filtered one (:377-378) — the result is the filtered signal."""
    energy = np.asarray(energy, dtype=np.float64)
    pi4 = math.sqrt(4.0 * math.pi)                                          # :323
    amp = np.zeros_like(energy)
    ok = np.abs(energy) >= 1e-6                                             # :343
    amp[ok] = energy[ok] / np.sqrt(energy[ok] * pi4)                        # :345
    ir = np.zeros(num_samples)
    for b in range(len(energy)):
        prev = amp[b] if b == 0 else amp[b - 1]                             # :347-355
        n = min(samples_per_bin, num_samples - b * samples_per_bin)         # :340
        for k in range(max(n, 0)):
            w = k / samples_per_bin                                         # :359
            ir[b * samples_per_bin + k] = (1.0 - w) * prev + w * amp[b]     # :360-362
    out = np.empty_like(ir)
    out[0] = ir[0]                                                          # :371
    for i in range(1, num_samples):
        out[i] = 0.25 * ir[i] + 0.75 * out[i - 1]                           # :372-375
    return out


def random_path(rng, oracle_mod):
    n = int(rng.integers(2, 19))
    scale = float(rng.choice([300.0, 1500.0, 4000.0, 12000.0]))            # short hops (skipped) and long ones
    pos = np.cumsum(rng.normal(size=(n, 3)) * scale, axis=0).astype(np.float32)
    has = rng.random(n) < 0.8
    mat = np.where(has, rng.integers(0, 4, size=n), oracle_mod.NO_MATERIAL)
    prob = np.where(rng.random(n) < 0.1, 1.0, rng.uniform(1e-3, 0.3, size=n)).astype(np.float32)
    nodes = [oracle_mod.make_node(pos[i], material=int(mat[i]), prob=float(prob[i])) for i in range(n)]
    return nodes, pos.astype(np.float64), has, mat, prob.astype(np.float64)


def test_evaluate_path_and_bin_rule_against_float64(oracle_mod):
    rng = np.random.default_rng(20250101)
    refl = rng.uniform(0.05, 0.9, size=(4, 1)).astype(np.float32)           # 4 materials, 1 band (slot 0 == Absorption[2])
    tri = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], np.float32)         # EvaluatePath never touches the geometry
    osc = oracle_mod.Scene(tri, np.zeros(1, np.uint16), refl)
    p = oracle_mod.default_params(num_pairs=1, depth=0)
    checked = edge = clamped = 0
    for _ in range(4000):
        nodes, pos, has, mat, prob = random_path(rng, oracle_mod)
        gains, delay = osc.evaluate_path(p, nodes)
        r = [float(refl[int(m), 0]) if h else 0.0 for m, h in zip(mat, has)]
        want_delay, want_gain = evaluate_path_f64(pos, r, has, prob)
        assert delay == pytest.approx(want_delay, rel=2e-6, abs=1e-12)
        assert float(gains[0]) == pytest.approx(want_gain, rel=3e-5, abs=1e-30)
        clamped += want_gain == 10.0
        x = want_delay * 1000.0
        if abs(x - round(x)) < 1e-5 * max(abs(x), 1.0):
            edge += 1                                                       # a bin edge within fp32 noise: either side is right
            continue
        buf = np.zeros(NUM_BINS, np.float32)
        assert oracle_mod.add_energy_at_delay(buf, delay, 1.0) == bin_of(want_delay)
        checked += 1
    assert checked > 3900 and edge < 100 and clamped > 50                                  # both branches of the clamp were exercised
    # the clamp ends of the bin rule (FSAC.h:89)
    buf = np.zeros(NUM_BINS, np.float32)
    for d, b in ((-0.5, 0), (0.0, 0), (0.0004, 0), (0.9995, 999), (1.7, 999), (0.0145772595, 14)):
        assert oracle_mod.add_energy_at_delay(buf, d, 1.0) == b == bin_of(d)


def test_reconstruct_against_float64(oracle_mod):
    rng = np.random.default_rng(7)
    for trial in range(12):
        e = np.zeros(NUM_BINS, np.float32)
        k = int(rng.integers(1, 400))
        idx = rng.integers(0, NUM_BINS, size=k)
        e[idx] = (10.0 ** rng.uniform(-8, 0.5, size=k)).astype(np.float32)  # straddles the 1e-6 threshold
        got = oracle_mod.reconstruct(e)
        want = reconstruct_f64(e)
        peak = np.abs(want).max()
        assert peak > 0
        assert np.abs(got - want).max() <= 2e-6 * peak
        assert np.array_equal(got[980 * 49:] != 0, want[980 * 49:] != 0)    # bins >= 980 write nothing, the filter tail decays
    assert reconstruct_f64(np.zeros(NUM_BINS)).max() == 0.0
