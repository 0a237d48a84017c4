"""Pins the CPU oracle (oracle/fs_oracle.c).

The reference ships no tests or golden vectors for this path ("parity unpinned", oracle/fs_oracle.h),
so the oracle is pinned by (1) the closed-form known-answer tests of SURVEY.md A.7, each derivable by
hand from the cited reference lines, (2) published Philox4x32-10 known answers (Random123 kat_vectors),
(3) self-consistency: BVH == brute force, shard-invariance, determinism.
"""
import ctypes as C
import math

import numpy as np
import pytest


# ---- RNG -------------------------------------------------------------------------------------------
PHILOX_KAT = [  # Random123 kat_vectors: philox4x32 10 <ctr x4> <key x2> -> <out x4>
    ((0x00000000,) * 4, (0x00000000,) * 2, (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


@pytest.mark.parametrize("ctr,key,out", PHILOX_KAT)
def test_philox_known_answers(oracle_mod, ctr, key, out):
    lib = oracle_mod.load()
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib.fso_philox4x32_10(c, k, o)
    assert tuple(o) == out


def test_u01_range(oracle_mod):
    lib = oracle_mod.load()
    assert lib.fso_u01(0) == 0.0
    assert lib.fso_u01(0xFFFFFFFF) == pytest.approx(1.0 - 2.0 ** -24, abs=0)
    assert lib.fso_u01(0x80000000) == 0.5


def test_sincos2pi_accuracy(oracle_mod):
    lib = oracle_mod.load()
    s, c = C.c_float(), C.c_float()
    worst = 0.0
    for u in np.linspace(0.0, 1.0, 4097, endpoint=False):
        lib.fso_sincos2pi(float(np.float32(u)), C.byref(s), C.byref(c))
        a = 2.0 * math.pi * float(np.float32(u))
        worst = max(worst, abs(s.value - math.sin(a)), abs(c.value - math.cos(a)))
    assert worst < 5e-7


def test_sampling_maps(oracle_mod):
    lib = oracle_mod.load()
    rng = np.random.default_rng(1)
    n = np.array([0.3, -0.5, 0.81], dtype=np.float32)
    n /= np.linalg.norm(n)
    nn = (C.c_float * 3)(*n)
    d = (C.c_float * 3)()
    cos_ref, cos_cos = [], []
    for _ in range(20000):
        U, V = float(np.float32(rng.random())), float(np.float32(rng.random()))
        lib.fso_sample_cone(nn, U, V, 0, d)
        v = np.array(list(d))
        assert abs(np.linalg.norm(v) - 1.0) < 1e-5
        cos_ref.append(float(v @ n))
        lib.fso_sample_cone(nn, U, V, 1, d)
        cos_cos.append(float(np.array(list(d)) @ n))
    cos_ref, cos_cos = np.array(cos_ref), np.array(cos_cos)
    assert cos_ref.min() > -1e-6  # hemisphere about n (ARTS.cpp:313: half angle 90 deg)
    # VRandCone(n, 90deg): polar density (sin+cos)/2 -> E[cos] = 1/4 + pi/8
    assert abs(cos_ref.mean() - (0.25 + math.pi / 8)) < 0.01
    assert abs(cos_cos.mean() - 2.0 / 3.0) < 0.01  # cosine-weighted alternative
    # sphere: unit vectors, zero mean
    acc = np.zeros(3)
    for i in range(5000):
        r = (C.c_uint32 * 4)(*[int(x) for x in rng.integers(0, 2 ** 32, 4, dtype=np.uint64)])
        lib.fso_sample_sphere(0x5EED, i, 0, 0, r, d)
        v = np.array(list(d))
        assert abs(np.linalg.norm(v) - 1.0) < 1e-5
        acc += v
    assert np.abs(acc / 5000).max() < 0.05


# ---- A.7 known answers ------------------------------------------------------------------------------
def _node(oracle_mod, pos, prob, material=0xFFFF):
    n = oracle_mod.Node()
    n.pos[:] = pos
    n.normal[:] = (0, 0, 0)
    n.material = material
    n.prob = prob
    return n


def test_kat_evaluate_path(oracle_mod, scene_factory):
    """A.7-1, by hand from ARTS.cpp:360-420."""
    sc = scene_factory("shoebox", 1)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, np.array([[0.5]], np.float32))
    p = oracle_mod.default_params()
    p0 = 0.9 / (4 * math.pi)
    nodes = [_node(oracle_mod, (0, 0, 0), 1.0), _node(oracle_mod, (2000, 0, 0), p0, material=0),
             _node(oracle_mod, (2000, 3000, 0), 0.5)]
    gains, delay = s.evaluate_path(p, nodes)
    e0 = 1 / (16 * math.pi) * math.exp(-0.1)
    assert e0 == pytest.approx(0.0180011685, rel=1e-7)
    e1 = e0 * (0.5 / math.pi) / (36 * math.pi) * math.exp(-0.15) / p0 ** 0.1
    assert e1 == pytest.approx(2.83805637e-5, rel=1e-6)
    assert gains[0] == pytest.approx(10 * e1, rel=2e-6)
    assert delay == pytest.approx(5 / 343, rel=1e-6)
    buf = np.zeros(1000, np.float32)
    b = oracle_mod.add_energy_at_delay(buf, delay, gains[0] / 1000.0)
    assert b == 14 and buf[14] == pytest.approx(2.83805637e-7, rel=2e-6)


def test_kat_skip_rule(oracle_mod, scene_factory):
    """A.7-2: all segments < 1000 cm -> E = 1 -> gain exactly 10 (ARTS.cpp:375-378, 410-413)."""
    sc = scene_factory("shoebox", 1)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle_mod.default_params()
    nodes = [_node(oracle_mod, (0, 0, 0), 1.0), _node(oracle_mod, (500, 0, 0), 0.07, 0),
             _node(oracle_mod, (500, 900, 0), 0.2, 0), _node(oracle_mod, (100, 900, 0), 0.1)]
    gains, delay = s.evaluate_path(p, nodes)
    assert gains[0] == 10.0
    assert delay == pytest.approx((0.5 + 0.9 + 0.4) / 343, rel=1e-6)


def test_kat_bin_index(oracle_mod):
    """A.7-3 (FSAC.h:87-91)."""
    for delay, want in ((0.0, 0), (0.9995, 999), (1.7, 999), (-0.3, 0), (0.0145772595, 14), (0.001, 1)):
        buf = np.zeros(1000, np.float32)
        assert oracle_mod.add_energy_at_delay(buf, delay, 1.0) == want
        assert buf[want] == 1.0 and buf.sum() == 1.0


def test_kat_sizes(oracle_mod):
    """A.7-5: NumBins = 1000, NumSamples = 48000, NumSamplesPerBin = 49 (fp32 ceil quirk, FSAC.cpp:324)."""
    lib = oracle_mod.load()
    assert lib.fso_num_bins(1.0, 0.001) == 1000
    assert lib.fso_num_samples(1.0, 48000) == 48000
    assert lib.fso_samples_per_bin(0.001, 48000) == 49
    assert np.float32(0.001) * np.float32(48000) > 48.0


def test_kat_reconstruct_one_hot(oracle_mod):
    """A.7-4 (FSAC.cpp:320-380)."""
    e = np.zeros(1000, np.float32)
    e[10] = 0.04
    ir = oracle_mod.reconstruct(e)
    a10 = 0.04 / math.sqrt(0.04 * math.sqrt(4 * math.pi))
    assert a10 == pytest.approx(0.1062251932, rel=1e-6)
    # unfiltered: ramp up over samples 490..538, ramp down over 539..587, zero elsewhere
    x = np.zeros(48000)
    for s in range(49):
        x[490 + s] = (s / 49.0) * a10
        x[539 + s] = (1 - s / 49.0) * a10
    y = np.zeros(48000)
    y[0] = x[0]
    for i in range(1, 48000):
        y[i] = 0.25 * x[i] + 0.75 * y[i - 1]
    assert np.abs(ir - y).max() < 2e-7
    assert ir[:490].max() == 0.0
    # bins >= 980 write nothing; bin 979 writes 29 samples (A.5)
    e2 = np.zeros(1000, np.float32)
    e2[985] = 1.0
    assert np.all(oracle_mod.reconstruct(e2) == 0.0)
    e3 = np.zeros(1000, np.float32)
    e3[979] = 1.0
    ir3 = oracle_mod.reconstruct(e3)
    assert np.all(ir3[: 979 * 49] == 0.0) and ir3[979 * 49 + 1] > 0.0
    # threshold 1e-6 and samples_per_bin override (compat flag j)
    e4 = np.zeros(1000, np.float32)
    e4[3] = 5e-7
    assert np.all(oracle_mod.reconstruct(e4) == 0.0)
    ir48 = oracle_mod.reconstruct(e, samples_per_bin=48)
    assert ir48[479] == 0.0 and ir48[481] > 0.0


# ---- ray queries: BVH == brute force -----------------------------------------------------------------
@pytest.mark.parametrize("name", ["shoebox", "starter_room", "old_mine"])
def test_bvh_equals_brute_force(oracle_mod, scene_factory, name):
    sc = scene_factory(name)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    rng = np.random.default_rng(7)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    n = 300 if name == "old_mine" else 1500
    hits = 0
    for i in range(n):
        o = sc.source + rng.normal(0, 60, 3) if i % 2 else rng.uniform(lo, hi)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        a = s.trace_closest(o, d, 1e6, brute=False)
        b = s.trace_closest(o, d, 1e6, brute=True)
        assert a[0] == b[0]
        if a[0]:
            hits += 1
            assert a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3])
            assert a[3] @ d <= 0  # normal faces the ray origin side
            tm = a[1] * rng.uniform(0.5, 1.5)
            assert s.trace_any(o, d, tm, brute=False) == s.trace_any(o, d, tm, brute=True) == (tm >= a[1])
    assert hits > n // 3


# ---- GeneratePath / ConnectSubpaths / UpdateSource -----------------------------------------------------
def test_generate_path_structure(oracle_mod, scene_factory):
    sc = scene_factory("starter_room")
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle_mod.default_params(depth=8)
    seg, dup, total = 0, 0, 0
    for pair in range(1500):
        nodes = s.generate_path(p, pair, 0, sc.source)
        assert 1 <= len(nodes) <= 9
        assert nodes[0].prob == 1.0 and nodes[0].material == 0xFFFF
        assert tuple(nodes[0].pos) == tuple(np.float32(sc.source))
        seg += len(nodes) - 1
        if len(nodes) > 1:
            assert nodes[1].prob == pytest.approx(0.9 / (4 * math.pi), rel=1e-6)  # ARTS.cpp:309-310
        for a, b in zip(nodes[:-1], nodes[1:]):
            total += 1
            if tuple(a.pos) == tuple(b.pos):
                dup += 1  # a miss (open door/window): duplicate node, ARTS.cpp:296 + :339 false
            else:
                assert b.material != 0xFFFF and abs(np.linalg.norm(list(b.normal)) - 1) < 1e-5
    # Russian roulette 0.9 with depth cap 8: E[segments] = sum_{k=1..8} 0.9^k = 5.13
    assert abs(seg / 1500 - sum(0.9 ** k for k in range(1, 9))) < 0.25
    assert 0 < dup < total // 4
    # unbounded (reference) walk: mean 9 segments; fixed depth: exactly `depth` segments
    pu = oracle_mod.default_params(depth=0)
    mean = np.mean([len(s.generate_path(pu, i, 1, sc.listener)) - 1 for i in range(1500)])
    assert abs(mean - 9.0) < 0.8
    pf = oracle_mod.default_params(depth=6, russian_roulette=0)
    assert all(len(s.generate_path(pf, i, 0, sc.source)) == 7 for i in range(50))


def test_compute_energy_consistency(oracle_mod, scene_factory):
    sc = scene_factory("starter_room")
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle_mod.default_params(num_pairs=2048, depth=8)
    e32, e64, c = s.compute_energy(p, sc.source, sc.listener)
    assert c.connected == c.deposits and 0 < c.connected < 2048
    assert c.closest_rays + 2 * 2048 == c.path_nodes or c.closest_rays <= c.path_nodes
    # determinism
    e32b, _, _ = s.compute_energy(p, sc.source, sc.listener)
    assert np.array_equal(e32, e32b)
    # brute force == BVH, bit for bit (the closest hit does not depend on the tree)
    pb = oracle_mod.default_params(num_pairs=2048, depth=8, flags=oracle_mod.FLAG_BRUTE_FORCE)
    e32c, _, _ = s.compute_energy(pb, sc.source, sc.listener, 0, 256)
    e32d, _, _ = s.compute_energy(p, sc.source, sc.listener, 0, 256)
    assert np.array_equal(e32c, e32d)
    # shard invariance: ranges sum to the whole (global pair index keys the RNG)
    parts = [s.compute_energy(p, sc.source, sc.listener, a, b)[1] for a, b in ((0, 700), (700, 1500), (1500, 2048))]
    assert np.allclose(sum(parts), e64, rtol=1e-12, atol=0)
    # quirk A.6-c: fixed normaliser 1/1000
    pn = oracle_mod.default_params(num_pairs=2048, depth=8, flags=oracle_mod.FLAG_FIXED_NORM_1000)
    _, e64n, _ = s.compute_energy(pn, sc.source, sc.listener)
    assert np.allclose(e64n, e64 * 2048 / 1000, rtol=1e-6)
    # every deposit is <= gain 10 / P per band
    assert e64.sum(axis=1).max() <= 10.0 * c.connected / 2048 + 1e-9
    # seeds matter
    p2 = oracle_mod.default_params(num_pairs=2048, depth=8, seed=1234)
    assert not np.array_equal(s.compute_energy(p2, sc.source, sc.listener)[0], e32)


def test_multithreaded_baseline_matches(oracle_mod, scene_factory):
    sc = scene_factory("shoebox", 2)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle_mod.default_params(num_pairs=3000, depth=6)
    _, e64, c = s.compute_energy(p, sc.source, sc.listener)
    _, e64m, cm = s.compute_energy_mt(p, sc.source, sc.listener, threads=4)
    assert np.allclose(e64, e64m, rtol=1e-12) and c.as_dict() == cm.as_dict()


# ---- a9: legacy forward tracer (UpdateSound / CastAudioRay / CastDirectAudioRay) ------------------------------
def test_kat_legacy_occlusion_closed_form(oracle_mod, scene_factory):
    """CastDirectAudioRay FSAC.cpp:209-280: the pawn hit returns exp(-0.0017 * metres travelled)."""
    sc = scene_factory("shoebox", 1)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    s.set_objects(sc.object_ids)
    r = s.update_sound(sc.source, sc.listener, listener_radius=34.0)
    d_cm = float(np.linalg.norm(sc.listener.astype(np.float64) - sc.source.astype(np.float64))) - 34.0
    assert r["occlusion_attenuation"] == pytest.approx(math.exp(-0.0017 * d_cm * 0.01), rel=2e-6)
    # a wall (one actor, two faces) between source and listener is passed through once (FSAC.cpp:272-276)
    wall = np.array([[[500, 0, 0], [500, 800, 0], [500, 800, 300]], [[500, 0, 0], [500, 800, 300], [500, 0, 300]],
                     [[510, 0, 0], [510, 800, 0], [510, 800, 300]], [[510, 0, 0], [510, 800, 300], [510, 0, 300]]],
                    dtype=np.float32)
    tri = np.concatenate([sc.triangles, wall])
    mat = np.concatenate([sc.material_ids, np.zeros(4, np.uint16)])
    s2 = oracle_mod.Scene(tri, mat, sc.absorption)
    s2.set_objects(np.concatenate([sc.object_ids, np.full(4, 7, np.uint32)]))
    r2 = s2.update_sound(sc.source, sc.listener, listener_radius=34.0)
    assert r2["occlusion_attenuation"] == pytest.approx(r["occlusion_attenuation"], rel=1e-5)
    # every face its own actor: the wall costs two of the ten pass-throughs, still audible ...
    s2.set_objects(None)
    assert s2.update_sound(sc.source, sc.listener)["occlusion_attenuation"] > 0.9
    # ... but six such walls (12 faces) exhaust them: occluded (FSAC.cpp:212)
    walls = np.concatenate([wall + np.array([20.0 * k, 0, 0], np.float32) for k in range(6)])
    s3 = oracle_mod.Scene(np.concatenate([sc.triangles, walls]),
                          np.concatenate([sc.material_ids, np.zeros(24, np.uint16)]), sc.absorption)
    assert s3.update_sound(sc.source, sc.listener)["occlusion_attenuation"] == 0.0
    # range cull: travel time > SimulatedDuration returns 0 (FSAC.cpp:171-175 / :260-263)
    assert s.update_sound(sc.source, sc.listener, raycast_distance=100.0)["occlusion_attenuation"] == 0.0


def test_legacy_forward_tracer_statistics(oracle_mod, scene_factory):
    sc = scene_factory("starter_room", 4)
    s = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    s.set_objects(sc.object_ids)
    r = s.update_sound(sc.source, sc.listener)
    assert 0 < r["rays_reaching_listener"] < 1500 and r["total_energy"] == pytest.approx(r["rays_reaching_listener"] / 1500)
    assert r["direct_hits"] > 0 and 0 < r["direct_energy_sum"] <= r["direct_hits"]
    assert 1500 <= r["traces"] <= 1500 * 20 + 10
    # the initial directions (VRandCone((0,-1,0), PI, PI)) are uniform on the sphere
    lib = oracle_mod.load()
    d = (C.c_float * 3)()
    acc = np.zeros(3)
    for i in range(4000):
        lib.fso_legacy_direction(0x5EED, i, d)
        v = np.array(list(d))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        acc += v
    assert np.abs(acc / 4000).max() < 0.05


def test_kat_all_connections_open_space(oracle_mod):
    """Row f3 by hand.  Nothing is ever hit (one speck of a triangle far away), Russian roulette off, D = 2:
    every walk step misses, so F0..F2 all sit at the source and B0..B2 at the listener (the duplicate nodes of
    ARTS.cpp:296) and all 9 (i, j) connections are visible.  Path (i, j): i zero-length segments (skipped,
    ARTS.cpp:375-378), the connection of d = 3, j zero-length segments => E = 1/(4 pi d^2) exp(-0.05 d) / P(Fi)^0.1
    with P(F0) = 1, P(Fi>0) = 0.9/(4 pi) (sphere pdf x roulette, ARTS.cpp:306-310); weight 1/N(i+j) with
    N(0..4) = 1, 2, 3, 2, 1; everything lands in bin floor(3/343 s / 1 ms) = 8."""
    far = np.array([[[1e5, 1e5, 1e5], [1e5 + 1e-3, 1e5, 1e5], [1e5, 1e5 + 1e-3, 1e5]]], np.float32)
    s = oracle_mod.Scene(far, np.zeros(1, np.uint16), np.array([[0.5]], np.float32))
    pairs = 4
    p = oracle_mod.default_params(num_pairs=pairs, depth=2, russian_roulette=0, flags=oracle_mod.FLAG_ALL_CONNECTIONS)
    e32, e64, c = s.compute_energy(p, (0, 0, 0), (3000, 0, 0))
    assert c.connected == 9 * pairs and c.deposits == 9 * pairs and c.any_rays == 9 * pairs
    base = 1.0 / (4 * math.pi * 9.0) * math.exp(-0.15)
    gain = [10.0 * min(1.0, base), 10.0 * min(1.0, base / (0.9 / (4 * math.pi)) ** 0.1)]
    weight = {0: 1.0, 1: 0.5, 2: 1.0 / 3.0, 3: 0.5, 4: 1.0}
    want = sum(gain[min(i, 1)] * weight[i + j] for i in range(3) for j in range(3))   # x pairs x norm (1/pairs)
    assert np.count_nonzero(e64) == 1 and e64[0, 8] == pytest.approx(want, rel=2e-6)
    assert e32[0, 8] == pytest.approx(want, rel=1e-5)
    # the end-to-end strategy alone (default mode) is the (2, 2) term with weight 1
    p1 = oracle_mod.default_params(num_pairs=pairs, depth=2, russian_roulette=0)
    _, e64b, c1 = s.compute_energy(p1, (0, 0, 0), (3000, 0, 0))
    assert c1.connected == pairs and e64b[0, 8] == pytest.approx(gain[1], rel=2e-6)
    # weights of one path length sum to 1 for any depth cap
    for D in (1, 2, 5, 8):
        for t in range(2 * D + 1):
            n = sum(1 for i in range(D + 1) for j in range(D + 1) if i + j == t)
            assert n == min(t, D) - max(0, t - D) + 1


# ---- row f3: balance-heuristic weights -------------------------------------------------------------------------
def _mis_by_definition(pos, nrm, s, D):
    """p_s = prod_{k<s} pf_k * prod_{k>s} pb_k written out strategy by strategy in float64 (the oracle evaluates the
    same sum in one pass).  pos[0] = source, pos[-1] = listener, nrm[k] = stored normal of surface vertex k."""
    t = len(pos) - 2
    lo, hi = max(0, t - D), min(t, D)
    seg = []
    for k in range(t + 1):
        d = pos[k + 1] - pos[k]
        l2 = float(d @ d)
        seg.append((d / math.sqrt(l2), l2))

    def pf(k):      # density of y_{k+1} generated from y_k, area measure
        d, l2 = seg[k]
        P = 1 / (4 * math.pi) if k == 0 else max(0.0, float(nrm[k] @ d)) / math.pi
        return P * abs(float(nrm[k + 1] @ d)) / l2

    def pb(k):      # density of y_k generated from y_{k+1}
        d, l2 = seg[k]
        P = 1 / (4 * math.pi) if k == t else max(0.0, -float(nrm[k + 1] @ d)) / math.pi
        return P * abs(float(nrm[k] @ d)) / l2

    def p(sp):
        v = 1.0
        for k in range(sp):
            v *= pf(k)
        for k in range(sp + 1, t + 1):
            v *= pb(k)
        return v

    return p(s) / sum(p(sp) for sp in range(lo, hi + 1))


def test_kat_mis_weight_one_bounce_by_hand(oracle_mod):
    """t = 1 by hand: floor z = -100 (normal +z), source (0,0,0), vertex (50,0,-100), listener (200,0,0).
    Generated from the source: p_1 = 1/(4 pi) cos1 / L0^2 with cos1 = 100/L0, L0^2 = 12500; from the listener:
    p_0 = 1/(4 pi) cos1' / L1^2 with cos1' = 100/L1, L1^2 = 32500.  w_1 = p_1 / (p_0 + p_1) = L1^3 / (L0^3 + L1^3)."""
    mk = oracle_mod.make_node
    nodes = [mk((0, 0, 0)), mk((50, 0, -100), (0, 0, 1)), mk((200, 0, 0))]
    L0, L1 = math.sqrt(12500.0), math.sqrt(32500.0)
    w1 = L1 ** 3 / (L0 ** 3 + L1 ** 3)
    assert oracle_mod.mis_weight(nodes, 1, 8) == pytest.approx(w1, rel=1e-14)
    assert oracle_mod.mis_weight(nodes, 0, 8) == pytest.approx(1 - w1, rel=1e-14)
    # direct path: one strategy, weight 1; depth cap 1 leaves both strategies of t = 1
    assert oracle_mod.mis_weight([mk((0, 0, 0)), mk((200, 0, 0))], 0, 8) == 1.0
    assert oracle_mod.mis_weight(nodes, 1, 1) == pytest.approx(w1, rel=1e-14)


def test_mis_weight_matches_definition_and_sums_to_one(oracle_mod):
    """Random reflection paths inside a unit-ish box: the one-pass evaluation equals the strategy-by-strategy
    definition, the weights of all strategies of one path sum to 1 (also when the depth cap removes some), a
    strategy that would have to leave a surface backwards has weight 0, degenerate segments give the uniform weight."""
    rng = np.random.default_rng(7)
    mk = oracle_mod.make_node
    for trial in range(200):
        t = int(rng.integers(1, 9))
        D = int(rng.integers(max(1, (t + 1) // 2), 9))
        pos = rng.uniform(-500, 500, (t + 2, 3))
        nrm = np.zeros((t + 2, 3))
        for k in range(1, t + 1):
            # a normal with both neighbours on its side (a reflection), as the walk produces
            a = (pos[k - 1] - pos[k]) / np.linalg.norm(pos[k - 1] - pos[k])
            b = (pos[k + 1] - pos[k]) / np.linalg.norm(pos[k + 1] - pos[k])
            n = a + b + rng.normal(0, 0.05, 3)
            if np.linalg.norm(a + b) < 0.2:        # nearly straight through: tilt the normal off the path
                n = np.cross(a, rng.normal(size=3)) + 0.3 * (a + b)
            nrm[k] = n / np.linalg.norm(n)
            if min(nrm[k] @ a, nrm[k] @ b) < 0.02:
                nrm[k] = np.cross(np.cross(a, b), a - b)      # bisector plane normal: equal, positive cosines
                nrm[k] /= np.linalg.norm(nrm[k])
                if nrm[k] @ a < 0:
                    nrm[k] = -nrm[k]
        pos32, nrm32 = pos.astype(np.float32), nrm.astype(np.float32)
        nodes = [mk(pos32[k], nrm32[k]) for k in range(t + 2)]
        lo, hi = max(0, t - D), min(t, D)
        ws = [oracle_mod.mis_weight(nodes, s, D) for s in range(lo, hi + 1)]
        for s, w in zip(range(lo, hi + 1), ws):
            assert w == pytest.approx(_mis_by_definition(pos32.astype(np.float64), nrm32.astype(np.float64), s, D),
                                      rel=1e-11, abs=1e-300)
        assert sum(ws) == pytest.approx(1.0, rel=1e-12)
    # vertex 1 faces away from the listener side: it cannot have been left towards vertex 2 by the forward walk,
    # so every strategy that generates vertex 2 from the source (s = 2) has density 0
    nodes = [mk((0, 0, 0)), mk((100, 0, -100), (0, 0, 1)), mk((200, 0, -200), (0, 0, 1)), mk((300, 0, 0))]
    assert oracle_mod.mis_weight(nodes, 2, 8) == pytest.approx(1.0 / 3.0)          # its own density is 0: uniform fallback
    assert oracle_mod.mis_weight(nodes, 0, 8) + oracle_mod.mis_weight(nodes, 1, 8) == pytest.approx(1.0)
    # a duplicate node (a walk step that missed, ARTS.cpp:296): uniform weight 1 / N(t)
    nodes = [mk((0, 0, 0)), mk((50, 0, -100), (0, 0, 1)), mk((50, 0, -100), (0, 0, 1)), mk((200, 0, 0))]
    assert oracle_mod.mis_weight(nodes, 1, 8) == pytest.approx(1.0 / 3.0)
    assert oracle_mod.mis_weight(nodes, 1, 1) == pytest.approx(1.0)                # t = 2, D = 1: only s = 1 is left


def test_mis_frame_properties(oracle_mod, scene_factory):
    """Frame level (cfg1): the balance heuristic reweights the same connected paths, so occupied bins, connection
    and deposit counts equal the uniform-weight mode's; the direct path (one strategy) is untouched; the flag
    implies all-connections."""
    sc = scene_factory("shoebox", 1)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    pu = oracle_mod.default_params(num_pairs=256, depth=4, seed=11, flags=oracle_mod.FLAG_ALL_CONNECTIONS)
    pm = oracle_mod.default_params(num_pairs=256, depth=4, seed=11, flags=oracle_mod.FLAG_MIS_BALANCE)
    pb = oracle_mod.default_params(num_pairs=256, depth=4, seed=11,
                                   flags=oracle_mod.FLAG_MIS_BALANCE | oracle_mod.FLAG_ALL_CONNECTIONS)
    eu, eu64, cu = osc.compute_energy(pu, sc.source, sc.listener)
    em, em64, cm = osc.compute_energy(pm, sc.source, sc.listener)
    eb, _, _ = osc.compute_energy(pb, sc.source, sc.listener)
    assert np.array_equal(em, eb)
    assert (cu.connected, cu.deposits, cu.any_rays) == (cm.connected, cm.deposits, cm.any_rays)
    assert np.array_equal(eu != 0, em != 0)
    assert not np.allclose(eu64, em64, rtol=1e-3)
    d = float(np.linalg.norm(sc.listener.astype(np.float64) - sc.source.astype(np.float64))) / 1000.0
    direct_bin = int(np.floor(d / 343.0 * 1000.0))
    assert np.flatnonzero(em[0]).min() == direct_bin


# ---- row f4: specular / diffuse / transmitted lobes in the walk ---------------------------------------------------
def _two_floors(z0=0.0, z1=300.0, half=1e5):
    """two huge horizontal quads (4 triangles): z = z0 (material 0) and z = z1 (material 1)"""
    def quad(z):
        a, b, c, d = (-half, -half, z), (half, -half, z), (half, half, z), (-half, half, z)
        return [[a, b, c], [a, c, d]]
    tri = np.array(quad(z0) + quad(z1), np.float32)
    return tri, np.array([0, 0, 1, 1], np.uint16)


def test_kat_lobe_table_by_hand(oracle_mod):
    """MaterialAcousticProcessor.cpp:51-72 per band: alpha = 0.4 -> Refl = 0.6; tau = 0.5 is clamped to 0.4 (Refl + tau
    <= 1); sigma = 0.25 -> specular 0.45, diffuse 0.15, transmitted 0.4; the gains sum to 1, so they are the
    selection probabilities too.  Second band alpha = 1: nothing reflected, tau = 0.2 kept."""
    tri, mat = _two_floors()
    s = oracle_mod.Scene(tri, mat, np.array([[0.4, 1.0], [0.3, 0.3]], np.float32),
                         transmission=np.array([[0.5, 0.2], [0.0, 0.0]], np.float32),
                         scattering=np.array([[0.25, 0.5], [1.0, 1.0]], np.float32))
    g, p = s.lobe_table(0)
    assert np.allclose(g[:, 0], [0.15, 0.45, 0.4], rtol=1e-6) and np.allclose(g[:, 1], [0.0, 0.0, 0.2], atol=1e-7)
    mean = g.mean(axis=1)
    assert np.allclose(p, mean / mean.sum(), rtol=1e-6)
    g1, p1 = s.lobe_table(1)                    # sigma = 1, tau = 0: the reference's purely diffuse walk
    assert np.allclose(g1[0], 0.7, rtol=1e-6) and not g1[1:].any() and np.array_equal(p1, [1.0, 0.0, 0.0])
    # without arrays: tau = 0, sigma = 1 for every material
    s0 = oracle_mod.Scene(tri, mat, np.array([[0.4, 1.0], [0.3, 0.3]], np.float32))
    g0, p0 = s0.lobe_table(0)
    assert np.allclose(g0[0], [0.6, 0.0], atol=1e-7) and not g0[1:].any() and np.array_equal(p0, [1.0, 0.0, 0.0])


def test_kat_specular_and_transmitted_directions(oracle_mod):
    """Mirror floor (sigma = 0, tau = 0): the walk leaves the floor in the mirror direction, so source -> floor ->
    ceiling keeps its horizontal velocity.  Fully transmitting floor (alpha = 1, tau = 1): it goes straight on to
    the plane below.  The lobe sits in material bits 16-17 of the vertex, its probability (1 here) times the
    roulette probability is the next node's probability."""
    src = (0.0, 0.0, 100.0)
    L = oracle_mod.FLAG_MATERIAL_LOBES
    tri, mat = _two_floors(0.0, 300.0)
    mirror = oracle_mod.Scene(tri, mat, np.array([[0.2], [0.2]], np.float32),
                              transmission=np.zeros((2, 1), np.float32), scattering=np.zeros((2, 1), np.float32))
    p = oracle_mod.default_params(num_pairs=64, depth=2, russian_roulette=0, flags=L)
    seen = 0
    for pair in range(64):
        nodes = mirror.generate_path(p, pair, 0, src)
        if len(nodes) < 3 or nodes[1].pos[2] > 1.0:          # first segment went up: the ceiling is a mirror too
            continue
        seen += 1
        n0, n1, n2 = (np.array(n.pos[:], np.float64) for n in nodes[:3])
        assert nodes[1].material == (0 | (oracle_mod.LOBE_SPECULAR << oracle_mod.LOBE_SHIFT))
        assert nodes[2].material == 1 and abs(n2[2] - 299.9) < 1e-3
        assert nodes[2].prob == pytest.approx(0.9)             # roulette probability x lobe probability 1
        slope_in = (n1[:2] - n0[:2]) / (n0[2] - 0.0)           # horizontal travel per unit of height, down
        slope_out = (n2[:2] - n1[:2]) / (300.0 - 0.0)          # and up again after the mirror
        assert np.allclose(slope_in, slope_out, rtol=2e-3, atol=2e-3)
    assert seen > 10
    tri, mat = _two_floors(0.0, -200.0)
    glass = oracle_mod.Scene(tri, mat, np.array([[1.0], [0.5]], np.float32),
                             transmission=np.ones((2, 1), np.float32), scattering=np.full((2, 1), 0.5, np.float32))
    seen = 0
    for pair in range(64):
        nodes = glass.generate_path(p, pair, 0, src)
        if len(nodes) < 3 or nodes[1].pos[2] > 1.0 or nodes[1].pos[2] < -1.0:
            continue
        seen += 1
        n0, n1, n2 = (np.array(n.pos[:], np.float64) for n in nodes[:3])
        assert nodes[1].material == (0 | (oracle_mod.LOBE_TRANSMIT << oracle_mod.LOBE_SHIFT))
        assert abs(n2[2] - (-199.9)) < 1e-3                    # stopped above the lower plane, coming from above
        assert np.allclose((n1[:2] - n0[:2]) / 100.0, (n2[:2] - n1[:2]) / 200.0, rtol=2e-3, atol=2e-3)
    assert seen > 10


def test_kat_lobe_gain_in_evaluate_path(oracle_mod):
    """EvaluatePath with the flag: the vertex factor is the gain of its lobe — diffuse over pi, specular and
    transmitted as they are — instead of Absorption / pi (ARTS.cpp:382-386)."""
    tri, mat = _two_floors()
    s = oracle_mod.Scene(tri, mat, np.array([[0.4], [0.3]], np.float32),
                         transmission=np.array([[0.5], [0.0]], np.float32), scattering=np.array([[0.25], [1.0]], np.float32))
    mk = oracle_mod.make_node
    p_on = oracle_mod.default_params(num_pairs=1, flags=oracle_mod.FLAG_MATERIAL_LOBES, energy_clamp=1e30)
    p_off = oracle_mod.default_params(num_pairs=1, energy_clamp=1e30)

    def gain(lobe, params):
        nodes = [mk((0, 0, 0)), mk((2000, 0, 0), (0, 0, 1), material=0 | (lobe << 16), prob=0.5), mk((2000, 3000, 0))]
        return float(s.evaluate_path(params, nodes)[0][0])

    base = gain(0, p_off) / (0.4 / math.pi)                   # everything but the vertex factor
    assert gain(0, p_on) == pytest.approx(base * 0.15 / math.pi, rel=1e-5)
    assert gain(1, p_on) == pytest.approx(base * 0.45, rel=1e-5)
    assert gain(2, p_on) == pytest.approx(base * 0.4, rel=1e-5)


def test_connection_into_a_collision_sphere_is_blocked():
    """SURVEY A.6-h by hand: ConnectSubpaths' trace runs from F to B - 0.1 cm * unit(B - F) and ignores no actor
    (ARTS.cpp:252-254); with the listener pawn's collision sphere (radius R > 0.1 cm) around B_0 the trace ends inside the
    sphere, so it crosses its surface: blocked.  Without the sphere the same connection through free air is open.  A walk
    that starts at the listener ignores the pawn (AddIgnoredActor) — only the OTHER end point's sphere can be hit."""
    import oracle
    sc_tris = np.array([[[-5000, -5000, -100], [5000, -5000, -100], [0, 5000, -100]]], dtype=np.float32)   # a floor far below
    s = oracle.Scene(sc_tris, np.zeros(1, np.uint16), np.full((1, 1), 0.5, np.float32))
    lib = oracle.load()
    f = oracle.Node(); b = oracle.Node()
    for k, v in enumerate((0.0, 0.0, 100.0)): f.pos[k] = v
    for k, v in enumerate((300.0, 0.0, 100.0)): b.pos[k] = v
    src = (oracle.C.c_float * 3)(0.0, 0.0, 100.0)
    lis = (oracle.C.c_float * 3)(300.0, 0.0, 100.0)
    for radius, want in ((0.0, 1), (0.05, 1), (0.5, 0), (34.0, 0)):
        p = oracle.default_params(listener_radius=radius)
        assert lib.fso_connect_ep(s.h, oracle.C.byref(p), oracle.C.byref(f), oracle.C.byref(b), src, lis, None) == want, radius
    # the source's own sphere blocks a connection that starts at F_0 just the same
    p = oracle.default_params(source_radius=20.0)
    assert lib.fso_connect_ep(s.h, oracle.C.byref(p), oracle.C.byref(f), oracle.C.byref(b), src, lis, None) == 0
    # ... but not one that passes 40 cm beside it
    for k, v in enumerate((0.0, 40.0, 100.0)): f.pos[k] = v
    for k, v in enumerate((300.0, 40.0, 100.0)): b.pos[k] = v
    assert lib.fso_connect_ep(s.h, oracle.C.byref(p), oracle.C.byref(f), oracle.C.byref(b), src, lis, None) == 1


def test_kat_a_walk_ignores_its_own_actor(oracle_mod):
    """AddIgnoredActor (ARTS.cpp:322-327) by hand: a source at the origin inside a box of half-width 30 (actor 7) inside a
    big box of half-width 1000 (other actors).  A source walk's first hit is on the small box (t <= 30 * sqrt(3)) without the
    actor set, and on the big one (t >= 1000 - 0.1) with source_object = 7; a listener-side walk from the same point still
    hits the small box (the listener's actor is another one)."""
    def box(h):
        v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)], np.float64)
        quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
        return np.array([[v[a], v[b], v[c]] for a, b, c, d in quads] + [[v[a], v[c], v[d]] for a, b, c, d in quads], np.float32)
    tris = np.concatenate([box(1000.0), box(30.0)])
    obj = np.concatenate([np.arange(12, dtype=np.uint32) + 100, np.full(12, 7, np.uint32)])
    osc = oracle_mod.Scene(tris, np.zeros(24, np.uint16), np.full((1, 1), 0.5, np.float32))
    osc.set_objects(obj)
    o = np.zeros(3, np.float32)
    for pair in range(40):
        plain = osc.generate_path(oracle_mod.default_params(depth=1, russian_roulette=0), pair, 0, o)
        own = osc.generate_path(oracle_mod.default_params(depth=1, russian_roulette=0, source_object=7), pair, 0, o)
        lis = osc.generate_path(oracle_mod.default_params(depth=1, russian_roulette=0, source_object=7), pair, 1, o)
        d_plain = np.linalg.norm(np.asarray(plain[1].pos[:], np.float64))
        d_own = np.linalg.norm(np.asarray(own[1].pos[:], np.float64))
        d_lis = np.linalg.norm(np.asarray(lis[1].pos[:], np.float64))
        assert d_plain <= 30.0 * np.sqrt(3.0) + 1e-3 and d_lis <= 30.0 * np.sqrt(3.0) + 1e-3
        assert 1000.0 - 0.2 <= d_own <= 1000.0 * np.sqrt(3.0)
