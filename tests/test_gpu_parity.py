"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs and against the committed golden fixtures.

Bars: ray hits, deposit bins and every integer bit-exact; energy within 1e-3 relative RMS per band
(BASELINE.json north_star) — in practice ~1e-6 because the path SET is bit-identical and only expf/powf
ulps and the fp32 summation order differ; impulse responses within 1e-5 of the peak.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-3      # north_star: "within 1e-3 RMS per band"
TIGHT_TOL = 2e-5    # identical path sets + fp32 atomics
IR_TOL = 1e-5       # relative to the IR peak


def rel_rms(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.sqrt(np.mean(ref ** 2))
    return float(np.sqrt(np.mean((a - ref) ** 2)) / max(den, 1e-300))


def make_ctx(pkg, sc, fast=False, **kw):
    ctx = pkg.Context(num_bands=sc.num_bands, **kw)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast=fast)   # fast: the tree is built on the device
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    return ctx, src


def check_energy(e_gpu, e32, e64, bands):
    assert np.array_equal(e_gpu != 0, e32 != 0)      # identical path set => identical occupied bins
    for b in range(bands):
        assert rel_rms(e_gpu[b], e64[b]) <= TIGHT_TOL, (b, rel_rms(e_gpu[b], e64[b]))
        assert rel_rms(e_gpu[b], e32[b]) <= RMS_TOL


# ---- a10: the engine line trace ---------------------------------------------------------------------------
@pytest.mark.parametrize("build", ["host_sah", "device_morton"])
@pytest.mark.parametrize("name", ["shoebox", "starter_room", "old_mine"])
def test_line_trace_matches_brute_force(pkg, oracle_mod, scene_factory, name, build):
    """closest and any hit against the oracle's BRUTE-FORCE scan, for the host-built (binned SAH) and the device-built
    (Morton / Karras, fs_scene_commit_fast) tree: the answer is a function of ray and triangles only"""
    sc = scene_factory(name)
    ctx, _ = make_ctx(pkg, sc, fast=build == "device_morton")
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    rng = np.random.default_rng(3)
    n = 2000 if name != "old_mine" else 600
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    o = np.where(rng.random((n, 1)) < 0.5, sc.source + rng.normal(0, 60, (n, 3)), rng.uniform(lo, hi, (n, 3)))
    o = o.astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, t, tri, nrm = ctx.trace_rays(o, d, 1e6)
    tm = np.empty(n, np.float32)
    for i in range(n):
        h, tt, ti, nn = osc.trace_closest(o[i], d[i], 1e6, brute=True)
        assert bool(hit[i]) == h, i
        if h:
            assert t[i] == np.float32(tt) and tri[i] == ti and np.array_equal(nrm[i], nn), i
        tm[i] = (tt if h else 500.0) * rng.uniform(0.6, 1.4)
    any_hit, *_ = ctx.trace_rays(o, d, tm, any_hit=True)
    for i in range(0, n, 3):
        assert bool(any_hit[i]) == osc.trace_any(o[i], d[i], float(tm[i]), brute=True), i
    assert hit.mean() > 0.3
    ctx.close()


SOUPS = ["uniform", "slivers", "duplicates", "huge_coordinates", "coplanar_grid", "one_triangle", "tiny_and_big", "zero_area"]


@pytest.mark.parametrize("build", ["host_sah", "device_morton"])
@pytest.mark.parametrize("kind", SOUPS)
def test_line_trace_fuzz_soups(pkg, oracle_mod, kind, build):
    """Random triangle soups that stress the builder and the tests at their edges: needle triangles, exact
    duplicates (ties broken by input index), coordinates around 1e6 cm, a coplanar grid (rays in the plane, rays
    through shared edges and vertices), a single triangle, four orders of magnitude of triangle sizes, triangles
    without area among ordinary ones.  Closest
    hits must equal the oracle's brute-force scan bit for bit, any-hits must agree."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(kind.encode()))   # stable across processes (str hashes are salted)
    if kind == "uniform":
        c = rng.uniform(-2000, 2000, (3000, 1, 3))
        tri = c + rng.normal(0, 60, (3000, 3, 3))
    elif kind == "slivers":
        c = rng.uniform(-1500, 1500, (2000, 1, 3))
        d = rng.normal(size=(2000, 1, 3))
        t = np.array([0.0, 1.0, 0.5]).reshape(1, 3, 1)
        tri = c + d * t * rng.uniform(50, 900, (2000, 1, 1)) + rng.normal(0, 0.02, (2000, 3, 3))
    elif kind == "duplicates":
        base = rng.uniform(-800, 800, (400, 1, 3)) + rng.normal(0, 80, (400, 3, 3))
        tri = np.concatenate([base, base, base[::2]], axis=0)
    elif kind == "huge_coordinates":
        c = rng.uniform(-300, 300, (1500, 1, 3)) + np.array([9.0e5, -7.5e5, 4.0e5])
        tri = c + rng.normal(0, 40, (1500, 3, 3))
    elif kind == "coplanar_grid":
        n = 24
        xs, ys = np.meshgrid(np.arange(n) * 50.0, np.arange(n) * 50.0, indexing="ij")
        p00 = np.stack([xs, ys, np.zeros_like(xs)], -1)[:-1, :-1]
        p10 = np.stack([xs, ys, np.zeros_like(xs)], -1)[1:, :-1]
        p01 = np.stack([xs, ys, np.zeros_like(xs)], -1)[:-1, 1:]
        p11 = np.stack([xs, ys, np.zeros_like(xs)], -1)[1:, 1:]
        tri = np.concatenate([np.stack([p00, p10, p11], -2).reshape(-1, 3, 3),
                              np.stack([p00, p11, p01], -2).reshape(-1, 3, 3)], axis=0)
    elif kind == "one_triangle":
        tri = np.array([[[0, 0, 0], [100, 0, 0], [0, 100, 0]]], dtype=np.float64)
    elif kind == "zero_area":
        # a level's collision mesh as exporters leave it: among ordinary triangles, triangles without area — two corners equal,
        # all three equal, three corners on a line.  No ray hits them (det = 0 exactly or the barycentric test fails) and the
        # builders must neither choke on their empty boxes nor on the normal 0 / 0 they store for them
        c = rng.uniform(-1500, 1500, (1800, 1, 3))
        tri = c + rng.normal(0, 70, (1800, 3, 3))
        tri[0::6, 1] = tri[0::6, 0]
        tri[1::6, 1] = tri[1::6, 0]; tri[1::6, 2] = tri[1::6, 0]
        tri[2::6, 2] = 0.5 * (tri[2::6, 0] + tri[2::6, 1])
    else:
        c = rng.uniform(-1000, 1000, (2500, 1, 3))
        size = 10.0 ** rng.uniform(-1.5, 2.5, (2500, 1, 1))
        tri = c + rng.normal(0, 1, (2500, 3, 3)) * size
    tri = tri.astype(np.float32)
    T = tri.shape[0]
    mat = np.zeros(T, np.uint16)
    absorption = np.full((1, 1), 0.5, np.float32)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(tri, mat, absorption, fast=build == "device_morton")
    osc = oracle_mod.Scene(tri, mat, absorption)
    lo, hi = tri.min(axis=(0, 1)).astype(np.float64), tri.max(axis=(0, 1)).astype(np.float64)
    ext = np.maximum(hi - lo, 1.0)
    n = 600
    o = rng.uniform(lo - 0.3 * ext, hi + 0.3 * ext, (n, 3))
    d = rng.normal(size=(n, 3))
    # a third of the rays aim at triangle vertices / edge midpoints / centroids: the borderline cases
    pick = rng.integers(0, T, n)
    w = rng.dirichlet((1, 1, 1), n)
    w[: n // 9] = np.array([1.0, 0.0, 0.0])
    w[n // 9: 2 * n // 9] = np.array([0.5, 0.5, 0.0])
    target = (tri[pick].astype(np.float64) * w[:, :, None]).sum(axis=1)
    aimed = np.arange(n) < n // 3
    d[aimed] = target[aimed] - o[aimed]
    if kind == "coplanar_grid":
        o[n // 3: n // 2, 2] = 0.0                                    # rays inside the plane of the grid
        d[n // 3: n // 2, 2] = 0.0
    o = o.astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, t, idx, nrm = ctx.trace_rays(o, d, 1e7)
    tm = np.empty(n, np.float32)
    for i in range(n):
        h, tt, ti, nn = osc.trace_closest(o[i], d[i], 1e7, brute=True)
        assert bool(hit[i]) == h, (kind, i)
        if h:
            assert t[i] == np.float32(tt) and idx[i] == ti and np.array_equal(nrm[i], nn), (kind, i, t[i], tt, idx[i], ti)
        tm[i] = (tt if h else 1000.0) * rng.uniform(0.5, 1.5)
    any_hit, *_ = ctx.trace_rays(o, d, tm, any_hit=True)
    for i in range(n):
        assert bool(any_hit[i]) == osc.trace_any(o[i], d[i], float(tm[i]), brute=True), (kind, i)
    # the cooperative traversal of small frames (round 4; its boxes are the quantised ones rounded outwards to fp16 —
    # infinite beyond the fp16 range): 1 ray per wave from memory, 4 per wave with the tree's top / all that fits in LDS
    for mode in (2, 7, 8):
        got = ctx.trace_rays(o, d, 1e7, any_hit=mode)
        assert all(np.array_equal(a, b) for a, b in zip(got, (hit, t, idx, nrm))), (kind, mode)
    st = ctx.stats()
    assert st["bvh_stack_need"] <= 64 and st["triangles"] == T
    ctx.close()


def test_million_triangle_scene(pkg, oracle_mod):
    """A scene ten times the headline size: 1 000 000 triangles in a 100 m x 100 m slab.  Its tree needs more than
    32 pending stack entries in the worst case, so the kernels' LDS stack (sized at run time from the committed
    tree) is deeper than for the BASELINE scenes.  Closest hits bit-exact, a small frame within the usual bars."""
    rng = np.random.default_rng(1)
    T = 1_000_000
    c = rng.uniform(0, 10000, (T, 1, 3))
    c[:, :, 2] *= 0.05
    tri = (c + rng.uniform(-30, 30, (T, 3, 3))).astype(np.float32)
    mat = (np.arange(T) % 3).astype(np.uint16)
    absorption = np.array([[0.3, 0.6], [0.5, 0.5], [0.8, 0.2]], np.float32)
    ctx = pkg.Context(num_bands=2)
    ctx.set_scene(tri, mat, absorption)
    st = ctx.stats()
    assert st["triangles"] == T and 32 < st["bvh_stack_need"] <= 64
    osc = oracle_mod.Scene(tri, mat, absorption)
    n = 400
    o = np.column_stack([rng.uniform(0, 10000, n), rng.uniform(0, 10000, n), rng.uniform(-100, 600, n)]).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d[:, 2] *= 0.2                                                     # mostly along the slab: long traversals
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    hit, t, idx, nrm = ctx.trace_rays(o, d, 1e7)
    for i in range(n):
        h, tt, ti, nn = osc.trace_closest(o[i], d[i], 1e7, brute=(i < 30))
        assert bool(hit[i]) == h, i
        if h:
            assert t[i] == np.float32(tt) and idx[i] == ti and np.array_equal(nrm[i], nn), i
    assert hit.mean() > 0.5
    src_pos = np.array([5000, 5000, 250], np.float32)
    lis_pos = np.array([5600, 4700, 250], np.float32)
    ctx.set_listener(lis_pos)
    src = ctx.create_source(src_pos)
    e_gpu = ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=8, seed=4, dist_divisor=100.0))
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=4096, depth=8, seed=4, dist_divisor=100.0),
                                       src_pos, lis_pos)
    if cnt.connected:
        check_energy(e_gpu, e32, e64, 2)
    else:
        assert not e_gpu.any()
    # the same million triangles through the device-side (Morton) builder: another tree, the same closest hits
    ctx.set_scene(tri, mat, absorption, fast=True)
    assert ctx.stats()["triangles"] == T
    hit2, t2, idx2, nrm2 = ctx.trace_rays(o, d, 1e7)
    assert np.array_equal(hit2, hit) and np.array_equal(t2[hit], t[hit])
    assert np.array_equal(idx2[hit], idx[hit]) and np.array_equal(nrm2[hit], nrm[hit])
    ctx.close()


# ---- a1-a5: ComputeEnergyResponse ----------------------------------------------------------------------------
CFGS = [  # BASELINE.json configs[0..2]: name, bands, rays, depth
    ("shoebox", 1, 1024, 4),
    ("starter_room", 4, 16384, 8),
    ("old_mine", 8, 262144, 8),
]


@pytest.mark.parametrize("name,bands,rays,depth", CFGS)
def test_energy_parity_baseline_configs(pkg, oracle_mod, scene_factory, name, bands, rays, depth):
    sc = scene_factory(name, bands)
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=rays, depth=depth, seed=0x5EED)
    e_gpu = ctx.compute_energy_response(src, p)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    op = oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=0x5EED)
    e32, e64, cnt = osc.compute_energy_mt(op, sc.source, sc.listener, 8) if rays > 100000 else \
        osc.compute_energy(op, sc.source, sc.listener)
    assert cnt.connected > 0
    # device-side work counters == the oracle's, exactly.  All three are OBSERVED work since round 4: the walkers count every
    # hit or miss they apply, the connect lanes every pair they test and every deposit; the plan pass's prediction from the
    # RNG stream alone (planned_segments) must agree with what the walk then did
    st = ctx.stats()
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    assert st["planned_segments"] == st["segments"]
    if rays > 100000:
        assert np.array_equal(e_gpu != 0, e64 != 0)
        for b in range(bands):
            assert rel_rms(e_gpu[b], e64[b]) <= TIGHT_TOL
    else:
        check_energy(e_gpu, e32, e64, bands)
    ctx.close()


VARIANTS = [
    # id, scene, bands, rays, depth, gpu kwargs, oracle kwargs
    ("unbounded_depth", "shoebox", 1, 2000, 0, {}, {}),
    ("fixed_depth_no_rr", "starter_room", 4, 4096, 6, {"russian_roulette": 0}, {"russian_roulette": 0}),
    ("metres_scale", "starter_room", 4, 8192, 8, {"dist_divisor": 100.0}, {"dist_divisor": 100.0}),
    ("cosine_sampling", "starter_room", 2, 4096, 8, {"flags": 4}, {"flags": 4}),
    ("fixed_norm_1000", "shoebox", 3, 3000, 5, {"flags": 1}, {"flags": 1}),
    ("ragged_pairs", "starter_room", 5, 2 * 777, 8, {}, {}),
    ("other_seed_air", "old_mine", 8, 8192, 12,
     {"seed": 0xABCDEF0123, "air_absorption": [0.01 * (i + 1) for i in range(8)], "dist_divisor": 200.0},
     {"seed": 0xABCDEF0123, "air_absorption": [0.01 * (i + 1) for i in range(8)], "dist_divisor": 200.0}),
    ("min_seg_zero", "shoebox", 1, 1024, 4, {"min_seg": 0.0, "dist_divisor": 100.0},
     {"min_seg": 0.0, "dist_divisor": 100.0}),
    ("one_pair", "shoebox", 1, 2, 8, {}, {}),
    # parameter extremes
    ("roulette_half", "starter_room", 4, 8192, 8, {"rr_prob": 0.5}, {"rr_prob": 0.5}),
    ("roulette_099_depth_32", "starter_room", 2, 2048, 32, {"rr_prob": 0.99}, {"rr_prob": 0.99}),
    ("no_offset_no_pullback", "starter_room", 4, 4096, 6, {"surface_offset": 0.0, "connect_pullback": 0.0},
     {"surface_offset": 0.0, "connect_pullback": 0.0}),
    ("short_traces", "starter_room", 4, 4096, 8, {"max_trace_dist": 300.0, "dist_divisor": 100.0},
     {"max_trace_dist": 300.0, "dist_divisor": 100.0}),
    ("clamp_and_gain", "shoebox", 1, 2048, 6, {"energy_clamp": 0.001, "energy_gain": 3.0, "prob_exponent": 1.0, "dist_divisor": 100.0},
     {"energy_clamp": 0.001, "energy_gain": 3.0, "prob_exponent": 1.0, "dist_divisor": 100.0}),
    ("slow_sound_late_bins", "old_mine", 8, 4096, 8, {"sound_speed": 30.0, "dist_divisor": 100.0},
     {"sound_speed": 30.0, "dist_divisor": 100.0}),
    ("reference_defaults_2000_rays", "starter_room", 1, 2000, 0, {"flags": 1}, {"flags": 1}),
    # row f3: every forward prefix x every backward prefix, uniform MIS weights (flag 16 on both sides)
    ("all_connections_cfg1", "shoebox", 1, 1024, 4, {"flags": 16}, {"flags": 16}),
    ("all_connections_cfg2", "starter_room", 4, 8192, 8, {"flags": 16}, {"flags": 16}),
    ("all_connections_mine_no_rr", "old_mine", 8, 4096, 5, {"flags": 16, "russian_roulette": 0, "dist_divisor": 200.0},
     {"flags": 16, "russian_roulette": 0, "dist_divisor": 200.0}),
    ("all_connections_unbounded", "shoebox", 2, 600, 0, {"flags": 16}, {"flags": 16}),
    ("all_connections_one_pair", "shoebox", 1, 2, 8, {"flags": 16}, {"flags": 16}),
    # row f3 with balance-heuristic weights (flag 32 on both sides; implies 16)
    ("mis_balance_cfg1", "shoebox", 1, 1024, 4, {"flags": 32}, {"flags": 32}),
    ("mis_balance_cfg2", "starter_room", 4, 8192, 8, {"flags": 32}, {"flags": 32}),
    ("mis_balance_mine_no_rr", "old_mine", 8, 4096, 5, {"flags": 32, "russian_roulette": 0, "dist_divisor": 200.0},
     {"flags": 32, "russian_roulette": 0, "dist_divisor": 200.0}),
    ("mis_balance_unbounded_cosine", "shoebox", 2, 600, 0, {"flags": 32 | 4}, {"flags": 32 | 4}),
    ("mis_balance_one_pair", "shoebox", 1, 2, 8, {"flags": 48}, {"flags": 48}),
    ("mis_balance_no_offset", "starter_room", 4, 2048, 6, {"flags": 32, "surface_offset": 0.0, "connect_pullback": 0.0},
     {"flags": 32, "surface_offset": 0.0, "connect_pullback": 0.0}),
]


@pytest.mark.parametrize("vid,name,bands,rays,depth,gkw,okw", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_energy_parity_variants(pkg, oracle_mod, scene_factory, vid, name, bands, rays, depth, gkw, okw):
    sc = scene_factory(name, bands)
    ctx, src = make_ctx(pkg, sc)
    gkw = dict(gkw)
    okw = dict(okw)
    gkw.setdefault("seed", 0x5EED)
    okw.setdefault("seed", 0x5EED)
    p = pkg.default_params(num_rays=rays, depth=depth, **gkw)
    e_gpu = ctx.compute_energy_response(src, p)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    op = oracle_mod.default_params(num_pairs=rays // 2, depth=depth, **okw)
    e32, e64, cnt = osc.compute_energy(op, sc.source, sc.listener)
    if cnt.connected == 0:
        assert not e_gpu.any()
    else:
        check_energy(e_gpu, e32, e64, bands)
    ctx.close()


def test_moved_source_and_listener(pkg, oracle_mod, scene_factory):
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = pkg.default_params(num_rays=4096, depth=8)
    op = oracle_mod.default_params(num_pairs=2048, depth=8)
    for s_pos, l_pos in (((1700, 300, 100), (200, 1400, 300)), ((1000, 100, 50), (1010, 120, 60))):
        ctx.set_source_position(src, s_pos)
        ctx.set_listener(l_pos)
        e_gpu = ctx.compute_energy_response(src, p)
        e32, e64, _ = osc.compute_energy(op, s_pos, l_pos)
        check_energy(e_gpu, e32, e64, 4)
    ctx.close()


def test_two_sources_one_context(pkg, oracle_mod, scene_factory):
    """cfg5 shape: several registered sources, one listener; buffers are per source."""
    sc = scene_factory("old_mine", 8)
    ctx, src0 = make_ctx(pkg, sc)
    src1 = ctx.create_source(sc.extra_sources[3])
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = pkg.default_params(num_rays=8192, depth=8)
    op = oracle_mod.default_params(num_pairs=4096, depth=8)
    e0 = ctx.compute_energy_response(src0, p)
    e1 = ctx.compute_energy_response(src1, p)
    r0 = osc.compute_energy(op, sc.source, sc.listener)
    r1 = osc.compute_energy(op, sc.extra_sources[3], sc.listener)
    check_energy(e0, r0[0], r0[1], 8)
    check_energy(e1, r1[0], r1[1], 8)
    assert np.array_equal(ctx.energy_buffer(src0), e0)   # untouched by the other source's update
    ctx.destroy_source(src1)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.compute_energy_response(src1, p)
    assert ei.value.code == pkg._capi.ERR_BAD_HANDLE
    ctx.close()


def test_frame_shapes_reuse_the_subpath_state(pkg, oracle_mod, scene_factory):
    """One context, frames of changing shape (rays x depth): the subpath state and the length-plan buckets are sized
    by two capacities (subpaths, subpaths x depth) and reused; every shape that fits them must still be exact —
    in particular many shallow walks after few deep ones (the plan needs (depth + 1) x subpaths slots)."""
    sc = scene_factory("starter_room", 2)
    ctx, src = make_ctx(pkg, sc)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    for rays, depth in ((4000, 1), (400, 20), (4000, 2), (128, 31), (3968, 2), (4000, 1), (2, 64), (4000, 3)):
        e = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=rays + depth, rr_prob=0.97))
        e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=rays + depth,
                                                                    rr_prob=0.97), sc.source, sc.listener)
        if cnt.connected:
            check_energy(e, e32, e64, 2)
        else:
            assert not e.any()
    ctx.close()


def test_force_update_sources_batches_all_active_sources(pkg, oracle_mod, scene_factory):
    """ForceUpdateSources / Tick of the mirrored subsystem (ARTS.cpp:55-85, 883-886) update every active source — here as
    one batched frame: each component ends up with the energy and IR of its own UpdateSource."""
    sc = scene_factory("starter_room", 4)
    sub = pkg.AudioRayTracingSubsystem(num_bands=4)
    sub.RegisterGeometry(sc.triangles, sc.material_ids)
    sub.SetMaterials(sc.absorption)
    sub.SetListenerLocation(sc.listener)
    comps = []
    for off in ((0, 0, 0), (180, -60, 30), (-250, 140, -20)):
        c = pkg.FrequenSeeAudioComponent(sc.source + np.array(off, np.float32))
        c.OnRegister(sub)
        comps.append(c)
    sub.params = pkg.default_params(num_rays=4096, depth=8, seed=5)
    one_by_one = [sub.UpdateSource(c, sub.params).copy() for c in comps]
    irs = [c.GetImpulseResponse()[0].copy() for c in comps]
    for c in comps:
        c.FlushEnergyBuffer()
    sub.Tick(0.016)
    for c, e, ir in zip(comps, one_by_one, irs):
        got = c.EnergyBuffer.reshape(e.shape)
        assert np.array_equal(got != 0, e != 0) and max(rel_rms(got[b], e[b].astype(np.float64)) for b in range(4)) <= TIGHT_TOL
        assert np.abs(c.GetImpulseResponse()[0] - ir).max() <= IR_TOL * max(np.abs(ir).max(), 1e-30)
    assert not np.array_equal(one_by_one[0], one_by_one[1])
    # the reference's own shape of Tick — one UpdateSource after the other — streamed through pipelined frames
    sub.SetPipelining(2)
    seed = sub.params.seed                      # Tick advanced it: the streamed frames use the next seed
    want = [sub.UpdateSource(c, sub.params).copy() for c in comps]
    want_ir = [c.GetImpulseResponse()[0].copy() for c in comps]
    for c in comps:
        c.FlushEnergyBuffer()
    assert sub.params.seed == seed
    sub.Tick(0.016)
    sub.ctx.synchronize()
    for c, e, ir in zip(comps, want, want_ir):
        got = c.EnergyBuffer.reshape(e.shape)
        assert np.array_equal(got != 0, e != 0) and max(rel_rms(got[b], e[b].astype(np.float64)) for b in range(4)) <= TIGHT_TOL
        assert np.abs(c.GetImpulseResponse()[0] - ir).max() <= IR_TOL * max(np.abs(ir).max(), 1e-30)
    sub.Deinitialize()


BATCH_CASES = [
    # id, scene, bands, rays per source, depth, sources, extra params
    ("three_sources_ragged", "starter_room", 4, 2 * 777, 8, 3, {}),
    ("eight_sources_mine", "old_mine", 8, 8192, 8, 8, {}),
    ("two_sources_dense", "old_mine", 8, 131072, 6, 2, {}),
    ("deterministic", "starter_room", 4, 4096, 8, 3, {"flags": 8}),
    ("lobes_no_rr", "starter_room", 4, 2048, 5, 4, {"flags": 64, "russian_roulette": 0}),
    ("all_connections_fallback", "shoebox", 1, 512, 4, 3, {"flags": 16}),
    ("one_source", "shoebox", 1, 1024, 6, 1, {}),
    ("unbounded_walks", "starter_room", 4, 8192, 0, 3, {}),
]


@pytest.mark.parametrize("cid,name,bands,rays,depth,nsrc,extra", BATCH_CASES, ids=[c[0] for c in BATCH_CASES])
def test_batched_sources_equal_separate_frames(pkg, oracle_mod, scene_factory, cid, name, bands, rays, depth, nsrc, extra):
    """fs_compute_energy_response_batch_async (UpdateSource over ActiveSources in one traced frame): every source gets
    the result of its own call — identical occupied bins, energies within the atomics' rounding (bit-identical in
    deterministic mode), the same work counters in total — and matches the oracle run per source position."""
    sc = scene_factory(name, bands)
    ctx, src0 = make_ctx(pkg, sc)
    rng = np.random.default_rng(nsrc)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    positions = [sc.source] + [(sc.source + rng.uniform(-0.1, 0.1, 3) * (hi - lo)).astype(np.float32) for _ in range(nsrc - 1)]
    srcs = [src0] + [ctx.create_source(pos) for pos in positions[1:]]
    p = pkg.default_params(num_rays=rays, depth=depth, seed=77, **extra)
    separate = []
    ctx.reset_stats()
    for s_ in srcs:
        separate.append(ctx.compute_energy_response(s_, p).copy())
    st_sep = ctx.stats()
    ctx.reset_stats()
    ctx.compute_energy_response_batch_async(srcs, p)
    for s_ in srcs:
        ctx.reconstruct_impulse_response_async(s_, p)
    ctx.synchronize()
    st_bat = ctx.stats()
    batched = [ctx.energy_buffer(s_).copy() for s_ in srcs]
    irs = [ctx.impulse_response(s_, 0).copy() for s_ in srcs]
    assert all(st_sep[k] == st_bat[k] for k in ("segments", "connections_tested", "deposits"))
    oflags = extra.get("flags", 0) & ~8
    okw = {k: v for k, v in extra.items() if k != "flags"}
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    for i, s_ in enumerate(srcs):
        assert np.array_equal(batched[i] != 0, separate[i] != 0), i
        if extra.get("flags", 0) & 8:
            assert np.array_equal(batched[i], separate[i]), i
        else:
            assert max(rel_rms(batched[i][b], separate[i][b].astype(np.float64)) for b in range(bands)) <= TIGHT_TOL
        e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=77, flags=oflags, **okw),
                                           positions[i], sc.listener)
        if cnt.connected:
            for b in range(bands):
                assert rel_rms(batched[i][b], e64[b]) <= TIGHT_TOL, (i, b)
        # the per-source reconstruct behind the batch: IR of the band-mean energy
        mean = (batched[i].sum(axis=0, dtype=np.float32) / np.float32(bands)).astype(np.float32) if bands > 1 else batched[i][0]
        want = oracle_mod.reconstruct(mean)
        assert np.abs(irs[i] - want).max() <= IR_TOL * max(np.abs(want).max(), 1e-30)
    if nsrc > 1:
        assert not np.array_equal(batched[0], batched[1])            # different positions, different results
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response_batch_async([srcs[0], srcs[0]], p)
    ctx.close()


def test_empty_scene_and_zero_rays(pkg, scene_factory):
    ctx = pkg.Context(num_bands=2)
    with pytest.raises(pkg.FrequenSeeError) as ei:          # not committed
        ctx.compute_energy_response(ctx.create_source((0, 0, 0)), pkg.default_params(num_rays=64, depth=4))
    assert ei.value.code == pkg._capi.ERR_NOT_COMMITTED
    ctx.set_scene(np.zeros((0, 3, 3), np.float32), np.zeros((0,), np.uint16), np.zeros((0, 2), np.float32))
    ctx.set_listener((100, 0, 0))
    src = ctx.create_source((0, 0, 0))
    # no geometry: every walk step misses, every pair connects over a 100 cm segment (skipped) -> gain 10
    e = ctx.compute_energy_response(src, pkg.default_params(num_rays=512, depth=4))
    assert e.shape == (2, 1000) and np.allclose(e[:, 0], 10.0, rtol=1e-5) and not e[:, 1:].any()
    e = ctx.compute_energy_response(src, pkg.default_params(num_rays=0, depth=4))
    assert not e.any()
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, pkg.default_params(num_rays=7, depth=4))     # odd
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, pkg.default_params(num_rays=8, depth=65))    # > FS_MAX_DEPTH
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, pkg.default_params(num_rays=(1 << 30) + 2, depth=4))   # 32-bit subpath indices
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, pkg.default_params(num_rays=8, depth=4, dist_divisor=0.0))
    bad = pkg.default_params(num_rays=8, depth=4)
    bad.struct_size = 12
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, bad)
    with pytest.raises(pkg.FrequenSeeError):                                          # non-finite vertex
        ctx.set_scene(np.full((1, 3, 3), np.nan, np.float32), np.zeros(1, np.uint16), np.zeros((1, 2), np.float32))
    assert "non-finite" in ctx.lib.fs_last_error(ctx.h).decode()
    ctx.close()


PLACEMENTS = [
    # id, source, listener (cm) in the starter room (2000 x 1600 x 400)
    ("same_point", (900.0, 700.0, 150.0), (900.0, 700.0, 150.0)),
    ("one_millimetre_apart", (900.0, 700.0, 150.0), (900.1, 700.0, 150.0)),
    ("source_outside_the_room", (-800.0, 700.0, 150.0), (900.0, 700.0, 150.0)),
    ("both_outside", (-800.0, 700.0, 150.0), (3000.0, -500.0, 900.0)),
    ("source_on_the_floor", (500.0, 500.0, 0.0), (1500.0, 1100.0, 160.0)),
    ("far_away_coordinates", (2.0e5, 3.0e5, 150.0), (900.0, 700.0, 150.0)),
]


@pytest.mark.parametrize("pid,spos,lpos", PLACEMENTS, ids=[p[0] for p in PLACEMENTS])
def test_energy_parity_unusual_placements(pkg, oracle_mod, scene_factory, pid, spos, lpos):
    """Source and listener where a level designer would not put them: coincident, outside the geometry, on a
    surface, hundreds of metres away.  The reference has no guards for any of these (ARTS.cpp:287-355 just
    walks); whatever the oracle does with them, the HIP path does the same."""
    sc = scene_factory("starter_room", 4)
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(lpos)
    src = ctx.create_source(spos)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    for flags in (0, 16):
        p = pkg.default_params(num_rays=4096, depth=6, seed=17, dist_divisor=100.0, flags=flags)
        e_gpu = ctx.compute_energy_response(src, p)
        e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=2048, depth=6, seed=17, dist_divisor=100.0, flags=flags),
                                           np.array(spos, np.float32), np.array(lpos, np.float32))
        assert np.all(np.isfinite(e_gpu))
        st = ctx.stats()
        if cnt.connected == 0:
            assert not e_gpu.any(), (pid, flags)
        else:
            check_energy(e_gpu, e32, e64, 4)
        ctx.reset_stats()
    ctx.close()


# ---- a6-a7: ReconstructImpulseResponse / GetImpulseResponse ---------------------------------------------------
def test_impulse_response_parity(pkg, oracle_mod, scene_factory):
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=16384, depth=8, dist_divisor=100.0)
    e = ctx.compute_energy_response(src, p)
    ctx.reconstruct_impulse_response(src, p)
    for b in range(4):
        ref = oracle_mod.reconstruct(e[b])
        got = ctx.band_impulse_response(src, b)
        assert np.abs(got - ref).max() <= IR_TOL * np.abs(ref).max()
    mean_e = (e.sum(axis=0, dtype=np.float32) / np.float32(4)).astype(np.float32)
    ref = oracle_mod.reconstruct(mean_e)
    ch0, ch1 = ctx.impulse_response(src, 0), ctx.impulse_response(src, 1)
    assert np.array_equal(ch0, ch1)                        # both channels identical (FSAC.cpp:331)
    assert np.abs(ch0 - ref).max() <= 2 * IR_TOL * np.abs(ref).max()
    # samples_per_bin = 48 compat flag (A.6-j)
    p48 = pkg.default_params(samples_per_bin=48)
    ctx.reconstruct_impulse_response(src, p48)
    ref48 = oracle_mod.reconstruct(e[1], samples_per_bin=48)
    assert np.abs(ctx.band_impulse_response(src, 1) - ref48).max() <= IR_TOL * np.abs(ref48).max()
    ctx.close()


def test_component_surface_like_the_reference(pkg, oracle_mod, scene_factory):
    """Drive the same sequence as ARTS.cpp:157-192 through the mirrored component interface."""
    sc = scene_factory("shoebox", 1)
    sub = pkg.AudioRayTracingSubsystem(num_bands=1)
    sub.RegisterGeometry(sc.triangles, sc.material_ids)
    sub.SetMaterials(sc.absorption)
    comp = pkg.FrequenSeeAudioComponent(sc.source)
    comp.OnRegister(sub)
    sub.SetListenerLocation(sc.listener)
    sub._commit()
    assert comp.NumBins == 1000 and comp.NumSamples == 48000
    ir0 = comp.GetImpulseResponse()
    assert len(ir0) == 2 and ir0[0].shape == (48000,) and not ir0[0].any()      # Init(0, NumSamples)
    # FlushEnergyBuffer + AddEnergyAtDelay + ReconstructImpulseResponse: the one-hot KAT (A.7-4)
    comp.FlushEnergyBuffer()
    comp.AddEnergyAtDelay(0.0145772595, 2.83805637e-7)      # A.7-1 -> bin 14
    comp.AddEnergyAtDelay(0.0105, 0.04)                     # bin 10
    comp.AddEnergyAtDelay(1.7, 0.5)                         # clamped to 999
    comp.AddEnergyAtDelay(-1.0, 0.25)                       # clamped to 0
    e = comp.EnergyBuffer
    assert e[14] == np.float32(2.83805637e-7) and e[10] == np.float32(0.04) and e[999] == 0.5 and e[0] == 0.25
    comp.ReconstructImpulseResponse()
    ref = oracle_mod.reconstruct(e)
    assert np.abs(comp.GetImpulseResponse()[0] - ref).max() <= IR_TOL * np.abs(ref).max()
    # UpdateEnergyBuffer: whole-buffer hand-off; wrong length -> the reference's check() -> error status
    new = np.zeros(1000, np.float32)
    new[10] = 0.04
    comp.UpdateEnergyBuffer(new)
    assert np.array_equal(comp.EnergyBuffer, new)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        comp.UpdateEnergyBuffer(np.zeros(999, np.float32))
    assert ei.value.code == pkg._capi.ERR_SIZE_MISMATCH
    comp.ReconstructImpulseResponse()
    a10 = 0.04 / np.sqrt(0.04 * np.sqrt(4 * np.pi))
    ir = comp.GetImpulseResponse()[0]
    assert not ir[:490].any() and ir[500] > 0 and ir[560] > 0              # ramps live in 490..587
    assert abs(ir.max() - oracle_mod.reconstruct(new).max()) < 1e-7 and ir.max() < a10
    # UpdateSource = trace + deposit + reconstruct (intent: deposit THEN reconstruct)
    p = pkg.default_params(num_rays=1024, depth=4)
    e = sub.UpdateSource(comp, p)
    ref = oracle_mod.reconstruct(e[0])
    assert e.any() and np.abs(comp.GetImpulseResponse()[1] - ref).max() <= IR_TOL * np.abs(ref).max()
    # quirk A.6-a reproduced literally: the second flush before reconstruct zeroes the IR (ARTS.cpp:191)
    pq = pkg.default_params(num_rays=1024, depth=4, flags=pkg._capi.FLAG_FLUSH_BEFORE_RECONSTRUCT)
    sub.UpdateSource(comp, pq)
    assert not comp.GetImpulseResponse()[0].any() and not comp.EnergyBuffer.any()
    # line trace through the subsystem
    hit, t, tri, n = sub.LineTraceSingle((500, 400, 150), (500, 400, 1000))
    assert hit and abs(t - 150.0) < 1e-3 and tuple(n) == (0.0, 0.0, -1.0)
    comp.OnUnregister()
    assert sub.ActiveSources == []
    sub.Deinitialize()


def test_published_ir_ring_is_stable(pkg, scene_factory):
    """GetImpulseResponse(): a returned pointer stays valid (unchanged) until the second-next publish."""
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=1024, depth=4)
    ctx.compute_energy_response(src, p)
    ctx.reconstruct_impulse_response(src, p)
    view1 = ctx.impulse_response_view(src, 0)
    snap1 = view1.copy()
    ctx.set_source_position(src, (600, 300, 100))
    ctx.compute_energy_response(src, p)
    ctx.reconstruct_impulse_response(src, p)
    view2 = ctx.impulse_response_view(src, 0)
    assert view2.ctypes.data != view1.ctypes.data
    assert np.array_equal(view1, snap1) and not np.array_equal(view2, snap1)
    for _ in range(5):                                    # async pipeline, front advances, no tearing
        ctx.compute_energy_response_async(src, p)
        ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    assert np.array_equal(ctx.impulse_response(src, 0), ctx.impulse_response_view(src, 0))
    assert ctx.impulse_response(src, 0).any()
    ctx.close()


# ---- multi-GPU sharding on one device: rank/world contexts sum to the single-rank frame -----------------------
@pytest.mark.parametrize("world", [2, 4, 8])
def test_shard_invariance_on_device(pkg, scene_factory, world):
    sc = scene_factory("starter_room", 4)
    p = pkg.default_params(num_rays=16384, depth=8)
    ctx, src = make_ctx(pkg, sc)
    full = ctx.compute_energy_response(src, p).astype(np.float64)
    ctx.close()
    acc = np.zeros_like(full)
    for r in range(world):
        c, s = make_ctx(pkg, sc, rank=r, world_size=world)
        acc += c.compute_energy_response(s, p)
        st = c.stats()
        a, b = pkg.sharding.pair_range(8192, r, world)
        assert st["pairs"] == b - a
        c.close()
    assert np.array_equal(acc != 0, full != 0)
    for b in range(4):
        assert rel_rms(acc[b], full[b]) <= TIGHT_TOL


def test_tail_stream_overlap_and_handoff(pkg, oracle_mod, scene_factory):
    """Frame f's tail (caller's collective, reconstruct, publish) runs on the tail stream while frame f+1 is
    already being traced on the compute stream into the source's OTHER energy buffer.  Every published IR must
    still be the reconstruct of its own frame's (reduced) energy.  The collective is stood in for by a
    device-to-device copy the test issues on the tail stream — exactly where bench.py issues the RCCL
    all-reduce — which replaces the frame's energy by that of a different, synchronously computed frame."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")   # the runtime libfrequensee.so is linked against (already loaded)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    ref_ctx, ref_src = make_ctx(pkg, sc)          # fully synchronous reference frames

    def params(seed):
        return pkg.default_params(num_rays=16384, depth=8, seed=seed, dist_divisor=100.0)

    frames = 6
    ctx.compute_energy_response_async(src, params(100))
    ptrs = set()
    for f in range(frames):
        e = ref_ctx.compute_energy_response(ref_src, params(200 + f))       # the "reduced" energy of frame f
        rptr, _ = ref_ctx.energy_device_ptr(ref_src)
        eptr, nbytes, tail = ctx.energy_handoff(src)
        assert nbytes == e.nbytes
        ptrs.add(eptr)
        assert hip.hipMemcpyAsync(eptr, rptr, nbytes, 3, tail) == 0         # hipMemcpyDeviceToDevice, on the tail stream
        ctx.reconstruct_impulse_response_async(src, params(100 + f))
        if f + 1 < frames:
            ctx.compute_energy_response_async(src, params(100 + f + 1))     # overlaps the tail of frame f
        ctx.synchronize()
        got = ctx.impulse_response(src, 0)
        mean = (e.sum(axis=0, dtype=np.float32) / np.float32(4)).astype(np.float32)
        want = oracle_mod.reconstruct(mean)
        assert np.abs(want).max() > 0
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max(), f
        if f + 1 < frames:   # and the overlapped trace of frame f+1 is a normal frame
            own = ref_ctx.compute_energy_response(ref_src, params(100 + f + 1))
            now = ctx.energy_buffer(src)
            assert np.array_equal(now != 0, own != 0) and max(rel_rms(now[b], own[b]) for b in range(4)) <= TIGHT_TOL
    assert len(ptrs) == frames                                      # every frame deposits into the next of the source's energy buffers (a rotation of 24)
    # energy helpers act on the current buffer and stay ordered against the tail: flush + one deposit + reconstruct
    ctx.check(ctx.lib.fs_flush_energy_buffer(ctx.h, src))
    ctx.check(ctx.lib.fs_add_energy_at_delay(ctx.h, src, 0, 0.0105, 4.0))
    ctx.reconstruct_impulse_response(src)
    e = ctx.energy_buffer(src)
    assert e[0, 10] == 4.0 and e.sum() == 4.0
    ctx.close()
    ref_ctx.close()


def test_unbounded_depth_walks_until_the_roulette_ends(pkg, oracle_mod, scene_factory, monkeypatch):
    """depth = 0 is the reference's while (true) (ARTS.cpp:294): a walk ends only when the roulette ends it.  At 65 536
    rays ~77 subpaths take more than FS_MAX_DEPTH = 64 steps (0.9^64); their later records live in the second tier.
    Oracle parity incl. the work counters, also for the all-connections weights (depth cap D = infinity); and a tier
    provisioned too small is grown and the frame traced again (FS_ERR_OVERFLOW on the async path)."""
    sc = scene_factory("starter_room", 4)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx, src = make_ctx(pkg, sc)
    for rays, flags in ((65536, 0), (2000, 0), (512, 16), (512, 32)):
        ctx.reset_stats()
        e = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=0, seed=rays, flags=flags))
        st = ctx.stats()
        e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=0, seed=rays, flags=flags),
                                           sc.source, sc.listener)
        check_energy(e, e32, e64, 4)
        assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
        if rays == 65536:
            assert cnt.closest_rays > 9.0 * rays * 0.98          # mean walk length 9 (not 8.99 capped at 64)
    ctx.close()
    # a second tier of ONE slot cannot hold the ~77 long walks of this frame
    monkeypatch.setenv("FS_OVER_CAP", "1")
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=65536, depth=0, seed=65536)
    ctx.compute_energy_response_async(src, p)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.synchronize()
    assert ei.value.code == pkg._capi.ERR_OVERFLOW
    e = ctx.compute_energy_response(src, p)                  # the synchronous entry point retraces by itself
    e32, e64, _ = osc.compute_energy(oracle_mod.default_params(num_pairs=32768, depth=0, seed=65536), sc.source, sc.listener)
    check_energy(e, e32, e64, 4)
    ctx.close()


def test_accumulate_energy_like_head(pkg, oracle_mod, scene_factory):
    """FS_FLAG_ACCUMULATE_ENERGY: at HEAD FlushEnergyBuffer is EnergyBuffer.SetNumZeroed(NumBins) (FSAC.h:76-79), which
    keeps existing values — every UpdateSource adds to the energy of all earlier ones.  Three frames with the flag equal
    the oracle's accumulated buffer; a frame without it starts from zero again."""
    sc = scene_factory("starter_room", 4)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx, src = make_ctx(pkg, sc)
    acc = None
    for f in range(3):
        e = ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=8, seed=40 + f, flags=128))
        op = oracle_mod.default_params(num_pairs=4096, depth=8, seed=40 + f, flags=128 if f else 0)
        acc = osc.compute_energy(op, sc.source, sc.listener, into=acc[:2] if acc else None)
        assert np.array_equal(e != 0, acc[0] != 0)
        assert max(rel_rms(e[b], acc[1][b]) for b in range(4)) <= TIGHT_TOL
        ctx.reconstruct_impulse_response(src)
    single = ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=8, seed=40))
    e32, e64, _ = osc.compute_energy(oracle_mod.default_params(num_pairs=4096, depth=8, seed=40), sc.source, sc.listener)
    check_energy(single, e32, e64, 4)
    ctx.close()
    ctx, src = make_ctx(pkg, sc, rank=0, world_size=2)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=8, flags=128))
    assert ei.value.code == pkg._capi.ERR_INVALID_ARGUMENT
    ctx.close()


def test_device_built_tree_gives_the_same_frames(pkg, oracle_mod, scene_factory):
    """fs_scene_commit_fast (RegisterGeometry / UnregisterGeometry at run time, ARTS.h:99-100): the tree is built on
    the device.  The set of paths does not depend on the tree: energy bins, counters and the IR equal the oracle's and
    the host-built tree's; a registration change (triangles added, then removed) is a fresh fast commit; moving
    geometry refits the device-built tree like the host-built one."""
    sc = scene_factory("starter_room", 4)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = pkg.default_params(num_rays=16384, depth=8, seed=31)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=8192, depth=8, seed=31), sc.source, sc.listener)
    ctx, src = make_ctx(pkg, sc, fast=True)
    st = ctx.stats()
    assert st["triangles"] == sc.num_triangles and 0 < st["bvh_stack_need"] <= 64 and st["bvh_nodes"] > sc.num_triangles // 8
    ctx.reset_stats()
    e = ctx.compute_energy_response(src, p)
    check_energy(e, e32, e64, 4)
    st = ctx.stats()
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    # an actor registers (a slab in the middle of the room), then unregisters again
    slab = np.array([[[900, 300, 0], [1100, 300, 0], [1100, 300, 380]], [[900, 300, 0], [1100, 300, 380], [900, 300, 380]]], np.float32)
    tri2 = np.concatenate([sc.triangles, slab]).astype(np.float32)
    mat2 = np.concatenate([sc.material_ids, np.zeros(2, np.uint16)])
    ctx.set_scene(tri2, mat2, sc.absorption, fast=True)
    e2 = ctx.compute_energy_response(src, p)
    o2 = oracle_mod.Scene(tri2, mat2, sc.absorption).compute_energy(oracle_mod.default_params(num_pairs=8192, depth=8, seed=31),
                                                                    sc.source, sc.listener)
    check_energy(e2, o2[0], o2[1], 4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast=True)
    check_energy(ctx.compute_energy_response(src, p), e32, e64, 4)
    # moving geometry on the device-built tree: shift the first 200 triangles, refit, compare with a fresh host build
    moved = sc.triangles.copy()
    moved[:200] += np.array([35.0, -20.0, 10.0], np.float32)
    ctx.update_triangles(0, moved[:200])
    em = ctx.compute_energy_response(src, p)
    ref, rsrc = make_ctx(pkg, sc)
    ref.set_scene(moved, sc.material_ids, sc.absorption)       # a fresh host (SAH) build of the moved geometry
    er = ref.compute_energy_response(rsrc, p)
    assert np.array_equal(em != 0, er != 0) and max(rel_rms(em[b], er[b]) for b in range(4)) <= TIGHT_TOL
    ref.close()
    ctx.close()


def test_progressive_commit_swaps_the_sah_tree_in(pkg, scene_factory):
    """fs_scene_commit_progressive: frames trace through the device-built tree at once; the host's SAH build runs on a
    background thread and is swapped in by the first trace after it has finished.  Nothing a caller can observe changes
    but the tree's statistics (and the speed): deterministic frames are bit-identical before and after the swap, also with
    triangles moved while the build was running, with pipelined frames held across the swap, and a new registration
    abandons an outstanding build."""
    sc = scene_factory("old_mine", 8)
    DETF = 8
    ref, rsrc = make_ctx(pkg, sc)                      # host-built SAH tree from the start
    host_nodes = ref.stats()["bvh_nodes"]
    ctx = pkg.Context(num_bands=8)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast="progressive")
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    assert ctx.stats()["bvh_nodes"] != host_nodes      # the Morton tree for now
    frames = [pkg.default_params(num_rays=32768, depth=8, seed=60 + i, flags=DETF) for i in range(6)]
    want = []
    moved = np.asarray(sc.triangles[:300], np.float32) + np.array([25.0, -15.0, 8.0], np.float32)
    for i, p in enumerate(frames):
        if i == 2:
            ref.update_triangles(0, moved)
        want.append(ref.compute_energy_response(rsrc, p).copy())
    ctx.set_pipelining(2)
    got = []
    for i, p in enumerate(frames):
        if i == 2:                                     # moving geometry while the background build may still be running
            ctx.update_triangles(0, moved)
        if i == 4:                                     # from here on the SAH tree: wait for the build, swap
            ctx.refine_wait()
            assert not ctx.refine_pending()
            assert ctx.stats()["bvh_nodes"] == host_nodes
        ctx.compute_energy_response_async(src, p)      # (frames 2 and 3 are held across the swap)
        if i != 3:
            ctx.synchronize()
            got.append(ctx.energy_buffer(src).copy())
        else:
            got.append(None)
    ctx.synchronize()
    for i, (g, w) in enumerate(zip(got, want)):
        if g is not None:
            assert np.array_equal(g, w), i
    assert np.array_equal(ctx.energy_buffer(src), want[-1])
    # without the explicit wait the swap happens by itself at a later trace
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast="progressive")
    import time
    t0 = time.time()
    while ctx.refine_pending() and time.time() - t0 < 30:
        ctx.compute_energy_response(src, frames[0])
    assert not ctx.refine_pending() and ctx.stats()["bvh_nodes"] == host_nodes
    ref.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    assert np.array_equal(ctx.compute_energy_response(src, frames[1]), ref.compute_energy_response(rsrc, frames[1]))
    # a new registration abandons an outstanding build
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, fast="progressive")
    ctx.set_scene(sc.triangles[:5000], sc.material_ids[:5000], sc.absorption, fast=True)
    assert not ctx.refine_pending() and ctx.stats()["triangles"] == 5000
    ctx.compute_energy_response(src, frames[0])
    ctx.close()
    ref.close()


def test_library_collective_one_rank(pkg, oracle_mod, scene_factory):
    """The RCCL all-reduce lives behind the C ABI (fs_comm_init): with a one-rank communicator attached every frame
    runs the library's collective on the tail stream — a sum over one rank — and the scene goes through the broadcast
    path of fs_scene_commit.  Energy, IR and counters must equal those of a plain context, over several overlapping
    frames, also in deterministic mode (uint64 all-reduce) and for a batched frame."""
    sc = scene_factory("starter_room", 4)
    plain, psrc = make_ctx(pkg, sc)
    ctx = pkg.Context(num_bands=4)
    assert ctx.comm_info() == (0, -1, 0)
    ctx.comm_init(pkg.Context.comm_unique_id())           # before set_scene: rank 0 builds + broadcasts the tree
    assert ctx.comm_info() == (1, 0, 1)                   # fs_comm_info asks RCCL itself: one rank, this is rank 0, ncclAllReduce
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    assert ctx.stats()["bvh_nodes"] == plain.stats()["bvh_nodes"]
    for f, flags in enumerate((0, 0, 8, 0, 8)):
        p = pkg.default_params(num_rays=16384, depth=8, seed=900 + f, flags=flags)
        ctx.compute_energy_response_async(src, p)
        ctx.reconstruct_impulse_response_async(src, p)
        if f + 1 < 5:   # the next frame's tracing overlaps this frame's reduce + reconstruct
            ctx.compute_energy_response_async(src, pkg.default_params(num_rays=4096, depth=8, seed=1))
        want_e = plain.compute_energy_response(psrc, p)
        plain.reconstruct_impulse_response(psrc, p)
        ctx.synchronize()
        got_ir, want_ir = ctx.impulse_response(src, 0), plain.impulse_response(psrc, 0)
        assert np.abs(want_ir).max() > 0
        if flags & 8:
            assert np.array_equal(got_ir, want_ir)        # integer sums: bit-identical
        else:
            assert np.abs(got_ir - want_ir).max() <= IR_TOL * np.abs(want_ir).max()
    # the reduced energy is what the helpers read back
    p = pkg.default_params(num_rays=16384, depth=8, seed=77)
    e = ctx.compute_energy_response(src, p)
    want = plain.compute_energy_response(psrc, p)
    assert np.array_equal(e != 0, want != 0) and max(rel_rms(e[b], want[b]) for b in range(4)) <= TIGHT_TOL
    assert np.array_equal(ctx.energy_buffer(src) != 0, want != 0)
    # closest hits through the broadcast tree
    rng = np.random.default_rng(5)
    o = np.tile(np.asarray(sc.source, np.float32), (512, 1))
    d = rng.normal(size=(512, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a, b = ctx.trace_rays(o, d, 1e6), plain.trace_rays(o, d, 1e6)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ctx.comm_detach()
    ctx.close()
    plain.close()


def test_sharded_context_refuses_a_partial_reconstruct(pkg, scene_factory):
    """ReconstructImpulseResponse is not linear in the energy: a rank of a sharded run must not publish the IR of
    its partial histogram.  Without a communicator the library says so (FS_ERR_COMM) unless the caller reduced the
    frame behind fs_energy_handoff; the partition each rank traces is the library's own fs_shard_range."""
    sc = scene_factory("starter_room", 4)
    p = pkg.default_params(num_rays=16384, depth=8)
    ctx, src = make_ctx(pkg, sc, rank=1, world_size=2)
    ctx.compute_energy_response(src, p)
    a, b = pkg.sharding.library_pair_range(16384, 1, 2)
    assert ctx.stats()["pairs"] == b - a
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.reconstruct_impulse_response(src, p)
    assert ei.value.code == pkg._capi.ERR_COMM and "partial" in str(ei.value)
    ctx.compute_energy_response_async(src, p)
    ctx.energy_handoff(src)                                # the caller's collective would run here, on the tail stream
    ctx.reconstruct_impulse_response(src, p)               # accepted now
    assert np.abs(ctx.impulse_response(src, 0)).max() > 0
    ctx.close()


# ---- row f3: all-prefix connections ----------------------------------------------------------------------------
def test_all_connections_properties_full_size(pkg, oracle_mod, scene_factory):
    """cfg3 size (262 144 rays, depth 8, 100 000 triangles) in all-connections mode: oracle parity on a pair
    subset through sharding (rank 0 of 16 traces the first 8 192 pairs with the full frame's normaliser), and
    size-independent properties of the whole frame: shard invariance, and the i = k, j = m strategy alone
    (weight 1/N) is contained in it, so energy(all) >= energy(end-to-end) / (D + 1) bin by bin."""
    sc = scene_factory("old_mine", 8)
    flags = pkg._capi.FLAG_ALL_CONNECTIONS
    p_all = pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=flags)
    ctx, src = make_ctx(pkg, sc)
    e_all = ctx.compute_energy_response(src, p_all).astype(np.float64)
    e_end = ctx.compute_energy_response(src, pkg.default_params(num_rays=262144, depth=8, seed=0x5EED)).astype(np.float64)
    ctx.close()
    assert e_all.sum() > e_end.sum() > 0
    assert np.all(e_all * (1 + 1e-4) + 1e-12 >= e_end / 9.0)
    acc = np.zeros_like(e_all)
    for r in range(2):
        c, s_ = make_ctx(pkg, sc, rank=r, world_size=2)
        acc += c.compute_energy_response(s_, p_all)
        c.close()
    assert np.array_equal(acc != 0, e_all != 0) and max(rel_rms(acc[b], e_all[b]) for b in range(8)) <= TIGHT_TOL
    c, s_ = make_ctx(pkg, sc, rank=0, world_size=16)
    e_r0 = c.compute_energy_response(s_, p_all)
    c.close()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    op = oracle_mod.default_params(num_pairs=131072, depth=8, seed=0x5EED, flags=oracle_mod.FLAG_ALL_CONNECTIONS)
    e32, e64, cnt = osc.compute_energy(op, sc.source, sc.listener, pair_begin=0, pair_end=8192)
    assert cnt.connected > 8192                                   # more than one connection per pair on average
    check_energy(e_r0, e32, e64, 8)
    c, s_ = make_ctx(pkg, sc, rank=0, world_size=16)
    c.compute_energy_response(s_, p_all)
    st = c.stats()
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    c.reset_stats()
    assert c.stats()["deposits"] == 0
    c.close()


def test_mis_balance_full_size(pkg, oracle_mod, scene_factory):
    """cfg3 size with balance-heuristic weights (FS_FLAG_MIS_BALANCE): oracle parity on the first 8 192 pairs
    (rank 0 of 16), work counters equal to the oracle's, and frame-level properties: the same paths are deposited as
    with uniform weights (identical occupied bins and counters, different energies), shard invariance, and the
    deterministic accumulation mode composes with it (bit-identical across shard counts)."""
    sc = scene_factory("old_mine", 8)
    mis, allc, det = pkg._capi.FLAG_MIS_BALANCE, pkg._capi.FLAG_ALL_CONNECTIONS, pkg._capi.FLAG_DETERMINISTIC
    p_mis = pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=mis)
    ctx, src = make_ctx(pkg, sc)
    e_mis = ctx.compute_energy_response(src, p_mis).astype(np.float64)
    st_mis = ctx.stats()
    ctx.reset_stats()
    e_uni = ctx.compute_energy_response(src, pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=allc)).astype(np.float64)
    st_uni = ctx.stats()
    ctx.reset_stats()
    e_det = ctx.compute_energy_response(src, pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=mis | det))
    ctx.close()
    assert np.array_equal(e_mis != 0, e_uni != 0)
    assert all(st_mis[k] == st_uni[k] for k in ("segments", "connections_tested", "deposits"))
    assert max(rel_rms(e_mis[b], e_uni[b]) for b in range(8)) > 1e-2           # the weights do differ
    assert max(rel_rms(e_det[b].astype(np.float64), e_mis[b]) for b in range(8)) <= TIGHT_TOL
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    p_det = pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=mis | det)

    def fixed_hist(c, s_):
        c.compute_energy_response_async(s_, p_det)
        ptr, nbytes, _ = c.energy_handoff(s_)
        c.synchronize()
        h = np.zeros(8 * 1000, np.uint64)
        assert nbytes == h.nbytes and hip.hipMemcpy(h.ctypes.data, ptr, nbytes, 2) == 0
        return h

    acc = np.zeros_like(e_mis)
    acc_det = np.zeros(8 * 1000, np.uint64)
    for r in range(2):
        c, s_ = make_ctx(pkg, sc, rank=r, world_size=2)
        acc += c.compute_energy_response(s_, p_mis)
        acc_det += fixed_hist(c, s_)
        c.close()
    assert np.array_equal(acc != 0, e_mis != 0) and max(rel_rms(acc[b], e_mis[b]) for b in range(8)) <= TIGHT_TOL
    assert np.array_equal((acc_det.astype(np.float64) * 2.0 ** -40).astype(np.float32).reshape(8, 1000), e_det)
    c, s_ = make_ctx(pkg, sc, rank=0, world_size=16)
    e_r0 = c.compute_energy_response(s_, p_mis)
    st = c.stats()
    c.close()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    op = oracle_mod.default_params(num_pairs=131072, depth=8, seed=0x5EED, flags=oracle_mod.FLAG_MIS_BALANCE)
    e32, e64, cnt = osc.compute_energy(op, sc.source, sc.listener, pair_begin=0, pair_end=8192)
    check_energy(e_r0, e32, e64, 8)
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)


# ---- row f4: specular / diffuse / transmitted lobes in the walk -------------------------------------------------------
LOBE_CASES = [
    # id, scene, bands, rays, depth, lobe arrays ("seeded" / "default" / (tau, sigma) constants), extra params
    ("cfg1_half_specular", "shoebox", 1, 2048, 6, (0.1, 0.5), {}),
    ("cfg2_seeded", "starter_room", 4, 8192, 8, "seeded", {}),
    ("cfg2_no_arrays", "starter_room", 4, 4096, 8, "default", {}),
    ("mine_no_rr", "old_mine", 8, 4096, 5, "seeded", {"russian_roulette": 0, "dist_divisor": 200.0}),
    ("mirror_room", "shoebox", 2, 2048, 8, (0.0, 0.0), {"dist_divisor": 100.0}),
    ("glass_room", "starter_room", 4, 4096, 8, (1.0, 0.5), {}),
    ("all_connections", "starter_room", 4, 2048, 6, "seeded", {"flags": 16}),
    ("cosine_unbounded", "shoebox", 1, 600, 0, (0.2, 0.7), {"flags": 4}),
    ("deterministic", "starter_room", 4, 4096, 8, "seeded", {"flags": 8}),
]


def lobe_arrays(pkg, sc, spec):
    if spec == "default":
        return None, None
    if spec == "seeded":
        return pkg.scenes.material_lobes(sc)
    tau, sigma = spec
    return np.full_like(sc.absorption, tau), np.full_like(sc.absorption, sigma)


@pytest.mark.parametrize("cid,name,bands,rays,depth,spec,extra", LOBE_CASES, ids=[c[0] for c in LOBE_CASES])
def test_material_lobes_parity(pkg, oracle_mod, scene_factory, cid, name, bands, rays, depth, spec, extra):
    """FS_FLAG_MATERIAL_LOBES against the oracle: identical path set (occupied bins, segment / connection / deposit
    counters) and energies within the usual tolerance, for mixed, purely specular, strongly transmitting and absent
    Transmission / Scattering arrays, alone and combined with the other modes."""
    sc = scene_factory(name, bands)
    tau, sigma = lobe_arrays(pkg, sc, spec)
    ctx = pkg.Context(num_bands=bands)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    extra = dict(extra)
    gflags = extra.pop("flags", 0) | pkg._capi.FLAG_MATERIAL_LOBES
    oflags = (gflags & ~pkg._capi.FLAG_DETERMINISTIC)              # the oracle's 8 is its brute-force switch
    e_gpu = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=0x5EED, flags=gflags, **extra))
    st = ctx.stats()
    e_off = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=0x5EED,
                                                                flags=gflags & ~pkg._capi.FLAG_MATERIAL_LOBES, **extra))
    ctx.close()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
    op = oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=0x5EED, flags=oflags, **extra)
    e32, e64, cnt = osc.compute_energy(op, sc.source, sc.listener)
    assert cnt.connected > 0
    if gflags & pkg._capi.FLAG_DETERMINISTIC:
        for b in range(bands):
            assert rel_rms(e_gpu[b], e64[b]) <= TIGHT_TOL
        assert np.abs(e_gpu - e64).max() <= 1e-9 + TIGHT_TOL * np.abs(e64).max()
    else:
        check_energy(e_gpu, e32, e64, bands)
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    assert not np.array_equal(e_gpu, e_off)                         # the mode does change the result


def test_material_lobes_full_size(pkg, oracle_mod, scene_factory):
    """cfg3 size (262 144 rays, depth 8, 100 000 triangles) with FS_FLAG_MATERIAL_LOBES and the seeded Transmission /
    Scattering arrays: oracle parity and equal work counters on the first 8 192 pairs (rank 0 of 16 with the full
    frame's normaliser), shard invariance of the whole frame, and nothing is deposited before the direct sound."""
    sc = scene_factory("old_mine", 8)
    tau, sigma = pkg.scenes.material_lobes(sc)
    lobes = pkg._capi.FLAG_MATERIAL_LOBES
    p = pkg.default_params(num_rays=262144, depth=8, seed=0x5EED, flags=lobes)

    def ctx_for(**kw):
        c = pkg.Context(num_bands=8, **kw)
        c.set_scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
        c.set_listener(sc.listener)
        return c, c.create_source(sc.source)

    ctx, src = ctx_for()
    e_full = ctx.compute_energy_response(src, p).astype(np.float64)
    ctx.close()
    assert e_full.sum() > 0
    acc = np.zeros_like(e_full)
    for r in range(2):
        c, s_ = ctx_for(rank=r, world_size=2)
        acc += c.compute_energy_response(s_, p)
        c.close()
    assert np.array_equal(acc != 0, e_full != 0) and max(rel_rms(acc[b], e_full[b]) for b in range(8)) <= TIGHT_TOL
    d = float(np.linalg.norm(sc.listener.astype(np.float64) - sc.source.astype(np.float64))) / 1000.0
    assert np.flatnonzero(e_full.sum(axis=0)).min() >= int(np.floor(d / 343.0 * 1000.0))
    c, s_ = ctx_for(rank=0, world_size=16)
    e_r0 = c.compute_energy_response(s_, p)
    st = c.stats()
    c.close()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
    op = oracle_mod.default_params(num_pairs=131072, depth=8, seed=0x5EED, flags=oracle_mod.FLAG_MATERIAL_LOBES)
    e32, e64, cnt = osc.compute_energy(op, sc.source, sc.listener, pair_begin=0, pair_end=8192)
    assert cnt.connected > 0
    check_energy(e_r0, e32, e64, 8)
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)


def test_material_lobes_rejects_mis(pkg, scene_factory):
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.compute_energy_response(src, pkg.default_params(num_rays=64, depth=4, flags=pkg._capi.FLAG_MATERIAL_LOBES |
                                                            pkg._capi.FLAG_MIS_BALANCE))
    assert ei.value.code == pkg._capi.ERR_INVALID_ARGUMENT
    ctx.close()


def test_physical_sanity_direct_sound(pkg, scene_factory):
    """Not a parity gate ("statistical tests"): nothing can arrive before the direct sound,
    and in all-connections mode the (i = 0, j = 0) strategy deposits exactly the direct path of every pair into
    the direct-sound bin: gain 10 * min(1, 1/(4 pi d^2) * exp(-0.05 d)) with d in the path's length unit."""
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    dist_cm = float(np.linalg.norm(sc.listener.astype(np.float64) - sc.source.astype(np.float64)))
    for div in (1000.0, 100.0):                                    # reference scale (ARTS.cpp:373) and metres
        d = dist_cm / div
        direct_bin = int(np.floor(d / 343.0 * 1000.0))
        e = ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=6, seed=3, dist_divisor=div))
        occupied = np.flatnonzero(e[0])
        assert occupied.size > 0 and occupied.min() >= direct_bin
        e_all = ctx.compute_energy_response(src, pkg.default_params(num_rays=8192, depth=6, seed=3, dist_divisor=div,
                                                                    flags=pkg._capi.FLAG_ALL_CONNECTIONS))
        assert np.flatnonzero(e_all[0]).min() == direct_bin
        direct = 10.0 * min(1.0, 1.0 / (4 * np.pi * d * d) * np.exp(-0.05 * d)) if d >= 1.0 else 10.0
        # every pair contributes the direct path once with weight 1 (norm = 1 / pairs); longer paths may share the bin
        assert e_all[0, direct_bin] >= direct * (1 - 1e-5)
        assert e_all[0, :direct_bin].sum() == 0
        # the histogram decays: the second half of the occupied range holds less energy than the first
        occ = np.flatnonzero(e_all[0])
        mid = (occ.min() + occ.max()) // 2
        assert e_all[0, mid + 1:].sum() < e_all[0, :mid + 1].sum()
    ctx.close()


def test_audio_thread_reads_while_frames_are_produced(pkg, scene_factory):
    """The unguarded race of the reference (game thread rewrites ImpulseBuffer while the audio thread reads it,
    FSAC.cpp:330-378 vs RVB.cpp:136) is what the published-IR ring removes.  A reader thread hammers the lock-free
    fs_get_impulse_response pointer while the producer publishes 300 frames alternating two parameter sets; in
    deterministic mode each set has one bit-exact IR, so every read must equal one of the two (or the initial
    zeros) — a torn or half-written buffer would match neither."""
    import threading
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    det = pkg._capi.FLAG_DETERMINISTIC
    params = [pkg.default_params(num_rays=8192, depth=8, seed=s_, dist_divisor=100.0, flags=det) for s_ in (5, 6)]
    irs = []
    for p in params:
        ctx.compute_energy_response(src, p)
        ctx.reconstruct_impulse_response(src, p)
        irs.append(ctx.impulse_response(src, 0).copy())
    assert not np.array_equal(irs[0], irs[1]) and irs[0].any()
    stop = threading.Event()
    seen = {"reads": 0, "bad": 0, "a": 0, "b": 0}

    def reader():
        while not stop.is_set():
            v = ctx.impulse_response_view(src, 0).copy()           # ctypes call + memcpy, both without the GIL held long
            seen["reads"] += 1
            if np.array_equal(v, irs[0]):
                seen["a"] += 1
            elif np.array_equal(v, irs[1]):
                seen["b"] += 1
            else:
                seen["bad"] += 1

    th = threading.Thread(target=reader)
    th.start()
    try:
        for f in range(300):
            ctx.compute_energy_response_async(src, params[f & 1])
            ctx.reconstruct_impulse_response_async(src, params[f & 1])
        ctx.synchronize()
    finally:
        stop.set()
        th.join()
    assert seen["bad"] == 0 and seen["reads"] > 20 and seen["a"] > 0 and seen["b"] > 0, seen
    assert np.array_equal(ctx.impulse_response(src, 0), irs[1])     # frame 299 used the second set
    ctx.close()


# ---- deterministic (fixed-point) accumulation ------------------------------------------------------
def test_deterministic_mode_is_bit_reproducible_and_shard_invariant(pkg, oracle_mod, scene_factory):
    """FS_FLAG_DETERMINISTIC: deposits are summed as u64 counts of 2^-40 quanta.  The histogram is then (i) within
    the usual tolerance of the oracle, (ii) bit-identical from run to run (the fp32 atomics are not), and (iii)
    bit-identical for every split of the pairs over ranks once the u64 buffers are integer-summed — the quantity a
    multi-GPU all-reduce sees through fs_energy_handoff."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    sc = scene_factory("starter_room", 4)
    det = pkg._capi.FLAG_DETERMINISTIC
    p = pkg.default_params(num_rays=16384, depth=8, seed=21, flags=det)
    nb = 4 * 1000

    def fixed_hist(ctx, src):
        ctx.compute_energy_response_async(src, p)
        ptr, nbytes, _ = ctx.energy_handoff(src)
        assert nbytes == 8 * nb                                   # u64 histogram, not the fp32 buffer
        ctx.synchronize()
        h = np.zeros(nb, np.uint64)
        assert hip.hipMemcpy(h.ctypes.data, ptr, nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return h

    ctx, src = make_ctx(pkg, sc)
    runs = [ctx.compute_energy_response(src, p).copy() for _ in range(3)]
    assert all(np.array_equal(r, runs[0]) for r in runs[1:])      # (ii)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, _ = osc.compute_energy(oracle_mod.default_params(num_pairs=8192, depth=8, seed=21), sc.source, sc.listener)
    # (i) a deposit is rounded to the nearest 2^-40 (9e-13; the reconstruct ignores bins below 1e-6), so a bin
    #     may differ from the exact sum by half a quantum per deposit
    for b in range(4):
        assert rel_rms(runs[0][b], e64[b]) <= TIGHT_TOL
    assert np.abs(runs[0] - e64).max() <= 1e-9 + TIGHT_TOL * np.abs(e64).max()
    full = fixed_hist(ctx, src)
    assert np.array_equal((full.astype(np.float64) * 2.0 ** -40).astype(np.float32).reshape(4, 1000), runs[0])
    # the reconstruct of a deterministic frame uses the (reduced) fixed-point histogram
    ctx.reconstruct_impulse_response(src, p)
    mean = (runs[0].sum(axis=0, dtype=np.float32) / np.float32(4)).astype(np.float32)
    want = oracle_mod.reconstruct(mean)
    assert np.abs(ctx.impulse_response(src, 0) - want).max() <= IR_TOL * np.abs(want).max()
    ctx.close()
    for world in (2, 3, 5):                                       # (iii)
        acc = np.zeros(nb, np.uint64)
        for r in range(world):
            c, s_ = make_ctx(pkg, sc, rank=r, world_size=world)
            acc += fixed_hist(c, s_)
            c.close()
        assert np.array_equal(acc, full), world
    # the default fp32 path is untouched by the flag's existence
    ctx, src = make_ctx(pkg, sc)
    e = ctx.compute_energy_response(src, pkg.default_params(num_rays=16384, depth=8, seed=21))
    check_energy(e, e32, e64, 4)
    assert np.abs(e - runs[0]).max() <= 1e-9 + TIGHT_TOL * np.abs(e64).max()
    _, nbytes, _ = ctx.energy_handoff(src)
    assert nbytes == 4 * nb
    ctx.close()


# ---- row f4: moving geometry, device refit instead of a rebuild ------------------------------------------------
def test_moving_geometry_refit(pkg, oracle_mod, scene_factory):
    """fs_scene_update_triangles + refit: after props move (also far outside the original bounds, and back)
    every result equals what the oracle computes on a FRESH structure over the moved triangles — closest hits
    bit for bit, energies as for any frame."""
    sc = scene_factory("starter_room", 4)
    tri0 = sc.triangles.copy()
    T = tri0.shape[0]
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=8192, depth=8, seed=11)
    rng = np.random.default_rng(8)

    def check(tris, tag):
        osc = oracle_mod.Scene(tris, sc.material_ids, sc.absorption)
        e_gpu = ctx.compute_energy_response(src, p)          # a pending refit runs before the trace
        e32, e64, _ = osc.compute_energy(oracle_mod.default_params(num_pairs=4096, depth=8, seed=11), sc.source, sc.listener)
        check_energy(e_gpu, e32, e64, 4)
        n = 400
        o = (sc.source + rng.normal(0, 80, (n, 3))).astype(np.float32)
        d = rng.normal(size=(n, 3))
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
        hit, t, tri, nrm = ctx.trace_rays(o, d, 1e6)
        for i in range(n):
            h, tt, ti, nn = osc.trace_closest(o[i], d[i], 1e6, brute=True)
            assert bool(hit[i]) == h, (tag, i)
            if h:
                assert t[i] == np.float32(tt) and tri[i] == ti and np.array_equal(nrm[i], nn), (tag, i)
        return e_gpu

    e0 = check(tri0, "committed")
    # 1. a block of props slides across the room
    a, b = T - 900, T - 300
    moved = tri0.copy()
    moved[a:b] += np.array([140.0, -75.0, 12.0], np.float32)
    ctx.update_triangles(a, moved[a:b])
    e1 = check(moved, "slide")
    assert not np.array_equal(e1, e0)
    # 2. a second, overlapping update: scaled about its centroid and lifted; explicit refit call
    c, d_ = T - 1200, T - 700
    cen = moved[c:d_].reshape(-1, 3).mean(axis=0)
    moved[c:d_] = ((moved[c:d_] - cen) * np.array([1.6, 0.7, 1.2], np.float32) + cen + np.array([0, 0, 30.0], np.float32)).astype(np.float32)
    ctx.update_triangles(c, moved[c:d_])
    ctx.refit()
    check(moved, "scale")
    # 3. far outside the committed bounds: every ancestor box up to the root has to grow
    far = moved.copy()
    far[a:a + 200] += np.array([9000.0, 4000.0, 500.0], np.float32)
    ctx.update_triangles(a, far[a:a + 200])
    check(far, "far")
    # 4. everything back where it was: the committed frame again (same path set, same bins)
    ctx.update_triangles(c, tri0[c:b])
    e_back = check(tri0, "back")
    assert np.array_equal(e_back != 0, e0 != 0) and max(rel_rms(e_back[k], e0[k]) for k in range(4)) <= TIGHT_TOL
    # errors: range outside the scene, update before commit
    with pytest.raises(pkg.FrequenSeeError):
        ctx.update_triangles(T - 1, tri0[:2])
    ctx.close()


# ---- golden fixtures -------------------------------------------------------------------------------------------
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "cfg*.npz")))   # frame fixtures


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_matches_golden(pkg, scene_factory, path):
    z = np.load(path)
    extra = {str(k): float(v) for k, v in zip(z["extra_keys"], z["extra_vals"])}
    for key in ("russian_roulette", "flags"):
        if key in extra:
            extra[key] = int(extra[key])
    bands = int(z["bands"])
    sc = scene_factory(str(z["scene"]), bands)
    ctx, src = make_ctx(pkg, sc)
    if extra.pop("lobes", 0):      # fixture key: seeded Transmission / Scattering arrays on the scene (row f4)
        tau, sigma = pkg.scenes.material_lobes(sc)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
    if "source" in extra:          # fixture key: the source sits at one of the scene's extra positions (cfg5)
        src = ctx.create_source(sc.extra_sources[int(extra.pop("source"))])
    p = pkg.default_params(num_rays=2 * int(z["pairs"]), depth=int(z["depth"]), seed=int(z["seed"]), **extra)
    e = ctx.compute_energy_response(src, p)
    assert np.array_equal(e != 0, z["energy_f32"] != 0)
    for b in range(bands):
        assert rel_rms(e[b], z["energy_f64"][b]) <= TIGHT_TOL
    # IR from the golden (oracle) energy: isolates the reconstruct kernel
    c = ctx
    gold_e = np.ascontiguousarray(z["energy_f32"], dtype=np.float32)   # keep alive across the call
    c.check(c.lib.fs_update_energy_buffer(c.h, src, gold_e.ctypes.data, gold_e.size))
    c.reconstruct_impulse_response(src, p)
    for b in range(bands):
        ir = c.band_impulse_response(src, b)
        assert np.abs(ir[::16] - z["ir_decimated"][b]).max() <= IR_TOL * max(float(z["ir_max"][b]), 1e-30)
        assert abs(float(ir.astype(np.float64).sum()) - float(z["ir_sum"][b])) <= 1e-5 * max(float(z["ir_abs_sum"][b]), 1e-30)
    hit, t, tri, nrm = ctx.trace_rays(z["ray_o"], z["ray_d"], 1e6)
    assert np.array_equal(hit, z["ray_hit"].astype(bool))
    assert np.array_equal(t[hit], z["ray_t"][hit]) and np.array_equal(tri, z["ray_tri"])
    assert np.array_equal(nrm, z["ray_n"])
    ctx.close()


# ---- BASELINE.json's largest sizes through size-independent properties ----------------------------------------------
def test_full_size_properties_cfg4(pkg, oracle_mod, scene_factory):
    """cfg4 shape (1 048 576 rays, depth 12, 8 bands): linearity (shards sum to the whole), mass bound, idempotence and
    sensitivity to the seed — and, since the oracle runs on native threads (a few CPU-seconds for this frame), the whole frame
    against the oracle with its observed work counters, like cfg1 - cfg3."""
    sc = scene_factory("old_mine", 8)
    p = pkg.default_params(num_rays=1048576, depth=12)
    ctx, src = make_ctx(pkg, sc)
    ps = pkg.default_params(num_rays=1048576, depth=12, seed=0x5EED)
    e_gpu = ctx.compute_energy_response(src, ps).copy()
    st = ctx.stats()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy_mt(oracle_mod.default_params(num_pairs=524288, depth=12, seed=0x5EED), sc.source, sc.listener, 8)
    assert cnt.connected > 50000
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    assert np.array_equal(e_gpu != 0, e64 != 0)
    for b in range(8):
        assert rel_rms(e_gpu[b], e64[b]) <= TIGHT_TOL, (b, rel_rms(e_gpu[b], e64[b]))
    full = ctx.compute_energy_response(src, p)
    again = ctx.compute_energy_response(src, p)
    assert np.array_equal(full != 0, again != 0)
    assert all(rel_rms(again[b], full[b]) < 1e-5 for b in range(8))        # only the atomics' order differs
    other = ctx.compute_energy_response(src, pkg.default_params(num_rays=1048576, depth=12, seed=99))
    assert not np.array_equal(other, full)
    assert all(rel_rms(other[b], full[b]) < 0.2 for b in range(8))         # same estimator, other samples
    # every deposit is at most gain 10 / P per band; the energy is non-negative
    assert full.min() >= 0.0 and full.sum(axis=1).max() <= 10.0 + 1e-3
    ctx.close()
    acc = np.zeros(full.shape, np.float64)
    for r in range(8):                                                      # the 8-GPU sharding of cfg4
        c, s = make_ctx(pkg, sc, rank=r, world_size=8)
        acc += c.compute_energy_response(s, p)
        c.close()
    assert np.array_equal(acc != 0, full != 0)
    assert all(rel_rms(acc[b], full[b]) <= TIGHT_TOL for b in range(8))


# ---- a9: legacy forward tracer ---------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["shoebox", "starter_room", "old_mine"])
def test_update_sound_parity(pkg, oracle_mod, scene_factory, name):
    """UpdateSound / CastAudioRay / CastDirectAudioRay (FSAC.cpp:132-306): counts exact, energies to fp32 ulps."""
    sc = scene_factory(name)
    ctx = pkg.Context(num_bands=sc.num_bands)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, object_ids=sc.object_ids)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    assert ctx.occlusion_attenuation(src) == 1.0            # OcclusionAttenuation = 1.f (FSAC.h:130)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    osc.set_objects(sc.object_ids)
    for kw in ({}, {"seed": 99, "raycasts_per_tick": 777, "raycast_bounces": 4, "listener_radius": 60.0},
               {"raycast_distance": 1500.0}):
        got = ctx.update_sound(src, pkg._capi.default_sound_params(**kw))
        ref = osc.update_sound(sc.source, sc.listener, **kw)
        for k in ("rays_reaching_listener", "direct_hits", "traces"):
            assert got[k] == ref[k], (k, got, ref)
        assert got["total_energy"] == ref["total_energy"]
        assert got["occlusion_attenuation"] == pytest.approx(ref["occlusion_attenuation"], rel=2e-6, abs=0)
        assert got["direct_energy_sum"] == pytest.approx(ref["direct_energy_sum"], rel=2e-5, abs=0)
        assert ctx.occlusion_attenuation(src) == got["occlusion_attenuation"]
    # default actors (every triangle its own) must agree as well
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    osc.set_objects(None)
    got, ref = ctx.update_sound(src), osc.update_sound(sc.source, sc.listener)
    assert got["traces"] == ref["traces"] and got["direct_hits"] == ref["direct_hits"]
    assert got["occlusion_attenuation"] == pytest.approx(ref["occlusion_attenuation"], rel=2e-6, abs=0)
    ctx.close()


def test_update_sound_matches_golden(pkg, scene_factory):
    """the same against the committed fixture (no oracle at run time)"""
    import json
    cases = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "legacy_update_sound.json")))
    for c in cases:
        sc = scene_factory(c["scene"])
        ctx = pkg.Context(num_bands=sc.num_bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, object_ids=sc.object_ids)
        ctx.set_listener(sc.listener)
        src = ctx.create_source(sc.source)
        got = ctx.update_sound(src, pkg._capi.default_sound_params(**c["params"]))
        want = c["result"]
        for k in ("rays_reaching_listener", "direct_hits", "traces"):
            assert got[k] == want[k], (c["scene"], k)
        assert got["total_energy"] == want["total_energy"]
        assert got["occlusion_attenuation"] == pytest.approx(want["occlusion_attenuation"], rel=2e-6, abs=0)
        assert got["direct_energy_sum"] == pytest.approx(want["direct_energy_sum"], rel=2e-5, abs=0)
        ctx.close()


def test_component_update_sound(pkg, scene_factory):
    sc = scene_factory("starter_room", 4)
    sub = pkg.AudioRayTracingSubsystem(num_bands=4)
    sub.RegisterGeometry(sc.triangles, sc.material_ids, sc.object_ids)
    sub.SetMaterials(sc.absorption)
    comp = pkg.FrequenSeeAudioComponent(sc.source)
    comp.OnRegister(sub)
    sub.SetListenerLocation(sc.listener)
    assert comp.GetOcclusionAttenuation() == 1.0
    r = comp.UpdateSound()
    assert 0.0 < comp.GetOcclusionAttenuation() <= 1.0 and r["traces"] > 1500
    sub.Deinitialize()


# ---- C++ host side: the headless harness drives the C ABI without Python -----------------------------------------
def test_cpp_harness_matches_oracle(pkg, oracle_mod, scene_factory, tmp_path):
    import json
    import subprocess
    exe = os.path.join(os.path.dirname(pkg._capi.LIB_PATH), "fs_harness")
    out = tmp_path / "saved_ir.txt"
    r = subprocess.run([exe, "3", "512", "4", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    sc = scene_factory("shoebox", 1)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e0 = osc.compute_energy(oracle_mod.default_params(num_pairs=512, depth=4, seed=0x5EED), sc.source, sc.listener)[1]
    assert j["energy_sum_first_frame"] == pytest.approx(float(e0.sum()), rel=1e-5)
    e_last = osc.compute_energy(oracle_mod.default_params(num_pairs=512, depth=4, seed=0x5EED + 2),
                                sc.source, sc.listener)[0]
    ir_ref = oracle_mod.reconstruct(e_last[0])
    ir = pkg._capi.load_float_array(out)                      # SaveArrayToFile text: six decimals
    assert ir.shape == (48000,) and np.abs(ir - ir_ref).max() <= 6e-7 + 1e-5 * np.abs(ir_ref).max()
    assert j["ir_peak"] == pytest.approx(float(np.abs(ir_ref).max()), rel=1e-4)
    osc.set_objects(np.zeros(12, np.uint32))
    assert j["occlusion_attenuation"] == pytest.approx(osc.update_sound(sc.source, sc.listener)["occlusion_attenuation"], rel=1e-5)
    assert j["material_fd_max_err"] <= 1e-5 * max(j["ir_peak"], 1e-3)   # MaterialAcousticProcessor through the C++ mirror
    # the same run with the mirror's streamed Tick (fs_set_pipelining 2 + fs_submit per tick): the same impulse response
    out2 = tmp_path / "saved_ir_streamed.txt"
    r2 = subprocess.run([exe, "3", "512", "4", str(out2), "2"], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0, r2.stderr
    ir2 = pkg._capi.load_float_array(out2)
    assert np.abs(ir2 - ir_ref).max() <= 6e-7 + 1e-5 * np.abs(ir_ref).max()
    assert json.loads(r2.stdout)["ir_peak"] == pytest.approx(j["ir_peak"], rel=1e-5)


def test_api_state_machine_stress():
    """tools/stress.py: a random sequence of frames in all modes, moving geometry, energy helpers, installed IRs,
    reverb callbacks and stats on one context, with checkpoints against a second, synchronous context."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fs_stress", os.path.join(root, "tools", "stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(400, 7) == 0


def test_api_state_machine_stress_with_grouped_pipelined_frames(monkeypatch):
    """The same random sequence with pipelined frames, three frames per launch (fs_set_frames_per_launch) and a 12-row LDS
    traversal stack (the deep store at work): every checkpoint still equals the synchronous context's frame."""
    import importlib.util
    monkeypatch.setenv("FS_STRESS_PIPELINE", "2")
    monkeypatch.setenv("FS_STRESS_FPL", "3")
    monkeypatch.setenv("FS_STACK_ROWS_CAP", "12")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fs_stress_grouped", os.path.join(root, "tools", "stress.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(300, 11) == 0
