"""Round 3: regression tests for the advisor's findings (held batched frames across a table growth, overflowed walks in
the connect pass, the roulette bound of uncapped walks, collective-consistent retries) and the round's new entry points."""
import os

import numpy as np
import pytest

from test_gpu_parity import IR_TOL, TIGHT_TOL, check_energy, make_ctx, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
DET = 8   # FS_FLAG_DETERMINISTIC


def test_context_advice_reports_the_hardware_queue_setting(pkg, scene_factory):
    """fs_context_advice: the library no longer sets GPU_MAX_HW_QUEUES behind the host's back (a load-time setenv is too
    late in a process that already uses HIP); it reports what it found.  The Python binding — the host here — exports 16
    before the runtime starts, so a context created through it has nothing to say."""
    ctx = pkg.Context(num_bands=1)
    assert int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) >= 8
    assert ctx.advice() == ""
    ctx.close()


def test_batch_table_growth_lets_held_frames_finish_first(pkg, scene_factory):
    """Pipelining on, a batch of 8 sources is held back, then a batch of 12 sources with FEWER rays each arrives: the
    per-frame pointer tables must grow (the subpath state does not), and the held frame still reads the old block.  The
    results equal those of an unpipelined context bit for bit (deterministic mode)."""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(5)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(sc.source + rng.uniform(-0.08, 0.08, 3) * (hi - lo)).astype(np.float32) for _ in range(12)]
    out = {}
    for mode in ("plain", "pipelined"):
        ctx, _ = make_ctx(pkg, sc)
        if mode == "pipelined":
            ctx.set_pipelining(2)
        srcs = [ctx.create_source(p) for p in pos]
        p8 = pkg.default_params(num_rays=8192, depth=8, seed=11, flags=DET)
        p12 = pkg.default_params(num_rays=2048, depth=8, seed=12, flags=DET)
        ctx.compute_energy_response_batch_async(srcs[:8], p8)      # held (pipelined): tables sized for 8 sources
        ctx.compute_energy_response_batch_async(srcs[:8], p8)      # a second held frame
        ctx.compute_energy_response_batch_async(srcs, p12)         # 12 sources: the table block must grow
        ctx.synchronize()
        out[mode] = [ctx.energy_buffer(s).copy() for s in srcs]
        ctx.close()
    for a, b in zip(out["plain"], out["pipelined"]):
        assert a.any() and np.array_equal(a, b)


def test_uncapped_walks_bound_the_roulette(pkg, oracle_mod, scene_factory):
    """depth = 0 (the reference's while (true), ARTS.cpp:294) walks until the roulette ends the walk; the record store
    covers 512 steps, which a survival probability above 0.95 would leave too often: refused with a message.  At the
    bound itself the second record tier is sized from rr^64 and the frame equals the oracle's."""
    sc = scene_factory("starter_room", 2)
    ctx, src = make_ctx(pkg, sc)
    with pytest.raises(pkg.FrequenSeeError) as ei:
        ctx.compute_energy_response(src, pkg.default_params(num_rays=1024, depth=0, rr_prob=0.97))
    assert ei.value.code == pkg._capi.ERR_INVALID_ARGUMENT and "0.95" in str(ei.value)
    ctx.compute_energy_response(src, pkg.default_params(num_rays=1024, depth=32, rr_prob=0.97))   # capped: fine
    rays = 8192
    ctx.reset_stats()
    e = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=0, rr_prob=0.95, seed=4))
    st = ctx.stats()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=0, rr_prob=0.95, seed=4),
                                       sc.source, sc.listener)
    check_energy(e, e32, e64, 2)
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    assert cnt.closest_rays > 18.0 * rays            # mean walk length 19
    ctx.close()


def test_overflowed_walks_are_not_connected(pkg, oracle_mod, scene_factory, monkeypatch):
    """A walk whose record misses both tiers raises the overflow word and keeps walking; the connect pass must not read
    records that were never written (out of bounds with a one-slot tier).  The async entry point reports the overflow, and
    the energy it leaves holds no contribution of such a pair: every occupied bin is one the oracle occupies too."""
    sc = scene_factory("starter_room", 4)
    monkeypatch.setenv("FS_OVER_CAP", "1")
    ctx, src = make_ctx(pkg, sc)
    p = pkg.default_params(num_rays=65536, depth=0, seed=9)
    for flags in (0, 16):
        q = pkg.default_params(num_rays=65536 if not flags else 1024, depth=0, seed=9, flags=flags)
        ctx.compute_energy_response_async(src, q)
        try:
            ctx.synchronize()
            overflowed = False
        except pkg.FrequenSeeError as e:
            overflowed = e.code == pkg._capi.ERR_OVERFLOW
        if not flags:
            assert overflowed
            partial = ctx.energy_buffer(src)
            osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
            e32, e64, _ = osc.compute_energy_mt(oracle_mod.default_params(num_pairs=32768, depth=0, seed=9), sc.source, sc.listener, 8)
            assert not np.any((partial != 0) & (e64 == 0))
    e = ctx.compute_energy_response(src, p)       # the synchronous entry point traces again with a grown tier
    assert e.any()
    ctx.close()


# ---- staged walks: depth = 0 frames in the frame pipeline --------------------------------------------------------
def _stream(pkg, ctx, src, params):
    for p in params:
        ctx.compute_energy_response_async(src, p)
        ctx.reconstruct_impulse_response_async(src, p)


@pytest.mark.parametrize("bounds", [None, [4], [3, 9, 30, 64, 65, 100], [64], [70, 90]], ids=["default", "one", "odd", "tier_edge", "beyond_tier"])
def test_staged_uncapped_walks_equal_the_unpipelined_frames(pkg, oracle_mod, scene_factory, bounds):
    """depth = 0 frames held at pipeline depth 2 walk in stages, one stage per launch, carried from launch to launch by
    continuation records.  A stream of such frames (new seed each, a capped frame in between) leaves exactly the energy,
    IR and work counters of an unpipelined context — bit for bit in deterministic mode — whatever the stage bounds are:
    inside the main record tier, on its edge, beyond it; and the last frame equals the oracle's."""
    sc = scene_factory("starter_room", 4)
    plain, ps = make_ctx(pkg, sc)
    pipe, qs = make_ctx(pkg, sc)
    pipe.set_pipelining(2)
    if bounds is not None:
        pipe.set_walk_stages(bounds)
    for flags in (DET, 0):
        params = [pkg.default_params(num_rays=16384, depth=0, seed=500 + i, flags=flags) for i in range(11)]
        params.insert(4, pkg.default_params(num_rays=8192, depth=8, seed=77, flags=flags))     # another shape mid-stream
        plain.reset_stats(); pipe.reset_stats()
        _stream(pkg, plain, ps, params)
        _stream(pkg, pipe, qs, params)
        plain.synchronize(); pipe.synchronize()
        want_e, got_e = plain.energy_buffer(ps), pipe.energy_buffer(qs)
        want_ir, got_ir = plain.impulse_response(ps, 0), pipe.impulse_response(qs, 0)
        a, b = plain.stats(), pipe.stats()
        assert all(a[k] == b[k] for k in ("segments", "connections_tested", "deposits", "frames", "rays"))
        assert want_e.any()
        if flags & DET:
            assert np.array_equal(got_e, want_e) and np.array_equal(got_ir, want_ir)
        else:
            assert np.array_equal(got_e != 0, want_e != 0)
            assert max(rel_rms(got_e[b_], want_e[b_]) for b_ in range(4)) <= TIGHT_TOL
            assert np.abs(got_ir - want_ir).max() <= IR_TOL * np.abs(want_ir).max()
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=8192, depth=0, seed=510), sc.source, sc.listener)
    check_energy(pipe.energy_buffer(qs), e32, e64, 4)
    plain.close(); pipe.close()


def test_staged_walks_with_reads_in_between_and_batches(pkg, scene_factory):
    """Every call that observes a held frame lets it finish alone (its remaining stages one after the other): energy reads
    between the frames of a staged stream see exactly what an unpipelined context holds; batched depth = 0 frames are
    staged like single ones."""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(21)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(sc.source + rng.uniform(-0.08, 0.08, 3) * (hi - lo)).astype(np.float32) for _ in range(3)]
    out = {}
    for mode in ("plain", "pipelined"):
        ctx, _ = make_ctx(pkg, sc)
        if mode == "pipelined":
            ctx.set_pipelining(2)
        srcs = [ctx.create_source(p) for p in pos]
        seen = []
        for i in range(9):
            p = pkg.default_params(num_rays=4096, depth=0, seed=900 + i, flags=DET)
            ctx.compute_energy_response_batch_async(srcs, p)
            if i in (2, 3, 7):
                seen.append(ctx.energy_buffer(srcs[i % 3]).copy())     # flushes the pipeline mid-stream
        ctx.synchronize()
        seen += [ctx.energy_buffer(s).copy() for s in srcs]
        out[mode] = seen
        ctx.close()
    for a, b in zip(out["plain"], out["pipelined"]):
        assert a.any() and np.array_equal(a, b)


def test_walk_stage_bounds_are_validated(pkg, scene_factory):
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    for bad in ([0, 5], [5, 5], [9, 3], [600], list(range(1, 9))):
        with pytest.raises(pkg.FrequenSeeError):
            ctx.set_walk_stages(bad)
    ctx.set_walk_stages([])            # depth = 0 frames are not held: still correct
    ctx.set_pipelining(2)
    ctx.compute_energy_response_async(src, pkg.default_params(num_rays=1024, depth=0, seed=1))
    ctx.synchronize()
    assert ctx.energy_buffer(src).any()
    ctx.close()


# ---- FS_FLAG_DOUBLE_POSITIONS: node positions like the reference's FVector -----------------------------------------
DPOS = 256


@pytest.mark.parametrize("name,bands,rays,depth,offset", [
    ("starter_room", 4, 16384, 8, 0.0),
    ("old_mine", 8, 65536, 8, 0.0),              # the cfg3 scene: coordinates to +-120 m
    ("old_mine", 8, 65536, 8, 1.0e6),            # ... 10 km from the origin: fp32 positions resolve 0.06 cm there
    ("starter_room", 2, 4096, 0, 4.0e6),         # uncapped walks, 40 km out
], ids=["room", "mine", "mine_10km", "room_40km_uncapped"])
def test_double_positions_match_the_double_position_oracle(pkg, oracle_mod, scene_factory, name, bands, rays, depth, offset):
    """FS_FLAG_DOUBLE_POSITIONS carries hit points, end points, the connection and every segment length in double and
    narrows where the reference narrows (ARTS.h:61, ARTS.cpp:372-373).  Its parity reference is the oracle's
    -DFSO_DOUBLE_POSITIONS build: the same walks, the same connections (not one visibility flip), identical bins, energy to
    the atomics' rounding — also where fp32 positions visibly fail (10 and 40 km from the origin)."""
    sc = scene_factory(name, bands)
    tri = (sc.triangles.astype(np.float64) + offset).astype(np.float32)
    src_pos = (np.asarray(sc.source, np.float64) + offset).astype(np.float32)
    lis_pos = (np.asarray(sc.listener, np.float64) + offset).astype(np.float32)
    ctx = pkg.Context(num_bands=bands)
    ctx.set_scene(tri, sc.material_ids, sc.absorption)
    ctx.set_listener(lis_pos)
    src = ctx.create_source(src_pos)
    ctx.reset_stats()
    e_gpu = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=0x5EED, flags=DPOS))
    st = ctx.stats()
    dlib = oracle_mod.load_dpos()
    osc = oracle_mod.Scene(tri, sc.material_ids, sc.absorption, lib=dlib)
    e32, e64, cnt = osc.compute_energy_mt(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=0x5EED), src_pos, lis_pos, 8)
    assert cnt.connected > 0
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)   # 0 flips
    assert np.array_equal(e_gpu != 0, e64 != 0)
    worst = max(rel_rms(e_gpu[b], e64[b]) for b in range(bands))
    assert worst <= 5e-6, worst
    # and the flag is what makes the difference far from the origin: the float-position frame there is another frame
    if offset >= 1.0e6:
        e_f32 = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=0x5EED))
        assert not np.array_equal(e_f32, e_gpu)
    with pytest.raises(pkg.FrequenSeeError):
        ctx.compute_energy_response(src, pkg.default_params(num_rays=1024, depth=4, flags=DPOS | 16))
    ctx.close()


# ---- the end points' collision spheres (HEAD's ECC_Pawn traces) ------------------------------------------
@pytest.mark.parametrize("name,bands,rays,depth,lr,sr,flags", [
    ("shoebox", 1, 2048, 6, 34.0, 0.0, 0),
    ("starter_room", 4, 16384, 8, 34.0, 20.0, 0),
    ("starter_room", 2, 4096, 0, 60.0, 60.0, 0),       # uncapped walks, big spheres: many walks land on them
    ("old_mine", 8, 32768, 8, 34.0, 34.0, 8),          # deterministic deposits
    ("starter_room", 2, 2048, 5, 34.0, 25.0, 16),      # every forward prefix x every backward prefix
    ("starter_room", 4, 8192, 8, 34.0, 20.0, 256),     # together with double positions
], ids=["listener_only", "both", "uncapped_big", "mine_det", "all_connections", "with_double_positions"])
def test_end_point_collision_spheres(pkg, oracle_mod, scene_factory, name, bands, rays, depth, lr, sr, flags):
    """At HEAD GeneratePath's and ConnectSubpaths' traces include ECC_Pawn: a walk can land on the OTHER end point's
    collision (and goes on from there, without a material), and every connection that starts or ends inside a sphere is
    blocked.  fs_params.listener_radius / source_radius reproduce that (0 = points, the default): same walks, same
    connections, same bins as the oracle; and the frame differs from the one without spheres."""
    sc = scene_factory(name, bands)
    ctx, src = make_ctx(pkg, sc)
    ctx.reset_stats()
    e_gpu = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=31, flags=flags,
                                                                listener_radius=lr, source_radius=sr))
    st = ctx.stats()
    lib = oracle_mod.load_dpos() if flags & 256 else None
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption, lib=lib)
    op = oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=31, flags=flags & ~(8 | 256), listener_radius=lr, source_radius=sr)
    e32, e64, cnt = osc.compute_energy_mt(op, sc.source, sc.listener, 8)
    assert (st["segments"], st["connections_tested"], st["deposits"]) == (cnt.closest_rays, cnt.any_rays, cnt.connected)
    assert np.array_equal(e_gpu != 0, e64 != 0)
    if cnt.connected:
        assert max(rel_rms(e_gpu[b], e64[b]) for b in range(bands)) <= TIGHT_TOL
    plain = ctx.compute_energy_response(src, pkg.default_params(num_rays=rays, depth=depth, seed=31, flags=flags))
    assert not np.array_equal(plain, e_gpu)
    ctx.close()


def test_collision_spheres_in_batches_and_pipelines(pkg, scene_factory):
    """Frames with spheres are never held by the frame pipeline and batch like any other frame (every source its own
    sphere): equal to the separate, unpipelined frames bit for bit in deterministic mode."""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(8)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(sc.source + rng.uniform(-0.08, 0.08, 3) * (hi - lo)).astype(np.float32) for _ in range(3)]
    p = pkg.default_params(num_rays=4096, depth=8, seed=5, flags=DET, listener_radius=34.0, source_radius=30.0)
    ctx, _ = make_ctx(pkg, sc)
    srcs = [ctx.create_source(q) for q in pos]
    want = [ctx.compute_energy_response(s, p).copy() for s in srcs]
    ctx.set_pipelining(2)
    ctx.compute_energy_response_batch_async(srcs, p)
    ctx.compute_energy_response_batch_async(srcs, p)
    ctx.synchronize()
    for s, w in zip(srcs, want):
        assert w.any() and np.array_equal(ctx.energy_buffer(s), w)
    ctx.close()


@pytest.mark.parametrize("cap", [12, 14, 21])
def test_bounded_lds_stack_spills_to_the_deep_store(pkg, scene_factory, monkeypatch, cap):
    """The LDS traversal stack holds FS_STACK_ROWS_CAP rows (default 21) when the tree's worst case needs more; a lane
    that fills them moves its oldest entries to the deep store in HBM and takes them back later (fs_device.hpp:
    trav_maintain).  With 12 rows on the mine's tree (worst case 30+) that happens all the time: closest
    hits, any hits, integer energy sums and the pipelined frames must not change by a bit."""
    sc = scene_factory("old_mine", 8)
    rng = np.random.default_rng(77)
    n = 50000
    o = np.tile(np.asarray(sc.source, np.float32), (n, 1)) + rng.normal(scale=5.0, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out = {}
    for mode in ("worst_case_rows", "capped"):
        if mode == "capped":
            monkeypatch.setenv("FS_STACK_ROWS_CAP", str(cap))
        else:
            monkeypatch.setenv("FS_STACK_ROWS_CAP", "64")
        ctx, src = make_ctx(pkg, sc)
        need = ctx.stats()["bvh_stack_need"]
        assert need + 1 > cap + 1, "the scene must need more rows than the cap for this test to mean anything"
        hit, t, tri, nrm = ctx.trace_rays(o, d, 1e6)
        anyhit = ctx.trace_rays(o, d, 3000.0, any_hit=True)[0]
        p = pkg.default_params(num_rays=65536, depth=8, seed=4242, flags=DET)
        e_plain = ctx.compute_energy_response(src, p).copy()
        p0 = pkg.default_params(num_rays=32768, depth=0, seed=4243, flags=DET)
        e_unbounded = ctx.compute_energy_response(src, p0).copy()
        pbig = pkg.default_params(num_rays=262144, depth=8, seed=4244, flags=DET)
        e_big_plain = ctx.compute_energy_response(src, pbig).copy()
        ctx.set_pipelining(2)
        for _ in range(4):
            ctx.compute_energy_response_async(src, p)      # a small launch: the wide flavour of the frame kernel (fs_frame.hip)
        ctx.synchronize()
        e_piped = ctx.energy_buffer(src).copy()
        for _ in range(3):
            ctx.compute_energy_response_async(src, pbig)   # 1024+ workgroups: the 128-VGPR flavour with the bounded stack
        ctx.synchronize()
        e_big_piped = ctx.energy_buffer(src).copy()
        assert np.array_equal(e_big_plain, e_big_piped)
        ctx.set_pipelining(0)
        p0big = pkg.default_params(num_rays=262144, depth=0, seed=4245, flags=DET)
        e_staged_plain = ctx.compute_energy_response(src, p0big).copy()
        ctx.set_pipelining(2)
        ctx.set_frames_per_launch(2)
        for _ in range(4):
            ctx.compute_energy_response_async(src, p0big)  # staged walks, two frames per launch, through the same kernel
        ctx.synchronize()
        assert np.array_equal(e_staged_plain, ctx.energy_buffer(src))
        out[mode] = (hit.copy(), t.copy(), tri.copy(), nrm.copy(), anyhit.copy(), e_plain, e_unbounded, e_piped, e_big_piped, e_staged_plain)
        ctx.close()
    assert out["capped"][0].any() and out["capped"][5].any() and out["capped"][6].any()
    for a, b in zip(out["worst_case_rows"], out["capped"]):
        assert np.array_equal(a, b)
    assert np.array_equal(out["capped"][5], out["capped"][7])     # the pipelined frames are the same frame four times


def test_staged_walks_of_a_million_ray_frame(pkg, scene_factory):
    """A stage that begins beyond step 64 visits every walk of the last schedule bucket (64 steps or more), not only the
    rr^begin that are still alive: with 917 504 rays that is 1 082 slots against the 1 072 a launch used to provision —
    every such frame ended in FS_ERR_OVERFLOW (found by tools/stage_sizes.py)."""
    sc = scene_factory("old_mine", 8)
    out = {}
    for depth in (0, 2):
        ctx, src = make_ctx(pkg, sc)
        ctx.set_pipelining(depth)
        p = pkg.default_params(num_rays=917504, depth=0, seed=31, flags=DET)
        for k in range(3):
            p.seed = 31 + k
            ctx.compute_energy_response_async(src, p)
        ctx.synchronize()                                   # used to raise status 9
        out[depth] = ctx.energy_buffer(src).copy()
        ctx.close()
    assert out[0].any() and np.array_equal(out[0], out[2])


@pytest.mark.parametrize("n", [2, 3, 4])
def test_frames_per_launch_gives_the_single_frames_results(pkg, scene_factory, n):
    """fs_set_frames_per_launch(n): plain pipelinable frames wait until n have come and are traced as ONE batched frame in
    which every item keeps its own seed, energy buffer and recorded reconstruct — also the same source several times.
    Energies (deterministic mode: bit for bit), published IRs, counters and the frame count must be those of the same calls
    with one frame per launch; reads in between send a partial group off."""
    sc = scene_factory("starter_room", 4)
    out = {}
    for per_launch in (1, n):
        ctx, s = make_ctx(pkg, sc)
        s2 = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(25.0))
        ctx.set_pipelining(2)
        ctx.set_frames_per_launch(per_launch)
        p = pkg.default_params(num_rays=16384, depth=8, flags=DET)
        got = []
        for i in range(11):
            p.seed = 900 + i
            src = s2 if i % 4 == 1 else s
            ctx.compute_energy_response_async(src, p)
            ctx.reconstruct_impulse_response_async(src, p)
            if i in (3, 8):                                   # an observer in the middle of a group
                got.append(ctx.energy_buffer(src).copy())
                ctx.synchronize()
                got.append(ctx.impulse_response(src, 0).copy())
        ctx.synchronize()
        for src in (s, s2):
            got.append(ctx.energy_buffer(src).copy())
            got.append(ctx.impulse_response(src, 0).copy())
        st = ctx.stats()
        got.append(np.asarray([st["frames"], st["segments"], st["connections_tested"], st["deposits"]], np.int64))
        out[per_launch] = got
        ctx.close()
    assert out[1][0].any() and np.abs(out[1][1]).max() > 0
    for a, b in zip(out[1], out[n]):
        assert np.array_equal(a, b)


def test_frames_per_launch_with_uncapped_walks_other_kinds_and_seed_words(pkg, scene_factory):
    """Groups of depth = 0 frames go through the staged walks; a frame of another kind (other parameters, a lobes frame, a
    different high seed word) sends the waiting ones off first and goes its own way; fs_submit and
    fs_set_frames_per_launch flush."""
    sc = scene_factory("starter_room", 4)
    tr, scat = pkg.scenes.material_lobes(sc)
    out = {}
    for per_launch in (1, 3):
        ctx = pkg.Context(num_bands=sc.num_bands)
        ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tr, scattering=scat)
        ctx.set_listener(sc.listener)
        s = ctx.create_source(sc.source)
        ctx.set_pipelining(2)
        ctx.set_frames_per_launch(per_launch)
        p0 = pkg.default_params(num_rays=8192, depth=0, flags=DET)
        p8 = pkg.default_params(num_rays=4096, depth=8, flags=DET)
        pl = pkg.default_params(num_rays=4096, depth=8, flags=DET | pkg._capi.FLAG_MATERIAL_LOBES)
        for i, p in enumerate([p0, p0, p8, p0, pl, p8, p8, p8, p0]):
            p.seed = (7 << 32) + 40 + i if i == 6 else 40 + i
            ctx.compute_energy_response_async(s, p)
            if i == 5:
                ctx.submit()
        ctx.set_frames_per_launch(1)
        out[per_launch] = [ctx.energy_buffer(s).copy(), np.asarray([ctx.stats()[k] for k in ("frames", "segments", "deposits")], np.int64)]
        ctx.close()
    assert out[1][0].any()
    for a, b in zip(out[1], out[3]):
        assert np.array_equal(a, b)


def test_frames_per_launch_keeps_the_positions_of_each_call(pkg, scene_factory):
    """A frame that waits in the group is traced with the source and listener positions of ITS call, not those at the time the
    group is sent off: a moving source stays in the group (per-item positions), a moving listener sends the group off (a
    batched frame has one listener)."""
    sc = scene_factory("starter_room", 4)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    out = {}
    for per_launch in (1, 3):
        ctx, s = make_ctx(pkg, sc)
        ctx.set_pipelining(2)
        ctx.set_frames_per_launch(per_launch)
        p = pkg.default_params(num_rays=8192, depth=8, flags=DET)
        got = []
        for i in range(7):
            p.seed = 70 + i
            ctx.set_source_position(s, np.asarray(sc.source, np.float32) + np.float32(3.0 * i) * np.asarray([1, 0.5, 0], np.float32))
            if i in (3, 4):
                ctx.set_listener(np.asarray(sc.listener, np.float32) + np.float32(10.0 * (i - 2)))
            ctx.compute_energy_response_async(s, p)
            ctx.reconstruct_impulse_response_async(s, p)
            if i in (1, 4):   # moved away again before the group goes: the waiting frames must not see this
                ctx.set_source_position(s, (0.5 * (lo + hi)).astype(np.float32))
        ctx.synchronize()
        out[per_launch] = [ctx.energy_buffer(s).copy(), ctx.impulse_response(s, 0).copy(),
                           np.asarray([ctx.stats()[k] for k in ("segments", "connections_tested", "deposits")], np.int64)]
        ctx.close()
    assert out[1][0].any()
    for a, b in zip(out[1], out[3]):
        assert np.array_equal(a, b)


def test_reverb_hears_the_irs_of_grouped_pipelined_frames(pkg, scene_factory):
    """The reverb callback convolves with the source's DEVICE-resident IR (fs_reverb_process).  With pipelined frames, several
    per launch and their reconstructs riding in the launches (fs_frame.hip: reconstruct parts; the frames of one source
    but the last write temporaries), that IR must at every callback be the one of the last frame issued — as it is with
    plain frames.  Callbacks are interleaved with the stream; every block must equal the plain context's block bit for bit."""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(12)
    blocks = [np.clip(rng.normal(0, 0.2, 2048), -1, 1).astype(np.float32) for _ in range(6)]
    out = {}
    for mode in ("plain", "grouped"):
        ctx, s = make_ctx(pkg, sc)
        ctx.reverb_init(s, 1024)
        if mode == "grouped":
            ctx.set_pipelining(2)
            ctx.set_frames_per_launch(3)
        p = pkg.default_params(num_rays=8192, depth=8, flags=DET, dist_divisor=100.0)
        got, k = [], 0
        for i in range(14):
            p.seed = 300 + i
            ctx.compute_energy_response_async(s, p)
            ctx.reconstruct_impulse_response_async(s, p)
            if i in (0, 3, 4, 8, 12, 13):
                ctx.synchronize()                       # the IR of frame i is the source's IR now
                got.append(np.asarray(ctx.reverb_process(s, blocks[k])).copy())
                k += 1
        ctx.synchronize()
        got.append(ctx.impulse_response(s, 0).copy())
        out[mode] = got
        ctx.close()
    assert max(np.abs(a).max() for a in out["plain"][:-1]) > 0          # (the first block may precede the first arrival)
    for a, b in zip(out["plain"], out["grouped"]):
        assert np.array_equal(a, b)


def test_headline_launch_shape_with_the_connect_part_first(pkg, scene_factory):
    """A launch whose walks need the chip's workgroup slots twice over (two 262 144-ray frames of roulette walks: 2 048 walk
    workgroups) puts 384 connect workgroups FIRST in the grid and lets them step through all the pairs (fs_frame.hip,
    FS_LAUNCH_FRAME).  Energies (deterministic mode: bit for bit), IRs and counters must be those of unpipelined frames."""
    sc = scene_factory("old_mine", 8)
    out = {}
    for mode in ("plain", "grouped"):
        ctx, s = make_ctx(pkg, sc)
        if mode == "grouped":
            ctx.set_pipelining(2)
            ctx.set_frames_per_launch(2)
        p = pkg.default_params(num_rays=262144, depth=8, flags=DET)
        got = []
        for i in range(8):
            p.seed = 4100 + i
            ctx.compute_energy_response_async(s, p)
            ctx.reconstruct_impulse_response_async(s, p)
            if i == 5:
                ctx.synchronize()
                got.append(ctx.energy_buffer(s).copy())
                got.append(ctx.impulse_response(s, 0).copy())
        ctx.synchronize()
        got.append(ctx.energy_buffer(s).copy())
        got.append(ctx.impulse_response(s, 0).copy())
        st = ctx.stats()
        got.append(np.asarray([st["frames"], st["segments"], st["connections_tested"], st["deposits"]], np.int64))
        out[mode] = got
        ctx.close()
    assert out["plain"][0].any()
    for a, b in zip(out["plain"], out["grouped"]):
        assert np.array_equal(a, b)


def test_batched_sources_with_the_connect_part_first(pkg, scene_factory):
    """The same launch shape through the batched frame (cfg5: several sources' pairs end to end, one energy buffer each):
    8 sources x 65 536 subpaths are 2 048 walk workgroups, so the 384 connect workgroups at the head of the grid step
    through (source, chunk) items and flush their LDS histogram at every change of source (connect_body, BATCH).  Energies
    of pipelined batches, bit for bit those of the sources traced one by one without pipelining."""
    sc = scene_factory("old_mine", 8)
    pos = [np.asarray(sc.source, np.float32) + np.asarray([40.0 * k, -25.0 * k, 6.0 * k], np.float32) for k in range(8)]
    out = {}
    for mode in ("one by one", "batched"):
        ctx, s0 = make_ctx(pkg, sc)
        srcs = [ctx.create_source(q) for q in pos]
        p = pkg.default_params(num_rays=65536, depth=8, flags=DET)
        if mode == "batched":
            ctx.set_pipelining(2)
        for i in range(4):
            p.seed = 5200 + i
            if mode == "batched":
                ctx.compute_energy_response_batch_async(srcs, p)
            else:
                for s in srcs:
                    ctx.compute_energy_response_async(s, p)
        ctx.synchronize()
        out[mode] = [ctx.energy_buffer(s).copy() for s in srcs]
        ctx.close()
    assert all(e.any() for e in out["one by one"])
    assert len({e.tobytes() for e in out["one by one"]}) == 8          # eight different sources
    for a, b in zip(out["one by one"], out["batched"]):
        assert np.array_equal(a, b)


# ---- round 4: ADVICE r3 -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("per_launch,bounds", [(4, None), (3, [6, 12, 20, 30, 44, 64, 96]), (2, [6, 12, 20, 30, 44, 64, 96])],
                         ids=["four_default_stages", "three_with_eight_stages", "two_with_eight_stages"])
def test_long_streams_of_staged_frames_never_share_an_energy_buffer(pkg, scene_factory, per_launch, bounds):
    """a staged depth = 0 frame owns its energy buffer from the launch that plans it to the launch that
    reconstructs it — stages + 3 launches.  Four such frames of ONE source per launch wrapped round 3's 24-buffer rotation
    after six launches: the plan part of a launch zeroed buffers its connect part was depositing an older frame into, and
    the next launch reconstructed the mix.  The tests of round 3 streamed at most 11 grouped frames.  Here 56 frames of one
    source stream through; every IR the stream publishes is picked up by its publish number and must be the IR of that frame
    from a context that traces one frame at a time — bit for bit (deterministic mode); at least every other launch must be seen
    (the collision hit all frames of a launch)."""
    sc = scene_factory("starter_room", 4)
    N = 56
    params = [pkg.default_params(num_rays=8192, depth=0, seed=7000 + i, flags=DET) for i in range(N)]
    ref, rs = make_ctx(pkg, sc)
    want_ir, want_e = [], []
    for p in params:
        ref.compute_energy_response_async(rs, p)
        ref.reconstruct_impulse_response_async(rs, p)
        ref.synchronize()
        want_ir.append(ref.impulse_response(rs, 0).copy())
        want_e.append(ref.energy_buffer(rs).copy())
    ref.close()
    assert len({a.tobytes() for a in want_ir}) == N      # every frame has an IR of its own
    ctx, s = make_ctx(pkg, sc)
    ctx.set_pipelining(2)
    ctx.set_frames_per_launch(per_launch)
    if bounds is not None:
        ctx.set_walk_stages(bounds)
    seen = {}

    def pick_up():
        k0 = ctx.impulse_response_sequence(s)
        if k0 == 0 or k0 in seen:
            return
        a = ctx.impulse_response_view(s, 0).copy()
        k1 = ctx.impulse_response_sequence(s)
        if k1 == k0:                                       # nothing was published meanwhile: the copy is publish k0
            seen[k0] = a

    for p in params:
        ctx.compute_energy_response_async(s, p)
        ctx.reconstruct_impulse_response_async(s, p)
        pick_up()
    ctx.synchronize()
    pick_up()
    assert ctx.impulse_response_sequence(s) == N
    assert np.array_equal(ctx.energy_buffer(s), want_e[-1])
    # a launch's IRs complete together (one of them is in front when the producer looks), and the first IR appears
    # stages + 3 launches after its call: at least half of the launches that publish while the stream runs must be seen
    stages = (len(bounds) if bounds is not None else 4) + 1
    assert len(seen) >= max(3, (N // per_launch - (stages + 2)) // 2), sorted(seen)
    bad = [k for k, a in sorted(seen.items()) if not np.array_equal(a, want_ir[k - 1])]
    assert not bad, f"published IRs {bad} are not those of their frames (seen: {sorted(seen)})"
    ctx.close()
