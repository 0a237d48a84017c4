"""Round 4: the cooperative traversal of small frames (a whole group of lanes searches one ray), the reference's tick,
staged walks of frames that are waited for, observed work counters, the walker's own actor."""
import os

import numpy as np
import pytest

from test_gpu_parity import IR_TOL, TIGHT_TOL, check_energy, make_ctx, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
DET = 8   # FS_FLAG_DETERMINISTIC


# ---- cooperative traversal ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("build", ["host_sah", "device_morton"])
@pytest.mark.parametrize("name", ["shoebox", "starter_room", "old_mine"])
def test_cooperative_line_trace_matches_brute_force(pkg, oracle_mod, scene_factory, name, build):
    """trav_coop — the G = 64 / R lanes of a group search one ray breadth first — must return the closest hit of the
    oracle's BRUTE-FORCE scan (t bit for bit, the triangle, the normal) for R = 1, 2, 4 rays per wave, on both builders'
    trees, and exactly what the lane-private traversal returns."""
    sc = scene_factory(name)
    ctx, _ = make_ctx(pkg, sc, fast=build == "device_morton")
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    rng = np.random.default_rng(11)
    n = 1500 if name != "old_mine" else 500
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    o = np.where(rng.random((n, 1)) < 0.5, sc.source + rng.normal(0, 60, (n, 3)), rng.uniform(lo, hi, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tm = np.where(rng.random(n) < 0.3, rng.uniform(50, 900, n), 1e6).astype(np.float32)    # short segments too
    ref = ctx.trace_rays(o, d, tm)
    for i in range(0, n, 7):     # the brute-force scan is slow: a sample
        h, t, tri, _nrm = osc.trace_closest(o[i], d[i], float(tm[i]), brute=True)
        assert bool(ref[0][i]) == bool(h)
        if h:
            assert ref[1][i] == np.float32(t) and ref[2][i] == tri
    for mode in (2, 3, 4, 5, 6, 7, 8):   # 1 / 2 / 4 rays per wave; records from memory, the top of the tree in LDS, all that fits in LDS
        got = ctx.trace_rays(o, d, tm, any_hit=mode)
        assert np.array_equal(got[0], ref[0]), mode
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), mode
    assert ref[0].sum() > n // 3
    ctx.close()


@pytest.mark.parametrize("leaf", ["1", "4"])
def test_cooperative_line_trace_with_other_leaf_sizes(pkg, scene_factory, monkeypatch, leaf):
    """leaves of one triangle, and of up to four (FS_BVH_LEAF): a lane requests the first two triangles of its hit leaf
    when it finds it and fetches the others when it tests them"""
    monkeypatch.setenv("FS_BVH_LEAF", leaf)
    sc = scene_factory("starter_room")
    ctx, _ = make_ctx(pkg, sc)
    rng = np.random.default_rng(5)
    n = 1200
    o = (sc.source + rng.normal(0, 80, (n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    ref = ctx.trace_rays(o, d, 1e6)
    for mode in (2, 3, 4, 6, 8):
        got = ctx.trace_rays(o, d, 1e6, any_hit=mode)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), mode
    ctx.close()


@pytest.mark.parametrize("name,bands,rays,depth", [("shoebox", 1, 1024, 4), ("starter_room", 4, 2000, 0), ("starter_room", 4, 6000, 8),
                                                   ("old_mine", 8, 2000, 0), ("old_mine", 8, 4096, 12)])
@pytest.mark.parametrize("rpw", [1, 2, 4])
def test_cooperative_walk_equals_the_sparse_walk_and_the_oracle(pkg, oracle_mod, scene_factory, monkeypatch, name, bands, rays, depth, rpw):
    """walk_kernel_coop (FS_WALK_COOP, default on for 1 / 2 / 4 subpaths per wave) against the sparse-wave kernel it
    replaces there: the same energies bit for bit in deterministic mode, the same work counters; and against the oracle."""
    sc = scene_factory(name, bands)
    out = {}
    for coop in ("0", "1"):
        monkeypatch.setenv("FS_WALK_COOP", coop)
        monkeypatch.setenv("FS_WALK_RAYS_PER_WAVE", str(rpw))
        ctx, s = make_ctx(pkg, sc)
        e = ctx.compute_energy_response(s, pkg.default_params(num_rays=rays, depth=depth, seed=321, flags=DET))
        st = ctx.stats()
        e2 = ctx.compute_energy_response(s, pkg.default_params(num_rays=rays, depth=depth, seed=321))
        out[coop] = (e.copy(), [st[k] for k in ("segments", "connections_tested", "deposits")], e2.copy())
        ctx.close()
    assert out["0"][0].any()
    assert np.array_equal(out["0"][0], out["1"][0]) and out["0"][1] == out["1"][1]
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=321), sc.source, sc.listener)
    check_energy(out["1"][2], e32, e64, bands)
    assert out["1"][1][2] == cnt.connected



# ---- the tick's reconstructs as one launch ------------------------------------------------------------------------
@pytest.mark.parametrize("flags", [0, DET])
def test_batched_reconstructs_equal_the_single_ones(pkg, scene_factory, flags):
    """fs_reconstruct_impulse_response_batch_async: one launch for every source of the tick, the channel views written
    straight into the published host buffers, one completion event.  Published IRs, per-band IRs and the publish counters
    must be those of one fs_reconstruct_impulse_response_async per source — over more ticks than the IR ring and the table
    ring have slots, mixed with single calls, with an observer between the ticks."""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(2)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(np.asarray(sc.source, np.float32) + rng.uniform(-0.05, 0.05, 3).astype(np.float32) * (hi - lo)).astype(np.float32) for _ in range(9)]
    out = {}
    for mode in ("single", "batch"):
        ctx, _ = make_ctx(pkg, sc)
        srcs = [ctx.create_source(p) for p in pos]
        got = []
        for tick in range(21):
            p = pkg.default_params(num_rays=2000, depth=0, seed=100 + tick, flags=flags | pkg._capi.FLAG_FIXED_NORM_1000)
            live = srcs if tick % 5 else srcs[:4]                     # the set of sources changes
            ctx.compute_energy_response_batch_async(live, p)
            if mode == "batch" and tick % 7 != 3:
                ctx.reconstruct_impulse_response_batch_async(live, p)
            else:
                for s in live:
                    ctx.reconstruct_impulse_response_async(s, p)
            if tick in (2, 11):                                       # an observer in between
                ctx.synchronize()
                got.append(ctx.impulse_response(srcs[1], 0).copy())
                got.append(ctx.band_impulse_response(srcs[2], 1).copy())
        ctx.synchronize()
        for s in srcs:
            got.append(ctx.impulse_response(s, 0).copy())
            got.append(ctx.band_impulse_response(s, 3).copy())
            got.append(np.asarray([ctx.impulse_response_sequence(s)], np.int64))
        out[mode] = got
        ctx.close()
    assert np.abs(out["single"][0]).max() > 0
    for a, b in zip(out["single"], out["batch"]):
        if flags & DET or a.dtype != np.float32:
            assert np.array_equal(a, b)
        else:     # fp32 atomics: the energies of two runs agree to rounding, and so do the IRs
            assert np.abs(a - b).max() <= IR_TOL * max(np.abs(a).max(), 1e-30)


def test_batched_reconstructs_with_reverb_and_pipelined_frames(pkg, scene_factory):
    """the reverb callback reads the device-resident IR behind a batched reconstruct WITHOUT a synchronize in between (its
    wait is the batch's event): the blocks equal those behind single reconstructs; and a batch call while frames are held
    back records every reconstruct with its frame, like the single call"""
    sc = scene_factory("starter_room", 2)
    rng = np.random.default_rng(4)
    blocks = [np.clip(rng.normal(0, 0.2, 2048), -1, 1).astype(np.float32) for _ in range(5)]
    out = {}
    for mode in ("single", "batch"):
        ctx, s0 = make_ctx(pkg, sc)
        s1 = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(30.0))
        ctx.reverb_init(s0, 1024)
        got = []
        for i in range(5):
            p = pkg.default_params(num_rays=4096, depth=8, seed=50 + i, flags=DET, dist_divisor=100.0)
            ctx.compute_energy_response_batch_async([s0, s1], p)
            if mode == "batch":
                ctx.reconstruct_impulse_response_batch_async([s0, s1], p)
            else:
                ctx.reconstruct_impulse_response_async(s0, p)
                ctx.reconstruct_impulse_response_async(s1, p)
            got.append(np.asarray(ctx.reverb_process(s0, blocks[i])).copy())   # no synchronize: ordered by the events alone
        ctx.synchronize()
        got += [ctx.impulse_response(s, 0).copy() for s in (s0, s1)]
        ctx.set_pipelining(2)
        p = pkg.default_params(num_rays=4096, depth=8, seed=99, flags=DET)
        ctx.compute_energy_response_async(s0, p)
        ctx.compute_energy_response_async(s1, p)
        if mode == "batch":
            ctx.reconstruct_impulse_response_batch_async([s0, s1], p)     # both frames are held: recorded with them
        else:
            ctx.reconstruct_impulse_response_async(s0, p)
            ctx.reconstruct_impulse_response_async(s1, p)
        ctx.synchronize()
        got += [ctx.impulse_response(s, 0).copy() for s in (s0, s1)]
        out[mode] = got
        ctx.close()
    assert max(np.abs(a).max() for a in out["single"]) > 0
    for a, b in zip(out["single"], out["batch"]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("depth", [0, 6])
def test_update_sources_equals_the_three_calls(pkg, scene_factory, depth):
    """fs_update_sources (UpdateSources, ARTS.cpp:100-126) = the batched frame + the batched reconstruct + the wait, with the
    reconstruct on the compute stream: the published IRs, per-band IRs, reverb blocks and publish counters over 20 ticks are
    those of the three calls — also when asynchronous ticks (their reconstructs on the tail stream, not waited for) and a
    tick of held-back pipelined frames sit between two fs_update_sources ticks"""
    sc = scene_factory("starter_room", 4)
    rng = np.random.default_rng(6)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(np.asarray(sc.source, np.float32) + rng.uniform(-0.05, 0.05, 3).astype(np.float32) * (hi - lo)).astype(np.float32) for _ in range(5)]
    block = np.clip(rng.normal(0, 0.2, 2048), -1, 1).astype(np.float32)
    out = {}
    for mode in ("three", "one"):
        ctx, _ = make_ctx(pkg, sc)
        srcs = [ctx.create_source(p) for p in pos]
        ctx.reverb_init(srcs[0], 1024)
        got = []
        for tick in range(20):
            p = pkg.default_params(num_rays=3000, depth=depth, seed=300 + tick, flags=DET | pkg._capi.FLAG_FIXED_NORM_1000)
            live = srcs if tick % 4 else srcs[:2]
            if tick == 13:
                ctx.set_pipelining(2)                                  # the frames of this tick are held back when the tick begins
            if mode == "one" and tick % 3 != 1:
                ctx.update_sources(live, p)
            else:
                ctx.compute_energy_response_batch_async(live, p)
                ctx.reconstruct_impulse_response_batch_async(live, p)
                if mode == "three" or tick % 6 == 1:                   # (mode "one": some ticks are left in flight)
                    ctx.synchronize()
            if tick == 13:
                ctx.set_pipelining(1)
            if tick % 5 == 2:
                got.append(np.asarray(ctx.reverb_process(srcs[0], block)).copy())
        ctx.synchronize()
        for s in srcs:
            got.append(ctx.impulse_response(s, 0).copy())
            got.append(ctx.band_impulse_response(s, 2).copy())
            got.append(np.asarray([ctx.impulse_response_sequence(s)], np.int64))
        out[mode] = got
        ctx.close()
    assert np.abs(out["three"][0]).max() > 0
    for a, b in zip(out["three"], out["one"]):
        assert np.array_equal(a, b)


# ---- staged walks of frames that are waited for -------------------------------------------------------------------
@pytest.mark.parametrize("bounds", ["16", "5,9,70", "64", "100"])
def test_staged_walks_of_waited_frames_equal_the_walk_in_one_piece(pkg, oracle_mod, scene_factory, monkeypatch, bounds):
    """A depth = 0 frame that is waited for (no pipelining) walks in stages too, one launch after the other on the stream —
    everybody on dense waves first, the survivors on cooperative waves (fs_capi_frame.cpp: frame_launch).  Energies
    (deterministic mode: bit for bit), counters and IRs are those of the walk in one piece, for bounds inside the main
    record tier, on its edge and beyond it; batched sources too; and the frame equals the oracle's."""
    sc = scene_factory("starter_room", 4)
    out = {}
    for mode in ("whole", "staged"):
        monkeypatch.setenv("FS_SYNC_WALK_STAGES", bounds if mode == "staged" else "")
        monkeypatch.setenv("FS_SYNC_STAGE_FROM", "1")
        ctx, s = make_ctx(pkg, sc)
        s2 = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(40.0))
        p = pkg.default_params(num_rays=24576, depth=0, seed=77, flags=DET)
        e = ctx.compute_energy_response(s, p).copy()
        ctx.reconstruct_impulse_response(s, p)
        got = [e, ctx.impulse_response(s, 0).copy()]
        ctx.compute_energy_response_batch_async([s, s2], pkg.default_params(num_rays=6000, depth=0, seed=78, flags=DET))
        ctx.synchronize()
        got += [ctx.energy_buffer(s).copy(), ctx.energy_buffer(s2).copy()]
        st = ctx.stats()
        got.append(np.asarray([st[k] for k in ("segments", "connections_tested", "deposits", "frames", "rays")], np.int64))
        got.append(ctx.compute_energy_response(s, pkg.default_params(num_rays=24576, depth=0, seed=77)).copy())
        out[mode] = got
        ctx.close()
    assert out["whole"][0].any() and out["whole"][3].any()
    for a, b in zip(out["whole"][:-1], out["staged"][:-1]):
        assert np.array_equal(a, b)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=12288, depth=0, seed=77), sc.source, sc.listener)
    check_energy(out["staged"][-1], e32, e64, 4)


# ---- cfg5 at its full size ---------------------------------------------------------------------------------------
def test_cfg5_full_size_batched_frame_equals_eight_frames(pkg, oracle_mod, scene_factory):
    """BASELINE.json configs[4] at full size: 8 sources x 131 072 rays, one listener, depth 8, 8 bands, as ONE batched
    frame (what bench.py times for cfg5 on one GPU) must give every source exactly the histogram and IR of its own frame —
    bit for bit in deterministic mode — also pipelined, and with the batched reconstruct."""
    sc = scene_factory("old_mine", 8)
    pos = [np.asarray(x, np.float32) for x in sc.extra_sources[:8]]
    p = pkg.default_params(num_rays=131072, depth=8, seed=0x5EED, flags=DET)
    ctx, _ = make_ctx(pkg, sc)
    srcs = [ctx.create_source(x) for x in pos]
    want_e, want_ir = [], []
    for s in srcs:
        want_e.append(ctx.compute_energy_response(s, p).copy())
        ctx.reconstruct_impulse_response(s, p)
        want_ir.append(ctx.impulse_response(s, 0).copy())
    assert all(e.any() for e in want_e) and not np.array_equal(want_e[0], want_e[1])
    for pipelined in (0, 2):
        ctx.set_pipelining(pipelined)
        ctx.compute_energy_response_batch_async(srcs, p)
        ctx.reconstruct_impulse_response_batch_async(srcs, p)
        ctx.synchronize()
        for s, e, ir in zip(srcs, want_e, want_ir):
            assert np.array_equal(ctx.energy_buffer(s), e) and np.array_equal(ctx.impulse_response(s, 0), ir)
    st = ctx.stats()
    assert st["rays"] == 3 * 8 * 131072
    # ... and the batched frame against the ORACLE at full size (round 5): source 4 of the eight, fp32 mode — the committed
    # fixture tests/golden/cfg5_multi_source.npz (the oracle's frame of that source alone) and the oracle itself on all cores
    import os
    from test_gpu_parity import check_energy
    pf = pkg.default_params(num_rays=131072, depth=8, seed=0x5EED)
    ctx.set_pipelining(0)
    ctx.compute_energy_response_batch_async(srcs, pf)
    ctx.synchronize()
    got = ctx.energy_buffer(srcs[4]).copy()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "cfg5_multi_source.npz"))
    assert int(z["pairs"]) == 65536 and int(z["depth"]) == 8 and int(z["seed"]) == 0x5EED
    check_energy(got, z["energy_f32"], z["energy_f64"], 8)
    assert ctx.stats()["deposits"] - st["deposits"] >= int(z["connected"])      # (the batch's counters cover all eight sources)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy_mt(oracle_mod.default_params(num_pairs=65536, depth=8, seed=0x5EED), pos[4], sc.listener, threads=8)
    assert cnt.connected == int(z["connected"])
    for b in range(8):
        assert rel_rms(got[b], e64[b]) <= TIGHT_TOL
    ctx.close()


# ---- the walk's own actor -----------------------------------------------------------------------------------------
def _room_with_a_box_around_the_source(pkg):
    """the shoebox room + a closed 60 cm box (actor 7) around the source: the source sits INSIDE its own mesh"""
    sc = pkg.scenes.shoebox(1)
    c = np.asarray(sc.source, np.float64)
    h = 30.0
    v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)]) + c
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    box = np.array([[v[a], v[b], v[cc]] for a, b, cc, d in quads] + [[v[a], v[cc], v[d]] for a, b, cc, d in quads], np.float32)
    tris = np.concatenate([np.asarray(sc.triangles, np.float32).reshape(-1, 3, 3), box])
    mats = np.concatenate([np.asarray(sc.material_ids, np.uint16), np.zeros(len(box), np.uint16)])
    obj = np.concatenate([np.arange(len(tris) - len(box), dtype=np.uint32) + 100, np.full(len(box), 7, np.uint32)])
    return sc, tris, mats, obj


@pytest.mark.parametrize("rays,depth", [(2000, 0), (32768, 6)])
def test_a_walk_ignores_the_actor_it_starts_from(pkg, oracle_mod, rays, depth):
    """GeneratePath ignores the walking actor and its mesh (AddIgnoredActor, ARTS.cpp:322-327); ConnectSubpaths ignores
    nothing (:252-254).  A source inside its own registered box mesh: without the actor set every source walk is trapped in
    the box and every connection from inside it is blocked by it — no energy; with fs_source_set_object the walks leave,
    the connections from F_1.. (outside) see the listener, and the frame equals the oracle's with the same source_object;
    a batched frame carries each source's own actor."""
    sc, tris, mats, obj = _room_with_a_box_around_the_source(pkg)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(tris, mats, sc.absorption, object_ids=obj)
    ctx.set_listener(sc.listener)
    s = ctx.create_source(sc.source)
    p = pkg.default_params(num_rays=rays, depth=depth, seed=31)
    trapped = ctx.compute_energy_response(s, p).copy()
    assert not trapped.any()
    ctx.set_source_object(s, 7)
    free = ctx.compute_energy_response(s, p).copy()
    st = ctx.stats()
    osc = oracle_mod.Scene(tris, mats, sc.absorption)
    osc.set_objects(obj)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=31), sc.source, sc.listener)
    assert cnt.connected == 0 and not e32.any()                       # the oracle's source is trapped too
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=31, source_object=7), sc.source, sc.listener)
    assert cnt.connected > 0
    check_energy(free, e32, e64, 1)
    assert st["deposits"] == cnt.connected
    # a batch: the first source with its actor, a second one (outside the box) without
    s2 = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(120.0))
    pd = pkg.default_params(num_rays=rays, depth=depth, seed=31, flags=DET)
    want = [ctx.compute_energy_response(q, pd).copy() for q in (s, s2)]
    ctx.compute_energy_response_batch_async([s, s2], pd)
    ctx.synchronize()
    assert np.array_equal(ctx.energy_buffer(s), want[0]) and np.array_equal(ctx.energy_buffer(s2), want[1]) and want[1].any()
    # the listener's actor: its walks ignore it, the source's walks do not
    ctx.set_source_object(s)                                           # back to none: trapped again
    ctx.set_listener_object(7)
    assert not ctx.compute_energy_response(s, p).any()
    ctx.close()


def test_lobes_survive_an_ignored_actor_and_end_point_spheres(pkg, oracle_mod):
    """the instantiations that carry the ignored actor / the end points' spheres read FS_FLAG_MATERIAL_LOBES at run time (they
    were compiled without lobes: a batched frame in which ONE source has an actor ran every source through them and
    dropped the lobes of the others — found by tools/stress.py).  Deterministic mode: a batched frame equals its sources'
    single frames bit for bit; and lobes change the result whether or not an actor is ignored."""
    sc = pkg.scenes.starter_room(4)
    tau, sigma = pkg.scenes.material_lobes(sc)
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma, object_ids=sc.object_ids)
    ctx.set_listener(sc.listener)
    s0 = ctx.create_source(sc.source)
    s1 = ctx.create_source(np.asarray(sc.source, np.float32) + np.array([150, -80, 20], np.float32))
    LOBES = pkg._capi.FLAG_MATERIAL_LOBES
    for radius in (0.0, 25.0):
        p = pkg.default_params(num_rays=4096, depth=8, seed=77, dist_divisor=100.0, flags=LOBES | DET, listener_radius=radius)
        plain = pkg.default_params(num_rays=4096, depth=8, seed=77, dist_divisor=100.0, flags=DET, listener_radius=radius)
        ctx.set_source_object(s0)                                      # no actor anywhere: the lobes instantiation
        ref1 = ctx.compute_energy_response(s1, p).copy()
        assert not np.array_equal(ref1, ctx.compute_energy_response(s1, plain))           # lobes matter in this scene
        ctx.set_source_object(s0, int(sc.object_ids[0]))               # s0's walks ignore an actor: the batch runs the EXT instantiations
        want0 = ctx.compute_energy_response(s0, p).copy()
        assert not np.array_equal(want0, ctx.compute_energy_response(s0, plain))          # ... which must still pick lobes
        ctx.compute_energy_response_batch_async([s0, s1], p)
        ctx.synchronize()
        assert np.array_equal(ctx.energy_buffer(s0), want0)
        assert np.array_equal(ctx.energy_buffer(s1), ref1)             # s1 has no actor: its frame is the one without any
    # and against the oracle: lobes + the walk's own actor, in fp32 mode
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption, transmission=tau, scattering=sigma)
    osc.set_objects(sc.object_ids)
    pf = pkg.default_params(num_rays=4096, depth=8, seed=77, dist_divisor=100.0, flags=LOBES)
    got = ctx.compute_energy_response(s0, pf).copy()
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=2048, depth=8, seed=77, dist_divisor=100.0, flags=oracle_mod.FLAG_MATERIAL_LOBES,
                                                                  source_object=int(sc.object_ids[0])), sc.source, sc.listener)
    check_energy(got, e32, e64, 4)
    ctx.close()
