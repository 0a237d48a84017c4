"""Round 5: the streamed path publishes its impulse responses from the launch itself (ring slot writes by the reconstruct
workgroups + a pinned host word, csrc/fs_device.hpp: publish_arrive) — the reference's contract is that the IR is in the
component's buffer when ReconstructImpulseResponse returns (FrequenSeeAudioComponent.cpp:377-378) and GetImpulseResponse
reads that buffer (FrequenSeeAudioComponent.h:113).  Results must not depend on which of the three publish mechanisms
(host word / tail-stream batch event / tail-stream copy + event) carried a frame."""
import os
import threading
import time

import numpy as np
import pytest

from test_gpu_parity import IR_TOL, TIGHT_TOL, make_ctx, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
DET = 8           # FS_FLAG_DETERMINISTIC
FLUSH2 = 2        # FS_FLAG_FLUSH_BEFORE_RECONSTRUCT: such a reconstruct always takes the tail stream (reconstruct_now)


def frames(pkg, n, rays=16384, flags=DET, seed0=500):
    return [pkg.default_params(num_rays=rays, depth=8, seed=seed0 + i, flags=flags) for i in range(n)]


@pytest.mark.parametrize("fpl", [1, 2, 4])
def test_streamed_frames_publish_from_the_launch_itself(pkg, scene_factory, fpl):
    """Steady state of a single-GPU stream of pipelined frames: nothing is enqueued on the tail stream, no cross-stream
    wait reaches the compute stream, every IR is published through the host word — and arrives without any host-side
    synchronisation call (the consumer only polls the sequence number)."""
    if os.environ.get("FS_FUSED_RECON") == "0":
        pytest.skip("the diagnostic switch FS_FUSED_RECON=0 sends every reconstruct to the tail stream: the counters asserted here are about the default")
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    ctx.set_pipelining(2)
    ctx.set_frames_per_launch(fpl)
    warm = frames(pkg, 8, seed0=100)
    for p in warm:
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    seq0 = ctx.impulse_response_sequence(src)
    assert seq0 == len(warm)
    c0 = ctx.pipeline_counters()
    ps = frames(pkg, 40)
    for p in ps:
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
    ctx.submit()                       # the held frames' last passes: enqueued, not waited for
    c1 = ctx.pipeline_counters()
    d = {k: c1[k] - c0[k] for k in c0}
    assert d["tail_stream_ops"] == 0 and d["publishes_by_event"] == 0 and d["stream_waits_enqueued"] == 0, d
    assert d["publishes_by_word"] == len(ps) and d["fused_launches"] >= len(ps) // fpl, d
    t0 = time.time()
    while ctx.impulse_response_sequence(src) < seq0 + len(ps):     # lock-free poll: no stream wait anywhere
        assert time.time() - t0 < 20.0, "the publishes never arrived in the host word"
    got = ctx.impulse_response_view(src, 0).copy()
    # the last frame against a context that waits for every frame
    plain, qs = make_ctx(pkg, sc)
    plain.compute_energy_response(qs, ps[-1]); plain.reconstruct_impulse_response(qs, ps[-1])
    assert np.array_equal(got, plain.impulse_response(qs, 0)) and np.abs(got).max() > 0
    ctx.synchronize()
    for b in range(4):                 # the device-resident set (the reverb's, fs_copy_band_impulse_response's) is the last frame's too
        assert np.array_equal(ctx.band_impulse_response(src, b), plain.band_impulse_response(qs, b))
    plain.close(); ctx.close()


def test_audio_thread_reads_while_pipelined_frames_are_produced(pkg, scene_factory):
    """The reader-thread test of test_gpu_parity.py on the streamed path: the reader takes whatever
    fs_get_impulse_response points at while fused launches write ring slots and the host word underneath it.  In
    deterministic mode each of the two parameter sets has one bit-exact IR: every read must equal one of them — a slot
    announced before its samples had landed, or recycled too early, would match neither."""
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    params = [pkg.default_params(num_rays=8192, depth=8, seed=s_, dist_divisor=100.0, flags=DET) for s_ in (5, 6)]
    irs = []
    for p in params:
        ctx.compute_energy_response(src, p); ctx.reconstruct_impulse_response(src, p)
        irs.append(ctx.impulse_response(src, 0).copy())
    assert not np.array_equal(irs[0], irs[1]) and irs[0].any()
    ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
    stop = threading.Event()
    seen = {"reads": 0, "bad": 0, "a": 0, "b": 0, "seq_max": 0}

    def reader():
        while not stop.is_set():
            seen["seq_max"] = max(seen["seq_max"], ctx.impulse_response_sequence(src))
            v = ctx.impulse_response_view(src, 0).copy()
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
        for f in range(400):
            ctx.compute_energy_response_async(src, params[(f >> 1) & 1])    # (two frames per launch: both sets reach the front)
            ctx.reconstruct_impulse_response_async(src, params[(f >> 1) & 1])
        ctx.synchronize()
    finally:
        stop.set()
        th.join()
    assert seen["bad"] == 0 and seen["reads"] > 20 and seen["a"] > 0 and seen["b"] > 0, seen
    assert ctx.impulse_response_sequence(src) == 402
    assert np.array_equal(ctx.impulse_response(src, 0), irs[1])     # frames 398, 399 used the second set
    ctx.close()


def test_publish_mechanisms_interleave(pkg, scene_factory):
    """Fused launches (host word), a literal second flush (tail stream: kernel + copy + event) and waited-for frames
    (compute-stream batch of one) in one stream of one source: every observation equals the same frame on a context that
    waits for everything — the published IR and the device-resident per-band set, whoever wrote it last."""
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    ref, rs = make_ctx(pkg, sc)
    ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
    ps = frames(pkg, 24, rays=8192)

    def expect(p, recon_flags=0):
        q = pkg.default_params(num_rays=p.num_rays, depth=p.depth, seed=p.seed, flags=p.flags | recon_flags)
        ref.compute_energy_response(rs, p); ref.reconstruct_impulse_response(rs, q)
        return ref.impulse_response(rs, 0).copy(), [ref.band_impulse_response(rs, b).copy() for b in range(4)]

    def check(p, recon_flags=0):
        ir, bands = expect(p, recon_flags)
        assert np.array_equal(ctx.impulse_response(src, 0), ir)
        for b in range(4):
            assert np.array_equal(ctx.band_impulse_response(src, b), bands[b]), b

    k = 0
    for rnd in range(3):
        for _ in range(5):             # fused launches
            ctx.compute_energy_response_async(src, ps[k]); ctx.reconstruct_impulse_response_async(src, ps[k]); k += 1
        ctx.synchronize()
        check(ps[k - 1])
        # the tail stream right behind fused launches that are still running (no wait in between)
        ctx.compute_energy_response_async(src, ps[k]); ctx.reconstruct_impulse_response_async(src, ps[k]); k += 1
        q = ps[k]; k += 1
        ctx.compute_energy_response_async(src, q)
        qf = pkg.default_params(num_rays=q.num_rays, depth=q.depth, seed=q.seed, flags=q.flags | FLUSH2)
        ctx.reconstruct_impulse_response_async(src, qf)            # reconstruct_now: all zeros (the literal second flush)
        ctx.synchronize()
        assert not ctx.impulse_response(src, 0).any()
        for b in range(4):
            assert not ctx.band_impulse_response(src, b).any()
        # ... and a fused launch right behind the tail stream's reconstruct
        ctx.compute_energy_response_async(src, ps[k]); ctx.reconstruct_impulse_response_async(src, ps[k]); k += 1
        ctx.synchronize()
        check(ps[k - 1])
    c = ctx.pipeline_counters()
    if os.environ.get("FS_FUSED_RECON") != "0":        # (the diagnostic switch sends every reconstruct to the tail stream)
        assert c["publishes_by_word"] > 0 and c["publishes_by_event"] == 3, c
    assert ctx.impulse_response_sequence(src) == k
    ref.close(); ctx.close()


def test_reverb_reads_behind_fused_reconstructs(pkg, scene_factory):
    """fs_reverb_process (audio thread, own stream) convolves with the device-resident IR: behind a stream of fused launches
    its output equals that of a context whose reconstructs are waited for.  (A source with a reverb gets an event on the
    compute stream behind every launch that rewrites its IR; one without does not — fs_reverb_init drains what is in flight.)"""
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    ref, rs = make_ctx(pkg, sc)
    ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
    ps = frames(pkg, 10, rays=8192)
    for p in ps[:4]:                   # launches in flight when the reverb is created
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
    ctx.reverb_init(src, 256); ref.reverb_init(rs, 256)
    rng = np.random.default_rng(3)
    audio = rng.standard_normal((6, 512)).astype(np.float32) * 0.1
    for i, p in enumerate(ps[4:]):
        ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
        ref.compute_energy_response(rs, p); ref.reconstruct_impulse_response(rs, p)
        if i % 2 == 1:                 # (two frames per launch: after an odd frame both have been sent off)
            ctx.submit()
            a = ctx.reverb_process(src, audio[i])
            b = ref.reverb_process(rs, audio[i])
            assert np.array_equal(a, b), i
        else:
            ref.reverb_process(rs, audio[i]); ctx.synchronize(); ctx.reverb_process(src, audio[i])   # keep the two histories in step
    ref.close(); ctx.close()


def test_pipeline_counters_argument_errors(pkg, scene_factory):
    import ctypes as C
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    c = pkg._capi.PipelineCounters()
    c.struct_size = 8
    assert ctx.lib.fs_get_pipeline_counters(ctx.h, C.byref(c)) == pkg._capi.ERR_INVALID_ARGUMENT
    assert ctx.lib.fs_get_pipeline_counters(ctx.h, None) == pkg._capi.ERR_INVALID_ARGUMENT
    d = ctx.pipeline_counters()
    assert set(d) >= {"fused_launches", "tail_stream_ops", "publishes_by_word", "publishes_by_event", "host_waits"}
    ctx.close()


# ---- the guards of the waited-for staged walk ---------------------------------------------------
@pytest.mark.parametrize("what", ["source_object", "listener_radius"])
def test_uncapped_frames_with_an_ignored_actor_or_end_point_spheres(pkg, oracle_mod, monkeypatch, what):
    """depth = 0, 32 768 subpaths (above the size from which a waited-for uncapped frame walks in stages), with the walk's own
    actor ignored (AddIgnoredActor, ARTS.cpp:322-327) or with end-point spheres: the frame equals the oracle's and — bit for
    bit in deterministic mode — the same frame on a context that never stages (FS_SYNC_WALK_STAGES empty)."""
    from test_round4 import _room_with_a_box_around_the_source
    from test_gpu_parity import check_energy
    sc, tris, mats, obj = _room_with_a_box_around_the_source(pkg)
    rays = 32768
    kw = dict(listener_radius=34.0, source_radius=20.0) if what == "listener_radius" else {}

    def run(flags):
        ctx = pkg.Context(num_bands=1)
        ctx.set_scene(tris, mats, sc.absorption, object_ids=obj)
        ctx.set_listener(sc.listener)
        s = ctx.create_source(sc.source)
        ctx.set_source_object(s, 7)     # (the source sits inside its own box: without its actor ignored nothing gets out)
        e = ctx.compute_energy_response(s, pkg.default_params(num_rays=rays, depth=0, seed=77, flags=flags, **kw)).copy()
        st = ctx.stats()
        ctx.close()
        return e, st

    got, st = run(0)
    det_staged, _ = run(DET)
    monkeypatch.setenv("FS_SYNC_WALK_STAGES", "")
    det_plain, _ = run(DET)
    assert got.any() and np.array_equal(det_staged, det_plain)
    osc = oracle_mod.Scene(tris, mats, sc.absorption)
    osc.set_objects(obj)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=0, seed=77, source_object=7, **kw), sc.source, sc.listener)
    check_energy(got, e32, e64, 1)
    assert st["deposits"] == cnt.connected


# ---- the walker's own actor on the fast path ------------------------------------------------------
@pytest.mark.parametrize("fpl", [1, 2])
@pytest.mark.parametrize("depth", [8, 0])
def test_pipelined_frames_ignore_the_walkers_own_actor(pkg, oracle_mod, fpl, depth):
    """GeneratePath always ignores the walking actor (AddIgnoredActor, ARTS.cpp:322-327).  Frames of a source with an actor of
    its own (and of a listener with one) are held by fs_set_pipelining like any other — the fused launch's EXT flavour,
    csrc/fs_frame_ext.hip — and give, bit for bit in deterministic mode, what a context that waits for every frame gives;
    in fp32 mode the frame equals the oracle's with the same source_object."""
    from test_round4 import _room_with_a_box_around_the_source
    from test_gpu_parity import check_energy
    sc, tris, mats, obj = _room_with_a_box_around_the_source(pkg)
    rays = 16384

    def make(pipe):
        ctx = pkg.Context(num_bands=1)
        ctx.set_scene(tris, mats, sc.absorption, object_ids=obj)
        ctx.set_listener(sc.listener)
        s = ctx.create_source(sc.source)
        ctx.set_source_object(s, 7)
        if pipe:
            ctx.set_pipelining(2); ctx.set_frames_per_launch(fpl)
        return ctx, s

    plain, ps = make(False)
    pipe, qs = make(True)
    c0 = pipe.pipeline_counters()
    for i in range(9):
        p = pkg.default_params(num_rays=rays, depth=depth, seed=900 + i, flags=DET)
        plain.compute_energy_response_async(ps, p); plain.reconstruct_impulse_response_async(ps, p)
        pipe.compute_energy_response_async(qs, p); pipe.reconstruct_impulse_response_async(qs, p)
    plain.synchronize(); pipe.synchronize()
    c1 = pipe.pipeline_counters()
    assert c1["fused_launches"] - c0["fused_launches"] >= 9 // fpl          # the frames WERE held
    want_e, got_e = plain.energy_buffer(ps), pipe.energy_buffer(qs)
    assert want_e.any() and np.array_equal(got_e, want_e)
    assert np.array_equal(pipe.impulse_response(qs, 0), plain.impulse_response(ps, 0))
    a, b = plain.stats(), pipe.stats()
    for k in ("frames", "rays", "segments", "connections_tested", "deposits"):
        assert a[k] == b[k], k
    # fp32 mode against the oracle, through the pipelined context (the blocking call lets the held frames finish)
    got = pipe.compute_energy_response(qs, pkg.default_params(num_rays=rays, depth=depth, seed=31)).copy()
    osc = oracle_mod.Scene(tris, mats, sc.absorption)
    osc.set_objects(obj)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, depth=depth, seed=31, source_object=7), sc.source, sc.listener)
    assert cnt.connected > 0
    check_energy(got, e32, e64, 1)
    # the listener's actor instead: its walks ignore the box, the source's (inside the box) do not — trapped, no energy, also when held
    pipe.set_source_object(qs)
    pipe.set_listener_object(7)
    p = pkg.default_params(num_rays=rays, depth=depth, seed=31, flags=DET)
    for _ in range(4):
        pipe.compute_energy_response_async(qs, p); pipe.reconstruct_impulse_response_async(qs, p)
    pipe.synchronize()
    assert not pipe.energy_buffer(qs).any()
    plain.close(); pipe.close()


def test_grouped_frames_keep_each_sources_actor(pkg, scene_factory):
    """Two sources stream through one pipelined context, one with an actor of its own, one without: frames of the two are
    never confused (a frame is traced with the actor of ITS source), whatever shares a launch with it."""
    from test_round4 import _room_with_a_box_around_the_source
    sc, tris, mats, obj = _room_with_a_box_around_the_source(pkg)

    def make(pipe):
        ctx = pkg.Context(num_bands=1)
        ctx.set_scene(tris, mats, sc.absorption, object_ids=obj)
        ctx.set_listener(sc.listener)
        a = ctx.create_source(sc.source)                                                  # inside the box: needs its actor ignored
        b = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(120.0))      # outside, no actor
        ctx.set_source_object(a, 7)
        if pipe:
            ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
        return ctx, a, b

    plain, pa, pb = make(False)
    pipe, qa, qb = make(True)
    for i in range(8):
        p = pkg.default_params(num_rays=8192, depth=8, seed=40 + i, flags=DET)
        for ctx, s in ((plain, pa if i % 3 else pb), (pipe, qa if i % 3 else qb)):
            ctx.compute_energy_response_async(s, p); ctx.reconstruct_impulse_response_async(s, p)
    plain.synchronize(); pipe.synchronize()
    for s_plain, s_pipe in ((pa, qa), (pb, qb)):
        assert plain.energy_buffer(s_plain).any()
        assert np.array_equal(pipe.energy_buffer(s_pipe), plain.energy_buffer(s_plain))
        assert np.array_equal(pipe.impulse_response(s_pipe, 0), plain.impulse_response(s_plain, 0))
    plain.close(); pipe.close()


# ---- the cooperative traversal's fp16 boxes on a large map --------------------------------------------
def test_small_frames_far_from_the_origin(pkg, oracle_mod, scene_factory):
    """A tick-sized frame (the cooperative traversal's domain) in a scene 300 m from the origin: beyond 16 384 units the
    fp16 world-space boxes of the cooperative records stop culling, so such scenes use the lane-private traversal for their
    small frames — results equal the oracle's either way (closest hits do not depend on the tree)."""
    from test_gpu_parity import check_energy
    sc = scene_factory("starter_room", 1)
    off = np.array([30000.0, -21000.0, 500.0], np.float32)
    tris = (np.asarray(sc.triangles, np.float32).reshape(-1, 3, 3) + off).astype(np.float32)
    ctx = pkg.Context(num_bands=1)
    ctx.set_scene(tris, sc.material_ids, sc.absorption)
    lis = (np.asarray(sc.listener, np.float32) + off).astype(np.float32)
    srcp = (np.asarray(sc.source, np.float32) + off).astype(np.float32)
    ctx.set_listener(lis)
    s = ctx.create_source(srcp)
    p = pkg.default_params(num_rays=2000, depth=0, seed=11, flags=pkg._capi.FLAG_FIXED_NORM_1000)
    got = ctx.compute_energy_response(s, p).copy()
    osc = oracle_mod.Scene(tris, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=1000, depth=0, seed=11, flags=oracle_mod.FLAG_FIXED_NORM_1000), srcp, lis)
    assert cnt.connected > 0
    check_energy(got, e32, e64, 1)
    ctx.close()


# ---- zero blocks of the published IR stay on the device side of the bus ------------------------------------------------
@pytest.mark.parametrize("path", ["waited", "streamed"])
def test_zero_blocks_of_the_ring_slots_are_rewritten_when_they_must_be(pkg, scene_factory, path):
    """The reconstruct workgroups skip the host write of a 4 096-sample block whose samples are all exactly zero when the ring
    slot's block is known to be zero already (csrc/fs_device.hpp: host_block_wanted; a room's IR ends after a quarter of the
    second).  The published IR must not depend on what the slot held eight publishes ago: energies with deposits in LATE bins
    (every block non-zero) and in early bins only alternate in runs longer and shorter than the ring, through the batch kernel
    (waited-for reconstructs), the fused launch (streamed frames with an installed energy buffer are not traced, so the streamed
    leg alternates traced frames of two distances scales), a copy command (fs_set_impulse_response) in between; every published
    IR equals the one a fresh context produces for the same energy."""
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    nb = ctx.num_bins
    rng = np.random.default_rng(5)

    def energy(kind):
        e = np.zeros((4, nb), np.float32)
        if kind == "late":
            e[:, rng.integers(0, nb, 40)] = rng.random(40).astype(np.float32) + 0.1
            e[:, nb - 1] = 0.5
        elif kind == "early":
            e[:, rng.integers(0, 30, 10)] = rng.random(10).astype(np.float32) + 0.1
        elif kind == "middle":
            e[:, 400 + rng.integers(0, 50, 10)] = rng.random(10).astype(np.float32) + 0.1
        return e

    def reference_ir(e):
        ref, rs = make_ctx(pkg, sc)
        ref.update_energy_buffer(rs, e)
        ref.reconstruct_impulse_response(rs)
        out = ref.impulse_response(rs, 0).copy()
        ref.close()
        return out

    if path == "waited":
        seq = ["late"] * 3 + ["early"] * 11 + ["zero"] * 2 + ["middle"] * 9 + ["late"] * 9 + ["early"] * 3 + ["set"] + ["early"] * 10 + ["zero"] * 9
        refs = {}
        for i, kind in enumerate(seq):
            if kind == "set":                                   # a copy command writes a whole slot behind the kernels' back
                ir = (rng.random(ctx.num_samples).astype(np.float32) - 0.5)
                ctx.set_impulse_response(src, ir)
                assert np.array_equal(ctx.impulse_response(src, 0), ir)
                continue
            e = energy(kind)
            ctx.update_energy_buffer(src, e)
            ctx.reconstruct_impulse_response(src)
            got = ctx.impulse_response(src, 0)
            want = reference_ir(e)
            assert np.array_equal(got, want), (i, kind, int(np.flatnonzero(got != want)[0]))
            if kind == "early":
                assert not got[3 * 4096:].any()                 # the tail really is exact zeros (what the rule relies on)
    else:
        # streamed frames, two per launch: long runs of far deposits (delays x 100 and a gain that lifts them over the amplitude threshold: blocks 4 - 6) then
        # of near ones (the defaults: block 0), each published IR picked up by its number and compared with an unpipelined context
        ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
        ref, rs = make_ctx(pkg, sc)
        far, near = dict(dist_divisor=10.0, energy_gain=1e12), dict()      # far: amplitudes in blocks 4 - 6 only; near: in block 0 only
        kinds = [far] * 12 + [near] * 20 + [far] * 3 + [near] * 13
        params = [pkg.default_params(num_rays=4096, depth=8, seed=800 + i, flags=DET, **kw) for i, kw in enumerate(kinds)]
        want = []
        for p in params:
            ref.compute_energy_response(rs, p); ref.reconstruct_impulse_response(rs, p)
            want.append(ref.impulse_response(rs, 0).copy())
        assert all(w[4 * 4096:7 * 4096].any() and not w[:3 * 4096].any() for w in want[:12])      # far frames: late blocks only
        assert all(w[:4096].any() and not w[2 * 4096:].any() for w in want[12:32])                 # near frames: the first block only
        seen = {}
        for p in params:
            ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            k0 = ctx.impulse_response_sequence(src)
            if k0 and k0 not in seen:
                a = ctx.impulse_response_view(src, 0).copy()
                if ctx.impulse_response_sequence(src) == k0:
                    seen[k0] = a
        ctx.synchronize()
        seen[len(params)] = ctx.impulse_response(src, 0).copy()
        assert len(seen) >= 8
        bad = [k for k, a in sorted(seen.items()) if not np.array_equal(a, want[k - 1])]
        assert not bad, (bad, sorted(seen))
        ref.close()
    ctx.close()


# ---- randomized parameters against the oracle ---------------------------------------------------------------------------
def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(n):
        rr = int(rng.random() < 0.8)
        depth = int(rng.choice([0, 1, 2, 3, 5, 8, 12, 17, 33])) if rr else int(rng.choice([1, 2, 4, 7]))
        kw = dict(depth=depth, russian_roulette=rr, seed=int(rng.integers(1, 1 << 48)),
                  rr_prob=float(rng.choice([0.5, 0.75, 0.9, 0.95])), dist_divisor=float(rng.choice([100.0, 300.0, 1000.0])),
                  surface_offset=float(rng.choice([0.05, 0.1, 0.5])), connect_pullback=float(rng.choice([0.05, 0.1, 1.0])),
                  min_seg=float(rng.choice([0.0, 0.5, 1.0])), prob_exponent=float(rng.choice([0.1, 0.25, 1.0])),
                  energy_clamp=float(rng.choice([0.5, 1.0, 100.0])), energy_gain=float(rng.choice([1.0, 10.0, 1e3])),
                  sound_speed=float(rng.choice([300.0, 343.0])), max_trace_dist=float(rng.choice([1500.0, 1e6])))
        flags = (4 if rng.random() < 0.3 else 0) | (1 if rng.random() < 0.3 else 0)      # cosine sampling, the fixed 1/1000 normaliser
        cases.append((str(rng.choice(["shoebox", "starter_room"])), int(rng.choice([2, 254, 1000, 4096, 16384])), flags, kw,
                      [float(x) for x in rng.choice([0.0, 0.01, 0.05, 0.2], 8)],
                      rng.random(6) if rng.random() < 0.5 else None))      # (half of the cases: source and listener somewhere else in the scene's box)
    return cases


@pytest.mark.parametrize("mode", ["waited", "streamed"])
def test_energy_and_published_ir_parity_over_random_parameters(pkg, oracle_mod, scene_factory, mode):
    """64 seeded random parameter sets (depth caps 0 .. 33, roulette on / off and its probability, distance scale, offsets, clamp
    and gain, air absorption per band, cosine sampling, the literal normaliser; 2 .. 16 384 subpaths, two scenes): the energy
    histogram against the oracle's (same occupied bins, <= TIGHT_TOL relative RMS per band) and the PUBLISHED impulse response
    against the oracle's reconstruct of the oracle's band-mean energy — frames waited for one by one, and the same frames
    streamed through the pipeline (two per launch) with the last one of every scene checked."""
    from test_gpu_parity import check_energy
    cases = _random_cases(64, 20250105)
    ctxs = {}
    for name, rays, flags, kw, air, where in cases:
        bands = 1 if name == "shoebox" else 4
        sc = scene_factory(name, bands)
        if name not in ctxs:
            ctx, src = make_ctx(pkg, sc)
            if mode == "streamed":
                ctx.set_pipelining(2); ctx.set_frames_per_launch(2)
            ctxs[name] = (ctx, src, oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption), sc)
        ctx, src, osc, sc = ctxs[name]
        spos, lpos = np.asarray(sc.source, np.float32), np.asarray(sc.listener, np.float32)
        if where is not None:
            lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
            spos = (lo + (0.1 + 0.8 * where[:3]) * (hi - lo)).astype(np.float32)
            lpos = (lo + (0.1 + 0.8 * where[3:]) * (hi - lo)).astype(np.float32)
        ctx.set_source_position(src, spos); ctx.set_listener(lpos)
        p = pkg.default_params(num_rays=rays, flags=flags, air_absorption=air, **kw)
        if mode == "waited":
            got = ctx.compute_energy_response(src, p).copy()
            ctx.reconstruct_impulse_response(src, p)
        else:
            ctx.compute_energy_response_async(src, p); ctx.reconstruct_impulse_response_async(src, p)
            ctx.synchronize()
            got = ctx.energy_buffer(src).copy()
        op = oracle_mod.default_params(num_pairs=rays // 2, flags=flags, air_absorption=air, **kw)
        e32, e64, cnt = osc.compute_energy(op, spos, lpos)
        check_energy(got, e32, e64, bands)
        mean_e = (e32.astype(np.float32).sum(axis=0, dtype=np.float32) / np.float32(bands)).astype(np.float32) if bands > 1 else e32[0]
        ir_ref = oracle_mod.reconstruct(mean_e)
        ir = ctx.impulse_response(src, 0)
        assert np.abs(ir - ir_ref).max() <= IR_TOL * max(float(np.abs(ir_ref).max()), 1e-30), (name, rays, kw)
    for ctx, *_ in ctxs.values():
        ctx.close()


def test_ticks_over_random_sources_and_parameters(pkg, oracle_mod, scene_factory):
    """fs_update_sources (UpdateSources, ARTS.cpp:100-126) with 1 .. 9 sources at random places and random parameters, 12 ticks:
    every source's energy equals the oracle's frame for that source and its published IR the oracle's reconstruct — also when the
    set of sources and the frame size change from tick to tick (tables, masks and ring slots are reused across ticks)."""
    from test_gpu_parity import check_energy
    sc = scene_factory("starter_room", 4)
    ctx, _ = make_ctx(pkg, sc)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    rng = np.random.default_rng(77)
    lo, hi = sc.triangles.min(axis=(0, 1)), sc.triangles.max(axis=(0, 1))
    pos = [(lo + (0.15 + 0.7 * rng.random(3)) * (hi - lo)).astype(np.float32) for _ in range(9)]
    srcs = [ctx.create_source(q) for q in pos]
    for tick in range(12):
        k = int(rng.integers(1, 10))
        pick = sorted(rng.choice(9, k, replace=False).tolist())
        rays = int(rng.choice([2, 500, 2000, 6000]))
        kw = dict(depth=int(rng.choice([0, 0, 4, 9])), seed=int(rng.integers(1, 1 << 40)), dist_divisor=float(rng.choice([100.0, 1000.0])),
                  energy_gain=float(rng.choice([10.0, 1e3])))
        flags = 1 if rng.random() < 0.5 else 0
        ctx.update_sources([srcs[i] for i in pick], pkg.default_params(num_rays=rays, flags=flags, **kw))
        for i in pick:
            e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=rays // 2, flags=flags, **kw), pos[i], sc.listener)
            got = ctx.energy_buffer(srcs[i])
            check_energy(got, e32, e64, 4)
            mean_e = (e32.astype(np.float32).sum(axis=0, dtype=np.float32) / np.float32(4)).astype(np.float32)
            ir_ref = oracle_mod.reconstruct(mean_e)
            ir = ctx.impulse_response(srcs[i], 0)
            assert np.abs(ir - ir_ref).max() <= IR_TOL * max(float(np.abs(ir_ref).max()), 1e-30), (tick, i)
    ctx.close()


# ---- the long-walk lane of waited-for uncapped frames --------------------------------------------------------------------------
@pytest.mark.parametrize("bounds,lane", [("24", "40,78"), ("24", "30,0"), ("16", "8,40"), ("24", "64,80"), ("24", "20,24"), ("5,9,70", "12,30"),
                                          ("24", "2,50"), (None, None)])
def test_the_long_walk_lane_changes_nothing_but_the_time(pkg, oracle_mod, scene_factory, monkeypatch, bounds, lane):
    """FS_SYNC_LANE = "len,end": the walks of `len` steps or more (the plan pass knows every walk's length) take cooperative waves
    of their own from step 0 on — steps [0, end) in the launch of the first stage, the rest in the second's (end 0: all of it in
    the first; no setting: length and first bound by the frame's size, sync_lane_plan in fs_capi_frame.cpp).  Whatever the setting — the lane ending inside the main record tier, beyond it, at the first bound, finishing in the
    first launch, holding most of the frame — energies (deterministic mode: bit for bit), IRs and counters are those of the walk
    in one piece; batched sources and a walk that ignores its own actor too; and the frame equals the oracle's."""
    from test_gpu_parity import check_energy
    from test_round4 import _room_with_a_box_around_the_source
    sc = scene_factory("starter_room", 4)
    scb, tris, mats, obj = _room_with_a_box_around_the_source(pkg)
    out = {}
    for mode in ("whole", "lane"):
        if mode == "lane" and bounds is None:   # the library's defaults (frames of 15 000 subpaths or more walk in stages)
            for k in ("FS_SYNC_WALK_STAGES", "FS_SYNC_LANE", "FS_SYNC_STAGE_FROM"):
                monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv("FS_SYNC_WALK_STAGES", bounds if mode == "lane" else "")
            monkeypatch.setenv("FS_SYNC_LANE", lane if mode == "lane" else "0")
            monkeypatch.setenv("FS_SYNC_STAGE_FROM", "1")
        ctx, s = make_ctx(pkg, sc)
        s2 = ctx.create_source(np.asarray(sc.source, np.float32) + np.float32(40.0))
        p = pkg.default_params(num_rays=24576, depth=0, seed=91, flags=DET)
        e = ctx.compute_energy_response(s, p).copy()
        ctx.reconstruct_impulse_response(s, p)
        got = [e, ctx.impulse_response(s, 0).copy()]
        for seed in (92, 93):   # (the second frame finds the first one's continuation records in the state set)
            ctx.compute_energy_response_batch_async([s, s2], pkg.default_params(num_rays=6000, depth=0, seed=seed, flags=DET))
            ctx.synchronize()
            got += [ctx.energy_buffer(s).copy(), ctx.energy_buffer(s2).copy()]
        st = ctx.stats()
        got.append(np.asarray([st[k] for k in ("segments", "connections_tested", "deposits", "frames", "rays")], np.int64))
        plain = ctx.compute_energy_response(s, pkg.default_params(num_rays=24576, depth=0, seed=91)).copy()
        assert ctx.pipeline_counters()["lane_launches"] == (0 if mode == "whole" else 2 if bounds is None else 4)
        ctx.close()
        # a walk that ignores the actor it starts from (the EXT instantiations of the same kernels)
        cb = pkg.Context(num_bands=1)
        cb.set_scene(tris, mats, scb.absorption, object_ids=obj)
        cb.set_listener(scb.listener)
        sb = cb.create_source(scb.source)
        cb.set_source_object(sb, 7)
        got.append(cb.compute_energy_response(sb, pkg.default_params(num_rays=32768, depth=0, seed=77, flags=DET)).copy())
        cb.close()
        out[mode] = got + [plain]
    assert out["whole"][0].any() and out["whole"][3].any() and out["whole"][-2].any()
    for a, b in zip(out["whole"][:-1], out["lane"][:-1]):
        assert np.array_equal(a, b)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=12288, depth=0, seed=91), sc.source, sc.listener)
    check_energy(out["lane"][-1], e32, e64, 4)
