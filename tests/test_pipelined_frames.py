"""Pipelined frames (fs_set_pipelining): the connect pass of frame f is held back and launched together with the walk of
frame f + 1 as ONE kernel.  Results must not depend on the setting — only when work reaches the GPU does."""
import numpy as np
import pytest

from test_gpu_parity import IR_TOL, TIGHT_TOL, make_ctx, rel_rms  # noqa: F401

pytestmark = pytest.mark.gpu
DET = 8   # FS_FLAG_DETERMINISTIC: integer deposits, bit-identical however the passes are scheduled


def frames(pkg, n, rays=16384, flags=DET):
    return [pkg.default_params(num_rays=rays, depth=8, seed=300 + i, flags=flags) for i in range(n)]


@pytest.mark.parametrize("depth", [1, 2])
def test_streamed_frames_equal_the_unpipelined_ones(pkg, scene_factory, depth):
    """A stream of frames on one source: every frame's energy and IR, read after the stream, equal those of a context
    without pipelining — bit-identical in deterministic mode, to tolerance with fp32 atomics; work counters equal."""
    sc = scene_factory("starter_room", 4)
    plain, ps = make_ctx(pkg, sc)
    pipe, qs = make_ctx(pkg, sc)
    pipe.set_pipelining(depth)
    for flags in (DET, 0):
        want_ir, got_ir = [], []
        plain.reset_stats(); pipe.reset_stats()
        for p in frames(pkg, 7, flags=flags):
            plain.compute_energy_response_async(ps, p); plain.reconstruct_impulse_response_async(ps, p)
            pipe.compute_energy_response_async(qs, p); pipe.reconstruct_impulse_response_async(qs, p)
        plain.synchronize(); pipe.synchronize()
        want_e, got_e = plain.energy_buffer(ps), pipe.energy_buffer(qs)
        want_ir, got_ir = plain.impulse_response(ps, 0), pipe.impulse_response(qs, 0)
        assert np.abs(want_ir).max() > 0
        if flags & DET:
            assert np.array_equal(got_e, want_e) and np.array_equal(got_ir, want_ir)
        else:
            assert np.array_equal(got_e != 0, want_e != 0)
            assert max(rel_rms(got_e[b], want_e[b]) for b in range(4)) <= TIGHT_TOL
            assert np.abs(got_ir - want_ir).max() <= IR_TOL * np.abs(want_ir).max()
        a, b = plain.stats(), pipe.stats()
        for k in ("frames", "rays", "segments", "connections_tested", "deposits"):
            assert a[k] == b[k], k
    plain.close(); pipe.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_every_observation_point_sees_the_finished_frame(pkg, scene_factory, oracle_mod, depth):
    """Whatever the caller does after an asynchronous compute — read the energy, the stats, the IR, move geometry, change
    the frame size, switch modes, trace another source — it sees exactly what a context without pipelining shows."""
    sc = scene_factory("starter_room", 4)
    plain, ps = make_ctx(pkg, sc)
    pipe, qs = make_ctx(pkg, sc)
    pipe.set_pipelining(depth)
    ps2, qs2 = plain.create_source(sc.source + np.float32(90.0)), pipe.create_source(sc.source + np.float32(90.0))
    tri = np.asarray(sc.triangles, np.float32)
    rng = np.random.default_rng(3)
    for step in range(60):
        op = int(rng.integers(0, 10))
        p = pkg.default_params(num_rays=int(rng.choice([2048, 8192, 16384, 65536])), depth=int(rng.choice([2, 8])),
                               seed=1000 + step, flags=DET | (1 if rng.random() < 0.3 else 0))
        for c, s1, s2 in ((plain, ps, ps2), (pipe, qs, qs2)):
            src = s2 if op == 7 else s1
            c.compute_energy_response_async(src, p)
            if op != 5:
                c.reconstruct_impulse_response_async(src, p)
            if op == 1:
                c.synchronize()
            elif op == 2:
                c._last = c.energy_buffer(src)
            elif op == 3:
                c._last = c.stats()["deposits"]
            elif op == 4:   # moving geometry between frames: the held-back frame saw the old positions
                c.update_triangles(0, tri[:32] + np.float32(step % 3))
            elif op == 6:   # a frame that is never held (all connections) behind one that is
                c.compute_energy_response_async(src, pkg.default_params(num_rays=2048, depth=4, seed=step, flags=DET | 32))
                c.reconstruct_impulse_response_async(src, p)
            elif op == 8:
                c.submit()
            elif op == 9 and c is pipe:   # change the depth in mid-stream
                c.set_pipelining(1 + (step % 2))
        if op in (2, 3):
            assert np.array_equal(np.asarray(plain._last), np.asarray(pipe._last)), (step, op)
    plain.synchronize(); pipe.synchronize()
    for a, b in ((ps, qs), (ps2, qs2)):
        assert np.array_equal(plain.energy_buffer(a), pipe.energy_buffer(b))
        assert np.array_equal(plain.impulse_response(a, 0), pipe.impulse_response(b, 0))
    sa, sb = plain.stats(), pipe.stats()
    for k in ("frames", "rays", "segments", "connections_tested", "deposits"):
        assert sa[k] == sb[k], k
    plain.close(); pipe.close()


def test_blocking_calls_and_the_oracle(pkg, scene_factory, oracle_mod):
    """The blocking variants never leave a frame held back: cfg2 against the oracle with pipelining on."""
    sc = scene_factory("starter_room", 4)
    ctx, src = make_ctx(pkg, sc)
    ctx.set_pipelining(2)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    for seed in (5, 6):
        e = ctx.compute_energy_response(src, pkg.default_params(num_rays=16384, depth=8, seed=seed))
        e32, e64, cnt = osc.compute_energy(oracle_mod.default_params(num_pairs=8192, depth=8, seed=seed), sc.source, sc.listener)
        assert np.array_equal(e != 0, e32 != 0)
        assert max(rel_rms(e[b], e64[b]) for b in range(4)) <= 1e-3
    ctx.set_pipelining(False)
    ctx.close()


@pytest.mark.parametrize("depth", [1, 2])
def test_batched_frames_are_pipelined_too(pkg, scene_factory, depth):
    """Several sources traced as one batched frame (ForceUpdateSources): the held-back passes carry the per-source tables;
    every source's energy and IR equal those of an unpipelined context, stream after stream, also with single-source
    frames in between."""
    sc = scene_factory("starter_room", 4)
    plain, ps = make_ctx(pkg, sc)
    pipe, qs = make_ctx(pkg, sc)
    pipe.set_pipelining(depth)
    pos = [np.asarray(sc.source, np.float32) + np.float32(40.0 * k) * np.array([1, -1, 0], np.float32) for k in range(1, 4)]
    pa = [ps] + [plain.create_source(q) for q in pos]
    qa = [qs] + [pipe.create_source(q) for q in pos]
    for rnd in range(3):
        for i in range(5):
            p = pkg.default_params(num_rays=8192, depth=8, seed=700 + 10 * rnd + i, flags=DET)
            for c, srcs in ((plain, pa), (pipe, qa)):
                if i == 3:      # a single-source frame between the batches
                    c.compute_energy_response_async(srcs[1], p)
                    c.reconstruct_impulse_response_async(srcs[1], p)
                else:
                    c.compute_energy_response_batch_async(srcs, p)
                    for s_ in (srcs if i != 2 else srcs[:2]):      # (one batch reconstructs only two of its sources)
                        c.reconstruct_impulse_response_async(s_, p)
        plain.synchronize(); pipe.synchronize()
        for a, b in zip(pa, qa):
            assert np.array_equal(plain.energy_buffer(a), pipe.energy_buffer(b)), rnd
            assert np.array_equal(plain.impulse_response(a, 0), pipe.impulse_response(b, 0)), rnd
        sa, sb = plain.stats(), pipe.stats()
        for k in ("frames", "rays", "segments", "connections_tested", "deposits"):
            assert sa[k] == sb[k], (rnd, k)
    plain.close(); pipe.close()


def test_argument_errors_of_the_new_calls(pkg, scene_factory):
    """fs_set_pipelining takes 0, 1 or 2; the gather needs a peer communicator; both say so instead of guessing."""
    sc = scene_factory("shoebox", 1)
    ctx, src = make_ctx(pkg, sc)
    for bad in (3, -1):
        with pytest.raises(pkg.FrequenSeeError) as e:
            ctx.set_pipelining(bad)
        assert e.value.code == pkg._capi.ERR_INVALID_ARGUMENT
    ctx.set_pipelining(2)
    ctx.compute_energy_response_async(src, pkg.default_params(num_rays=1024, depth=4))
    with pytest.raises(pkg.FrequenSeeError) as e:      # (the held frame is flushed first; then the missing communicator is reported)
        ctx.gather_energy(src)
    assert e.value.code == pkg._capi.ERR_COMM
    ctx.submit()
    ctx.synchronize()
    assert ctx.energy_buffer(src).sum() > 0
    ctx.close()
