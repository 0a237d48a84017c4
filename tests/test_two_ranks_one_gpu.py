"""Two ranks of the library's multi-GPU protocol on ONE GPU ("a 2-rank GPU test that compares every rank's IR
with the single-rank IR").  Real RCCL refuses two ranks on one device, so the ranks' collectives go through the test
double tests/fake_rccl.cpp ($FS_RCCL_LIB) — everything above it is the product: fs_comm_unique_id / fs_comm_init,
rank 0 building the tree and fs_scene_commit broadcasting it, each rank tracing the library's own partition of the
pairs, the per-frame all-reduce on the tail stream (fp32 and the deterministic mode's uint64), reconstruct + publish.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IR_TOL = 2e-5


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "fake_rccl.cpp"), "-o", str(out), "-L/opt/rocm/lib", "-lamdhip64", "-lrt",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return str(out)


def test_fake_rccl_builds_and_exports_what_the_library_opens(fake_rccl):
    """CPU tier: the test double compiles and carries every symbol RcclApi (csrc/fs_capi.cpp) looks up."""
    syms = subprocess.run(["nm", "-D", "--defined-only", fake_rccl], check=True, capture_output=True, text=True).stdout
    for s in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclCommCount", "ncclCommUserRank",
              "ncclAllReduce", "ncclBroadcast", "ncclAllGather", "ncclGetErrorString"):
        assert f" T {s}" in syms, s


@pytest.mark.gpu
@pytest.mark.parametrize("world,pipelined,oneshot,fpl", [(2, False, False, 1), (3, False, False, 1), (2, True, False, 1), (2, False, True, 1),
                                                         (3, True, True, 1), (2, True, False, 3), (2, True, True, 2)],
                         ids=["two", "three", "two_pipelined", "two_oneshot", "three_pipelined_oneshot", "two_pipelined_3_per_launch",
                              "two_pipelined_oneshot_2_per_launch"])
def test_every_rank_publishes_the_single_rank_ir(pkg, fake_rccl, tmp_path, world, pipelined, oneshot, fpl):
    """oneshot: fs_comm_enable_oneshot — the per-frame sum goes through the ranks' HIP-IPC mailboxes (one peer-write
    exchange, fs_oneshot.hip) instead of the communicator's all-reduce; the communicator still carries the scene broadcast
    and the handle exchange.  Everything a rank publishes must be what the all-reduce path publishes."""
    # (FS_ONESHOT_COARSE_OK: the library insists on fine-grained mailbox memory — peer writes over xGMI into coarse-grained
    # memory need not be visible to the polling kernel; ranks that share ONE device, as here, see one L2 and may use plain memory
    # where the runtime cannot share a fine-grained allocation between processes)
    env = dict(os.environ, FS_RCCL_LIB=fake_rccl, FAKE_RCCL_TIMEOUT_S="60", FS_TEST_PIPELINE="1" if pipelined else "0",
               FS_TEST_ONESHOT="1" if oneshot else "0", FS_TEST_FPL=str(fpl), FS_ONESHOT_COARSE_OK="1")
    id_file = str(tmp_path / "comm_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(r), str(world),
                               id_file, str(tmp_path / f"rank{r}.npz")], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for pr in procs:
        try:
            logs.append(pr.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]

    # the single-rank answer, in this process, through the same library without a communicator
    sc = pkg.scenes.starter_room(4)
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    src = ctx.create_source(sc.source)
    assert all(int(rk["bvh_nodes"]) == ctx.stats()["bvh_nodes"] for rk in ranks)      # the broadcast tree is rank 0's tree
    for k, (flags, seed) in enumerate([(0, 901), (8, 902), (0, 903)]):
        p = pkg.default_params(num_rays=16384, depth=8, seed=seed, flags=flags)
        ctx.compute_energy_response(src, p)
        ctx.reconstruct_impulse_response(src, p)
        want = ctx.impulse_response(src, 0)
        assert np.abs(want).max() > 0
        for r, rk in enumerate(ranks):
            got = rk[f"ir{k}"]
            if flags & 8:
                assert np.array_equal(got, want), (k, r)              # integer sums: bit-identical on every rank
            else:
                assert np.abs(got - want).max() <= IR_TOL * np.abs(want).max(), (k, r)
            assert np.array_equal(got, ranks[0][f"ir{k}"]), (k, r)    # and all ranks publish the same samples
    for seed in (911, 912, 913, 914):   # the run of same-kind frames (grouped on the ranks when fpl > 1)
        pg = pkg.default_params(num_rays=8192, depth=8, seed=seed, flags=8)
        ctx.compute_energy_response(src, pg)
        ctx.reconstruct_impulse_response(src, pg)
    for r, rk in enumerate(ranks):
        assert np.array_equal(rk["energy_run"], ctx.energy_buffer(src)), r
        assert np.array_equal(rk["ir_run"], ctx.impulse_response(src, 0)), r
    p = pkg.default_params(num_rays=16384, depth=8, seed=77)
    want_e = ctx.compute_energy_response(src, p)
    rng = np.random.default_rng(5)
    o = np.tile(np.asarray(sc.source, np.float32), (256, 1))
    d = rng.normal(size=(256, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    hit, t, idx, _ = ctx.trace_rays(o, d, 1e6)
    ctx.update_triangles(0, np.asarray(sc.triangles[:64], np.float32) + np.float32(3.0))
    want_moved = ctx.compute_energy_response(src, p)
    assert not np.array_equal(want_moved, want_e)
    for r, rk in enumerate(ranks):
        for got, want in ((rk["energy"], want_e), (rk["energy_moved"], want_moved)):
            assert np.array_equal(got != 0, want != 0), r
            num = np.sqrt(((got.astype(np.float64) - want) ** 2).sum(axis=1))
            assert (num <= 2e-5 * np.sqrt((want.astype(np.float64) ** 2).sum(axis=1))).all(), r
        assert np.array_equal(rk["hit"], hit) and np.array_equal(rk["t"], t) and np.array_equal(rk["idx"], idx), r
    ctx.close()


@pytest.mark.gpu
def test_one_source_per_rank_and_any_rank_serves_any_source(pkg, fake_rccl, tmp_path):
    """cfg5's arrangement: every rank owns one source, nothing of a frame is sharded or reduced; the peer
    communicator's all-gather hands every rank all histograms, and a histogram installed on a mirror source
    reconstructs to exactly the IR its owner publishes."""
    world = 3
    env = dict(os.environ, FS_RCCL_LIB=fake_rccl, FAKE_RCCL_TIMEOUT_S="60")
    id_file = str(tmp_path / "comm_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_two_rank_worker.py"), str(r), str(world),
                               id_file, str(tmp_path / f"g{r}.npz"), "gather"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for pr in procs:
        try:
            logs.append(pr.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    ranks = [np.load(tmp_path / f"g{r}.npz") for r in range(world)]
    sc = pkg.scenes.starter_room(4)
    ctx = pkg.Context(num_bands=4)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    p = pkg.default_params(num_rays=8192, depth=8, seed=41, flags=8)       # the gathered frame (deterministic: exact)
    want = []
    for r in range(world):
        pos = np.asarray(sc.source, np.float32) + np.float32(60.0 * r) * np.array([1, -1, 0], np.float32)
        src = ctx.create_source(pos)
        want.append(ctx.compute_energy_response(src, p).copy())
    assert not np.array_equal(want[0], want[1])                            # the sources really differ
    for r, rk in enumerate(ranks):
        g = rk["gathered"]
        assert g.shape == (world, 4, ctx.num_bins)
        for q in range(world):
            assert np.array_equal(g[q], want[q]), (r, q)                  # every rank holds every source's histogram
        assert np.abs(rk["own_ir"]).max() > 0
        assert np.array_equal(rk["peer_ir"], ranks[(r + 1) % world]["own_ir"]), r   # ... and serves the peer's IR
    ctx.close()
