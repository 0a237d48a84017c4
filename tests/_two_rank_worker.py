"""One rank of tests/test_two_ranks_one_gpu.py (not collected by pytest): a world_size-N context on cuda:0 whose
collectives go through the test double tests/fake_rccl.cpp ($FS_RCCL_LIB), driven exactly like a rank of a real run.
usage: python tests/_two_rank_worker.py <rank> <world_size> <id_file> <out.npz>"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

rank, world, id_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
mode = sys.argv[5] if len(sys.argv) > 5 else "shard"
pkg = graft.load_package()
assert os.environ.get("FS_RCCL_LIB"), "the worker must run on the test double"
sc = pkg.scenes.starter_room(4)
ctx = pkg.Context(num_bands=4, rank=rank, world_size=world) if mode == "shard" else pkg.Context(num_bands=4)
if rank == 0:                                   # the id travels by any host-side transport: here a file
    uid = pkg.Context.comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        assert time.time() - t0 < 60, "rank 0 never wrote the id"
        time.sleep(0.01)
    with open(id_file, "rb") as f:
        uid = f.read()
if mode == "gather":
    # cfg5's arrangement: every rank owns ONE source and traces all of its pairs; the peers only exchange histograms
    ctx.peers_init(uid, rank, world)
    ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
    ctx.set_listener(sc.listener)
    my_pos = np.asarray(sc.source, np.float32) + np.float32(60.0 * rank) * np.array([1, -1, 0], np.float32)
    src = ctx.create_source(my_pos)
    res = {}
    for k, flags in enumerate((0, 8)):
        p = pkg.default_params(num_rays=8192, depth=8, seed=40 + k, flags=flags)
        ctx.compute_energy_response_async(src, p)
        ctx.reconstruct_impulse_response_async(src, p)
        if k == 0:                                     # the next frame is already tracing while the gather runs
            ctx.compute_energy_response_async(src, pkg.default_params(num_rays=2048, depth=4, seed=9))
            ctx.reconstruct_impulse_response_async(src, pkg.default_params(num_rays=2048, depth=4, seed=9))
        # note: gather_energy returns the CURRENT frame's histograms: gather right behind the frame that matters
    p = pkg.default_params(num_rays=8192, depth=8, seed=41, flags=8)
    ctx.compute_energy_response_async(src, p)
    ctx.reconstruct_impulse_response_async(src, p)
    allE = ctx.gather_energy(src)                      # collective: [world][B][bins]
    ctx.synchronize()
    res["gathered"] = allE
    res["own_ir"] = ctx.impulse_response(src, 0).copy()
    # serve the NEXT rank's source from here: install its histogram on a mirror source and reconstruct
    peer = (rank + 1) % world
    mirror = ctx.create_source(my_pos)
    ctx.update_energy_buffer(mirror, allE[peer])
    ctx.reconstruct_impulse_response(mirror, p)
    res["peer_ir"] = ctx.impulse_response(mirror, 0).copy()
    ctx.peers_detach()
    ctx.close()
    np.savez(out, **res)
    sys.exit(0)
assert ctx.comm_info() == (0, -1, 0)                                  # fs_comm_info without a communicator
ctx.comm_init(uid)
assert ctx.comm_info() == (world, rank, 1), ctx.comm_info()           # what the communicator itself says (ncclCommCount / UserRank), all-reduce
if os.environ.get("FS_TEST_ONESHOT") == "1":
    ctx.comm_enable_oneshot()     # the sum over the ranks through the peers' IPC-mapped mailboxes instead of ncclAllReduce
    assert ctx.comm_info() == (world, rank, 2), ctx.comm_info()
if os.environ.get("FS_TEST_PIPELINE") == "1":
    ctx.set_pipelining(2)         # held-back connect passes: the all-reduce and the reconstruct follow them
FPL = int(os.environ.get("FS_TEST_FPL", "1"))
if FPL > 1:
    ctx.set_frames_per_launch(FPL)   # same-kind frames share a launch; each item is reduced and reconstructed on its own
# every rank registers the same triangles; rank 0 builds the tree, the others receive it (fs_scene_commit)
ctx.set_scene(sc.triangles, sc.material_ids, sc.absorption)
ctx.set_listener(sc.listener)
src = ctx.create_source(sc.source)
res = {"bvh_nodes": np.int64(ctx.stats()["bvh_nodes"])}
frames = [(0, 901), (8, 902), (0, 903)]         # (flags, seed): fp32 frame, deterministic frame, fp32 again
for k, (flags, seed) in enumerate(frames):
    p = pkg.default_params(num_rays=16384, depth=8, seed=seed, flags=flags)
    ctx.compute_energy_response_async(src, p)
    ctx.reconstruct_impulse_response_async(src, p)
    if k + 1 < len(frames):                     # the next frame's tracing is queued behind this frame's reduce
        ctx.compute_energy_response_async(src, pkg.default_params(num_rays=4096, depth=8, seed=5))
    ctx.synchronize()
    res[f"ir{k}"] = ctx.impulse_response(src, 0).copy()
# a run of same-kind frames (what fs_set_frames_per_launch groups): every frame's sum over the ranks and IR
for k, seed in enumerate((911, 912, 913, 914)):
    pg = pkg.default_params(num_rays=8192, depth=8, seed=seed, flags=8)
    ctx.compute_energy_response_async(src, pg)
    ctx.reconstruct_impulse_response_async(src, pg)
ctx.synchronize()
res["ir_run"] = ctx.impulse_response(src, 0).copy()
res["energy_run"] = ctx.energy_buffer(src).copy()
p = pkg.default_params(num_rays=16384, depth=8, seed=77)
res["energy"] = ctx.compute_energy_response(src, p).copy()        # the helpers read the summed buffer
rng = np.random.default_rng(5)
o = np.tile(np.asarray(sc.source, np.float32), (256, 1))
d = rng.normal(size=(256, 3)).astype(np.float32)
d /= np.linalg.norm(d, axis=1, keepdims=True)
hit, t, idx, nrm = ctx.trace_rays(o, d, 1e6)                      # through the broadcast tree
res.update(hit=hit, t=t, idx=idx)
# moving geometry on a sharded context: every rank refits its own copy
ctx.update_triangles(0, np.asarray(sc.triangles[:64], np.float32) + np.float32(3.0))
res["energy_moved"] = ctx.compute_energy_response(src, p).copy()
ctx.comm_detach()
ctx.close()
np.savez(out, **res)
