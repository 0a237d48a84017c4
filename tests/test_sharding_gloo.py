"""N>1 path on CPU: world_size-2 gloo.  Each rank asks libfrequensee.so itself for its share of the pairs
(fs_shard_range: the host code fs_compute_energy_response* partitions a frame with), traces it with the oracle
standing in for the kernels (no device here), the [bands][bins] histograms are all-reduced, and the sum must equal
the single-rank frame.  The library-side collective (fs_comm_*) is covered on the GPU: tests/test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pair_ranges_partition(pkg):
    for P in (0, 1, 7, 512, 131072, 524288):
        for W in (1, 2, 3, 4, 8):
            r = pkg.sharding.all_ranges(P, W)
            assert r[0][0] == 0 and r[-1][1] == P
            assert all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    with pytest.raises(ValueError):
        pkg.sharding.pair_range(10, 2, 2)


def test_library_partition_is_the_documented_one(pkg):
    """fs_shard_range (C ABI, host only) == sharding.pair_range for every rank, and the ranks tile [0, P) exactly"""
    for rays in (0, 2, 14, 2000, 16384, 262144, 1048576, 2 * 777):
        for W in (1, 2, 3, 4, 5, 8):
            got = [pkg.sharding.library_pair_range(rays, r, W) for r in range(W)]
            assert got == pkg.sharding.all_ranges(rays // 2, W)
            assert got[0][0] == 0 and got[-1][1] == rays // 2
    with pytest.raises(ValueError):
        pkg.sharding.library_pair_range(7, 0, 2)      # odd ray count
    with pytest.raises(ValueError):
        pkg.sharding.library_pair_range(8, 2, 2)      # rank out of range


def _worker(rank, world, port, pairs, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = graft.load_package()
    sc = pkg.scenes.by_name("starter_room", 4)
    osc = oracle.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle.default_params(num_pairs=pairs, depth=8, seed=0x5EED)
    a, b = pkg.sharding.library_pair_range(2 * pairs, rank, world)   # the library's own partition of the frame
    assert (a, b) == pkg.sharding.pair_range(pairs, rank, world)
    _, e64, _ = osc.compute_energy(p, sc.source, sc.listener, a, b)
    t = torch.from_numpy(e64.copy())
    dist.all_reduce(t)                      # the energy-buffer sum (RCCL on the GPU box, gloo here)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pairs", [(2, 3000), (3, 3001)])    # (3 ranks, a pair count none of them divides: ragged shares)
def test_world_size_2_gloo_matches_single_rank(pkg, oracle_mod, tmp_path, world, pairs):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(world, port, pairs, out), nprocs=world, join=True)
    sc = pkg.scenes.by_name("starter_room", 4)
    osc = oracle_mod.Scene(sc.triangles, sc.material_ids, sc.absorption)
    p = oracle_mod.default_params(num_pairs=pairs, depth=8, seed=0x5EED)
    _, e64, _ = osc.compute_energy(p, sc.source, sc.listener)
    got = np.load(out)
    assert np.array_equal(got != 0, e64 != 0)
    assert np.allclose(got, e64, rtol=1e-12, atol=0)
