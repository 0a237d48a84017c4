"""Multi-GPU partition of the BDPT pairs.

Pairs are independent (GenerateFullPaths' loop, AudioRayTracingSubsystem.cpp:215-230, carries no state
between iterations once the RNG is keyed by the global pair index), and the deposit is a commutative
sum, so rank r of W traces pairs [P*r/W, P*(r+1)/W) and the per-rank [bands][bins] histograms are
summed (one all-reduce of bands*bins floats).  The same split is used inside libfrequensee.so
(fs_config.rank / world_size).
"""
from __future__ import annotations


def pair_range(num_pairs: int, rank: int, world_size: int) -> tuple[int, int]:
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    return (num_pairs * rank) // world_size, (num_pairs * (rank + 1)) // world_size


def all_ranges(num_pairs: int, world_size: int) -> list[tuple[int, int]]:
    return [pair_range(num_pairs, r, world_size) for r in range(world_size)]


def library_pair_range(num_rays: int, rank: int, world_size: int) -> tuple[int, int]:
    """the same rule as libfrequensee.so applies it (fs_shard_range: host code, needs no device)"""
    import ctypes as C

    from . import _capi
    b, n = C.c_uint32(), C.c_uint32()
    rc = _capi.load().fs_shard_range(num_rays, rank, world_size, C.byref(b), C.byref(n))
    if rc != _capi.OK:
        raise ValueError("bad num_rays / rank / world_size")
    return b.value, b.value + n.value
