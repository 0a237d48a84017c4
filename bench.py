#!/usr/bin/env python3
"""bench.py — rays/s + IR-frames/s of the MI355X-native FrequenSee BDPT path.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one complete frame of the hot path over one batch of synthetic input:
  fs_compute_energy_response (walk + connect/deposit kernels)  [-> RCCL all-reduce of the energy
  buffer when N>1]  ->  fs_reconstruct_impulse_response (energy -> 1 s / 48 kHz IR, published to host).
Workload at N=1 = BASELINE.json configs[2] (the config the >=10 M rays/s target is quoted on):
Scene_OldMine stand-in (100 000 triangles, procedural), 262 144 rays/frame (source + listener
subpaths), depth 8, 8 bands.  N>1: weak scaling, 262 144 rays per GPU per frame, pairs sharded by
global pair index, one all-reduce of [8][1000] fp32 per frame.

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (walk_kernel, the
dominant kernel; algorithmic bytes from the oracle's counters) and `cpu_baseline` (the oracle timed on
this box's host cores on the same frame).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (scene, bands, rays per GPU per frame, depth)
    "cfg2_starter_room": ("starter_room", 4, 16384, 8),
    "cfg3_old_mine": ("old_mine", 8, 262144, 8),
    "cfg4_old_mine_d12": ("old_mine", 8, 131072, 12),          # cfg4's per-GPU share at 8 GPUs
    "cfg4_old_mine_1m_d12": ("old_mine", 8, 1048576, 12),      # all of cfg4 on one GPU
    # cfg5: 8 sources x 131 072 rays, 1 listener; sources are dealt round-robin to the ranks (one per GPU at
    # N = 8), each on its own context/stream, no reduce; total work is fixed => strong scaling
    "cfg5_multi_source": ("old_mine", 8, 131072, 8),
}
MULTI_SOURCE = {"cfg5_multi_source": 8}


class _CudaArray:
    """expose a foreign device pointer to torch through __cuda_array_interface__ (zero copy)"""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def algorithmic_bytes(cnt, rays, bands):
    """SURVEY.md §8(d) per-frame figure, split per kernel.  cnt = the oracle's counters for
    the frame on its reference BVH2: 32 B per node visit, 48 B per triangle test; per subpath a 24-B terminal
    state (written by walk, read by connect); per walk segment a 12-B record (length, probability, material)
    written once and read once for connected pairs; a deposit = fp32 read-modify-write per band."""
    walk_nodes = cnt["node_visits"] - cnt["any_node_visits"]
    walk_tris = cnt["tri_tests"] - cnt["any_tri_tests"]
    segments = cnt["closest_rays"]
    pairs = max(rays // 2, 1)
    walk = 32 * walk_nodes + 48 * walk_tris + 24 * rays + 12 * segments
    connect = (32 * cnt["any_node_visits"] + 48 * cnt["any_tri_tests"] + 24 * rays
               + 12 * segments * cnt["connected"] // pairs + 8 * bands * cnt["deposits"] + 4 * bands * 1000)
    return walk, connect


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3_old_mine", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fixed-depth", action="store_true", help="Russian roulette off: every subpath takes `depth` segments")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED)
    ap.add_argument("--inflight", type=int, default=None,
                    help="independent frames in flight (one context + HIP stream each); 1 = strictly sequential frames")
    ap.add_argument("--all-connections", action="store_true",
                    help="FS_FLAG_ALL_CONNECTIONS (row f3): every forward prefix x every backward prefix per pair")
    ap.add_argument("--mis-balance", action="store_true",
                    help="FS_FLAG_MIS_BALANCE (row f3): all-connections mode with balance-heuristic weights")
    ap.add_argument("--deterministic", action="store_true",
                    help="FS_FLAG_DETERMINISTIC: u64 fixed-point deposits, integer all-reduce (bit-identical for any N)")
    ap.add_argument("--fixed-seed", action="store_true",
                    help="every frame traces the same sample set (default: a new RNG seed every frame, as an application "
                         "does; the frame checked against the oracle uses --seed itself)")
    ap.add_argument("--no-batch", action="store_true",
                    help="multi-source workloads: one context and one frame per source, all in flight, instead of one "
                         "batched frame over this rank's sources (fs_compute_energy_response_batch_async)")
    ap.add_argument("--no-pipelined", action="store_true",
                    help="skip the extra region that times the same frames with two in flight")
    args = ap.parse_args()

    # stdout carries exactly one line, the JSON result: libraries that print banners to fd 1 (RCCL does at
    # communicator creation) are sent to stderr for the lifetime of the process
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # A context overlaps frame f's tail (reduce, reconstruct, publish) with frame f+1's tracing on two HIP streams.
    # The runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); with RCCL's own streams
    # in the process the two land on one queue and serialise (measured 0.64 vs 0.57 ms per frame).  Must be set
    # before the HIP runtime initialises.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    # rehearsal switches (never set by the driver): run the N>1 control flow on ONE GPU with gloo
    backend = os.environ.get("FS_BENCH_BACKEND", "nccl")
    if os.environ.get("FS_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # rehearsal switch (never set by the driver): take the RCCL all-reduce path with a single rank
    force_reduce = os.environ.get("FS_BENCH_FORCE_REDUCE") == "1"
    if world > 1 or force_reduce:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = graft.load_package()
    scene_name, bands, rays_per_gpu, depth = WORKLOADS[args.workload]
    sc = pkg.scenes.by_name(scene_name, bands)
    n_sources = MULTI_SOURCE.get(args.workload, 0)
    if n_sources:
        # independent sources: rank r owns sources r, r+N, ...; every source traces all of its pairs here
        my_sources = [i for i in range(n_sources) if i % world == rank]
        total_rays = rays_per_gpu * n_sources            # per step, over all ranks
        p = pkg.default_params(num_rays=rays_per_gpu, depth=depth, seed=args.seed,
                               russian_roulette=0 if args.fixed_depth else 1)
        positions = [np.asarray(x, np.float32) for x in sc.extra_sources]   # 8 spots on a ~15 m spacing
        assert len(positions) >= n_sources
    else:
        my_sources = []
        total_rays = rays_per_gpu * world
        p = pkg.default_params(num_rays=total_rays, depth=depth, seed=args.seed,
                               russian_roulette=0 if args.fixed_depth else 1)
    if args.deterministic:
        p.flags |= pkg._capi.FLAG_DETERMINISTIC
    if args.all_connections:
        p.flags |= pkg._capi.FLAG_ALL_CONNECTIONS
    if args.mis_balance:
        p.flags |= pkg._capi.FLAG_MIS_BALANCE
    lanes = []   # one (stream, context, source) per frame in flight; multi-source: one per source of this rank,
    #              the sources dealt round-robin to `--inflight` contexts (each context = one compute + one tail stream)
    ctxs = []
    batch_sources = bool(n_sources) and not args.no_batch and len(my_sources) > 1
    if args.inflight is None:
        # independent sources: one batched frame on one context (measured best), else all sources in flight
        args.inflight = (1 if batch_sources else len(my_sources)) if n_sources else 1
    n_ctx = max(1, args.inflight) if not n_sources else max(1, min(args.inflight, len(my_sources)))
    for i in range(n_ctx):
        st_i = torch.cuda.current_stream() if i == 0 else torch.cuda.Stream()
        if n_sources:
            c = pkg.Context(num_bands=bands, device=local_rank, stream=st_i.cuda_stream)
        else:
            c = pkg.Context(num_bands=bands, device=local_rank, rank=rank, world_size=world, stream=st_i.cuda_stream)
        c.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        c.set_listener(sc.listener)
        ctxs.append((st_i, c))
    if n_sources:
        for j, si in enumerate(my_sources):
            st_i, c = ctxs[j % n_ctx]
            lanes.append((st_i, c, c.create_source(positions[si])))
    else:
        for st_i, c in ctxs:
            lanes.append((st_i, c, c.create_source(sc.source)))
    stream, ctx, src = lanes[0]
    frame_no = [0]
    tensors = {}      # device pointer -> torch view of that energy buffer (a source alternates between two)
    tails = {}        # tail stream handle -> torch stream object

    def frame():
        if n_sources:   # one step = one update of every source (this rank's share), all in flight together
            frame_no[0] += 1
            if not args.fixed_seed:
                p.seed = args.seed + frame_no[0]
            if batch_sources and len(ctxs) == 1:   # all of this rank's sources in one traced frame
                lanes[0][1].compute_energy_response_batch_async([s_i for _, _, s_i in lanes], p)
                for _, c, s_i in lanes:
                    c.reconstruct_impulse_response_async(s_i, p)
                return
            for _, c, s_i in lanes:
                c.compute_energy_response_async(s_i, p)
                c.reconstruct_impulse_response_async(s_i, p)
            return
        st_i, c, s_i = lanes[frame_no[0] % len(lanes)]
        frame_no[0] += 1
        if not args.fixed_seed:
            p.seed = args.seed + frame_no[0]
        c.compute_energy_response_async(s_i, p)
        if (world > 1 or force_reduce) and os.environ.get("FS_BENCH_SKIP_REDUCE") != "2":
            # RCCL sum of the [bands][bins] fp32 energy buffer on the context's tail stream: it and the
            # reconstruct behind it overlap the next frame's tracing on the compute stream
            eptr, ebytes, tail = c.energy_handoff(s_i)
            if eptr not in tensors:   # fp32 energy, or the i64 fixed-point histogram in deterministic mode
                tensors[eptr] = torch.as_tensor(_CudaArray(eptr, (bands * c.num_bins,),
                                                           "<i8" if ebytes == 8 * bands * c.num_bins else "<f4"),
                                                device=f"cuda:{local_rank}")
            if tail not in tails:
                tails[tail] = torch.cuda.ExternalStream(tail, device=f"cuda:{local_rank}")
            if os.environ.get("FS_BENCH_SKIP_REDUCE") != "1":   # rehearsal switch: everything but the collective
                with torch.cuda.stream(tails[tail]):
                    dist.all_reduce(tensors[eptr])
        c.reconstruct_impulse_response_async(s_i, p)

    def barrier():
        if world > 1 or force_reduce:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame()
    for _, c in ctxs:
        c.synchronize()
        c.reset_stats()
        c.set_profiling(1)                     # HIP events around the dominant kernel on its launch stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    st = None
    for _, c in ctxs:
        c.synchronize()
        s1 = c.stats()
        c.set_profiling(0)
        if st is None:
            st = s1
        else:
            for k in ("walk_kernel_ms_sum", "timed_frames", "segments", "connections_tested", "deposits"):
                st[k] += s1[k]
    work = [float(st["segments"]), float(st["connections_tested"]), float(st["deposits"])]   # device-side counters
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        w = torch.tensor(work, dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(w)
        work = [float(x) for x in w.tolist()]

    # untimed: the frame the oracle leg checks (seed = --seed), then a few sequential frames with events around
    # EVERY kernel for the per-kernel breakdown
    p.seed = args.seed
    ctx.compute_energy_response_async(src, p)
    ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    e_gpu = ctx.energy_buffer(src)
    ir = ctx.impulse_response(src, 0)
    ctx.reset_stats()
    ctx.set_profiling(2)
    for i in range(10):
        if not args.fixed_seed:
            p.seed = args.seed + 1000003 + i
        ctx.compute_energy_response_async(src, p)
        ctx.reconstruct_impulse_response_async(src, p)
    ctx.synchronize()
    p.seed = args.seed
    st_all = ctx.stats()
    ctx.set_profiling(0)

    result = None
    if rank == 0:
        rays_s = total_rays * args.steps / elapsed
        ms_step = 1e3 * elapsed / args.steps
        walk_ms = st["walk_kernel_ms_sum"] / max(st["timed_frames"], 1)
        conn_ms = st_all["connect_kernel_ms_sum"] / max(st_all["timed_connects"], 1)
        rec_ms = st_all["reconstruct_ms_sum"] / max(st_all["timed_reconstructs"], 1)
        result = {
            "metric": "rays/sec + IR-frames/sec (1s IR, depth 8)",
            "value": rays_s,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "strong" if n_sources else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" + ("" if not args.fixed_seed else " (same sample set every frame)"),
            "config": {"workload": f"{args.workload}: {scene_name} {sc.num_triangles} tris, {rays_per_gpu} rays/frame/GPU "
                                   f"(source+listener subpaths), depth {depth}, {bands} bands, "
                                   f"{'fixed depth' if args.fixed_depth else 'Russian roulette 0.9'}",
                       "rays_per_frame": total_rays, "pairs_per_frame": total_rays // 2, "depth": depth,
                       "bands": bands, "triangles": sc.num_triangles,
                       "sharding": f"{n_sources} sources round-robin over {world} ranks" if n_sources else f"pairs/{world}",
                       "frames_in_flight": len(ctxs),
                       **({"sources_per_batched_frame": len(lanes)} if batch_sources and len(ctxs) == 1 else {})},
            "ir_frames_per_s": args.steps / elapsed,
            # the same rate in the other units one might mean by "rays" (device-side counters)
            "pairs_per_s": total_rays / 2 * args.steps / elapsed,
            "segments_per_s": (work[0] + work[1]) / elapsed,            # closest-hit + any-hit queries
            "contributions_per_s": work[2] / elapsed,                  # unobstructed connections deposited
            "kernel_ms": {"walk": walk_ms, "connect": conn_ms, "reconstruct+publish": rec_ms},
        }

    # ---- extra region (N=1): the same frames with TWO in flight on separate HIP streams — one cfg3 frame
    #      cannot fill the chip (occupancy decays as walks end); independent frames/sources overlap ------------
    if world == 1 and len(lanes) == 1 and not n_sources and not args.no_pipelined:
        st2 = torch.cuda.Stream()
        c2 = pkg.Context(num_bands=bands, device=local_rank, stream=st2.cuda_stream)
        c2.set_scene(sc.triangles, sc.material_ids, sc.absorption)
        c2.set_listener(sc.listener)
        s2 = c2.create_source(sc.source)
        both = [(ctx, src), (c2, s2)]
        k2 = max(2, args.steps)
        for i in range(6):
            c, s_i = both[i % 2]
            c.compute_energy_response_async(s_i, p)
            c.reconstruct_impulse_response_async(s_i, p)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(k2):
            c, s_i = both[i % 2]
            if not args.fixed_seed:
                p.seed = args.seed + 2000003 + i
            c.compute_energy_response_async(s_i, p)
            c.reconstruct_impulse_response_async(s_i, p)
        torch.cuda.synchronize()
        el2 = time.perf_counter() - t1
        p.seed = args.seed
        c2.close()
        result["pipelined"] = {"frames_in_flight": 2, "steps": k2, "value": total_rays * k2 / el2, "unit": "rays/s",
                               "ms_per_step": 1e3 * el2 / k2, "ir_frames_per_s": k2 / el2}

    # ---- oracle leg (rank 0, N=1 only): parity check, algorithmic bytes, CPU baseline ---------------------
    if rank == 0 and world == 1 and not n_sources and not args.no_cpu_baseline:
        import oracle  # checker / CPU baseline only

        lib = None
        kind_note = "portable build"
        try:
            so = oracle.build(native=True, outdir=os.path.join("/tmp", f"fs_oracle_{os.getpid()}"))
            lib = oracle.load(so)
            kind_note = "-O3 -march=native build"
        except Exception:
            lib = oracle.load()
        osc = oracle.Scene(sc.triangles, sc.material_ids, sc.absorption, lib=lib)
        oflags = (oracle.FLAG_ALL_CONNECTIONS if args.all_connections else 0) | \
            (oracle.FLAG_MIS_BALANCE if args.mis_balance else 0)
        op = oracle.default_params(num_pairs=total_rays // 2, depth=depth, seed=args.seed, flags=oflags,
                                   russian_roulette=0 if args.fixed_depth else 1)
        # 1 thread on a bounded sample (the first 1/8 of the frame's pairs)
        sample_pairs = max(1, (total_rays // 2) // 8)
        t1 = time.perf_counter()
        osc.compute_energy(op, sc.source, sc.listener, 0, sample_pairs)
        t_1t = time.perf_counter() - t1
        # all host cores: the frame itself (counters + parity reference), then more frames with other seeds so
        # that the timed sample is ~10-30 s of CPU work
        cores = min(os.cpu_count() or 1, 16)
        t1 = time.perf_counter()
        e32, e64, cnt = osc.compute_energy_mt(op, sc.source, sc.listener, cores)
        t_first = time.perf_counter() - t1
        extra_frames = int(max(0, min(15, round(12.0 / max(t_first * cores, 1e-3)) - 1)))
        t1 = time.perf_counter()
        for i in range(extra_frames):
            op_i = oracle.default_params(num_pairs=total_rays // 2, depth=depth, seed=args.seed + 1 + i, flags=oflags,
                                         russian_roulette=0 if args.fixed_depth else 1)
            osc.compute_energy_mt(op_i, sc.source, sc.listener, cores)
        t_mt = (t_first + (time.perf_counter() - t1)) / (1 + extra_frames)
        cnt = cnt.as_dict()
        rms = [float(np.sqrt(np.mean((e_gpu[b].astype(np.float64) - e64[b]) ** 2)) /
                     max(np.sqrt(np.mean(e64[b] ** 2)), 1e-300)) for b in range(bands)]
        if max(rms) > 1e-3:
            raise SystemExit(f"parity failure: relative RMS per band {rms}")
        wb, cb = algorithmic_bytes(cnt, total_rays, bands)
        walk_s = 1e-3 * result["kernel_ms"]["walk"]
        achieved = wb / walk_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.workload, {}).get("walk_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result["roofline"] = {"bound": "hbm", "kernel": "walk_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                              "algorithmic_bytes_per_launch": wb, "avg_launch_ms": result["kernel_ms"]["walk"],
                              "frame_algorithmic_bytes": wb + cb,
                              "frame_achieved_GBs": (wb + cb) / (1e-3 * (result["kernel_ms"]["walk"] +
                                                                        result["kernel_ms"]["connect"])) / 1e9,
                              "note": "achieved = SURVEY 8(d) algorithmic bytes (oracle BVH2 node/triangle fetches) / "
                                      "launch time; the tree is served from L2 / Infinity Cache, so frac can exceed 1 — "
                                      "`traffic` is the HBM traffic measured by PMC (profiles/traffic.json); the "
                                      "kernel is bound by VALU issue, DESIGN.md section 5"}
        result["cpu_baseline"] = {"value": total_rays / t_mt, "unit": "rays/s", "cores": cores, "kind": "port",
                                  "sample": f"{1 + extra_frames} full frames ({total_rays} rays each) on {cores} threads "
                                            f"(~{t_mt * (1 + extra_frames) * cores:.0f} s of CPU work), oracle {kind_note}; "
                                            f"1 thread on the first {2 * sample_pairs} rays: "
                                            f"{2 * sample_pairs / t_1t:.0f} rays/s",
                                  "value_1_thread": 2 * sample_pairs / t_1t}
        result["parity"] = {"max_rel_rms_per_band": max(rms), "connected_pairs": cnt["connected"],
                            "connections_tested": cnt["any_rays"],
                            "segments": cnt["closest_rays"], "node_visits": cnt["node_visits"],
                            "tri_tests": cnt["tri_tests"]}
    if rank == 0:
        result_out.write(json.dumps(result) + "\n")
        result_out.flush()
    for _, c in ctxs:
        c.close()
    if world > 1 or force_reduce:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
