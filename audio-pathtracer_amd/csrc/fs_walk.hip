// fs_walk.hip — plan pass and walk kernels (GeneratePath, AudioRayTracingSubsystem.cpp:279-355) + their launchers.
//   plan_kernel        the length of a subpath under Russian roulette depends only on the RNG stream: bucket the
//                      subpaths by length before tracing, so that every walk wave holds equal-length walks;
//                      also FlushEnergyBuffer (:157-161).
//   walk_kernel_*      _shared: one subpath per lane, length-sorted schedule, closest-hit queries shared within the wave
//                      (default);  _sparse: the same on waves that own only a few subpaths (small frames).
#include "fs_device.hpp"

namespace fs {
namespace {

__global__ __launch_bounds__(kBlock) void plan_kernel(KParams kp, unsigned* __restrict__ scratch,
                                                      uint32_t* __restrict__ perm, float* __restrict__ energy,
                                                      int energy_words, float* const* __restrict__ energy_tab,
                                                      int energy_count) {
    if (kp.plan_coop) plan_coop_body(blockIdx.x, gridDim.x, kp, scratch, perm, energy, energy_words, energy_tab, energy_count);
    else plan_body(blockIdx.x, gridDim.x, kp, scratch, perm, energy, energy_words, energy_tab, energy_count);
}

#ifdef FS_EXPERIMENTS   // diagnostic builds only (-DFS_TRAV_STATS, tests/trav_stats.py): no work sharing, per-lane step counts
// ---------------------------------------------------------------------------------------------------
// walk_kernel_simple: one subpath per lane (reference variant)
// ---------------------------------------------------------------------------------------------------
template <int LOBES>
__global__ __launch_bounds__(kBlock) void walk_kernel_simple(DeviceScene sc, KParams kp, SubpathState st,
                                                             const unsigned* __restrict__ scratch,
                                                             const uint32_t* __restrict__ perm) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock]
    int* s_stack = s_dyn;
    __shared__ unsigned s_cnt[kPlanBuckets];
    if (perm) {   // wave-uniform: bucket counts of the plan pass
        for (int i = threadIdx.x; i <= min(kp.depth, FS_MAX_DEPTH); i += kBlock) s_cnt[i] = scratch[1 + i];
        __syncthreads();
    }
    const uint32_t slot = blockIdx.x * kBlock + threadIdx.x;
    if (slot >= 2u * kp.num_local) return;
    // length-sorted schedule (plan pass) or identity
    const uint32_t g = perm ? planned_subpath(slot, min(kp.depth, FS_MAX_DEPTH), 2u * kp.num_local, s_cnt, perm) : slot;
    int* stack = &s_stack[threadIdx.x];
    Walker w;
    walker_start(w, g, slot, kp, st);
    Ray ray;
    while (walker_next_ray<LOBES>(w, kp, sc, st, ray)) {
        Trav T;
        trav_init(T, kp.max_trace_dist, sc.num_nodes > 0);
        trav_deep_reset(sc, stack);
#ifdef FS_TRAV_STATS
        const unsigned steps = (unsigned)trav_run<false>(sc, ray, T, stack);
        if (g_step_buf) g_step_buf[(size_t)w.k * (2u * (size_t)kp.num_local) + w.g] = (unsigned short)steps;
#else
        trav_run<false>(sc, ray, T, stack);
#endif
        walker_apply_hit(w, kp, sc, st, ray, T);
    }
    walker_finish(w, st);
}
#endif

template <int LOBES, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(kBlock) void walk_kernel_shared(DeviceScene sc, KParams kp, SubpathState st,
                                                             const unsigned* __restrict__ scratch,
                                                             const uint32_t* __restrict__ perm, WalkStage stage) {
    walk_shared_body<LOBES, COUNT, EXT>(blockIdx.x, sc, kp, st, scratch, perm, stage);
}

template <int LOBES, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(kBlock) void walk_kernel_sparse(DeviceScene sc, KParams kp, SubpathState st,
                                                             const unsigned* __restrict__ scratch,
                                                             const uint32_t* __restrict__ perm, int rays_per_wave,
                                                             WalkStage stage, WalkLane lane) {
    walk_sparse_body<LOBES, COUNT, EXT>(blockIdx.x, sc, kp, st, scratch, perm, rays_per_wave, stage, lane);
}

template <int LOBES, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(kBlock) void walk_kernel_coop(DeviceScene sc, CoopView cv, KParams kp, SubpathState st,
                                                           const unsigned* __restrict__ scratch,
                                                           const uint32_t* __restrict__ perm, int rays_per_wave,
                                                           WalkStage stage, WalkLane lane) {
    walk_coop_body<LOBES, COUNT, EXT>(blockIdx.x, sc, cv, kp, st, scratch, perm, rays_per_wave, stage, lane);
}
// the default instantiation with eight waves per workgroup: one workgroup per CU keeps sixteen hundred more resident
// records in its LDS than two workgroups of four waves could (fs_device.hpp: coop_lds_bytes)
constexpr int kCoopBigWaves = 8;
__global__ __launch_bounds__(64 * kCoopBigWaves) void walk_kernel_coop_big(DeviceScene sc, CoopView cv, KParams kp, SubpathState st,
                                                                           const unsigned* __restrict__ scratch,
                                                                           const uint32_t* __restrict__ perm, int rays_per_wave,
                                                                           WalkStage stage, WalkLane lane) {
    walk_coop_body<0, false, false>(blockIdx.x, sc, cv, kp, st, scratch, perm, rays_per_wave, stage, lane);
}

// A first stage with a long-walk lane (WalkLane): the first lane_blocks workgroups are cooperative waves of ONE walk each
// — the frame's longest walks, whose chain of queries sets the frame's time, at the cooperative search's latency from step 0 on —
// the others walk everybody else on sparse waves as before.  One launch: the two run side by side.
template <int LOBES, bool COUNT, bool EXT = false>
__global__ __launch_bounds__(kBlock) void walk_kernel_lane(DeviceScene sc, CoopView cv, KParams kp, SubpathState st,
                                                           const unsigned* __restrict__ scratch, const uint32_t* __restrict__ perm,
                                                           int rays_per_wave, WalkStage stage, WalkLane lane, uint32_t lane_blocks) {
    if (blockIdx.x < lane_blocks) {
        lane.mode = kLaneOnly;
        __builtin_amdgcn_s_setprio(3);   // (the chain's waves ahead of the first stage's at the issue port: 1 % of the 262 144-ray frame)
        walk_coop_body<LOBES, COUNT, EXT>(blockIdx.x, sc, cv, kp, st, scratch, perm, 1, stage, lane);
    } else {
        lane.mode = kLaneSkip;
        walk_sparse_body<LOBES, COUNT, EXT>(blockIdx.x - lane_blocks, sc, kp, st, scratch, perm, rays_per_wave, stage, lane);
    }
}

}  // namespace

const uint32_t* launch_plan(const KParams& kp, const WalkLaunch& wl, float* energy, int energy_words,
                            float* const* energy_tab, int energy_count, hipStream_t s) {
    // Without roulette every walk takes kp.depth segments: nothing to sort, and the caller counts the segments on
    // the host.  With roulette the pass always runs — it is also what counts the frame's walk segments — but it
    // only produces the length-sorted schedule when that is enabled and can matter.
    uint32_t full = 0;
    bool sort = false;
    if (!plan_shape(kp, wl, &full, &sort)) return nullptr;
    hipLaunchKernelGGL(plan_kernel, dim3(full), dim3(kBlock), 0, s, kp, wl.queue_head, sort ? wl.perm : nullptr, energy,
                       energy_words, energy_tab, energy_count);
    return sort ? wl.perm : nullptr;
}

bool plan_shape(const KParams& kp, const WalkLaunch& wl, uint32_t* blocks, bool* sort) {
    const uint32_t lanes = 2u * kp.num_local;
    if (lanes == 0 || !kp.russian_roulette) return false;
    if (blocks) *blocks = kp.plan_coop ? (lanes + (kBlock / 64) * plan_coop_items(lanes) - 1) / ((kBlock / 64) * plan_coop_items(lanes))   // a wave per 8 .. 32 subpaths
                                                : (lanes + kBlock * kPlanItems - 1) / (kBlock * kPlanItems);
    if (sort) *sort = wl.plan && kp.depth > 1 && wl.perm;
    return true;
}

uint32_t walk_stage_slots(const KParams& kp, int begin) {
    const uint32_t lanes = 2u * kp.num_local;
    if (begin <= 0 || !kp.russian_roulette) return lanes;
    // (a stage that begins beyond FS_MAX_DEPTH visits every slot of the last schedule bucket — all walks of FS_MAX_DEPTH
    // steps or more; pricing it at rr^begin overflowed frames of 900 000+ rays)
    const double expect = (double)lanes * std::pow((double)kp.rr_prob, (double)std::min(begin, FS_MAX_DEPTH));
    return (uint32_t)std::min<double>((double)lanes, (double)std::max(kp.stage_margin, 1.3f) * expect + 1024.0);
}

bool walk_lane_possible(const DeviceScene& sc, const KParams& kp, const WalkLaunch& first, const WalkLaunch& late, const uint32_t* perm) {
    if (!perm || !kp.russian_roulette || kp.count) return false;
    if (!FS_SHARED_WALK(first) || !first.coop || first.rays_per_wave <= 0 || first.rays_per_wave >= 64) return false;
    if (!FS_SHARED_WALK(late) || !late.coop || !coop_rays_per_wave(late.rays_per_wave)) return false;
    return coop_view(sc, 1) != nullptr && coop_view(sc, late.rays_per_wave) != nullptr;
}

void launch_walk(const DeviceScene& sc_in, const KParams& kp, const SubpathState& st, const WalkLaunch& wl,
                 const uint32_t* perm, hipStream_t s, const WalkStage& stage_in, const WalkLane& lane_in) {
    DeviceScene sc = sc_in;
    WalkStage stage = stage_in;
    WalkLane lane = lane_in;
    if (stage.begin > 0 && stage.slots_cap == 0xFFFFFFFFu) stage.slots_cap = walk_stage_slots(kp, stage.begin);
    uint32_t lanes = stage.begin > 0 ? stage.slots_cap : 2u * kp.num_local;   // a later stage only has lanes for the walks still alive
    if (lanes == 0) return;
    uint32_t full = (lanes + kBlock - 1) / kBlock;
    const bool shared = FS_SHARED_WALK(wl);
    // record-fetch counting (fs_set_profiling level 3) exists for the default walk only (no lobes)
#define FS_LAUNCH_WALK(K, GRID, ...)                                                                        \
    do {                                                                                                    \
        if (kp.dpos || kp.ignore_on || kp.listener_radius > 0.0f || kp.source_radius > 0.0f) { allow_lds(K<-1, false, true>, lds_ext); hipLaunchKernelGGL((K<-1, false, true>), dim3(GRID), dim3(kBlock), lds_ext, s, __VA_ARGS__); } /* (lobes: read from kp.lobes) */ \
        else if (kp.lobes) { allow_lds(K<1, false>, lds); hipLaunchKernelGGL((K<1, false>), dim3(GRID), dim3(kBlock), lds, s, __VA_ARGS__); }   \
        else if (kp.count) { allow_lds(K<0, true>, lds); hipLaunchKernelGGL((K<0, true>), dim3(GRID), dim3(kBlock), lds, s, __VA_ARGS__); }     \
        else { allow_lds(K<0, false>, lds); hipLaunchKernelGGL((K<0, false>), dim3(GRID), dim3(kBlock), lds, s, __VA_ARGS__); }                 \
    } while (0)
    if (lane.len > 0 && lane.mode == kLaneSplit) {   // a first stage on sparse waves + the long-walk lane on cooperative ones
        const CoopView* lv = shared && wl.coop && perm && wl.rays_per_wave > 0 && wl.rays_per_wave < 64 ? coop_view(sc, 1) : nullptr;
        if (lv && lane.cap > 0) {
            CoopView cv = *lv;
            const size_t lds_walk = stack_bytes(sc) + kShareLdsBytes, lds_walk_ext = stack_bytes(sc) + kShareIgnLdsBytes;
            const uint32_t lane_blocks = (lane.cap + kBlock / 64 - 1) / (kBlock / 64);
            const uint32_t waves = (lanes + (uint32_t)wl.rays_per_wave - 1) / (uint32_t)wl.rays_per_wave;
            const uint32_t blocks = lane_blocks + (waves + kBlock / 64 - 1) / (kBlock / 64);
            // the lane's resident records: what fits into the LDS the sparse workgroups of the launch take anyway
            const size_t lane_fixed = kCoopWaveBytes * (size_t)(kBlock / 64);
            cv.lds_nodes = lds_walk > lane_fixed ? (int)std::min<size_t>((size_t)std::max(cv.nodes, 0), (lds_walk - lane_fixed) / ((size_t)16 << cv.wshift)) : 0;
            const size_t lds = std::max(lds_walk, coop_lds_bytes(kBlock / 64, cv)), lds_ext = std::max(lds_walk_ext, coop_lds_bytes(kBlock / 64, cv));
            if (!attach_deep(sc, blocks)) return;
            FS_LAUNCH_WALK(walk_kernel_lane, blocks, sc, cv, kp, st, wl.queue_head, perm, wl.rays_per_wave, stage, lane, lane_blocks);
            return;
        }
        lane.len = 0;   // (no cooperative view of this tree, or dense waves: everybody walks the stage as before)
    }
    const CoopView* cvp = shared && wl.coop && coop_rays_per_wave(wl.rays_per_wave) ? coop_view(sc, wl.rays_per_wave) : nullptr;
    if (cvp) {   // a handful of subpaths per wave: every query searched by a whole group of lanes
        CoopView cv = *cvp;
        const uint32_t waves = (lanes + (uint32_t)wl.rays_per_wave - 1) / (uint32_t)wl.rays_per_wave;
        const bool plain = !(kp.dpos || kp.ignore_on || kp.listener_radius > 0.0f || kp.source_radius > 0.0f || kp.lobes || kp.count);
        // workgroups of eight waves when the frame needs more than four waves per CU (and the instantiation exists)
        const int W = plain && waves > 4u * (uint32_t)std::max(wl.num_cus, 1) ? kCoopBigWaves : kBlock / 64;
        const uint32_t blocks = (waves + (uint32_t)W - 1) / (uint32_t)W;
        cv.lds_nodes = coop_resident_nodes(cv, W, blocks, wl.num_cus);
        static const int resident_max = std::getenv("FS_COOP_RESIDENT_MAX") ? std::atoi(std::getenv("FS_COOP_RESIDENT_MAX")) : -1;   // (experiments: fewer staged records)
        if (resident_max >= 0) cv.lds_nodes = std::min(cv.lds_nodes, resident_max);
        const size_t lds = coop_lds_bytes(W, cv), lds_ext = lds;
        static const bool dbg = std::getenv("FS_DEBUG_SCENE_INFO") != nullptr;   // (read once)
        if (dbg) std::fprintf(stderr, "[frequensee] cooperative walk: %u lanes, %d per wave, %u workgroups of %d waves, %d-wide nodes, %d of %d resident in LDS (%zu bytes)\n",
                              lanes, wl.rays_per_wave, blocks, W, 1 << cv.wshift, cv.lds_nodes, cv.nodes, lds);
        if (W == kCoopBigWaves) {
            allow_lds(walk_kernel_coop_big, lds);
            hipLaunchKernelGGL(walk_kernel_coop_big, dim3(blocks), dim3(64 * W), lds, s, sc, cv, kp, st, wl.queue_head, perm, wl.rays_per_wave, stage, lane);
        } else {
            FS_LAUNCH_WALK(walk_kernel_coop, blocks, sc, cv, kp, st, wl.queue_head, perm, wl.rays_per_wave, stage, lane);
        }
        return;
    }
    if (shared && wl.rays_per_wave > 0 && wl.rays_per_wave < 64) {   // small frame: sparse waves, idle lanes help
        const size_t lds = stack_bytes(sc) + kShareLdsBytes, lds_ext = stack_bytes(sc) + kShareIgnLdsBytes;   // (EXT: + the ignored actor per ray)
        const uint32_t waves = (lanes + (uint32_t)wl.rays_per_wave - 1) / (uint32_t)wl.rays_per_wave;
        const uint32_t blocks = (waves + kBlock / 64 - 1) / (kBlock / 64);
        if (!attach_deep(sc, blocks)) return;
        FS_LAUNCH_WALK(walk_kernel_sparse, blocks, sc, kp, st, wl.queue_head, perm, wl.rays_per_wave, stage, lane);
        return;
    }
    if (!attach_deep(sc, full)) return;
    if (shared) {   // the lobes of FS_FLAG_MATERIAL_LOBES are compiled out of the default instantiation
        const size_t lds = stack_bytes(sc) + kShareLdsBytes, lds_ext = stack_bytes(sc) + kShareIgnLdsBytes;
        FS_LAUNCH_WALK(walk_kernel_shared, full, sc, kp, st, wl.queue_head, perm, stage);
        return;
    }
#undef FS_LAUNCH_WALK
#ifdef FS_EXPERIMENTS
    if (kp.lobes) {
        allow_lds(walk_kernel_simple<1>, stack_bytes(sc));
        hipLaunchKernelGGL(walk_kernel_simple<1>, dim3(full), dim3(kBlock), stack_bytes(sc), s, sc, kp, st, wl.queue_head, perm);
    } else {
        allow_lds(walk_kernel_simple<0>, stack_bytes(sc));
        hipLaunchKernelGGL(walk_kernel_simple<0>, dim3(full), dim3(kBlock), stack_bytes(sc), s, sc, kp, st, wl.queue_head, perm);
    }
#endif
}

}  // namespace fs
