// fs_frame.hip — pipelined frames: ONE launch for the walk of one frame, the connect pass of an older one and the plan
// pass of the newest (fs_set_pipelining; fs_capi_frame.cpp: frame_launch, fs_capi_pipeline.cpp: drain_fused).
//
// Compiled twice (Makefile): as it stands — the kernel limited to 128 VGPRs and the LDS stack capped, so that four
// workgroups share a CU, for launches that fill the chip — and through fs_frame_wide.hip with FS_FRAME_WIDE defined:
// no register limit, the tree's worst-case stack rows in LDS and none of the deep-store logic in the traversal, for
// launches too small for occupancy to matter (cfg2: 0.105 -> 0.097 ms, cfg4 at 131 072 rays: 400 -> 435 M rays/s).
//
// And once more per flavour with FS_FRAME_EXT defined (fs_frame_ext.hip, fs_frame_wide_ext.hip): the walk parts are the EXT
// instantiations, whose queries skip the triangles of the actor the walk starts from (AddIgnoredActor, ARTS.cpp:322-327 — what the
// reference's GeneratePath ALWAYS does; fs_source_set_object / fs_listener_set_object).  Until round 5 such frames were never
// held by fs_set_pipelining: a plugin whose source actors carry registered meshes got the unpipelined rate.
#ifdef FS_FRAME_EXT
#define FS_FRAME_EXT_ON true
#else
#define FS_FRAME_EXT_ON false
#endif
#ifdef FS_FRAME_WIDE
#define FS_DEEP_NO_CHECK 1
#define FS_FRAME_MIN_WAVES 1
#ifdef FS_FRAME_EXT
#define frame_kernel frame_kernel_wide_ext
#define FS_LAUNCH_FRAME launch_frame_wide_ext
#else
#define frame_kernel frame_kernel_wide
#define FS_LAUNCH_FRAME launch_frame_wide
#endif
#elif defined(FS_FRAME_EXT)
#define frame_kernel frame_kernel_ext
#define FS_LAUNCH_FRAME launch_frame_narrow_ext
#else
#define FS_LAUNCH_FRAME launch_frame_narrow
#endif
#include "fs_device.hpp"

namespace fs {
namespace {

// One launch for several frames (pipelined frames, fs_capi_frame.cpp): the first workgroups run walk parts — whole walks
// of a frame planned by an earlier launch, or one STAGE (steps [begin, end)) of the walks of a depth = 0 frame, every
// stage from a different frame — the next connect_blocks connect the pairs of a frame whose walks are complete, the rest
// run the plan pass of the newest frame.  The thin single round of the connect pass, the thin tails of the walks'
// longest waves and the short plan pass fill each other's idle wave slots, and the kernel boundaries between them
// disappear.  The frames share nothing but the scene: each has its own subpath state, schedule, frame scratch and
// energy buffer.
struct WalkArgs {
    KParams kp; SubpathState st;
    const unsigned* scratch; const uint32_t* perm;
    int rays_per_wave; WalkStage stage;
    uint32_t block_end;   // one past the last workgroup of this part
};
struct FrameArgs {
    int num_walk;
    WalkArgs walk[kMaxWalkParts];
    uint32_t connect_blocks;
    KParams kpc; SubpathState stc; float* energy; unsigned long long* fixed; unsigned* scratch_c; int pairs_per_wave;
    float* const* energy_tab; unsigned long long* const* fixed_tab;
    KParams kpp; unsigned* scratch_p; uint32_t* perm_p; float* zero_p; int zero_words_p; float* const* zero_tab_p; int zero_count_p;
    uint32_t plan_blocks;
    // reconstruct parts (behind the plan part): (B + 1) rows x recon_cb blocks of chunks per item
    int num_recon; const ReconItem* recon_tab;   // (items in pinned host memory: FrameParts::recon_tab)
    int recon_B, recon_nb, recon_samples; uint32_t recon_cb;
    PublishWord pub;
    uint32_t connect_first;   // the connect part takes the FIRST workgroups of the grid (see FS_LAUNCH_FRAME)
};
static_assert(sizeof(FrameArgs) + sizeof(DeviceScene) <= 4096, "kernel arguments are limited to 4 KB");

// 4 waves per SIMD (at most 128 VGPRs; the walk part spills a dozen registers outside its loops): with the bounded LDS stack
// (DeviceScene.stack_limit) four workgroups share a CU: 0.313 -> 0.296 ms per cfg3 launch (profiles/r03_occupancy_ab.log)
#ifndef FS_FRAME_MIN_WAVES
#define FS_FRAME_MIN_WAVES 4
#endif
template <int B, bool BATCH>
__global__ __launch_bounds__(kBlock, FS_FRAME_MIN_WAVES) void frame_kernel(DeviceScene sc, FrameArgs a) {
    uint32_t b = blockIdx.x;
    if (a.connect_first) {
        if (b < a.connect_blocks) {
            connect_body<B, 0, BATCH, false>(b, a.connect_blocks, sc, a.kpc, a.stc, a.energy, a.fixed, a.scratch_c,
                                             a.pairs_per_wave, a.energy_tab, a.fixed_tab);
            return;
        }
        b -= a.connect_blocks;
    }
    const uint32_t cb = a.connect_first ? 0u : a.connect_blocks;   // connect workgroups behind the walk parts
    // which walk part this workgroup belongs to (a scalar search; the walk itself runs OUTSIDE the loop: inside it every value
    // that is live across the walk was loop-carried as far as the register allocator could tell)
    uint32_t first = 0;
    int part = -1;
    for (int i = 0; i < a.num_walk; ++i) {
        if (b < a.walk[i].block_end) { part = i; break; }
        first = a.walk[i].block_end;
    }
    if (part >= 0) {
        const WalkArgs& w = a.walk[part];
        if (w.rays_per_wave < 64) walk_sparse_body<0, false, false, FS_FRAME_EXT_ON>(b - first, sc, w.kp, w.st, w.scratch, w.perm, w.rays_per_wave, w.stage);
        else walk_shared_body<0, false, false, FS_FRAME_EXT_ON>(b - first, sc, w.kp, w.st, w.scratch, w.perm, w.stage);
        return;
    }
    if (b < first + cb) {
        connect_body<B, 0, BATCH, false>(b - first, a.connect_blocks, sc, a.kpc, a.stc, a.energy, a.fixed, a.scratch_c,
                                         a.pairs_per_wave, a.energy_tab, a.fixed_tab);
    } else if (b < first + cb + a.plan_blocks) {
        if (a.kpp.plan_coop)
            plan_coop_body(b - first - cb, a.plan_blocks, a.kpp, a.scratch_p, a.perm_p, a.zero_p, a.zero_words_p, a.zero_tab_p, a.zero_count_p);
        else
            plan_body(b - first - cb, a.plan_blocks, a.kpp, a.scratch_p, a.perm_p, a.zero_p, a.zero_words_p, a.zero_tab_p, a.zero_count_p);
    } else {   // reconstruct part: item, row, block of chunks
        extern __shared__ __attribute__((aligned(16))) int s_dyn_r[];
        const uint32_t rb = b - first - cb - a.plan_blocks, per_item = (uint32_t)(a.recon_B + 1) * a.recon_cb;
        const uint32_t item = rb / per_item, in_item = rb - item * per_item;
        const ReconItem it = a.recon_tab[item];
        float* const s_amp = reinterpret_cast<float*>(s_dyn_r);
        reconstruct_body((int)(in_item / a.recon_cb), (int)(in_item % a.recon_cb), it.energy, a.recon_B, a.recon_nb, a.recon_samples, it.spb,
                         it.ir_bands, it.ir_mono, s_amp, it.host, s_amp + a.recon_nb, it.mask);
        publish_arrive(a.pub.tickets, (uint32_t)a.num_recon * per_item, a.pub.host_word, a.pub.id);
    }
}

template <int B>
void launch_frame_t(const DeviceScene& sc_in, uint32_t blocks, size_t lds, const FrameArgs& a, bool batch, hipStream_t s) {
    DeviceScene sc = sc_in;
    if (!attach_deep(sc, blocks)) return;
    if (batch) {
        allow_lds(frame_kernel<B, true>, lds);
        hipLaunchKernelGGL((frame_kernel<B, true>), dim3(blocks), dim3(kBlock), lds, s, sc, a);
    } else {
        allow_lds(frame_kernel<B, false>, lds);
        hipLaunchKernelGGL((frame_kernel<B, false>), dim3(blocks), dim3(kBlock), lds, s, sc, a);
    }
}

}  // namespace

// blocks_only: report the grid this frame would be launched with and return (nothing is launched)
bool FS_LAUNCH_FRAME(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s, uint32_t* blocks_only) {
    FrameArgs a{};
    uint32_t blocks = 0;
    size_t lds = 0;
    if (f.num_walk > kMaxWalkParts) return false;
    for (int i = 0; i < f.num_walk; ++i) {
        const WalkPart& p = f.walk[i];
        if (!FS_SHARED_WALK(p.wl)) return false;
        if (p.kp.lobes || p.kp.count || p.kp.dpos || (p.kp.ignore_on && !FS_FRAME_EXT_ON) || p.kp.listener_radius > 0.0f || p.kp.source_radius > 0.0f || p.kp.num_local == 0)
            return false;   // the default instantiations only (the EXT flavour: + the walk's own actor)
        WalkArgs& w = a.walk[a.num_walk++];
        w.kp = p.kp; w.st = p.st; w.scratch = p.wl.queue_head; w.perm = p.perm; w.stage = p.stage;
        w.rays_per_wave = p.wl.rays_per_wave > 0 && p.wl.rays_per_wave < 64 ? p.wl.rays_per_wave : 64;
        if (w.stage.begin > 0 && w.stage.slots_cap == 0xFFFFFFFFu) w.stage.slots_cap = walk_stage_slots(p.kp, w.stage.begin);
        const uint32_t lanes = w.stage.begin > 0 ? w.stage.slots_cap : 2u * p.kp.num_local;
        const uint32_t waves = (lanes + (uint32_t)w.rays_per_wave - 1) / (uint32_t)w.rays_per_wave;
        blocks += (waves + kBlock / 64 - 1) / (kBlock / 64);
        w.block_end = blocks;
        lds = std::max(lds, stack_bytes(sc) + (FS_FRAME_EXT_ON ? kShareIgnLdsBytes : kShareLdsBytes));
    }
    if (f.has_connect) {
        if (f.kpc.lobes || f.kpc.count || f.kpc.dpos || f.kpc.num_local == 0 || f.ppw < 1 || f.ppw > 64) return false;
        const uint32_t per_block = (uint32_t)f.ppw * (kBlock / 64);
        const uint32_t want = f.energy_tab ? (f.kpc.num_local / f.kpc.pairs_per_source) * ((f.kpc.pairs_per_source + per_block - 1) / per_block)
                                           : (f.kpc.num_local + per_block - 1) / per_block;
        a.connect_blocks = std::min<uint32_t>(want, 1024u);
        // Where the connect workgroups sit in the grid (tools/launch_timeline.py, profiles/r03_launch_timeline_*.json,
        // profiles/r03_connect_first.log).  Behind the walks, the connect pass starts when the last walk workgroup has been
        // dispatched and is the launch's tail: 55 of 597 us at 40 % of the wave slots when two frames share the launch.  A
        // launch whose walk workgroups need the chip's 1 024 workgroup slots twice over starts 384 connect workgroups FIRST
        // instead (they step through all the pairs): they finish while the long walks run and the tail is the
        // shortest walks (950 -> 965 M rays/s; 256: 962, 512: 946, 1 024: 921).  A launch whose walks fit the chip once
        // keeps the connect pass behind them: there every long walk has to start at t = 0 (858 -> 815 M with 384 first).
        // FS_FRAME_CONNECT_FIRST = n > 0: always n first; < 0: never.  (cfg5's eight sources 894 -> 912 M, deterministic mode
        // 835 -> 870 M, three frames per launch 818 -> 852 M; cfg4 and staged walks unchanged.)
        static const int cap_env = std::getenv("FS_FRAME_CONNECT_FIRST") ? std::atoi(std::getenv("FS_FRAME_CONNECT_FIRST")) : 0;
        // (walks of one length — Russian roulette off — leave no short walks for the tail: 660 -> 606 M with connect first)
        const int cap_first = cap_env != 0 ? cap_env : (blocks >= 2048u && f.walk[0].kp.russian_roulette ? 384 : -1);
        if (cap_first > 0 && f.num_walk > 0) { a.connect_blocks = std::min<uint32_t>(a.connect_blocks, (uint32_t)cap_first); a.connect_first = 1u; }
        a.kpc = f.kpc; a.stc = f.stc; a.energy = f.energy; a.fixed = f.fixed; a.scratch_c = f.scratch_c; a.pairs_per_wave = f.ppw;
        a.energy_tab = f.energy_tab; a.fixed_tab = f.fixed_tab;
        blocks += a.connect_blocks;
        lds = std::max(lds, stack_bytes(sc) + sizeof(float) * (size_t)B * (size_t)f.kpc.hist_window + kShareAnyLdsBytes);
    }
    if (f.has_plan) {
        uint32_t pb = 0;
        bool sort = false;
        if (!plan_shape(f.kpp, f.wl_p, &pb, &sort)) return false;
        a.kpp = f.kpp; a.scratch_p = f.scratch_p; a.perm_p = f.perm_p; a.zero_p = f.zero_p; a.zero_words_p = f.zero_words_p;
        a.zero_tab_p = f.zero_tab_p; a.zero_count_p = f.zero_count_p;
        a.plan_blocks = pb;
        blocks += pb;
    }
    if (f.num_recon > 0) {
        if (f.num_recon > kMaxReconParts || f.recon_tab == nullptr || f.recon_B < 1 || f.recon_nb < 1 || f.recon_samples < 1) return false;
        a.num_recon = f.num_recon;
        a.recon_tab = f.recon_tab;
        a.recon_B = f.recon_B; a.recon_nb = f.recon_nb; a.recon_samples = f.recon_samples;
        a.pub = f.pub;
        const uint32_t chunks = (uint32_t)((f.recon_samples + kChunk - 1) / kChunk);
        a.recon_cb = (chunks + kBlock - 1) / kBlock;
        blocks += (uint32_t)f.num_recon * (uint32_t)(f.recon_B + 1) * a.recon_cb;
        lds = std::max(lds, sizeof(float) * ((size_t)f.recon_nb + (size_t)kBlock * (kChunk + 1)));   // the amplitudes | the block's samples staged for 16-byte stores
    }
    if (blocks == 0) return false;
    if (blocks_only) { *blocks_only = blocks; return true; }
    const bool batch = f.has_connect && f.energy_tab;
    switch (B) {   // the band counts in use; 0 = kp.num_bands at run time (fs_connect.hip)
        case 1: launch_frame_t<1>(sc, blocks, lds, a, batch, s); break;
        case 4: launch_frame_t<4>(sc, blocks, lds, a, batch, s); break;
        case 8: launch_frame_t<8>(sc, blocks, lds, a, batch, s); break;
        default: launch_frame_t<0>(sc, blocks, lds, a, batch, s); break;
    }
    return true;
}

#if !defined(FS_FRAME_WIDE) && !defined(FS_FRAME_EXT)
bool launch_frame_wide(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s, uint32_t* blocks_only);   // fs_frame_wide.hip
bool launch_frame_narrow_ext(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s, uint32_t* blocks_only);   // fs_frame_ext.hip
bool launch_frame_wide_ext(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s, uint32_t* blocks_only);     // fs_frame_wide_ext.hip

// Launches with fewer workgroups than the chip holds at four per CU take the wide flavour (see the top of the file), and
// so do small frames that only reach that many workgroups because their walks run on sparse waves (a few subpaths per
// wave, the other lanes help): there the fourth workgroup of a CU adds helpers, not work.
constexpr uint32_t kFrameNarrowFromBlocks = 1024;
bool launch_frame(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s) {
    uint32_t blocks = 0;
    bool ext = false;   // some walk of the launch ignores the actor it starts from: the EXT flavours
    for (int i = 0; i < f.num_walk; ++i) ext = ext || f.walk[i].kp.ignore_on != 0;
    const auto narrow = ext ? launch_frame_narrow_ext : launch_frame_narrow;
    const auto wide = ext ? launch_frame_wide_ext : launch_frame_wide;
    if (!narrow(B, sc, f, s, &blocks)) return false;
    static const bool dbg = std::getenv("FS_DEBUG_FRAME") != nullptr;
    if (dbg) std::fprintf(stderr, "[launch_frame] blocks %u stack_worst %d rows %d limit %d\n", blocks, sc.stack_worst, sc.stack_rows, sc.stack_limit);
    bool dense = f.num_walk == 0;
    for (int i = 0; i < f.num_walk; ++i) dense = dense || !(f.walk[i].wl.rays_per_wave > 0 && f.walk[i].wl.rays_per_wave < 64);
    if (sc.stack_worst > 0 && (blocks < kFrameNarrowFromBlocks || !dense)) {
        DeviceScene w = sc;
        w.stack_rows = sc.stack_worst; w.stack_limit = sc.stack_worst; w.stack_attn = 0x7FFFFFFFu;
        w.deep = nullptr; w.deep_lanes = 0; w.deep_owner = nullptr;
        return wide(B, w, f, s, nullptr);
    }
    return narrow(B, sc, f, s, nullptr);
}
#endif

}  // namespace fs
