// fs_frame.hip — pipelined frames: ONE launch for the walk of one frame, the connect pass of an older one and the plan
// pass of the newest (fs_set_pipelining; fs_capi_frame.cpp).
#include "fs_device.hpp"

namespace fs {
namespace {

// One launch for up to three frames (pipelined frames, fs_capi.cpp): workgroups [0, walk_blocks) walk the subpaths of
// frame f - 1 (planned by the previous launch), the next connect_blocks connect the pairs of frame f - 2 (walked by the
// previous launch), the rest run the plan pass of frame f.  The thin single round of the connect pass, the thin tail of
// the walk's longest waves and the short plan pass fill each other's idle wave slots — and two kernel boundaries per
// frame disappear.  The frames share nothing but the scene: each has its own subpath state, schedule, frame scratch
// and energy buffer.
template <int B, bool BATCH>
__global__ __launch_bounds__(kBlock) void frame_kernel(DeviceScene sc, uint32_t walk_blocks, uint32_t connect_blocks,
                                                       KParams kpw, SubpathState stw, const unsigned* __restrict__ scratch_w,
                                                       const uint32_t* __restrict__ perm, int rays_per_wave,
                                                       KParams kpc, SubpathState stc, float* __restrict__ energy,
                                                       unsigned long long* __restrict__ fixed, unsigned* scratch_c,
                                                       int pairs_per_wave, float* const* __restrict__ energy_tab,
                                                       unsigned long long* const* __restrict__ fixed_tab,
                                                       KParams kpp, unsigned* __restrict__ scratch_p, uint32_t* __restrict__ perm_p,
                                                       float* __restrict__ zero_p, int zero_words_p,
                                                       float* const* __restrict__ zero_tab_p, int zero_count_p) {
    const uint32_t b = blockIdx.x;
    if (b < walk_blocks) {   // (walks first: starting the connect pass before the short walks measured slower)
        if (rays_per_wave < 64) walk_sparse_body<0, false>(b, sc, kpw, stw, scratch_w, perm, rays_per_wave);
        else walk_shared_body<0, false>(b, sc, kpw, stw, scratch_w, perm);
    } else if (b < walk_blocks + connect_blocks) {
        connect_body<B, 0, BATCH, false>(b - walk_blocks, connect_blocks, sc, kpc, stc, energy, fixed, scratch_c,
                                         pairs_per_wave, energy_tab, fixed_tab);
    } else {
        plan_body(b - walk_blocks - connect_blocks, gridDim.x - walk_blocks - connect_blocks, kpp, scratch_p, perm_p, zero_p,
                  zero_words_p, zero_tab_p, zero_count_p);
    }
}

template <int B, bool BATCH>
void launch_frame_tb(const DeviceScene& sc, uint32_t wb, uint32_t cb, uint32_t pb, size_t lds, const FrameParts& f, int rpw,
                     bool sort, hipStream_t s) {
    allow_lds(frame_kernel<B, BATCH>, lds);
    hipLaunchKernelGGL((frame_kernel<B, BATCH>), dim3(wb + cb + pb), dim3(kBlock), lds, s, sc, wb, cb, f.kpw, f.stw,
                       f.wl.queue_head, f.perm, rpw, f.kpc, f.stc, f.energy, f.fixed, f.scratch_c, f.ppw, f.energy_tab,
                       f.fixed_tab, f.kpp, f.scratch_p, sort ? f.perm_p : nullptr, f.zero_p, f.zero_words_p, f.zero_tab_p,
                       f.zero_count_p);
}

template <int B>
void launch_frame_t(const DeviceScene& sc, uint32_t wb, uint32_t cb, uint32_t pb, size_t lds, const FrameParts& f, int rpw,
                    bool sort, hipStream_t s) {
    if (f.has_connect && f.energy_tab) launch_frame_tb<B, true>(sc, wb, cb, pb, lds, f, rpw, sort, s);
    else launch_frame_tb<B, false>(sc, wb, cb, pb, lds, f, rpw, sort, s);
}

}  // namespace

bool launch_frame(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s) {
    if (!FS_SHARED_WALK(f.wl)) return false;
    uint32_t wb = 0, cb = 0, pb = 0;
    int rpw = 64;
    bool sort = false;
    size_t lds = 0;
    if (f.has_walk) {
        if (f.kpw.lobes || f.kpw.count || f.kpw.num_local == 0) return false;   // the default instantiations only
        rpw = f.wl.rays_per_wave > 0 && f.wl.rays_per_wave < 64 ? f.wl.rays_per_wave : 64;
        const uint32_t waves = (2u * f.kpw.num_local + (uint32_t)rpw - 1) / (uint32_t)rpw;
        wb = (waves + kBlock / 64 - 1) / (kBlock / 64);
        lds = std::max(lds, stack_bytes(sc) + kShareLdsBytes);
    }
    if (f.has_connect) {
        if (f.kpc.lobes || f.kpc.count || f.kpc.num_local == 0 || f.ppw < 1 || f.ppw > 64) return false;
        const uint32_t per_block = (uint32_t)f.ppw * (kBlock / 64);
        const uint32_t want = f.energy_tab ? (f.kpc.num_local / f.kpc.pairs_per_source) * ((f.kpc.pairs_per_source + per_block - 1) / per_block)
                                           : (f.kpc.num_local + per_block - 1) / per_block;
        cb = std::min<uint32_t>(want, 1024u);
        lds = std::max(lds, stack_bytes(sc) + sizeof(float) * (size_t)B * (size_t)f.kpc.hist_window + kShareAnyLdsBytes);
    }
    if (f.has_plan) {
        if (!plan_shape(f.kpp, f.wl, &pb, nullptr)) return false;
        sort = f.perm_p != nullptr;
    }
    if (wb + cb + pb == 0) return false;
    switch (B) {   // the band counts in use; 0 = kp.num_bands at run time (fs_connect.hip)
        case 1: launch_frame_t<1>(sc, wb, cb, pb, lds, f, rpw, sort, s); break;
        case 4: launch_frame_t<4>(sc, wb, cb, pb, lds, f, rpw, sort, s); break;
        case 8: launch_frame_t<8>(sc, wb, cb, pb, lds, f, rpw, sort, s); break;
        default: launch_frame_t<0>(sc, wb, cb, pb, lds, f, rpw, sort, s); break;
    }
    return true;
}

}  // namespace fs
