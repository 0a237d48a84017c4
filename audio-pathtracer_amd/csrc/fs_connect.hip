// fs_connect.hip — ConnectSubpaths + EvaluatePath + deposit kernels (AudioRayTracingSubsystem.cpp:235-277, 360-420;
// FrequenSeeAudioComponent.h:87-91), the all-prefix variant (draft :518-546, row f3) and the fixed-point rounding pass.
#include "fs_device.hpp"

namespace fs {
namespace {

template <int B, int LOBES, bool BATCH, bool COUNT, bool EXT = false, int AHEAD = 1>
__global__ __launch_bounds__(kBlock, AHEAD > 4 ? 2 : 4) void connect_kernel(DeviceScene sc, KParams kp, SubpathState st,
                                                         float* __restrict__ energy,
                                                         unsigned long long* __restrict__ fixed, unsigned* queue_head,
                                                         int pairs_per_wave, float* const* __restrict__ energy_tab,
                                                         unsigned long long* const* __restrict__ fixed_tab) {
    connect_body<B, LOBES, BATCH, COUNT, EXT, AHEAD>(blockIdx.x, gridDim.x, sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab);
}

// ---------------------------------------------------------------------------------------------------
// connect_all_kernel (row f3): the reference's unfinished "naive connections" (Is_NaiveConnections,
// ARTS.cpp:518-546: every bounce of the forward sample x every bounce of the backward sample, "Equation 12").
// For pair p with forward nodes F0..Fk and backward nodes B0..Bm every (i, j) in [0,k] x [0,m] is a
// candidate path F0..Fi, Bj..B0: ConnectSubpaths' visibility test Fi -> Bj, EvaluatePath over the stored
// segment records in path order, uniform multiple-importance weight 1 / N(i + j) with N(t) = number of
// (i', j') in [0, D]^2, i' + j' = t (D = depth cap).  One WAVE per pair, one lane per (i, j): the up to
// (D+1)^2 visibility rays of a pair start and end at neighbouring nodes, so the wave traverses coherently.
// ---------------------------------------------------------------------------------------------------
// Balance-heuristic weight of strategy i (vertices y_1..y_i generated from the source, y_t..y_{i+1} from the
// listener, t = i + j) among the strategies [max(0, t-D), min(t, D)] that give the same path — the intent of the
// draft's MISEnergy (ARTS.cpp:571-597); build-owned definition, DESIGN.md section 8: forward density of y_{k+1}
// given y_k  pf_k = Pf(k) |n_{k+1}.d_k| / L_k^2 with Pf(0) = 1/4pi, Pf(k) = max(0, n_k.d_k)/pi; backward density of
// y_k given y_{k+1}  pb_k = Pb(k+1) |n_k.d_k| / L_k^2 with Pb(t+1) = 1/4pi, Pb(k) = max(0, -n_k.d_{k-1})/pi;
// p_s = prod_{k<s} pf_k prod_{k>s} pb_k, w_i = p_i / sum_s p_s.  One pass over the t + 1 segments in double:
// T_k = T_{k-1} pb_k + [lo <= k <= hi] PF_k ends as sum_s p_s, Q likewise as p_i; uniform weight when a segment
// is degenerate or the ratio is not finite and positive.
// Vertex k of the connected path: 0 = source, 1..i = forward nodes, i+1..t = backward nodes j..1, t+1 = listener.
struct MisVertex { double x, y, z, nx, ny, nz; };
__device__ __forceinline__ MisVertex mis_vertex(const KParams& kp, const SubpathState& st, uint32_t total, uint32_t sf,
                                                uint32_t sl, int i, int t, int k) {
    MisVertex v;
    if (k == 0) { v.x = kp.src[0]; v.y = kp.src[1]; v.z = kp.src[2]; v.nx = v.ny = v.nz = 0.0; return v; }
    if (k == t + 1) { v.x = kp.lis[0]; v.y = kp.lis[1]; v.z = kp.lis[2]; v.nx = v.ny = v.nz = 0.0; return v; }
    const float4 q = k <= i ? load_pos(st, total, k - 1, sf) : load_pos(st, total, t - k, sl);
    const float4 m = k <= i ? load_nrm(st, total, k - 1, sf) : load_nrm(st, total, t - k, sl);
    v.x = q.x; v.y = q.y; v.z = q.z; v.nx = m.x; v.ny = m.y; v.nz = m.z;
    return v;
}
__device__ float mis_weight(const KParams& kp, const SubpathState& st, uint32_t total, uint32_t sf, uint32_t sl, int i,
                            int j) {
    const int t = i + j, D = kp.mis_depth;
    const int lo = t - D > 0 ? t - D : 0, hi = t < D ? t : D;
    const double uniform = 1.0 / (double)(hi - lo + 1);
    if (t <= 0) return (float)uniform;
    const double inv4pi = 1.0 / (4.0 * 3.14159265358979323846), invpi = 1.0 / 3.14159265358979323846;
    double PF = 1.0, T = 0.0, Q = 0.0;
    MisVertex a = mis_vertex(kp, st, total, sf, sl, i, t, 0);
    for (int k = 0; k <= t; ++k) {
        const MisVertex b = mis_vertex(kp, st, total, sf, sl, i, t, k + 1);
        double dx = b.x - a.x, dy = b.y - a.y, dz = b.z - a.z;
        const double l2 = dx * dx + dy * dy + dz * dz;
        if (!(l2 > 1e-8)) return (float)uniform;
        const double inv = 1.0 / sqrt(l2);
        dx *= inv; dy *= inv; dz *= inv;
        const double ca = k > 0 ? a.nx * dx + a.ny * dy + a.nz * dz : 0.0;
        const double cb = k < t ? b.nx * dx + b.ny * dy + b.nz * dz : 0.0;
        if (k >= 1) {
            const double Pb = k == t ? inv4pi : (cb < 0.0 ? -cb : 0.0) * invpi;
            const double pb = Pb * fabs(ca) / l2;
            T *= pb;
            if (k > i) Q *= pb;
        }
        if (k >= lo && k <= hi) T += PF;
        if (k == i) Q = PF;
        if (k < t) {
            const double Pf = k == 0 ? inv4pi : (ca > 0.0 ? ca : 0.0) * invpi;
            PF *= Pf * fabs(cb) / l2;
        }
        a = b;
    }
    const double w = Q / T;
    if (!(Q > 0.0) || !(T > 0.0) || !(w <= 1.0)) return (float)uniform;
    return (float)w;
}

template <int B>
__global__ __launch_bounds__(kBlock) void connect_all_kernel(DeviceScene sc, KParams kp, SubpathState st,
                                                             float* __restrict__ energy,
                                                             unsigned long long* __restrict__ fixed,
                                                             unsigned* queue_head) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock] stack | [B][hist_window] histogram
    int* s_stack = s_dyn;
    float* s_hist = reinterpret_cast<float*>(s_dyn + (size_t)sc.stack_rows * kBlock);
    const int nb = kp.num_bins, W = kp.hist_window, NB = band_count<B>(kp);
    int* s_share = reinterpret_cast<int*>(s_hist + (size_t)NB * W);   // work-sharing area of trav_any_shared
    __shared__ int s_lo, s_hi;
    for (int i = threadIdx.x; i < NB * W; i += kBlock) s_hist[i] = 0.0f;
    if (threadIdx.x == 0) { s_lo = nb; s_hi = -1; }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 1 + 2 * kPlanBuckets; i += kBlock) queue_head[i] = 0u;
    __syncthreads();

    const uint32_t n = kp.num_local;
    const uint32_t total = 2u * n;
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t wave = threadIdx.x >> 6, waves = kBlock / 64;
    unsigned my_deposits = 0, my_tests = 0, my_segments = 0;
    for (uint32_t li = blockIdx.x * waves + wave; li < n; li += gridDim.x * waves) {
        const uint32_t sf = slot_of(st, li), sl = slot_of(st, n + li);
        const uint2 Fm = st.end_misc[sf];
        const uint2 Lm = st.end_misc[sl];
        const int kf = (int)Fm.y, kl = (int)Lm.y;
        if (lane == 0) my_segments += (unsigned)(kf + kl);   // fs_stats.segments: the steps the two walks took
        // depth = 0 only: a walk that outlived the record store (overflow word raised, the frame is traced again)
        if (st.over_levels && !(rec_fits(st, kf - 1, sf) && rec_fits(st, kl - 1, sl))) continue;
        const int combos = (kf + 1) * (kl + 1);
        for (int c0 = 0; c0 < combos; c0 += 64) {   // wave-uniform trip count: all lanes share the visibility queries
            const bool active = c0 + lane < combos;
            const int c = active ? c0 + lane : 0;
            if (active) ++my_tests;
            const int i = c / (kl + 1), j = c - i * (kl + 1);
            // node Fi (position, material, probability) and node Bj (position)
            float fx = kp.src[0], fy = kp.src[1], fz = kp.src[2];
            if (i > 0) { const float4 q = load_pos(st, total, i - 1, sf); fx = q.x; fy = q.y; fz = q.z; }
            float bx = kp.lis[0], by = kp.lis[1], bz = kp.lis[2];
            if (j > 0) { const float4 q = load_pos(st, total, j - 1, sl); bx = q.x; by = q.y; bz = q.z; }
            uint32_t fmat; float fprob;
            if (i < kf) { fmat = load_mat(st, total, i, sf); fprob = load_np(st, total, i, sf).y; }
            else { fmat = Fm.x; fprob = st.end_pos[sf].w; }
            fmat &= 0xFFFFu;   // a connection vertex scatters diffusely whatever lobe the walk took there later (row f4)
            float dx = bx - fx, dy = by - fy, dz = bz - fz;
            float l2 = dx * dx + dy * dy + dz * dz;
            float len = sqrtf(l2);
            float inv = 1.0f / len;
            float tmax = len - kp.connect_pullback;
            bool has_ray = active && (l2 > 1e-8f) && (tmax > 0.0f);
            Ray ray = make_ray(fx, fy, fz, dx * inv, dy * inv, dz * inv);
            bool sphere_blocked = false;   // the end points' collision spheres: ConnectSubpaths ignores no actor
            if (has_ray && (kp.listener_radius > 0.0f || kp.source_radius > 0.0f)) {
                float ts;
                sphere_blocked = (kp.listener_radius > 0.0f && sphere_hit(ray, kp.lis, kp.listener_radius, tmax, ts)) ||
                                 (kp.source_radius > 0.0f && sphere_hit(ray, kp.src, kp.source_radius, tmax, ts));
                if (sphere_blocked) has_ray = false;
            }
            const bool hit = trav_any_shared(sc, has_ray, ray, tmax, &s_stack[threadIdx.x], s_share);
            if (!active || hit || sphere_blocked) continue;
            ++my_deposits;
            float E[Bands<B>::kMax];
#pragma unroll
            for (int b = 0; b < Bands<B>::kMax; ++b) E[b] = 1.0f;
            float sd = 0.0f;
            for (int a = 0; a < i; ++a) {                                 // F_a -> F_a+1
                const float2 np = load_np(st, total, a, sf);
                sd += np.x;
                apply_segment<B>(E, np.x, load_mat(st, total, a, sf), np.y, kp, sc);
            }
            {                                                             // Fi -> Bj
                float nd = sqrtf(l2) / kp.dist_divisor;
                sd += nd;
                apply_segment<B>(E, nd, fmat, fprob, kp, sc);
            }
            for (int a = j - 1; a >= 0; --a) {                            // B_a+1 -> B_a
                const float2 np = load_np(st, total, a, sl);
                sd += np.x;
                uint32_t bmat = load_mat(st, total, a, sl);
                if (a == j - 1) bmat &= 0xFFFFu;                          // Bj is the other connection vertex
                apply_segment<B>(E, np.x, bmat, np.y, kp, sc);
            }
            const int t = i + j, D = kp.mis_depth;
            const int lo_t = t - D > 0 ? t - D : 0, hi_t = t < D ? t : D;
            float w = 1.0f / (float)(hi_t - lo_t + 1);
            if (kp.mis) w = mis_weight(kp, st, total, sf, sl, i, j);
            float delay = sd / kp.sound_speed;
            float x = (delay * 1000.f) / 1.0f;
            float fl = floorf(x);
            int bin = !(fl > 0.0f) ? 0 : (fl >= (float)(nb - 1) ? nb - 1 : (int)fl);
            const bool near = bin < W;
            if (!fixed && near) {
                atomicMin(&s_lo, bin);
                atomicMax(&s_hi, bin);
            }
#pragma unroll
            for (int b = 0; b < Bands<B>::kMax; ++b) {
                if (B == 0 && b >= NB) break;
                float e = E[b];
                e = (e < kp.energy_clamp) ? e : kp.energy_clamp;
                e *= kp.energy_gain;
                e *= kp.norm;
                e *= w;
                if (fixed)
                    atomicAdd(&fixed[b * nb + bin], (unsigned long long)__double2ll_rn((double)e * kFixedScale));
                else if (near)
                    atomicAdd(&s_hist[b * W + bin], e);   // ds_add_f32.  (Summing the equal-bin deposits of a wave first —
                else                                      // ballot per distinct bin + butterfly per band — measured slower:
                    atomicAdd(&energy[b * nb + bin], e);  // 2.12 -> 2.47 ms at cfg3; a pair's paths rarely share a bin.)
            }
        }
    }
    {   // work counters: one atomic per wave
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(queue_head + kCounterWord);
        unsigned d = my_deposits, t = my_tests;
        for (int o = 32; o > 0; o >>= 1) { d += __shfl_down(d, o); t += __shfl_down(t, o); }
        if (lane == 0) {
            if (my_segments) atomicAdd(&counters[0], (unsigned long long)my_segments);
            if (d) atomicAdd(&counters[2], (unsigned long long)d);
            if (t) atomicAdd(&counters[1], (unsigned long long)t);
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    if (hi < lo) return;
    const int span = hi - lo + 1;
    for (int i = threadIdx.x; i < NB * span; i += kBlock) {
        int b = i / span, bin = lo + (i - b * span);
        float v = s_hist[b * W + bin];
        if (v != 0.0f) atomicAdd(&energy[b * nb + bin], v);
    }
}

// deterministic mode: fixed-point histogram -> the fp32 energy buffer (one rounding per bin, after all sums)
__global__ __launch_bounds__(kBlock) void fixed_to_energy_kernel(const unsigned long long* __restrict__ fixed,
                                                                 float* __restrict__ energy, int words) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < words) energy[i] = (float)((double)fixed[i] * (1.0 / kFixedScale));
}

#ifndef FS_CONNECT_AHEAD_N
#define FS_CONNECT_AHEAD_N 4
#endif
template <int B>
void launch_connect_t(const DeviceScene& sc_in, const KParams& kp, const SubpathState& st, float* energy,
                      unsigned long long* fixed, unsigned* queue_head, int pairs_per_wave, float* const* energy_tab,
                      unsigned long long* const* fixed_tab, hipStream_t s) {
    if (kp.num_local == 0) return;
    DeviceScene sc = sc_in;
    if (pairs_per_wave < 1 || pairs_per_wave > 64) pairs_per_wave = 64;
    const uint32_t per_block = (uint32_t)pairs_per_wave * (kBlock / 64);
    const bool batch = energy_tab != nullptr;
    uint32_t blocks = batch ? (kp.num_local / kp.pairs_per_source) * ((kp.pairs_per_source + per_block - 1) / per_block)
                            : (kp.num_local + per_block - 1) / per_block;
    if (blocks > 1024) blocks = 1024;
    if (!attach_deep(sc, blocks)) return;
    size_t lds = stack_bytes(sc) + sizeof(float) * (size_t)kp.num_bands * (size_t)kp.hist_window + kShareAnyLdsBytes;
#define FS_LAUNCH_CONNECT(L, BT, CN)                                                                                 \
    do {                                                                                                             \
        allow_lds(connect_kernel<B, L, BT, CN>, lds);                                                                \
        hipLaunchKernelGGL((connect_kernel<B, L, BT, CN>), dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy,   \
                           fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab);                                \
    } while (0)
    // FS_FLAG_DOUBLE_POSITIONS / end-point collision spheres: one instantiation pair with the run-time band count and run-time lobes
    if (kp.dpos || kp.listener_radius > 0.0f || kp.source_radius > 0.0f) {
        if (B != 0) return launch_connect_t<0>(sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab, s);
        if (batch) {
            allow_lds(connect_kernel<0, -1, true, false, true>, lds);
            hipLaunchKernelGGL((connect_kernel<0, -1, true, false, true>), dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy, fixed,
                               queue_head, pairs_per_wave, energy_tab, fixed_tab);
        } else {
            allow_lds(connect_kernel<0, -1, false, false, true>, lds);
            hipLaunchKernelGGL((connect_kernel<0, -1, false, false, true>), dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy, fixed,
                               queue_head, pairs_per_wave, energy_tab, fixed_tab);
        }
        return;
    }
    // uncapped walks (the waited-for frames; the pipelined ones connect inside the fused launch): paths of up to a few hundred
    // segments, evaluated by ONE lane when the wave is dense — four records in flight (connect_body's AHEAD)
    static const int ahead = std::getenv("FS_CONNECT_AHEAD") ? std::atoi(std::getenv("FS_CONNECT_AHEAD")) : 4;   // (0 / 1: one at a time)
    if (st.over_levels != 0 && !kp.lobes && !kp.count && ahead > 1) {
        if (batch) {
            allow_lds(connect_kernel<B, 0, true, false, false, FS_CONNECT_AHEAD_N>, lds);
            hipLaunchKernelGGL((connect_kernel<B, 0, true, false, false, FS_CONNECT_AHEAD_N>), dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy, fixed, queue_head,
                               pairs_per_wave, energy_tab, fixed_tab);
        } else {
            allow_lds(connect_kernel<B, 0, false, false, false, FS_CONNECT_AHEAD_N>, lds);
            hipLaunchKernelGGL((connect_kernel<B, 0, false, false, false, FS_CONNECT_AHEAD_N>), dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy, fixed, queue_head,
                               pairs_per_wave, energy_tab, fixed_tab);
        }
        return;
    }
    // record-fetch counting (fs_set_profiling level 3) exists for the default frame shape only
    if (batch) { if (kp.lobes) FS_LAUNCH_CONNECT(1, true, false); else FS_LAUNCH_CONNECT(0, true, false); }
    else if (kp.lobes) FS_LAUNCH_CONNECT(1, false, false);
    else if (kp.count) FS_LAUNCH_CONNECT(0, false, true);
    else FS_LAUNCH_CONNECT(0, false, false);
#undef FS_LAUNCH_CONNECT
}

template <int B>
void launch_connect_all_t(const DeviceScene& sc_in, const KParams& kp, const SubpathState& st, float* energy,
                          unsigned long long* fixed, unsigned* queue_head, hipStream_t s) {
    if (kp.num_local == 0) return;
    DeviceScene sc = sc_in;
    uint32_t blocks = (kp.num_local + 3) / 4;   // one wave per pair, 4 waves per workgroup
    if (blocks > 4096) blocks = 4096;
    if (!attach_deep(sc, blocks)) return;
    size_t lds = stack_bytes(sc) + sizeof(float) * (size_t)kp.num_bands * (size_t)kp.hist_window + kShareAnyLdsBytes;
    allow_lds(connect_all_kernel<B>, lds);
    hipLaunchKernelGGL(connect_all_kernel<B>, dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy, fixed, queue_head);
}

}  // namespace

void launch_connect(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                    unsigned long long* fixed, unsigned* queue_head, int pairs_per_wave, float* const* energy_tab,
                    unsigned long long* const* fixed_tab, hipStream_t s) {
    // instantiated for the band counts in use (the reference: 1; BASELINE.json's configurations: 4 and 8); B = 0 reads
    // kp.num_bands at run time
    switch (B) {
        case 1: launch_connect_t<1>(sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab, s); break;
        case 4: launch_connect_t<4>(sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab, s); break;
        case 8: launch_connect_t<8>(sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab, s); break;
        default: launch_connect_t<0>(sc, kp, st, energy, fixed, queue_head, pairs_per_wave, energy_tab, fixed_tab, s); break;
    }
}

void launch_connect_all(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                        unsigned long long* fixed, unsigned* queue_head, hipStream_t s) {
    switch (B) {
        case 1: launch_connect_all_t<1>(sc, kp, st, energy, fixed, queue_head, s); break;
        case 4: launch_connect_all_t<4>(sc, kp, st, energy, fixed, queue_head, s); break;
        case 8: launch_connect_all_t<8>(sc, kp, st, energy, fixed, queue_head, s); break;
        default: launch_connect_all_t<0>(sc, kp, st, energy, fixed, queue_head, s); break;
    }
}

void launch_fixed_to_energy(const unsigned long long* fixed, float* energy, int words, hipStream_t s) {
    if (words <= 0) return;
    hipLaunchKernelGGL(fixed_to_energy_kernel, dim3((unsigned)((words + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, fixed,
                       energy, words);
}

}  // namespace fs
