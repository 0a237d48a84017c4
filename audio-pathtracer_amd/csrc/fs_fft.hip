// fs_fft.hip — frequency-domain material filter for gfx950 (row f4).
//
// Replaces UMaterialAcousticProcessor::ApplyMaterialFD
// (Plugins/FrequenSee/Source/FrequenSee/Private/MaterialAcousticProcessor.cpp:8-107): one forward FFT of the
// zero-padded block, per-bin specular / diffuse / transmitted gains (:51-72), three inverse FFTs, scale 1/N.
//
// Design: a radix-2 complex FFT pair that never runs a bit-reversal pass.
//   forward  = decimation in frequency, natural order in  -> bit-reversed order out;
//   the gains are applied in bit-reversed order (the bin of position p is brev(p));
//   inverse  = decimation in time, bit-reversed order in -> natural order out.
// Stages whose butterfly span fits a 2048-point chunk run inside LDS (one workgroup per chunk, 11 stages per
// launch); wider spans are one global pass each.  For the plugin's block sizes (N <= 65536) that is
// 1 + 5 + 1 launches forward and the same inverse, the three inverse transforms batched over blockIdx.y.
// Twiddles come from a table computed in double precision on the host (W[k] = exp(-2 pi i k / N), k < N/2).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr int kFftBlock = 256;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 w) {
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}
__device__ __forceinline__ float2 cmul_conj(float2 a, float2 w) {   // a * conj(w)
    return make_float2(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y);
}

// MAP.cpp:51-72 — the gain of output `which` (0 specular, 1 diffuse, 2 transmitted) at one bin
__device__ __forceinline__ float material_gain(int which, float alpha, float tau, float sigma) {
    const float refl = 1.0f - alpha;
    if (refl + tau > 1.0f) tau = 1.0f - refl;   // energy conservation
    return which == 0 ? refl * (1.0f - sigma) : (which == 1 ? refl * sigma : tau);
}

// ---- forward, spans >= one chunk: one butterfly per thread ----
// in_real != nullptr: first pass, reads the zero-padded real block instead of x
__global__ __launch_bounds__(kFftBlock) void fft_dif_global(const float* __restrict__ in_real, int L,
                                                            float2* __restrict__ x, const float2* __restrict__ W,
                                                            int n, int ls) {
    const uint32_t t = blockIdx.x * kFftBlock + threadIdx.x;
    const uint32_t half = 1u << (n - 1);
    if (t >= half) return;
    const uint32_t s = 1u << ls;
    const uint32_t j = t & (s - 1u);
    const uint32_t i0 = ((t >> ls) << (ls + 1)) + j, i1 = i0 + s;
    float2 a, b;
    if (in_real) {
        a = make_float2(i0 < (uint32_t)L ? in_real[i0] : 0.0f, 0.0f);
        b = make_float2(i1 < (uint32_t)L ? in_real[i1] : 0.0f, 0.0f);
    } else {
        a = x[i0]; b = x[i1];
    }
    x[i0] = cadd(a, b);
    x[i1] = cmul(csub(a, b), W[(size_t)j << (n - 1 - ls)]);
}

// ---- forward, the last c stages inside LDS on chunks of 2^c points ----
__global__ __launch_bounds__(kFftBlock) void fft_dif_local(const float* __restrict__ in_real, int L,
                                                           float2* __restrict__ x, const float2* __restrict__ W,
                                                           int n, int c) {
    extern __shared__ __attribute__((aligned(16))) float2 sh[];
    const uint32_t C = 1u << c;
    const uint32_t base = blockIdx.x << c;
    for (uint32_t i = threadIdx.x; i < C; i += kFftBlock) {
        const uint32_t p = base + i;
        sh[i] = in_real ? make_float2(p < (uint32_t)L ? in_real[p] : 0.0f, 0.0f) : x[p];
    }
    for (int ls = c - 1; ls >= 0; --ls) {
        __syncthreads();
        const uint32_t s = 1u << ls;
        for (uint32_t t = threadIdx.x; t < (C >> 1); t += kFftBlock) {
            const uint32_t j = t & (s - 1u);
            const uint32_t i0 = ((t >> ls) << (ls + 1)) + j, i1 = i0 + s;
            const float2 a = sh[i0], b = sh[i1];
            sh[i0] = cadd(a, b);
            sh[i1] = cmul(csub(a, b), W[(size_t)j << (n - 1 - ls)]);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < C; i += kFftBlock) x[base + i] = sh[i];
}

// ---- inverse, first c stages inside LDS; applies the gains while loading.  blockIdx.y = output ----
// c == n: the whole transform fits one chunk and the kernel writes the scaled real block itself.
__global__ __launch_bounds__(kFftBlock) void fft_dit_local(const float2* __restrict__ X, float2* __restrict__ y,
                                                           const float* __restrict__ absorption,
                                                           const float* __restrict__ transmission,
                                                           const float* __restrict__ scattering,
                                                           const float2* __restrict__ W, int n, int c, int L,
                                                           float* __restrict__ out, float scale) {
    extern __shared__ __attribute__((aligned(16))) float2 sh[];
    const int which = blockIdx.y;
    const uint32_t C = 1u << c;
    const uint32_t N = 1u << n;
    const uint32_t base = blockIdx.x << c;
    for (uint32_t i = threadIdx.x; i < C; i += kFftBlock) {
        const uint32_t p = base + i;
        const uint32_t k = n ? (__brev(p) >> (32 - n)) : 0u;   // the bin stored at position p
        const uint32_t kk = k <= (N >> 1) ? k : N - k;          // real input: X[N-k] = conj(X[k]), same gain
        const float g = material_gain(which, absorption[kk], transmission[kk], scattering[kk]);
        const float2 v = X[p];
        sh[i] = make_float2(v.x * g, v.y * g);
    }
    for (int ls = 0; ls < c; ++ls) {
        __syncthreads();
        const uint32_t s = 1u << ls;
        for (uint32_t t = threadIdx.x; t < (C >> 1); t += kFftBlock) {
            const uint32_t j = t & (s - 1u);
            const uint32_t i0 = ((t >> ls) << (ls + 1)) + j, i1 = i0 + s;
            const float2 a = sh[i0];
            const float2 b = cmul_conj(sh[i1], W[(size_t)j << (n - 1 - ls)]);
            sh[i0] = cadd(a, b);
            sh[i1] = csub(a, b);
        }
    }
    __syncthreads();
    if (c == n) {
        for (uint32_t i = threadIdx.x; i < C && i < (uint32_t)L; i += kFftBlock)
            out[(size_t)which * L + i] = sh[i].x * scale;
    } else {
        for (uint32_t i = threadIdx.x; i < C; i += kFftBlock) y[(size_t)which * N + base + i] = sh[i];
    }
}

// ---- inverse, spans >= one chunk; the last stage (ls == n-1) writes the scaled real block ----
__global__ __launch_bounds__(kFftBlock) void fft_dit_global(float2* __restrict__ y, const float2* __restrict__ W, int n,
                                                            int ls, int L, float* __restrict__ out, float scale) {
    const uint32_t t = blockIdx.x * kFftBlock + threadIdx.x;
    const uint32_t half = 1u << (n - 1);
    if (t >= half) return;
    const int which = blockIdx.y;
    float2* yy = y + ((size_t)which << n);
    const uint32_t s = 1u << ls;
    const uint32_t j = t & (s - 1u);
    const uint32_t i0 = ((t >> ls) << (ls + 1)) + j, i1 = i0 + s;
    const float2 a = yy[i0];
    const float2 b = cmul_conj(yy[i1], W[(size_t)j << (n - 1 - ls)]);
    if (ls == n - 1) {
        if (i0 < (uint32_t)L) out[(size_t)which * L + i0] = (a.x + b.x) * scale;
        if (i1 < (uint32_t)L) out[(size_t)which * L + i1] = (a.x - b.x) * scale;
    } else {
        yy[i0] = cadd(a, b);
        yy[i1] = csub(a, b);
    }
}

}  // namespace

// x: [N] complex work buffer, y: [3][N], W: [max(N/2,1)] twiddles, resp: absorption | transmission | scattering
// ([3][N/2+1]), out: [3][L] (specular | diffuse | transmitted).  n = log2 N.
void launch_apply_material_fd(const float* in, int L, int n, float2* x, float2* y, const float2* W, const float* resp,
                              float* out, hipStream_t s) {
    const int N = 1 << n;
    const int bins = N / 2 + 1;
    const int c = n < kFftChunkLog ? n : kFftChunkLog;
    const size_t lds = sizeof(float2) << c;
    const unsigned half_blocks = n ? (unsigned)(((N >> 1) + kFftBlock - 1) / kFftBlock) : 0u;
    const float scale = 1.0f / (float)N;   // MAP.cpp:88
    // forward
    for (int ls = n - 1; ls >= c; --ls)
        hipLaunchKernelGGL(fft_dif_global, dim3(half_blocks), dim3(kFftBlock), 0, s, ls == n - 1 ? in : nullptr, L, x, W,
                           n, ls);
    hipLaunchKernelGGL(fft_dif_local, dim3(1u << (n - c)), dim3(kFftBlock), lds, s, c == n ? in : nullptr, L, x, W, n, c);
    // gains + inverse, three outputs at once
    hipLaunchKernelGGL(fft_dit_local, dim3(1u << (n - c), 3), dim3(kFftBlock), lds, s, x, y, resp, resp + bins,
                       resp + 2 * bins, W, n, c, L, out, scale);
    for (int ls = c; ls < n; ++ls)
        hipLaunchKernelGGL(fft_dit_global, dim3(half_blocks, 3), dim3(kFftBlock), 0, s, y, W, n, ls, L, out, scale);
}

}  // namespace fs
