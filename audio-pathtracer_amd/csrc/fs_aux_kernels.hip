// fs_aux_kernels.hip — reconstruct_kernel (ReconstructImpulseResponse, FrequenSeeAudioComponent.cpp:320-380), the engine
// line trace for tests / tools, the legacy forward tracer (UpdateSound, :132-306), the reverb plugin's per-callback
// convolution (FrequenSeeAudioReverbPlugin.cpp:118-213), AddEnergyAtDelay on the device-resident buffer.
#include "fs_device.hpp"

namespace fs {
namespace {

// ---------------------------------------------------------------------------------------------------
// reconstruct_kernel: ReconstructImpulseResponse (FSAC.cpp:320-380)
//   rows 0..B-1 = bands, row B = band-mean energy -> the channel view (channels are identical,
//   FSAC.cpp:331).  The one-pole filter y[i] = 0.25 x[i] + 0.75 y[i-1] (FSAC.cpp:366-375) is evaluated
//   per kChunk-sample chunk after a kWarm-sample warm-up: 0.75^96 ~ 1e-12 is far below fp32 resolution.
//   The interpolated sample x[i] is produced incrementally (bin / in-bin counters), no division by spb.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void reconstruct_kernel(const float* __restrict__ energy, int B, int nb,
                                                             int num_samples, int spb, float* __restrict__ ir_bands,
                                                             float* __restrict__ ir_mono) {
    extern __shared__ __attribute__((aligned(16))) float s_amp[];  // reconstruct_body_fast's layout
    reconstruct_body_fast((int)blockIdx.y, (int)blockIdx.x, energy, B, nb, num_samples, spb, ir_bands, ir_mono, s_amp, nullptr);
}

// many sources' reconstructs as one launch: block -> (item, row, block of chunks); the table lives in pinned host memory
__global__ __launch_bounds__(kBlock) void reconstruct_batch_kernel(const ReconItem* __restrict__ table, int B, int nb, int num_samples,
                                                                   uint32_t cb, PublishWord pub) {
    extern __shared__ __attribute__((aligned(16))) float s_rb[];  // reconstruct_body_fast's layout
    const uint32_t per_item = (uint32_t)(B + 1) * cb;
    const uint32_t item = blockIdx.x / per_item, in_item = blockIdx.x - item * per_item;
    const ReconItem it = table[item];
    reconstruct_body_fast((int)(in_item / cb), (int)(in_item % cb), it.energy, B, nb, num_samples, it.spb, it.ir_bands, it.ir_mono, s_rb, it.host, it.mask);
    publish_arrive(pub.tickets, gridDim.x, pub.host_word, pub.id);
}

// ---------------------------------------------------------------------------------------------------
// trace_rays_kernel: the engine line trace (tests / tools)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(DeviceScene sc, const float* __restrict__ o,
                                                            const float* __restrict__ d,
                                                            const float* __restrict__ tmax, int N, int any_hit,
                                                            int32_t* hit, float* t, int32_t* tri, float* normal) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock]
    int* s_stack = s_dyn;
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    Ray r = make_ray(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    Trav T;
    trav_init(T, tmax[i], sc.num_nodes > 0);
    trav_deep_reset(sc, &s_stack[threadIdx.x]);
    if (any_hit) {
        trav_run<true>(sc, r, T, &s_stack[threadIdx.x]);
        hit[i] = T.leaf_index >= 0;
        return;
    }
    trav_run<false>(sc, r, T, &s_stack[threadIdx.x]);
    hit[i] = T.leaf_index >= 0;
    if (T.leaf_index >= 0) {
        float nx, ny, nz;
        uint32_t mat;
        hit_surface(sc, T.leaf_index, r, nx, ny, nz, mat);
        t[i] = T.t;
        tri[i] = (int32_t)T.id;
        normal[3 * i] = nx; normal[3 * i + 1] = ny; normal[3 * i + 2] = nz;
    } else {
        t[i] = tmax[i];
        tri[i] = -1;
        normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f;
    }
}

// the same query through the cooperative traversal (trav_coop): a wave takes R = 1, 2 or 4 rays, each searched by a group of 64 / R lanes
__global__ __launch_bounds__(kBlock) void trace_rays_coop_kernel(DeviceScene sc, CoopView cv, const float* __restrict__ o,
                                                                 const float* __restrict__ d,
                                                                 const float* __restrict__ tmax, int N, int R,
                                                                 int32_t* hit, float* t, int32_t* tri, float* normal, unsigned* overflow) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [lds_nodes][16] records | [waves][kCoopWaveWords]
    coop_stage_nodes(cv, s_dyn);
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int i = wave * R + lane;
    const bool own = lane < R && i < N;
    const int ic = own ? i : 0;
    Ray r = make_ray(o[3 * ic], o[3 * ic + 1], o[3 * ic + 2], d[3 * ic], d[3 * ic + 1], d[3 * ic + 2]);
    Trav T;
    trav_coop<false, false>(sc, cv, R, own, r, tmax[ic], 0xFFFFFFFFu, T, s_dyn, coop_wave_words(cv, s_dyn), overflow);
    if (!own) return;
    hit[i] = T.leaf_index >= 0;
    if (T.leaf_index >= 0) {
        float nx, ny, nz;
        uint32_t mat;
        hit_surface(sc, T.leaf_index, r, nx, ny, nz, mat);
        t[i] = T.t;
        tri[i] = (int32_t)T.id;
        normal[3 * i] = nx; normal[3 * i + 1] = ny; normal[3 * i + 2] = nz;
    } else {
        t[i] = tmax[i];
        tri[i] = -1;
        normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------
// update_sound_shared_kernel: the legacy per-frame forward tracer (UpdateSound FrequenSeeAudioComponent.cpp:283-306,
// CastAudioRay :132-207, CastDirectAudioRay :209-280).  Ray i < N follows specular chain i (with the
// listener-directed transmission ray at every bounce); ray N computes OcclusionAttenuation (:295-299).
// Build-owned engine semantics: actors = object id per triangle, the player pawn = a sphere.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kNoObject = 0xFFFFFFFFu;
constexpr uint32_t kPawnObject = 0xFFFFFFFEu;

struct LegacyHit { float t; uint32_t object; float nx, ny, nz; };

// ---------------------------------------------------------------------------------------------------
// The legacy tracer on sparse waves.  UpdateSound is 1501 rays, each a CHAIN of up to ~20 dependent closest-hit
// queries: on one lane per ray that is 24 waves on a 1024-SIMD chip and the call takes the latency of the longest
// chain (0.9 ms at 100 000 triangles).  update_sound_shared_kernel gives every wave only `rays_per_wave` rays and
// lets the other lanes help: every query of the wave is searched by all 64 lanes (the wave work sharing of
// trav_run_shared, here with the per-ray ignored actor), so a query takes about as many steps as its deepest
// root-to-leaf descent instead of its total node count.  Each lane runs CastAudioRay / CastDirectAudioRay as a
// small state machine (main trace | direct trace | done) so that the whole wave meets at every query.
//   LDS behind the stack rows: ShareArea<false, true>.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void update_sound_shared_kernel(DeviceScene sc, SoundKParams sp, SoundAccum* acc,
                                                                     int rays_per_wave) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock] | share area
    int* stack = &s_dyn[threadIdx.x];
    int* share = s_dyn + (size_t)sc.stack_rows * kBlock;
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
    const int N = sp.raycasts_per_tick;
    const int i = wave * rays_per_wave + lane;
    const bool mine = lane < rays_per_wave && i <= N;
    enum { MAIN = 0, DIRECT = 1, DONE = 2 };
    int mode = DONE;
    bool is_occl = false;
    unsigned long long traces = 0;
    // CastAudioRay state (FSAC.cpp:132-207)
    float px = 0.f, py = 0.f, pz = 0.f, ddx = 0.f, ddy = 0.f, ddz = 0.f;
    float max_distance = sp.raycast_distance;
    int bounces = sp.raycast_bounces;
    const float energy = 1.0f;
    float result = 0.0f, direct_sum = 0.0f;
    unsigned direct_hits = 0;
    // CastDirectAudioRay state (FSAC.cpp:209-280)
    float qx = 0.f, qy = 0.f, qz = 0.f, ex = 0.f, ey = 0.f, ez = 0.f, dmax = 0.f, denergy = 0.f;
    int dbounces = 0;
    uint32_t dactor = kNoObject;

    if (mine && i == N) {                                                  // FSAC.cpp:295-299
        is_occl = true;
        float dx = sp.lis[0] - sp.src[0], dy = sp.lis[1] - sp.src[1], dz = sp.lis[2] - sp.src[2];
        float l2 = dx * dx + dy * dy + dz * dz;
        if (l2 > 0.0f) {
            float inv = 1.0f / sqrtf(l2);
            ex = dx * inv; ey = dy * inv; ez = dz * inv;
            qx = sp.src[0]; qy = sp.src[1]; qz = sp.src[2];
            dmax = sp.raycast_distance; dbounces = 10; denergy = 1.0f; dactor = kNoObject;
            mode = DIRECT;
        } else {
            acc->occlusion = 0.0f;
        }
    } else if (mine) {
        // initial direction: FMath::VRandCone((0,-1,0), PI, PI) FSAC.cpp:291 == theta = 2 pi U, phi = acos(2V-1)
        const uint4 rnd = philox((uint32_t)i, 0u, 0u, sp.seed_lo, sp.seed_hi, 0x46533032u);
        const float U = u01(rnd.x), V = u01(rnd.y);
        const float x = fmaf(V, 2.0f, -1.0f);
        const float sphi = sqrtf(fmaxf(0.0f, fmaf(-x, x, 1.0f)));
        float st, ct;
        sincos2pi(U, st, ct);
        const float nx = 0.0f, ny = -1.0f, nz = 0.0f;
        float sg = copysignf(1.0f, nz);
        float a = -1.0f / (sg + nz);
        float b = nx * ny * a;
        float t0 = fmaf(sg * nx * nx, a, 1.0f), t1 = sg * b, t2 = -sg * nx;
        float b0 = b, b1 = fmaf(ny * ny, a, sg), b2 = -ny;
        float lx = sphi * ct, ly = sphi * st;
        float d0 = fmaf(lx, t0, fmaf(ly, b0, x * nx));
        float d1 = fmaf(lx, t1, fmaf(ly, b1, x * ny));
        float d2 = fmaf(lx, t2, fmaf(ly, b2, x * nz));
        float l2 = d0 * d0 + d1 * d1 + d2 * d2;
        float inv = 1.0f / sqrtf(l2);
        ddx = d0 * inv; ddy = d1 * inv; ddz = d2 * inv;
        px = sp.src[0]; py = sp.src[1]; pz = sp.src[2];
        mode = MAIN;
    }

    while (true) {
        // settle everything that needs no trace: a finished direct ray returns to its caller, a walk out of bounces ends
        for (int guard = 0; guard < 2; ++guard) {
            if (mode == DIRECT && (dbounces == 0 || denergy <= 0.0f)) {   // FSAC.cpp:212: the direct ray died
                if (is_occl) { acc->occlusion = 0.0f; mode = DONE; } else { mode = MAIN; }
            }
            if (mode == MAIN && bounces == 0) mode = DONE;                // FSAC.cpp:134
        }
        const bool has = mode != DONE;
        if (__ballot(has) == 0ull) break;
        // the next query of this lane
        Ray r;
        float tmax = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
        uint32_t ign = kNoObject;
        if (mode == MAIN) {
            float l2 = ddx * ddx + ddy * ddy + ddz * ddz;                 // GetSafeNormal FSAC.cpp:141
            float inv = 1.0f / sqrtf(l2);
            dx = ddx * inv; dy = ddy * inv; dz = ddz * inv;
            r = make_ray(px, py, pz, dx, dy, dz);
            tmax = max_distance;
        } else {
            dx = ex; dy = ey; dz = ez;
            r = make_ray(fmaf(ex, 0.1f, qx), fmaf(ey, 0.1f, qy), fmaf(ez, 0.1f, qz), ex, ey, ez);   // FSAC.cpp:232
            tmax = dmax;
            ign = dactor;
        }
        Trav T;
        trav_shared<false, true>(sc, has, r, tmax, ign, T, stack, share);
        if (!has) continue;
        // legacy_trace: closest of the triangles and the pawn sphere
        ++traces;
        LegacyHit h;
        bool hit;
        {
            float ts;
            const bool hs = sphere_hit(r, sp.lis, sp.listener_radius, tmax, ts);
            const bool ht = T.leaf_index >= 0;
            hit = ht || hs;
            if (hs && (!ht || ts <= T.t)) { h.t = ts; h.object = kPawnObject; h.nx = h.ny = h.nz = 0.f; }
            else if (ht) {
                uint32_t mat;
                hit_surface(sc, T.leaf_index, r, h.nx, h.ny, h.nz, mat);
                h.t = T.t;
                h.object = __float_as_uint(sc.tris[T.leaf_index].c.w);
            }
        }
        if (mode == MAIN) {
            if (!hit) { mode = DONE; continue; }                          // FSAC.cpp:192-196
            const float ipx = fmaf(h.t, dx, px), ipy = fmaf(h.t, dy, py), ipz = fmaf(h.t, dz, pz);
            const float left = max_distance - h.t;                        // DistanceLeft FSAC.cpp:167
            const float tx = sp.lis[0] - ipx, ty = sp.lis[1] - ipy, tz = sp.lis[2] - ipz;
            const float dist_to_player = sqrtf(tx * tx + ty * ty + tz * tz);
            const float travel_time = (sp.raycast_distance - left + dist_to_player) * 0.01f / 343.0f;   // :171
            if (travel_time > sp.simulated_duration) { mode = DONE; continue; }
            if (h.object == kPawnObject) { result = energy; mode = DONE; continue; }   // FSAC.cpp:177-181
            // the reflection (FSAC.cpp:186-187) does not depend on the direct ray: set the next main segment up now
            const float dn = dx * h.nx + dy * h.ny + dz * h.nz;
            ddx = fmaf(-2.0f * dn, h.nx, dx);
            ddy = fmaf(-2.0f * dn, h.ny, dy);
            ddz = fmaf(-2.0f * dn, h.nz, dz);
            px = fmaf(h.nx, 0.5f, ipx); py = fmaf(h.ny, 0.5f, ipy); pz = fmaf(h.nz, 0.5f, ipz);
            max_distance = left;
            bounces -= 1;
            if (dist_to_player > 0.0f) {                                  // FSAC.cpp:184-185: one direct ray to the listener
                const float invp = 1.0f / dist_to_player;
                ex = tx * invp; ey = ty * invp; ez = tz * invp;
                qx = ipx; qy = ipy; qz = ipz;
                dmax = left; dbounces = 1; denergy = energy; dactor = kNoObject;
                mode = DIRECT;
            }
        } else {                                                          // CastDirectAudioRay FSAC.cpp:209-280
            float de = 0.0f;
            bool finished = true;
            if (hit) {
                if (h.object == kPawnObject) {                            // FSAC.cpp:253-270
                    float travel = sp.raycast_distance - dmax + h.t;
                    travel *= 0.01f;
                    float time = travel / 343.0f;
                    if (!(time > sp.simulated_duration)) de = denergy * expf(-0.0017f * travel);
                } else {                                                  // through the obstacle, FSAC.cpp:272-276
                    qx = fmaf(h.t, ex, r.ox); qy = fmaf(h.t, ey, r.oy); qz = fmaf(h.t, ez, r.oz);
                    dmax = dmax - h.t;
                    dbounces -= 1;
                    dactor = h.object;
                    finished = false;
                }
            }
            if (finished) {
                if (is_occl) { acc->occlusion = de; mode = DONE; }
                else { if (de > 0.0f) { ++direct_hits; direct_sum += de; } mode = MAIN; }
            }
        }
    }
    if (mine && !is_occl) {
        if (result > 0.0f) atomicAdd(&acc->reaching, 1u);
        if (direct_hits) { atomicAdd(&acc->direct_hits, direct_hits); atomicAdd(&acc->direct_energy_sum, direct_sum); }
    }
    if (traces) atomicAdd(&acc->traces, traces);
}

// ---------------------------------------------------------------------------------------------------
// Row f2 — the reverb plugin's per-callback convolution (FFrequenSeeAudioReverbPlugin::ProcessSourceAudio,
// FrequenSeeAudioReverbPlugin.cpp:118-170, ConvolveFFT :172-213).  The reference zero-pads the last
// 47 999 + 1 024 samples and the 48 000-tap IR to 65 536 and multiplies three KissFFT spectra; only output
// samples [47 999, 49 023) are kept, for which the circular product equals the plain convolution
//   out[s] = sum_k IR[k] * u[47 999 + s - k].
// On this chip 2 x 1024 x 48 000 MACs are a few microseconds of fp32 FMA, so the kernel evaluates that sum
// directly (no FFT, no 65 536-point scratch, deterministic order): thread t owns a contiguous 192-tap slice
// and slides a 31-sample register window over it (47 loads per 256 FMAs), partial sums meet in LDS.
//   u[j] = j < tail ? ring[(head - tail + j) & mask] : cur[j - tail]
// ---------------------------------------------------------------------------------------------------
constexpr int kRevOut = 16;      // outputs per workgroup
constexpr int kRevRing = 65536;  // history ring length per channel (power of two >= 47 999)

__global__ void reverb_prepare_kernel(const float* __restrict__ in, float* __restrict__ cur, int frame, int literal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame) return;
    // RVB.cpp:147-148 copies the first `frame` floats of the INTERLEAVED buffer into both mono tails
    cur[i] = literal ? in[i] : in[2 * i];
    cur[frame + i] = literal ? in[i] : in[2 * i + 1];
}

__global__ __launch_bounds__(kBlock) void reverb_conv_kernel(const float* __restrict__ ir, int ir_size,
                                                             const float* __restrict__ ring, unsigned head,
                                                             const float* __restrict__ cur, int frame,
                                                             float* __restrict__ out_interleaved) {
    __shared__ float s_part[kRevOut][kBlock + 1];
    const int ch = blockIdx.y;
    const int s0 = blockIdx.x * kRevOut;
    const int tail = ir_size - 1;
    const float* rg = ring + (size_t)ch * kRevRing;
    const float* cu = cur + (size_t)ch * frame;
    const unsigned base = head - (unsigned)tail;   // ring index of u[0]
    const int slice = ((ir_size + kBlock - 1) / kBlock + 15) & ~15;
    const int k0 = (int)threadIdx.x * slice;
    const int k1 = min(k0 + slice, ir_size);
    float acc[kRevOut];
#pragma unroll
    for (int o = 0; o < kRevOut; ++o) acc[o] = 0.0f;
    for (int kb = k0; kb < k1; kb += 16) {
        float w[31], h[16];
        const int j0 = tail + s0 - kb - 15;   // u index of w[0]
#pragma unroll
        for (int i = 0; i < 31; ++i) {
            const int j = j0 + i;
            float v = 0.0f;
            if (j >= 0 && j < tail + frame) v = j < tail ? rg[(base + (unsigned)j) & (unsigned)(kRevRing - 1)] : cu[j - tail];
            w[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = (kb + i) < ir_size ? ir[kb + i] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int o = 0; o < kRevOut; ++o) acc[o] = fmaf(h[i], w[15 - i + o], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < kRevOut; ++o) s_part[o][threadIdx.x] = acc[o];
    __syncthreads();
    for (int stride = kBlock / 2; stride > 0; stride >>= 1) {
        if ((int)threadIdx.x < stride)
#pragma unroll
            for (int o = 0; o < kRevOut; ++o) s_part[o][threadIdx.x] += s_part[o][threadIdx.x + stride];
        __syncthreads();
    }
    if (threadIdx.x < kRevOut && s0 + (int)threadIdx.x < frame) {
        float v = s_part[threadIdx.x][0];
        v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);                 // FMath::Clamp RVB.cpp:165-167, MixAlpha = 1
        out_interleaved[2 * (s0 + (int)threadIdx.x) + ch] = v;
    }
}

// AudioTailBuffer{Left,Right}.AddSamples(in, frame, ch, 2)  RVB.cpp:144-145
__global__ void reverb_push_kernel(const float* __restrict__ in, float* __restrict__ ring, unsigned head, int frame) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame) return;
    ring[(head + (unsigned)i) & (unsigned)(kRevRing - 1)] = in[2 * i];
    ring[kRevRing + ((head + (unsigned)i) & (unsigned)(kRevRing - 1))] = in[2 * i + 1];
}

// AddEnergyAtDelay on the device-resident buffer (FSAC.h:87-91)
__global__ void add_energy_kernel(float* row, int nb, float delay, float e) {
    float x = (delay * 1000.f) / 1.0f;
    float fl = floorf(x);
    int bin = !(fl > 0.0f) ? 0 : (fl >= (float)(nb - 1) ? nb - 1 : (int)fl);
    row[bin] += e;
}

}  // namespace

void launch_reconstruct(const float* energy, int B, int num_bins, int sample_rate, int num_samples, int spb,
                        float* ir_bands, float* ir_mono, hipStream_t s) {
    (void)sample_rate;
    int chunks = (num_samples + kChunk - 1) / kChunk;
    dim3 grid((chunks + kBlock - 1) / kBlock, B + 1);
    const size_t lds = sizeof(float) * ((size_t)num_bins + (size_t)kBlock * kChunk + kWarm + (size_t)kBlock * (kChunk + 1));
    allow_lds(reconstruct_kernel, lds);
    hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(kBlock), lds, s, energy, B, num_bins, num_samples, spb, ir_bands, ir_mono);
}

void launch_reconstruct_batch(const ReconItem* table, int count, int B, int num_bins, int num_samples, hipStream_t s, const PublishWord& pub) {
    if (count <= 0) return;
    const uint32_t chunks = (uint32_t)((num_samples + kChunk - 1) / kChunk), cb = (chunks + kBlock - 1) / kBlock;
    const size_t lds = sizeof(float) * ((size_t)num_bins + (size_t)kBlock * kChunk + kWarm + (size_t)kBlock * (kChunk + 1));
    allow_lds(reconstruct_batch_kernel, lds);
    hipLaunchKernelGGL(reconstruct_batch_kernel, dim3((uint32_t)count * (uint32_t)(B + 1) * cb), dim3(kBlock), lds, s, table, B, num_bins,
                       num_samples, cb, pub);
}

void launch_trace_rays(const DeviceScene& sc_in, const float* o, const float* d, const float* tmax, int N, int any_hit,
                       int32_t* hit, float* t, int32_t* tri, float* normal, hipStream_t s) {
    if (N <= 0) return;
    DeviceScene sc = sc_in;
    if (any_hit >= 2) {   // closest hit through the cooperative traversal: 2 / 3 / 4 = 1, 2, 4 rays per wave, every record from global memory;
        //                   5 / 6 / 7 = the same with the top of the tree resident in LDS (5 sixteen-wide nodes: both kinds of fetch in one query) ; 8: all that fit.
        //                   `hit` doubles as the overflow word's home (hit[N])
        const int m = any_hit >= 8 ? 2 : (any_hit - 2) % 3;
        const int R = m == 0 ? 1 : (m == 1 ? 2 : 4);
        const int waves = (N + R - 1) / R;
        const uint32_t blocks = (uint32_t)((waves + kBlock / 64 - 1) / (kBlock / 64));
        const CoopView* cvp = coop_view(sc, R);
        if (!cvp) { launch_trace_rays(sc_in, o, d, tmax, N, 0, hit, t, tri, normal, s); return; }   // (a tree the cooperative stack cannot hold: the lane-private traversal)
        CoopView cv = *cvp;
        cv.lds_nodes = any_hit < 5 ? 0 : (any_hit < 8 ? std::min(cv.nodes, 5) : coop_resident_nodes(cv, kBlock / 64, blocks, 0));
        const size_t lds = coop_lds_bytes(kBlock / 64, cv);
        allow_lds(trace_rays_coop_kernel, lds);
        hipLaunchKernelGGL(trace_rays_coop_kernel, dim3(blocks), dim3(kBlock), lds, s, sc, cv, o, d, tmax, N, R,
                           hit, t, tri, normal, reinterpret_cast<unsigned*>(hit + N));
        return;
    }
    if (!attach_deep(sc, (uint32_t)((N + kBlock - 1) / kBlock))) return;
    allow_lds(trace_rays_kernel, stack_bytes(sc));
    hipLaunchKernelGGL(trace_rays_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), stack_bytes(sc), s, sc, o, d, tmax, N,
                       any_hit, hit, t, tri, normal);
}

void launch_update_sound(const DeviceScene& sc_in, const SoundKParams& sp, SoundAccum* acc, int rays_per_wave, hipStream_t s) {
    int lanes = sp.raycasts_per_tick + 1;
    if (rays_per_wave > 64 || rays_per_wave <= 0) rays_per_wave = 64;   // 64 = one ray per lane (finished lanes still help)
    const int waves = (lanes + rays_per_wave - 1) / rays_per_wave;
    DeviceScene sc = sc_in;
    if (!attach_deep(sc, (uint32_t)((waves + kBlock / 64 - 1) / (kBlock / 64)))) return;
    const size_t lds = stack_bytes(sc) + kShareIgnLdsBytes;
    allow_lds(update_sound_shared_kernel, lds);
    hipLaunchKernelGGL(update_sound_shared_kernel, dim3((waves + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), lds, s, sc, sp,
                       acc, rays_per_wave);
}

void launch_reverb(const float* ir, int ir_size, float* ring, unsigned head, const float* in, float* cur, float* out,
                   int frame, int literal_tail, hipStream_t s) {
    const int tb = 256;
    hipLaunchKernelGGL(reverb_prepare_kernel, dim3((frame + tb - 1) / tb), dim3(tb), 0, s, in, cur, frame, literal_tail);
    hipLaunchKernelGGL(reverb_conv_kernel, dim3((frame + kRevOut - 1) / kRevOut, 2), dim3(kBlock), 0, s, ir, ir_size,
                       ring, head, cur, frame, out);
    hipLaunchKernelGGL(reverb_push_kernel, dim3((frame + tb - 1) / tb), dim3(tb), 0, s, in, ring, head, frame);
}

#ifdef FS_WAVE_TIMELINE
extern "C" void fs_debug_wave_buffer(unsigned long long* device_ptr) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_wave_buf), &device_ptr, sizeof(device_ptr));
}
extern "C" void fs_debug_connect_buffer(unsigned long long* device_ptr) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_conn_buf), &device_ptr, sizeof(device_ptr));
}
#endif
#ifdef FS_TRAV_STATS
extern "C" void fs_debug_trav_stats(unsigned long long* out, int reset) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trav_stats), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trav_stats), z, sizeof(z)); }
}
extern "C" void fs_debug_step_buffer(unsigned short* device_ptr) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_step_buf), &device_ptr, sizeof(device_ptr));
}
#endif

size_t traversal_lds_bytes(int stack_rows, int bands, int num_bins) {
    const size_t stack = sizeof(int) * (size_t)stack_rows * (size_t)kBlock;
    const size_t walk = stack + std::max(kShareLdsBytes, kShareIgnLdsBytes);
    const size_t connect = stack + sizeof(float) * (size_t)bands * (size_t)std::min(num_bins, default_hist_window(bands)) + kShareAnyLdsBytes;
    return std::max(walk, connect) + 1024;   // + the kernels' small static arrays
}

// Histogram bins the connect part keeps in LDS by default: as many as fit into what the walk part of the same launch
// needs anyway (stack + its share area), so that the histogram never decides how many workgroups a CU holds — with a
// 256-bin window at 8 bands a tree needing 33+ stack rows dropped from three resident workgroups to two (-20 %,
// profiles/r03_ab_tree.log); bins beyond the window take global atomics, and a cfg3 frame touches bins 0..60.
int default_hist_window(int bands) {
    const size_t spare = kShareLdsBytes > kShareAnyLdsBytes ? kShareLdsBytes - kShareAnyLdsBytes : 0;
    const int fit = (int)(spare / (sizeof(float) * (size_t)std::max(bands, 1))) & ~15;
    return std::max(64, std::min(kHistWindow, fit));
}

void launch_add_energy(float* energy_row, int num_bins, float delay_s, float e, hipStream_t s) {
    hipLaunchKernelGGL(add_energy_kernel, dim3(1), dim3(1), 0, s, energy_row, num_bins, delay_s, e);
}

}  // namespace fs
