// fs_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, wave64) of the FrequenSee BDPT path.
//
//   walk_kernel        GeneratePath (AudioRayTracingSubsystem.cpp:279-355) for every source and listener
//                      subpath, one subpath per lane, with the EvaluatePath (:360-420) segment terms applied
//                      in-register as the walk proceeds (the path need not be stored).
//   connect_kernel     ConnectSubpaths (:235-277) any-hit visibility ray per pair, the connection
//                      segment's EvaluatePath term, clamp/gain (:410-413), normalisation (:164-170) and
//                      AddEnergyAtDelay (FrequenSeeAudioComponent.h:87-91) into an LDS-privatised
//                      [bands][bins] histogram flushed with global float atomics.
//   reconstruct_kernel ReconstructImpulseResponse (FrequenSeeAudioComponent.cpp:320-380).
//   trace_rays_kernel  the engine line trace itself (closest / any hit), for tests and tools.
//
// The triangle test, the hit point/normal/offset arithmetic and the sampling maps use a fixed
// operation order with explicit fmaf and are compiled with -ffp-contract=off: the path geometry is a
// pure function of (scene, seed, pair index) and does not depend on launch geometry or on the BVH.
#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr float kPi = 3.1415926535897932f;
constexpr uint32_t kNoMat = FS_NO_MATERIAL;

// ---------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = (pair, bounce<<1|side, block, 'FS01'), key = seed
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox(uint32_t pair, uint32_t bs, uint32_t block, uint32_t k0, uint32_t k1) {
    uint32_t c0 = pair, c1 = bs, c2 = block, c3 = 0x46533031u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

// sin/cos(2 pi u): quadrant reduction + fixed fmaf polynomials (bit-reproducible, unlike sinf/cosf)
__device__ __forceinline__ void sincos2pi(float u, float& s_out, float& c_out) {
    float q = floorf(fmaf(u, 4.0f, 0.5f));
    float a = fmaf(q, -0.25f, u);
    float x = a * 6.283185307179586f;
    float x2 = x * x;
    float sp = 2.7557319e-06f;
    sp = fmaf(sp, x2, -1.9841270e-04f);
    sp = fmaf(sp, x2, 8.3333333e-03f);
    sp = fmaf(sp, x2, -1.6666667e-01f);
    float s = fmaf(sp * x2, x, x);
    float cp = 2.4801587e-05f;
    cp = fmaf(cp, x2, -1.3888889e-03f);
    cp = fmaf(cp, x2, 4.1666667e-02f);
    cp = fmaf(cp, x2, -0.5f);
    float c = fmaf(cp, x2, 1.0f);
    int k = ((int)q) & 3;
    s_out = (k == 0) ? s : (k == 1) ? c : (k == 2) ? -s : -c;
    c_out = (k == 0) ? c : (k == 1) ? -s : (k == 2) ? -c : s;
}

// FMath::VRand: cube rejection until 1e-4 < |v|^2 <= 1, normalise (ARTS.cpp:308)
__device__ __forceinline__ void sample_sphere(uint32_t pair, uint32_t bs, uint4 r0, uint32_t k0, uint32_t k1,
                                              float& dx, float& dy, float& dz) {
    uint32_t a = r0.y, b = r0.z, c = r0.w;
    dx = 0.f; dy = 0.f; dz = 1.f;
    for (uint32_t attempt = 0; attempt < 16; ++attempt) {
        if (attempt > 0) {
            uint4 r = philox(pair, bs, attempt, k0, k1);
            a = r.x; b = r.y; c = r.z;
        }
        float x = fmaf(u01(a), 2.0f, -1.0f);
        float y = fmaf(u01(b), 2.0f, -1.0f);
        float z = fmaf(u01(c), 2.0f, -1.0f);
        float l2 = x * x + y * y + z * z;
        if (l2 > 1e-4f && l2 <= 1.0f) {
            float inv = 1.0f / sqrtf(l2);
            dx = x * inv; dy = y * inv; dz = z * inv;
            return;
        }
    }
}

// FMath::VRandCone(n, 90 deg) (ARTS.cpp:313) or cosine-weighted (compat flag)
__device__ __forceinline__ void sample_cone(float nx, float ny, float nz, float U, float V, int cosine, float& dx,
                                            float& dy, float& dz) {
    float cphi, sphi;
    if (cosine) {
        cphi = sqrtf(1.0f - V);
        sphi = sqrtf(V);
    } else {
        float x = fmaf(V, 2.0f, -1.0f);
        float r = sqrtf(fmaxf(0.0f, fmaf(-x, x, 1.0f)));
        if (x > 0.0f) { cphi = x; sphi = r; } else { cphi = r; sphi = -x; }
    }
    float st, ct;
    sincos2pi(U, st, ct);
    float sg = copysignf(1.0f, nz);
    float a = -1.0f / (sg + nz);
    float b = nx * ny * a;
    float t0 = fmaf(sg * nx * nx, a, 1.0f), t1 = sg * b, t2 = -sg * nx;
    float b0 = b, b1 = fmaf(ny * ny, a, sg), b2 = -ny;
    float lx = sphi * ct, ly = sphi * st;
    float d0 = fmaf(lx, t0, fmaf(ly, b0, cphi * nx));
    float d1 = fmaf(lx, t1, fmaf(ly, b1, cphi * ny));
    float d2 = fmaf(lx, t2, fmaf(ly, b2, cphi * nz));
    float l2 = d0 * d0 + d1 * d1 + d2 * d2;
    float inv = 1.0f / sqrtf(l2);
    dx = d0 * inv; dy = d1 * inv; dz = d2 * inv;
}

// ---------------------------------------------------------------------------------------------------
// ray / triangle / box
// ---------------------------------------------------------------------------------------------------
struct Ray {
    float ox, oy, oz, dx, dy, dz;
    float ix, iy, iz;     // safe reciprocals for the slab test
    float nox, noy, noz;  // -o * inv: slab distances become one fma per plane
};

// Box tests only need to be conservative (boxes are padded far beyond this error), so the hardware
// reciprocal approximation is fine here; the triangle test uses IEEE division.
__device__ __forceinline__ float safe_rcp(float x) {
    if (fabsf(x) < 1e-20f) x = copysignf(1e-20f, x);
    return __builtin_amdgcn_rcpf(x);
}

__device__ __forceinline__ Ray make_ray(float ox, float oy, float oz, float dx, float dy, float dz) {
    Ray r;
    r.ox = ox; r.oy = oy; r.oz = oz; r.dx = dx; r.dy = dy; r.dz = dz;
    r.ix = safe_rcp(dx); r.iy = safe_rcp(dy); r.iz = safe_rcp(dz);
    r.nox = -(ox * r.ix); r.noy = -(oy * r.iy); r.noz = -(oz * r.iz);
    return r;
}

// Moeller-Trumbore, two-sided, accepts t in (0, tmax].  Operation order is part of the spec.
__device__ __forceinline__ bool tri_hit(const float4 A, const float4 Bq, const float4 Cq, const Ray& r, float tmax,
                                        float& t_out) {
    const float v0x = A.x, v0y = A.y, v0z = A.z;
    const float e1x = A.w, e1y = Bq.x, e1z = Bq.y;
    const float e2x = Bq.z, e2y = Bq.w, e2z = Cq.x;
    float px = fmaf(r.dy, e2z, -(r.dz * e2y));
    float py = fmaf(r.dz, e2x, -(r.dx * e2z));
    float pz = fmaf(r.dx, e2y, -(r.dy * e2x));
    float det = fmaf(e1x, px, fmaf(e1y, py, e1z * pz));
    if (det == 0.0f) return false;
    // barycentric tests on the un-normalised values, sign-normalised by det (exact: sign-bit xor);
    // the one IEEE division is only paid by rays that are inside the triangle
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    const float ad = fabsf(det);
    float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    float U = fmaf(sx, px, fmaf(sy, py, sz * pz));
    float us = __uint_as_float(__float_as_uint(U) ^ sgn);
    if (!(us >= 0.0f && us <= ad)) return false;
    float qx = fmaf(sy, e1z, -(sz * e1y));
    float qy = fmaf(sz, e1x, -(sx * e1z));
    float qz = fmaf(sx, e1y, -(sy * e1x));
    float V = fmaf(r.dx, qx, fmaf(r.dy, qy, r.dz * qz));
    float vs = __uint_as_float(__float_as_uint(V) ^ sgn);
    if (!(vs >= 0.0f && (us + vs) <= ad)) return false;
    float t = fmaf(e2x, qx, fmaf(e2y, qy, e2z * qz)) / det;
    if (!(t > 0.0f && t <= tmax)) return false;
    t_out = t;
    return true;
}

__device__ __forceinline__ float slab(float lx, float ly, float lz, float hx, float hy, float hz, const Ray& r,
                                      float tmax, bool& hit) {
    float t0x = fmaf(lx, r.ix, r.nox), t1x = fmaf(hx, r.ix, r.nox);
    float t0y = fmaf(ly, r.iy, r.noy), t1y = fmaf(hy, r.iy, r.noy);
    float t0z = fmaf(lz, r.iz, r.noz), t1z = fmaf(hz, r.iz, r.noz);
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    hit = tn <= tf;
    return tn;
}

struct HitRec {
    float t;
    int32_t leaf_index;  // index into the leaf-ordered triangle array, -1 = miss
    uint32_t id;         // input triangle index (tie-break key)
};

// Stack-based BVH2 walk, one ray per lane, "while-while" form: every lane first descends inner nodes
// until it holds a leaf (or is done), THEN the wave intersects leaves together — box tests and triangle
// tests no longer serialise against each other inside one loop body.  `stack` is this lane's column of
// the workgroup's LDS stack (element i at stack[i * kBlock]).  ANY = stop at the first hit.
template <bool ANY>
__device__ __forceinline__ HitRec traverse(const DeviceScene& sc, const Ray& r, float tmax, int* stack) {
    constexpr int kDone = (int)0x80000000;
    HitRec best;
    best.t = tmax; best.leaf_index = -1; best.id = 0xFFFFFFFFu;
    if (sc.num_nodes == 0) return best;
    int sp = 0;
    int cur = 0;  // >= 0: inner node index; < 0: leaf code (~cur = first*4 + count-1) or kDone
    const float4* nodes4 = reinterpret_cast<const float4*>(sc.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
    while (cur != kDone) {
        // ---- phase 1: inner nodes. one 64-B fetch decides both children ----
        while (cur >= 0) {
            const float4 q0 = nodes4[4 * cur + 0];
            const float4 q1 = nodes4[4 * cur + 1];
            const float4 q2 = nodes4[4 * cur + 2];
            const float4 q3 = nodes4[4 * cur + 3];
            bool h0, h1;
            float t0 = slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, best.t, h0);
            float t1 = slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r, best.t, h1);
            const int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (h0 && h1) {
                const bool sw = t1 < t0;
                stack[sp * kBlock] = sw ? c0 : c1;  // far child waits
                ++sp;
                cur = sw ? c1 : c0;
            } else if (h0) {
                cur = c0;
            } else if (h1) {
                cur = c1;
            } else if (sp > 0) {
                --sp;
                cur = stack[sp * kBlock];
            } else {
                cur = kDone;
            }
        }
        // ---- phase 2: leaf (1..4 contiguous 48-B triangle records) ----
        if (cur != kDone) {
            const int code = ~cur;
            const int ft = code >> 2, cnt = (code & 3) + 1;
            for (int i = 0; i < cnt; ++i) {
                const float4 A = tris4[3 * (ft + i) + 0];
                const float4 Bq = tris4[3 * (ft + i) + 1];
                const float4 Cq = tris4[3 * (ft + i) + 2];
                float t;
                if (tri_hit(A, Bq, Cq, r, best.t, t)) {
                    if (ANY) { best.t = t; best.leaf_index = ft + i; return best; }
                    uint32_t id = __float_as_uint(Cq.z);
                    if (t < best.t || best.leaf_index < 0 || (t == best.t && id < best.id)) {
                        best.t = t; best.leaf_index = ft + i; best.id = id;
                    }
                }
            }
            if (sp > 0) { --sp; cur = stack[sp * kBlock]; } else { cur = kDone; }
        }
    }
    return best;
}

// geometric normal of the hit triangle, unit, flipped to face the ray origin side (ImpactNormal)
__device__ __forceinline__ void hit_normal(const float4 A, const float4 Bq, const float4 Cq, const Ray& r, float& nx,
                                           float& ny, float& nz) {
    const float e1x = A.w, e1y = Bq.x, e1z = Bq.y;
    const float e2x = Bq.z, e2y = Bq.w, e2z = Cq.x;
    float x = fmaf(e1y, e2z, -(e1z * e2y));
    float y = fmaf(e1z, e2x, -(e1x * e2z));
    float z = fmaf(e1x, e2y, -(e1y * e2x));
    float l2 = x * x + y * y + z * z;
    float inv = 1.0f / sqrtf(l2);
    x *= inv; y *= inv; z *= inv;
    float dn = fmaf(x, r.dx, fmaf(y, r.dy, z * r.dz));
    if (dn > 0.0f) { x = -x; y = -y; z = -z; }
    nx = x; ny = y; nz = z;
}

// one EvaluatePath segment term on E[b] (ARTS.cpp:381-398), in the reference's operation order
template <int B>
__device__ __forceinline__ void apply_segment(float (&E)[B], float nd, uint32_t mat, float prob, const KParams& kp,
                                              const DeviceScene& sc) {
    if (nd < kp.min_seg) return;  // ARTS.cpp:375-378
    float nd2 = nd * nd;
    float geo = 1.0f / (4 * kPi * nd2);            // ARTS.cpp:391
    float pw = powf(prob, kp.prob_exponent);       // ARTS.cpp:398
    bool has = (mat != kNoMat) && ((int32_t)mat < sc.num_materials);
#pragma unroll
    for (int b = 0; b < B; ++b) {
        float bsdf = has ? sc.absorption[mat * B + b] / kPi : 1.0f;   // ARTS.cpp:382-386
        float e = E[b];
        e *= bsdf;
        e *= geo;
        e *= expf(-kp.air[b] * nd);                // ARTS.cpp:395-397
        e /= pw;
        E[b] = e;
    }
}

// ---------------------------------------------------------------------------------------------------
// walk_kernel: GeneratePath for 2 * num_local subpaths, one per lane
// ---------------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(kBlock) void walk_kernel(DeviceScene sc, KParams kp, SubpathState st) {
    __shared__ int s_stack[kStackDepth * kBlock];
    const uint32_t n = kp.num_local;
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= 2u * n) return;
    const uint32_t side = g >= n ? 1u : 0u;
    const uint32_t li = g - side * n;
    const uint32_t pair = kp.pair_begin + li;
    int* stack = &s_stack[threadIdx.x];

    // state variables ARTS.cpp:287-291
    float px = side ? kp.lis[0] : kp.src[0];
    float py = side ? kp.lis[1] : kp.src[1];
    float pz = side ? kp.lis[2] : kp.src[2];
    float nx = 0.f, ny = 0.f, nz = 0.f;
    bool has_normal = false;
    uint32_t mat = kNoMat;
    float prob = 1.0f;
    float E[B];
#pragma unroll
    for (int b = 0; b < B; ++b) E[b] = 1.0f;
    float sd = 0.0f;
    int k = 0;
    for (; k < kp.depth; ++k) {
        const uint32_t bs = ((uint32_t)k << 1) | side;
        uint4 r = philox(pair, bs, 0, kp.seed_lo, kp.seed_hi);
        if (kp.russian_roulette && !(u01(r.x) < kp.rr_prob)) break;   // ARTS.cpp:300-301, 349-353
        float dx, dy, dz, prob_new;
        if (!has_normal) {                                            // ARTS.cpp:306-310
            sample_sphere(pair, bs, r, kp.seed_lo, kp.seed_hi, dx, dy, dz);
            float pdf = 1.0f / (4.0f * kPi);
            prob_new = pdf * kp.rr_prob;
        } else {                                                      // ARTS.cpp:311-318
            sample_cone(nx, ny, nz, u01(r.y), u01(r.z), kp.cosine, dx, dy, dz);
            float cos_theta = dx * nx + dy * ny + dz * nz;
            float pdf = cos_theta / kPi;
            prob_new = pdf * kp.rr_prob;
        }
        Ray ray = make_ray(px, py, pz, dx, dy, dz);
        HitRec h = traverse<false>(sc, ray, kp.max_trace_dist, stack);  // ARTS.cpp:339-342
        float qx = px, qy = py, qz = pz;
        uint32_t mat_new = mat;
        if (h.leaf_index >= 0) {                                      // ARTS.cpp:345-347
            const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
            const float4 A = tris4[3 * h.leaf_index + 0];
            const float4 Bq = tris4[3 * h.leaf_index + 1];
            const float4 Cq = tris4[3 * h.leaf_index + 2];
            hit_normal(A, Bq, Cq, ray, nx, ny, nz);
            qx = fmaf(kp.surface_offset, nx, fmaf(h.t, dx, px));
            qy = fmaf(kp.surface_offset, ny, fmaf(h.t, dy, py));
            qz = fmaf(kp.surface_offset, nz, fmaf(h.t, dz, pz));
            has_normal = true;
            mat_new = __float_as_uint(Cq.y);
        }
        // the segment just added (zero length on a miss: the duplicate node of ARTS.cpp:296)
        float ddx = qx - px, ddy = qy - py, ddz = qz - pz;
        float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);        // ARTS.cpp:372
        float nd = dist / kp.dist_divisor;                            // ARTS.cpp:373
        if (side == 0) {
            sd += nd;                                                 // ARTS.cpp:374, path order
            apply_segment<B>(E, nd, mat, prob, kp, sc);               // node i = departure node
        } else {
            st.seg_nd[(size_t)k * n + li] = nd;                       // summed in path order by connect
            apply_segment<B>(E, nd, mat_new, prob_new, kp, sc);       // node i = arrival node (reversed walk)
        }
        px = qx; py = qy; pz = qz;
        mat = mat_new;
        prob = prob_new;
    }
    st.pos_sd[g] = make_float4(px, py, pz, sd);
    st.misc[g] = make_float4(prob, __uint_as_float(mat), __int_as_float(k), 0.f);
#pragma unroll
    for (int b = 0; b < B; ++b) st.energy[(size_t)b * 2u * n + g] = E[b];
}

// ---------------------------------------------------------------------------------------------------
// connect_kernel: ConnectSubpaths + connection term + clamp/gain + deposit
// ---------------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(kBlock) void connect_kernel(DeviceScene sc, KParams kp, SubpathState st,
                                                         float* __restrict__ energy) {
    __shared__ int s_stack[kStackDepth * kBlock];
    extern __shared__ __attribute__((aligned(16))) float s_hist[];  // [B][num_bins]
    __shared__ int s_lo, s_hi;
    const int nb = kp.num_bins;
    for (int i = threadIdx.x; i < B * nb; i += kBlock) s_hist[i] = 0.0f;
    if (threadIdx.x == 0) { s_lo = nb; s_hi = -1; }
    __syncthreads();

    const uint32_t n = kp.num_local;
    for (uint32_t li = blockIdx.x * kBlock + threadIdx.x; li < n; li += gridDim.x * kBlock) {
        const float4 F = st.pos_sd[li];
        const float4 Fm = st.misc[li];
        const float4 L = st.pos_sd[n + li];
        const float4 Lm = st.misc[n + li];
        // visibility F_k -> B_m - 0.1 * unit(B_m - F_k) (ARTS.cpp:252-254); visible iff NO hit
        float dx = L.x - F.x, dy = L.y - F.y, dz = L.z - F.z;
        float l2 = dx * dx + dy * dy + dz * dz;
        bool visible = true;
        if (l2 > 1e-8f) {
            float len = sqrtf(l2);
            float inv = 1.0f / len;
            float tmax = len - kp.connect_pullback;
            if (tmax > 0.0f) {
                Ray ray = make_ray(F.x, F.y, F.z, dx * inv, dy * inv, dz * inv);
                HitRec h = traverse<true>(sc, ray, tmax, &s_stack[threadIdx.x]);
                visible = h.leaf_index < 0;
            }
        }
        if (!visible) continue;
        // connected path F0..Fk, Bm..B0 (ARTS.cpp:262-267): connection segment uses F_k's material/prob
        float E[B];
#pragma unroll
        for (int b = 0; b < B; ++b) E[b] = st.energy[(size_t)b * 2u * n + li];
        float dist = sqrtf(l2);
        float nd = dist / kp.dist_divisor;
        float sd = F.w;
        sd += nd;
        apply_segment<B>(E, nd, __float_as_uint(Fm.y), Fm.x, kp, sc);
        // listener-side segments in path order B_m -> ... -> B_0 (reverse of the walk)
        const int segs = __float_as_int(Lm.z);
        for (int j = segs - 1; j >= 0; --j) sd += st.seg_nd[(size_t)j * n + li];
        float delay = sd / kp.sound_speed;                            // ARTS.cpp:419
        float x = (delay * 1000.f) / 1.0f;                            // FSAC.h:89, BinSizeMs = 1
        float fl = floorf(x);
        int bin = !(fl > 0.0f) ? 0 : (fl >= (float)(nb - 1) ? nb - 1 : (int)fl);
        atomicMin(&s_lo, bin);
        atomicMax(&s_hi, bin);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float e = E[b] * st.energy[(size_t)b * 2u * n + n + li];
            e = (e < kp.energy_clamp) ? e : kp.energy_clamp;          // FMath::Min ARTS.cpp:410
            e *= kp.energy_gain;                                      // ARTS.cpp:413
            e *= kp.norm;                                             // ARTS.cpp:164-170
            atomicAdd(&s_hist[b * nb + bin], e);                      // ds_add_f32
        }
    }
    __syncthreads();
    const int lo = s_lo, hi = s_hi;
    if (hi < lo) return;
    const int span = hi - lo + 1;
    for (int i = threadIdx.x; i < B * span; i += kBlock) {
        int b = i / span, bin = lo + (i - b * span);
        float v = s_hist[b * nb + bin];
        if (v != 0.0f) atomicAdd(&energy[b * nb + bin], v);           // global_atomic_add_f32
    }
}

// ---------------------------------------------------------------------------------------------------
// reconstruct_kernel: ReconstructImpulseResponse (FSAC.cpp:320-380)
//   rows 0..B-1 = bands, row B = band-mean energy -> the channel view (channels are identical,
//   FSAC.cpp:331).  The one-pole filter y[i] = 0.25 x[i] + 0.75 y[i-1] (FSAC.cpp:366-375) is evaluated
//   per 32-sample chunk after a 160-sample warm-up: 0.75^160 ~ 1e-20 is far below fp32 resolution.
// ---------------------------------------------------------------------------------------------------
constexpr int kChunk = 32;
constexpr int kWarm = 160;

__global__ __launch_bounds__(kBlock) void reconstruct_kernel(const float* __restrict__ energy, int B, int nb,
                                                             int num_samples, int spb, float* __restrict__ ir_bands,
                                                             float* __restrict__ ir_mono) {
    extern __shared__ __attribute__((aligned(16))) float s_amp[];  // [nb] amplitude per bin of this row
    const int row = blockIdx.y;
    const float Pi4 = sqrtf(4.0f * kPi);                           // FSAC.cpp:323
    for (int i = threadIdx.x; i < nb; i += kBlock) {
        float e;
        if (row < B) e = energy[row * nb + i];
        else {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += energy[b * nb + i];
            e = s / (float)B;
        }
        float a = 0.0f;
        if (fabsf(e) >= 1e-6f) a = e / sqrtf(e * Pi4);             // FSAC.cpp:343-345
        s_amp[i] = a;
    }
    __syncthreads();
    const int chunk = blockIdx.x * kBlock + threadIdx.x;
    const int s0 = chunk * kChunk;
    if (s0 >= num_samples) return;
    float* out = row < B ? ir_bands + (size_t)row * num_samples : ir_mono;
    const float inv_spb = (float)spb;
    auto sample = [&](int i) -> float {
        int bin = i / spb;
        if (bin >= nb) return 0.0f;
        int bs = i - bin * spb;
        float cur = s_amp[bin];
        float prev = bin == 0 ? cur : s_amp[bin - 1];             // FSAC.cpp:347-355
        float w = (float)bs / inv_spb;                            // FSAC.cpp:359
        float a = (1.0f - w) * prev;
        float b = w * cur;
        return a + b;                                             // FSAC.cpp:360
    };
    int w0 = s0 - kWarm;
    float y;
    int i;
    if (w0 <= 0) { y = sample(0); i = 1; if (s0 == 0) out[0] = y; }  // Filtered[0] = IR[0] FSAC.cpp:371
    else { y = 0.0f; i = w0; }
    const int s1 = min(s0 + kChunk, num_samples);
    for (; i < s1; ++i) {
        float a = 0.25f * sample(i);
        float b = (1.0f - 0.25f) * y;
        y = a + b;                                                // FSAC.cpp:374
        if (i >= s0) out[i] = y;
    }
}

// ---------------------------------------------------------------------------------------------------
// trace_rays_kernel: the engine line trace (tests / tools)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(DeviceScene sc, const float* __restrict__ o,
                                                            const float* __restrict__ d,
                                                            const float* __restrict__ tmax, int N, int any_hit,
                                                            int32_t* hit, float* t, int32_t* tri, float* normal) {
    __shared__ int s_stack[kStackDepth * kBlock];
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    Ray r = make_ray(o[3 * i], o[3 * i + 1], o[3 * i + 2], d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if (any_hit) {
        HitRec h = traverse<true>(sc, r, tmax[i], &s_stack[threadIdx.x]);
        hit[i] = h.leaf_index >= 0;
        return;
    }
    HitRec h = traverse<false>(sc, r, tmax[i], &s_stack[threadIdx.x]);
    hit[i] = h.leaf_index >= 0;
    if (h.leaf_index >= 0) {
        const float4* tris4 = reinterpret_cast<const float4*>(sc.tris);
        const float4 A = tris4[3 * h.leaf_index + 0];
        const float4 Bq = tris4[3 * h.leaf_index + 1];
        const float4 Cq = tris4[3 * h.leaf_index + 2];
        float nx, ny, nz;
        hit_normal(A, Bq, Cq, r, nx, ny, nz);
        t[i] = h.t;
        tri[i] = (int32_t)h.id;
        normal[3 * i] = nx; normal[3 * i + 1] = ny; normal[3 * i + 2] = nz;
    } else {
        t[i] = tmax[i];
        tri[i] = -1;
        normal[3 * i] = 0.f; normal[3 * i + 1] = 0.f; normal[3 * i + 2] = 0.f;
    }
}

// AddEnergyAtDelay on the device-resident buffer (FSAC.h:87-91)
__global__ void add_energy_kernel(float* row, int nb, float delay, float e) {
    float x = (delay * 1000.f) / 1.0f;
    float fl = floorf(x);
    int bin = !(fl > 0.0f) ? 0 : (fl >= (float)(nb - 1) ? nb - 1 : (int)fl);
    row[bin] += e;
}

template <int B>
void launch_walk_t(const DeviceScene& sc, const KParams& kp, const SubpathState& st, hipStream_t s) {
    uint32_t lanes = 2u * kp.num_local;
    if (lanes == 0) return;
    dim3 grid((lanes + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(walk_kernel<B>, grid, dim3(kBlock), 0, s, sc, kp, st);
}

template <int B>
void launch_connect_t(const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                      hipStream_t s) {
    if (kp.num_local == 0) return;
    uint32_t blocks = (kp.num_local + kBlock - 1) / kBlock;
    if (blocks > 1024) blocks = 1024;
    size_t lds = sizeof(float) * (size_t)B * (size_t)kp.num_bins;
    hipLaunchKernelGGL(connect_kernel<B>, dim3(blocks), dim3(kBlock), lds, s, sc, kp, st, energy);
}

}  // namespace

#define FS_DISPATCH_B(B, CALL)            \
    switch (B) {                          \
        case 1: CALL(1); break;           \
        case 2: CALL(2); break;           \
        case 3: CALL(3); break;           \
        case 4: CALL(4); break;           \
        case 5: CALL(5); break;           \
        case 6: CALL(6); break;           \
        case 7: CALL(7); break;           \
        default: CALL(8); break;          \
    }

void launch_walk(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, hipStream_t s) {
#define CALL(N) launch_walk_t<N>(sc, kp, st, s)
    FS_DISPATCH_B(B, CALL)
#undef CALL
}

void launch_connect(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                    hipStream_t s) {
#define CALL(N) launch_connect_t<N>(sc, kp, st, energy, s)
    FS_DISPATCH_B(B, CALL)
#undef CALL
}

void launch_reconstruct(const float* energy, int B, int num_bins, int sample_rate, int num_samples, int spb,
                        float* ir_bands, float* ir_mono, hipStream_t s) {
    (void)sample_rate;
    int chunks = (num_samples + kChunk - 1) / kChunk;
    dim3 grid((chunks + kBlock - 1) / kBlock, B + 1);
    hipLaunchKernelGGL(reconstruct_kernel, grid, dim3(kBlock), sizeof(float) * (size_t)num_bins, s, energy, B,
                       num_bins, num_samples, spb, ir_bands, ir_mono);
}

void launch_trace_rays(const DeviceScene& sc, const float* o, const float* d, const float* tmax, int N, int any_hit,
                       int32_t* hit, float* t, int32_t* tri, float* normal, hipStream_t s) {
    if (N <= 0) return;
    hipLaunchKernelGGL(trace_rays_kernel, dim3((N + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sc, o, d, tmax, N,
                       any_hit, hit, t, tri, normal);
}

void launch_add_energy(float* energy_row, int num_bins, float delay_s, float e, hipStream_t s) {
    hipLaunchKernelGGL(add_energy_kernel, dim3(1), dim3(1), 0, s, energy_row, num_bins, delay_s, e);
}

}  // namespace fs
