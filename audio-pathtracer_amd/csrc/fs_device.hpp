// fs_device.hpp — device-side code of the FrequenSee BDPT path shared by the kernel translation units (gfx950, wave64):
// RNG and sampling maps, ray / triangle / box arithmetic, the BVH traversal step and its wave work sharing, the walk
// (GeneratePath, AudioRayTracingSubsystem.cpp:279-355), ConnectSubpaths + EvaluatePath + deposit (:235-277, :360-420),
// the plan pass.  Everything is inline in an anonymous namespace: fs_walk.hip, fs_connect.hip, fs_frame.hip and
// fs_aux_kernels.hip each instantiate the kernels they launch (built side by side; fs_kernels_all.hip is the same code as
// one unit for the diagnostic builds, whose device-side debug symbols must be shared by all kernels).
//
// The triangle test, the hit point/normal/offset arithmetic and the sampling maps use a fixed
// operation order with explicit fmaf and are compiled with -ffp-contract=off: the path geometry is a
// pure function of (scene, seed, pair index) and does not depend on launch geometry or on the BVH.
#pragma once
#include <algorithm>
#include <atomic>

#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr float kPi = 3.1415926535897932f;
constexpr uint32_t kNoMat = FS_NO_MATERIAL;
constexpr uint32_t kLobeDiffuse = 0u, kLobeSpecular = 1u, kLobeTransmit = 2u;
constexpr int kLobeShift = 16;   // segment record: material id | lobe << 16
constexpr int kDone = (int)0x80000000;  // traversal cursor: nothing left
constexpr uint32_t kMissKey = 0xFFFFFFFCu;   // node test: sort key of a child the ray misses (| slot)
constexpr double kFixedScale = 1099511627776.0;   // 2^40: quantum of the deterministic (fixed-point) energy sum
#ifdef FS_WAVE_TIMELINE   // diagnostic build only (tools/wave_timeline.py): when every walk wave ran and what it spent its cycles on
__device__ unsigned long long* g_wave_buf;       // [waves][8]: start, end (100 MHz), cycles in traversal, cycles in all, iterations, segments, hw id, slot
__device__ unsigned long long* g_conn_buf;       // [waves][8]: connect kernel: start, set-up done, visibility done, evaluated, end (100 MHz), lane-0 deposits
#endif
#ifdef FS_TRAV_STATS
__device__ unsigned long long g_trav_stats[32];  // closest-hit queries at [0..15], any-hit at [16..31]: [0] step calls, [1] node iterations,
                                                 // [2] node lanes, [3] tri iterations, [4] tri lanes, [5..7] node visits by children hit,
                                                 // [8] busy lanes, [9] lanes on taken work, [10] sharing-loop iterations, [11] lanes with both kinds
__device__ unsigned short* g_step_buf;           // optional [depth][2P]: traversal iterations of every walk segment
#endif

// ---------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = (pair, bounce<<1|side, block, 'FS01'), key = seed
// ---------------------------------------------------------------------------------------------------
// (legacy tracer: counter = (ray, 0, 0, 'FS02'))
__device__ __forceinline__ uint4 philox(uint32_t pair, uint32_t bs, uint32_t block, uint32_t k0, uint32_t k1,
                                        uint32_t domain = 0x46533031u) {
    uint32_t c0 = pair, c1 = bs, c2 = block, c3 = domain;
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
// (two halves: the sample in the cone's own frame depends on the two uniforms only — the cooperative walk lets idle lanes
// compute it for 64 bounces at a time — the turn into the world on the surface normal; sample_cone = one after the other)
__device__ __forceinline__ void cone_local(float U, float V, int cosine, float& lx, float& ly, float& cphi) {
    float sphi;
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
    lx = sphi * ct; ly = sphi * st;
}
__device__ __forceinline__ void cone_world(float nx, float ny, float nz, float lx, float ly, float cphi, float& dx, float& dy, float& dz) {
    float sg = copysignf(1.0f, nz);
    float a = -1.0f / (sg + nz);
    float b = nx * ny * a;
    float t0 = fmaf(sg * nx * nx, a, 1.0f), t1 = sg * b, t2 = -sg * nx;
    float b0 = b, b1 = fmaf(ny * ny, a, sg), b2 = -ny;
    float d0 = fmaf(lx, t0, fmaf(ly, b0, cphi * nx));
    float d1 = fmaf(lx, t1, fmaf(ly, b1, cphi * ny));
    float d2 = fmaf(lx, t2, fmaf(ly, b2, cphi * nz));
    float l2 = d0 * d0 + d1 * d1 + d2 * d2;
    float inv = 1.0f / sqrtf(l2);
    dx = d0 * inv; dy = d1 * inv; dz = d2 * inv;
}
__device__ __forceinline__ void sample_cone(float nx, float ny, float nz, float U, float V, int cosine, float& dx,
                                            float& dy, float& dz) {
    float lx, ly, cphi;
    cone_local(U, V, cosine, lx, ly, cphi);
    cone_world(nx, ny, nz, lx, ly, cphi, dx, dy, dz);
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
    // barycentric tests on the un-normalised values, sign-normalised by det (exact: sign-bit xor).  All of
    // it is straight-line code behind ONE branch (bitwise &, no short-circuit exits: with a dozen lanes in
    // the test some lane nearly always needs every term, and each early exit costs exec-mask bookkeeping);
    // the one IEEE division is only paid by rays that are inside the triangle.
    const uint32_t sgn = __float_as_uint(det) & 0x80000000u;
    const float ad = fabsf(det);
    float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    float U = fmaf(sx, px, fmaf(sy, py, sz * pz));
    float us = __uint_as_float(__float_as_uint(U) ^ sgn);
    float qx = fmaf(sy, e1z, -(sz * e1y));
    float qy = fmaf(sz, e1x, -(sx * e1z));
    float qz = fmaf(sx, e1y, -(sy * e1x));
    float V = fmaf(r.dx, qx, fmaf(r.dy, qy, r.dz * qz));
    float vs = __uint_as_float(__float_as_uint(V) ^ sgn);
    float tn = fmaf(e2x, qx, fmaf(e2y, qy, e2z * qz));
    const bool inside = (det != 0.0f) & (us >= 0.0f) & (us <= ad) & (vs >= 0.0f) & ((us + vs) <= ad);
    if (!inside) return false;
    float t = tn / det;
    t_out = t;
    return (t > 0.0f) & (t <= tmax);
}


// ---------------------------------------------------------------------------------------------------
// BVH traversal, one ray per lane, as a resumable single loop.  Every call of trav_step a busy lane
// advances on BOTH fronts it has work on: it tests one triangle of its pending leaf AND visits its next
// inner node (4 child boxes).  A wave executes both code paths in most iterations anyway (its lanes are
// in different phases), so letting one lane use both halves the iterations a ray needs — per ray about
// max(node visits, triangle tests) instead of their sum.  The closest hit does not depend on the order
// of the tests, so results are unchanged.  A lane can be parked/resumed between any two steps.
// `stack` is this lane's column of the workgroup's LDS stack (element i at stack[i * kBlock]).
// ---------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

struct Trav {
    int cur;        // next node: >= 0 inner index, < 0 leaf code (~cur = first*4 + count-1), kDone = none
    int sp;         // stack top (entries live in [sb, sp))
    int sb;         // stack bottom: 0 unless entries were given away from the bottom (wave work sharing)
    int tri_i, tri_n;  // pending triangles [tri_i, tri_n) of the current leaf
    float t;        // closest hit so far (init: tmax)
    int leaf_index; // hit triangle (leaf order), -1 = none
    uint32_t id;    // its input index (tie-break key)
    uint32_t nv, nt;   // COUNT instantiations only (fs_set_profiling level 3): node records / triangle records this lane fetched
    uint32_t ni, nl, nd;   // ... and (one lane per wave) node-request instructions, their active lanes, the distinct records among those
};

__device__ __forceinline__ void trav_init(Trav& T, float tmax, bool scene_nonempty) {
    T.cur = scene_nonempty ? 0 : kDone;
    T.sp = 0; T.sb = 0; T.tri_i = 0; T.tri_n = 0;
    T.t = tmax; T.leaf_index = -1; T.id = 0xFFFFFFFFu;
    T.nv = 0u; T.nt = 0u; T.ni = 0u; T.nl = 0u; T.nd = 0u;
}
__device__ __forceinline__ bool trav_busy(const Trav& T) { return T.tri_i < T.tri_n || T.cur != kDone; }

// The step in pieces, ordered so that as little as possible lies between the arrival of a lane's records and the
// request for its next ones (round 3: one extra L1-hit load per step costs the walk as much as 20 more vector
// instructions — every instruction of a wave between `wait` and the next `issue` is on its serial critical path):
//   trav_wait       the records have arrived
//   trav_node_part  4 child boxes of the lane's node against the bound known so far, sort, pushes, next node
//   trav_settle     a pending leaf becomes the triangle cursor and the next node is popped right away
//   trav_issue      request the node and / or triangle record the lane needs NEXT (the triangle into the other register set)
//   trav_tri_part   test the triangle that arrived with this step — in the shadow of the fetch just issued
// The node test uses the bound from before this step's triangle test: a looser bound only admits more candidates,
// and the (t, id) key decides among them, so the closest hit is unchanged bit for bit.
// The loads are inline asm under the lanes' own exec mask, waited for once behind both groups: a lane without a
// pending triangle (or node) requests nothing, both records of a lane are in flight together.  (Round 1 let every lane
// fetch a dummy record 0 with plain loads instead, because inside `if (has_node)` / `if (has_tri)` blocks the compiler
// sinks the first arithmetic on the loaded words into the block of the loads, i.e. waits for one record before it
// requests the other: 448 lane-loads per wave iteration for 207 useful ones.)
typedef float v4f __attribute__((ext_vector_type(4)));
struct NodeRegs {
    v4f q0, q1, q2, q3;
#if defined(FS_SENS_XLOADS)   // sensitivity builds only (tools/build_variant.sh): extra L1-hit loads per node visit
    v4f x0, x1;
#endif
};
struct TriRegs { v4f a, b, c; };
// Node stride in bytes: 64 in the product.  FS_NODE_STRIDE=128 is a sensitivity build (host commit path only) in which
// every node owns a whole 128-B cache line.
#ifndef FS_NODE_STRIDE
#define FS_NODE_STRIDE 64
#endif

// ---- bounded LDS stack with a deep store in HBM (DeviceScene.deep, fs_internal.hpp) ---------------------------
// The logical stack of a lane is  deep[0, count)  followed by  LDS rows [sb, sp).  Entries move between the two in
// chunks of kDeepChunk, oldest first out, newest first back, so pops keep their order.  All of it happens in ONE place,
// trav_maintain, called at the top of a step for the lanes that need it; the pops and pushes of the step are the plain
// LDS ones.  A lane with entries in the deep store carries T.sb = kDeepSb (-2; its real bottom is row 0 and it gives
// nothing away to idle lanes meanwhile) and is kept at >= 2 LDS entries at the top of every step — a step pops at most
// twice — so its pop test `sp > sb` never fails while the deep store still holds something.  One unsigned compare finds
// both kinds of lane:  (unsigned)(sp + sb) >= limit - 4  (DeviceScene.stack_attn) is true for a lane close to its last rows (sb >= 0; two rows
// early, or earlier for a lane that has given entries away — trav_maintain looks again) and for a flagged lane with
// sp < 2 (sp - 2 wraps) or close to its last rows.  (Tests in the pops themselves cost 2 % of the walk, a deep count
// read from LDS on every empty pop 7 %: profiles/r03_occupancy_ab.log.)
constexpr int kDeepSb = -2;
__device__ __forceinline__ int* trav_deep_count(const DeviceScene& sc, int* stack) { return stack + (size_t)sc.stack_limit * kBlock; }
__device__ __forceinline__ void trav_deep_reset(const DeviceScene& sc, int* stack) {
    if (sc.deep != nullptr) *trav_deep_count(sc, stack) = 0;
}
__device__ __forceinline__ bool trav_needs_maintenance(const DeviceScene& sc, const Trav& T) {
    return (unsigned)(T.sp + T.sb) >= sc.stack_attn;   // stack_limit - 4 with a deep store, else never
}
// (rolled loops: these paths are as good as never taken, their code should stay small inside the traversal loop)
__device__ __forceinline__ void trav_maintain(const DeviceScene& sc, Trav& T, int* stack) {
    if (sc.deep == nullptr) return;   // (without a deep store stack_limit covers the tree's worst case)
    int* cnt = trav_deep_count(sc, stack);
    int32_t* col = sc.deep + (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (T.sb < 0) {
        if (T.sp < 2) {   // the newest chunk comes back, below the entry that may be left
            const int have = *cnt;
            if (T.sp == 1) stack[kDeepChunk * kBlock] = stack[0];
#pragma unroll 1
            for (int i = 0; i < kDeepChunk; ++i) stack[i * kBlock] = col[(size_t)(have - kDeepChunk + i) * sc.deep_lanes];
            T.sp += kDeepChunk;
            *cnt = have - kDeepChunk;
            if (have == kDeepChunk) T.sb = 0;
            return;
        }
    } else if (T.sb > 0) {   // close the gap left by donated entries
        const int n = T.sp - T.sb;
#pragma unroll 1
        for (int i = 0; i < n; ++i) stack[i * kBlock] = stack[(T.sb + i) * kBlock];
        T.sb = 0; T.sp = n;
    }
    if (T.sp + 2 >= sc.stack_limit) {   // a node visit writes up to row sp + 2: the oldest chunk goes out
        const int have = *cnt;
#pragma unroll 1
        for (int i = 0; i < kDeepChunk; ++i) col[(size_t)(have + i) * sc.deep_lanes] = stack[i * kBlock];
#pragma unroll 1
        for (int i = kDeepChunk; i < T.sp; ++i) stack[(i - kDeepChunk) * kBlock] = stack[i * kBlock];
        T.sp -= kDeepChunk;
        T.sb = kDeepSb;
        *cnt = have + kDeepChunk;
    }
}
// next pending entry into T.cur (kDone: none left)
__device__ __forceinline__ void trav_pop(const DeviceScene& sc, Trav& T, int* stack) {
    if (T.sp > T.sb) { --T.sp; T.cur = stack[T.sp * kBlock]; }
    else T.cur = kDone;
}

__device__ __forceinline__ void trav_settle(const DeviceScene& sc, Trav& T, int* stack) {
    if (T.tri_i >= T.tri_n && T.cur < 0 && T.cur != kDone) {
        const int code = ~T.cur;
        T.tri_i = code >> 2;
        T.tri_n = T.tri_i + (code & 3) + 1;
        trav_pop(sc, T, stack);
    }
}

// The request.  ONE asm statement, executed by every lane that reaches it, with every destination register tied in and
// out ("+v"): the lanes that want a record are selected by writing their ballot to EXEC inside the statement.  Both
// matter.  (1) The compiler does not know that the destinations are still being written until the next trav_wait; with
// conditionally executed "=v" outputs the old and the new value meet in a phi, and register allocation is free to
// resolve that phi with copies placed right behind the request — reading registers whose data has not arrived (round 3
// lost a day's first build to exactly that; tools/check_isa_hazards.py now proves the absence of such accesses on the
// final ISA).  A tied operand chain issue -> wait -> use has no phi to resolve.  (2) Every wave executes the same
// number of vector memory instructions per step whatever its lanes need, so counted waits stay possible.
// cache-policy bits of the step's triangle / node requests (-DFS_TRI_NT=1: nt, 2: sc0, 3: sc1, 4: sc0 sc1; FS_NODE_NT alike):
// measured, none kept
#if FS_TRI_NT == 1
#define FS_TRI_LOAD_POLICY " nt"
#elif FS_TRI_NT == 2
#define FS_TRI_LOAD_POLICY " sc0"
#elif FS_TRI_NT == 3
#define FS_TRI_LOAD_POLICY " sc1"
#elif FS_TRI_NT == 4
#define FS_TRI_LOAD_POLICY " sc0 sc1"
#else
#define FS_TRI_LOAD_POLICY ""
#endif
#if FS_NODE_NT == 1
#define FS_NODE_LOAD_POLICY " nt"
#elif FS_NODE_NT == 2
#define FS_NODE_LOAD_POLICY " sc0"
#elif FS_NODE_NT == 3
#define FS_NODE_LOAD_POLICY " sc1"
#else
#define FS_NODE_LOAD_POLICY ""
#endif
__device__ __forceinline__ void trav_issue(const DeviceScene& sc, const Trav& T, NodeRegs& N, TriRegs& X) {
    const unsigned long long mn = __ballot(T.cur >= 0), mt = __ballot(T.tri_i < T.tri_n);   // subsets of EXEC
    // (addresses of lanes that want nothing are never dereferenced)
    const char* np = reinterpret_cast<const char*>(sc.nodes) + (size_t)(uint32_t)T.cur * FS_NODE_STRIDE;
    const Tri48* tp = sc.tris + (uint32_t)T.tri_i;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[mn]\n\t"
                 "global_load_dwordx4 %[q0], %[np], off" FS_NODE_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %[q1], %[np], off offset:16" FS_NODE_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %[q2], %[np], off offset:32" FS_NODE_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %[q3], %[np], off offset:48" FS_NODE_LOAD_POLICY "\n\t"
#if defined(FS_SENS_XLOADS)
                 "global_load_dwordx4 %[x0], %[np], off\n\t"
#if FS_SENS_XLOADS >= 2
                 "global_load_dwordx4 %[x1], %[np], off offset:32\n\t"
#endif
#endif
                 "s_mov_b64 exec, %[mt]\n\t"
#ifndef FS_NO_TRI_SKIP
                 "s_cbranch_execz 1f\n\t"      // one wave step in four has no lane with a pending triangle
#endif
                 "global_load_dwordx4 %[ta], %[tp], off" FS_TRI_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %[tb], %[tp], off offset:16" FS_TRI_LOAD_POLICY "\n\t"
                 "global_load_dwordx4 %[tc], %[tp], off offset:32" FS_TRI_LOAD_POLICY "\n"
                 "1:\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [q0] "+&v"(N.q0), [q1] "+&v"(N.q1), [q2] "+&v"(N.q2), [q3] "+&v"(N.q3),
#if defined(FS_SENS_XLOADS)
                   [x0] "+&v"(N.x0), [x1] "+&v"(N.x1),
#endif
                   [ta] "+&v"(X.a), [tb] "+&v"(X.b), [tc] "+&v"(X.c), [sv] "=&s"(sv)
                 : [np] "v"(np), [tp] "v"(tp), [mn] "s"(mn), [mt] "s"(mt)
                 : "memory");
}

__device__ __forceinline__ void trav_wait(NodeRegs& N, TriRegs& X) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(N.q0), "+v"(N.q1), "+v"(N.q2), "+v"(N.q3), "+v"(X.a), "+v"(X.b), "+v"(X.c));
#if defined(FS_SENS_XLOADS)
    asm volatile("" : "+v"(N.x0), "+v"(N.x1));
#endif
}

// the triangle that arrived with this step (leaf-order index `tested`): Moeller-Trumbore, closest-hit update as selects
template <bool ANY, bool IGN = false>
__device__ __forceinline__ void trav_tri_part(const Ray& r, Trav& T, const TriRegs& X, const int tested,
                                              uint32_t ignore_object = 0xFFFFFFFFu) {
    const float4 a = make_float4(X.a.x, X.a.y, X.a.z, X.a.w), b = make_float4(X.b.x, X.b.y, X.b.z, X.b.w),
                 c = make_float4(X.c.x, X.c.y, X.c.z, X.c.w);   // triangle: v0 | e1 | e2 (+ material, id, object)
    float t = 0.0f;
    // IGN: FCollisionQueryParams::AddIgnoredActor — triangles of one actor (object id in c.w) are skipped
    bool hit = tri_hit(a, b, c, r, T.t, t);
    if (IGN) hit = hit & (__float_as_uint(c.w) != ignore_object);
    const uint32_t id = __float_as_uint(c.z);
    if (ANY) {
        if (hit) {
            T.t = t; T.leaf_index = tested; T.id = id;
            T.tri_i = 0; T.tri_n = 0; T.cur = kDone; T.sp = 0; T.sb = 0;  // first hit ends the query (records already requested are ignored)
        }
    } else {
        // closest hit, ties to the lower input index — as selects, not branches.  (t, id) compares as ONE 64-bit
        // key: t > 0, so its bits order like an integer, and a traversal without a hit yet carries id = ~0
        // (equivalent to t < T.t | no hit yet | (t == T.t & id < T.id); one v_cmp_lt_u64 instead of five compares)
        const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | id;
        const unsigned long long cur = ((unsigned long long)__float_as_uint(T.t) << 32) | T.id;
        const bool better = hit & (key < cur);
        T.t = better ? t : T.t;
        T.leaf_index = better ? tested : T.leaf_index;
        T.id = better ? id : T.id;
    }
}

// the lane's inner node (T.cur >= 0): 4 child boxes, near-first order, far children to the stack, next node
__device__ __forceinline__ void trav_node_part(const DeviceScene& sc, const Ray& r, Trav& T, int* stack, const NodeRegs& N) {
    const float4 q0 = make_float4(N.q0.x, N.q0.y, N.q0.z, N.q0.w), q1 = make_float4(N.q1.x, N.q1.y, N.q1.z, N.q1.w),
                 q2 = make_float4(N.q2.x, N.q2.y, N.q2.z, N.q2.w), q3 = make_float4(N.q3.x, N.q3.y, N.q3.z, N.q3.w);
    // ---- 4-wide node, child boxes on the node's 8-bit grid: plane distance = fma(q, step*inv, (origin-o)*inv)
    const float sx = q0.w * r.ix, sy = q2.z * r.iy, sz = q2.w * r.iz;   // grid step (a power of two) / direction
    const float bx = fmaf(q0.x, r.ix, r.nox);
    const float by = fmaf(q0.y, r.iy, r.noy);
    const float bz = fmaf(q0.z, r.iz, r.noz);
    const uint32_t lox = __float_as_uint(q1.x), loy = __float_as_uint(q1.y), loz = __float_as_uint(q1.z);
    const uint32_t hix = __float_as_uint(q1.w), hiy = __float_as_uint(q2.x), hiz = __float_as_uint(q2.y);
    // the ray's direction signs pick the entry / exit plane words once per node
    const uint32_t nxw = r.ix < 0.0f ? hix : lox, fxw = r.ix < 0.0f ? lox : hix;
    const uint32_t nyw = r.iy < 0.0f ? hiy : loy, fyw = r.iy < 0.0f ? loy : hiy;
    const uint32_t nzw = r.iz < 0.0f ? hiz : loz, fzw = r.iz < 0.0f ? loz : hiz;
    const v2f sx2 = {sx, sx}, sy2 = {sy, sy}, sz2 = {sz, sz}, bx2 = {bx, bx}, by2 = {by, by}, bz2 = {bz, bz};
    uint32_t key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // (entry, exit) plane distances per axis as one packed fma each (v_pk_fma_f32)
        const v2f qx = {(float)((nxw >> (8 * c)) & 0xFFu), (float)((fxw >> (8 * c)) & 0xFFu)};
        const v2f qy = {(float)((nyw >> (8 * c)) & 0xFFu), (float)((fyw >> (8 * c)) & 0xFFu)};
        const v2f qz = {(float)((nzw >> (8 * c)) & 0xFFu), (float)((fzw >> (8 * c)) & 0xFFu)};
        const v2f tx = __builtin_elementwise_fma(qx, sx2, bx2);
        const v2f ty = __builtin_elementwise_fma(qy, sy2, by2);
        const v2f tz = __builtin_elementwise_fma(qz, sz2, bz2);
        const float tnx = tx.x, tfx = tx.y, tny = ty.x, tfy = ty.y, tnz = tz.x, tfz = tz.y;
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, T.t));
        const bool h = tn <= tf;
        // entry distance (>= 0, so its bits order like an integer) with the slot in the low 2 bits; a missed child
        // sorts behind every hit one (kMissKey | slot)
        key[c] = h ? ((__float_as_uint(tn) & ~3u) | (uint32_t)c) : (kMissKey | (uint32_t)c);
    }
    int ref0 = __float_as_int(q3.x), ref1 = __float_as_int(q3.y), ref2 = __float_as_int(q3.z),
        ref3 = __float_as_int(q3.w);
    // sort the 4 (key, child reference) pairs, nearest first: 5-comparator network, branch-free.  (Measured in
    // round 2: choosing only the nearest child and pushing the rest in slot order — also no ordering at all for
    // visibility rays — saves a dozen instructions per visit and costs as much in extra visits: walk 0.356 ->
    // 0.361 ms, connect 0.075 -> 0.078 ms.)
#define FS_CSWAP(a, b) { const bool sw_ = key[b] < key[a]; const uint32_t lo_ = min(key[a], key[b]); \
                         const uint32_t hi_ = max(key[a], key[b]); key[a] = lo_; key[b] = hi_; \
                         const int ra_ = sw_ ? ref##b : ref##a; const int rb_ = sw_ ? ref##a : ref##b; \
                         ref##a = ra_; ref##b = rb_; }
    FS_CSWAP(0, 1) FS_CSWAP(2, 3) FS_CSWAP(0, 2) FS_CSWAP(1, 3) FS_CSWAP(1, 2)
#undef FS_CSWAP
    // the number of children hit, read off the sorted keys: at least k + 1 <=> key[k] is a hit
    const bool h1 = key[0] < kMissKey, h2 = key[1] < kMissKey, h3 = key[2] < kMissKey, h4 = key[3] < kMissKey;
#ifdef FS_TRAV_STATS
    atomicAdd(&g_trav_stats[5 + (h2 ? 2 : (h1 ? 1 : 0))], 1ull);   // [5] visits with no child hit, [6] one, [7] two or more
#endif
    // far children wait on the stack, farthest pushed first, at sp .. sp + hits - 2: with two hits all three stores
    // land on sp and the last one (the second nearest) stays, with three hits the first two share sp — no store
    // goes above the new top, so the stack needs exactly the tree's worst-case number of rows.  With fewer than two
    // hits the three stores write (unused) words to the free row above the top: cheaper than branching around them,
    // a wave nearly always has a lane that pushes.
    const int p3 = T.sp;
    const int p2 = p3 + (h4 ? 1 : 0);
    const int p1 = p2 + (h3 ? 1 : 0);
    stack[p3 * kBlock] = ref3;
    stack[p2 * kBlock] = ref2;
    stack[p1 * kBlock] = ref1;
    T.sp = p1 + (h2 ? 1 : 0);
    if (h1) T.cur = ref0;
    else trav_pop(sc, T, stack);
}

// One whole step of a busy lane, in the pipelined order: node part, advance, request the next records (the
// triangle into `nxt`), then the triangle part on `cur` while they are in flight.  Entry: the records of (T.cur,
// T.tri_i) have arrived in (N, cur).
template <bool ANY, bool IGN = false, bool COUNT = false>
__device__ __forceinline__ void trav_advance(const DeviceScene& sc, const Ray& r, Trav& T, int* stack, NodeRegs& N,
                                             TriRegs& cur, TriRegs& nxt, uint32_t ignore_object = 0xFFFFFFFFu) {
    const bool has_tri = T.tri_i < T.tri_n;
    const bool has_node = T.cur >= 0;
    const int tested = T.tri_i;
    if (COUNT) { T.nv += has_node ? 1u : 0u; T.nt += has_tri ? 1u : 0u; }
#ifdef FS_TRAV_STATS   // diagnostic build only (tests/trav_stats.py): SIMD occupancy of the two step kinds
    {
        const unsigned long long mt = __ballot(has_tri), mn = __ballot(has_node);
        unsigned long long* gs = g_trav_stats + (ANY ? 16 : 0);
        if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(true)) - 1)) {
            atomicAdd(&gs[0], 1ull);
            if (mn) { atomicAdd(&gs[1], 1ull); atomicAdd(&gs[2], (unsigned long long)__popcll(mn)); }
            if (mt) { atomicAdd(&gs[3], 1ull); atomicAdd(&gs[4], (unsigned long long)__popcll(mt)); }
            atomicAdd(&gs[11], (unsigned long long)__popcll(mn & mt));
        }
    }
#endif
    // bounded LDS stack: a node visit writes up to row sp + 2.  Checked here, ahead of the node arithmetic and as one
    // scalar branch for the wave, so that the node part stays a single basic block; lanes of trees without a deep store
    // (stack_limit = worst case + 1) may pass the test near their worst case and return at once.
#ifndef FS_DEEP_NO_CHECK   // compiled out in the wide flavour of the frame kernel (worst-case rows, fs_frame.hip)
    if (__builtin_expect(__ballot(trav_needs_maintenance(sc, T)) != 0ull, 0)) {
        if (trav_needs_maintenance(sc, T)) trav_maintain(sc, T, stack);
    }
#endif
    if (has_node) trav_node_part(sc, r, T, stack, N);
    if (has_tri) ++T.tri_i;
    trav_settle(sc, T, stack);
    if (COUNT) {   // counting instantiation only: how coherent is this wave's node request?  lanes that take part, distinct 64-B records among them
        const unsigned long long mn = __ballot(T.cur >= 0);
        if (mn != 0ull) {
            unsigned distinct = 0;
            for (unsigned long long rest = mn; rest != 0ull; rest &= rest - 1ull) {
                const int l = __ffsll((long long)rest) - 1;
                const int v = __builtin_amdgcn_readlane(T.cur, l);
                const unsigned long long same = __ballot(T.cur == v) & mn;
                distinct += (__ffsll((long long)same) - 1) == l ? 1u : 0u;
            }
            if ((threadIdx.x & 63u) == (unsigned)(__ffsll((long long)__ballot(true)) - 1)) {
                T.ni += 1u; T.nl += (uint32_t)__popcll(mn); T.nd += distinct;
            }
        }
    }
    trav_issue(sc, T, N, nxt);
    // the triangle test must stay BEHIND the requests: it is plain arithmetic on registers, which the compiler would
    // otherwise move in front of the (to it unrelated) load instructions — and then fold the two register sets into one
    asm volatile("" : "+v"(cur.a), "+v"(cur.b), "+v"(cur.c));
    if (has_tri) trav_tri_part<ANY, IGN>(r, T, cur, tested, ignore_object);
}

// Before a traversal returns, every record it has requested must have landed: the compiler knows nothing of loads in
// flight and would hand their destination registers to other values (an any-hit query ends with requests outstanding).
__device__ __forceinline__ void trav_drain(NodeRegs& N, TriRegs& X, TriRegs& Y) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(N.q0), "+v"(N.q1), "+v"(N.q2), "+v"(N.q3), "+v"(X.a), "+v"(X.b), "+v"(X.c),
                                        "+v"(Y.a), "+v"(Y.b), "+v"(Y.c));
#if defined(FS_SENS_XLOADS)
    asm volatile("" : "+v"(N.x0), "+v"(N.x1));
#endif
}

// one ray per lane without work sharing (tests, tools, diagnostic builds); returns the number of steps taken
template <bool ANY, bool IGN = false>
__device__ __forceinline__ int trav_run(const DeviceScene& sc, const Ray& r, Trav& T, int* stack,
                                        uint32_t ignore_object = 0xFFFFFFFFu) {
    NodeRegs N;
    TriRegs X, Y;
    int steps = 0;
    trav_settle(sc, T, stack);
    trav_issue(sc, T, N, X);
    // The loop is wave-uniform (lanes whose ray is finished idle along): a per-lane exit would make the compiler carry
    // every lane's register sets out of the loop through copies — of registers that may still be in flight.
    while (true) {
        if (__ballot(trav_busy(T)) == 0ull) break;
        trav_wait(N, X);
        if (trav_busy(T)) { trav_advance<ANY, IGN>(sc, r, T, stack, N, X, Y, ignore_object); ++steps; }
        if (__ballot(trav_busy(T)) == 0ull) break;
        trav_wait(N, Y);
        if (trav_busy(T)) { trav_advance<ANY, IGN>(sc, r, T, stack, N, Y, X, ignore_object); ++steps; }
    }
    trav_drain(N, X, Y);
    return steps;
}

// ImpactNormal: the record's unit geometric normal, flipped to face the ray origin side; material of the hit
__device__ __forceinline__ void hit_surface(const DeviceScene& sc, int leaf_index, const Ray& r, float& nx, float& ny,
                                            float& nz, uint32_t& mat) {
    const float4 c = sc.tris[leaf_index].c;
    const float4 d = sc.tri_nrm[leaf_index];
    float x = d.x, y = d.y, z = d.z;
    float dn = fmaf(x, r.dx, fmaf(y, r.dy, z * r.dz));
    if (dn > 0.0f) { x = -x; y = -y; z = -z; }
    nx = x; ny = y; nz = z;
    mat = __float_as_uint(c.y);
}

// one EvaluatePath segment term on E[b] (ARTS.cpp:381-398), in the reference's operation order
// LOBES: 0 / 1 = FS_FLAG_MATERIAL_LOBES known at compile time (the default connect kernel), -1 = read kp.lobes
// Band count of the connect kernels: a template constant for the counts in use (1, 4, 8: fully unrolled loops, the
// band energies stay in registers) or B = 0: kp.num_bands at run time (every other count: the same arithmetic, unrolled
// to FS_MAX_BANDS under a predicate).
template <int B> struct Bands { static constexpr int kMax = B ? B : FS_MAX_BANDS; };
template <int B> __device__ __forceinline__ int band_count(const KParams& kp) { return B ? B : kp.num_bands; }

template <int B, int LOBES = -1>
__device__ __forceinline__ void apply_segment(float (&E)[Bands<B>::kMax], float nd, uint32_t mat, float prob, const KParams& kp,
                                              const DeviceScene& sc) {
    const int NB = band_count<B>(kp);
    const bool lobes = LOBES < 0 ? kp.lobes != 0 : LOBES != 0;
    if (nd < kp.min_seg) return;  // ARTS.cpp:375-378
    float nd2 = nd * nd;
    float geo = 1.0f / (4 * kPi * nd2);            // ARTS.cpp:391
    float pw = powf(prob, kp.prob_exponent);       // ARTS.cpp:398
    // FS_FLAG_MATERIAL_LOBES (row f4): bits 16-17 of the record = lobe the walk took at this vertex (0 = diffuse,
    // also at a connection vertex); its gain replaces Absorption (the diffuse one still over pi)
    const uint32_t lobe = (lobes && mat != kNoMat) ? ((mat >> kLobeShift) & 3u) : 0u;
    if (lobes && mat != kNoMat) mat &= 0xFFFFu;
    bool has = (mat != kNoMat) && ((int32_t)mat < sc.num_materials);
    const float* coeff = lobes ? sc.lobe_gain + ((size_t)mat * 3 + lobe) * NB : sc.absorption + (size_t)mat * NB;
    const bool over_pi = !lobes || lobe == kLobeDiffuse;
#pragma unroll
    for (int b = 0; b < Bands<B>::kMax; ++b) {
        if (B == 0 && b >= NB) break;
        float bsdf = 1.0f;                                            // ARTS.cpp:382-386
        if (has) bsdf = over_pi ? coeff[b] / kPi : coeff[b];
        float e = E[b];
        e *= bsdf;
        e *= geo;
        e *= expf(-kp.air[b] * nd);                // ARTS.cpp:395-397
        e /= pw;
        E[b] = e;
    }
}

// The factors apply_segment multiplies in, computed apart from the running product: one lane per SEGMENT of a connected path
// evaluates them (the pow, the exponentials, the divisions), and the product over the path — strictly in the reference's
// order — is then four multiplications and a division per segment and band (connect_body, one pair per wave: a path of
// 160 segments took 40 us of one lane's time).  The same expressions as apply_segment, so the same bits.
template <int B>
struct SegFactors { float geo, pw; bool live; float bsdf[Bands<B>::kMax], ex[Bands<B>::kMax]; };
template <int B, int LOBES = -1>
__device__ __forceinline__ void segment_factors(SegFactors<B>& f, float nd, uint32_t mat, float prob, const KParams& kp, const DeviceScene& sc) {
    const int NB = band_count<B>(kp);
    const bool lobes = LOBES < 0 ? kp.lobes != 0 : LOBES != 0;
    f.live = !(nd < kp.min_seg);                   // ARTS.cpp:375-378
    float nd2 = nd * nd;
    f.geo = 1.0f / (4 * kPi * nd2);                // ARTS.cpp:391
    f.pw = powf(prob, kp.prob_exponent);           // ARTS.cpp:398
    const uint32_t lobe = (lobes && mat != kNoMat) ? ((mat >> kLobeShift) & 3u) : 0u;
    if (lobes && mat != kNoMat) mat &= 0xFFFFu;
    bool has = (mat != kNoMat) && ((int32_t)mat < sc.num_materials);
    const float* coeff = lobes ? sc.lobe_gain + ((size_t)mat * 3 + lobe) * NB : sc.absorption + (size_t)mat * NB;
    const bool over_pi = !lobes || lobe == kLobeDiffuse;
#pragma unroll
    for (int b = 0; b < Bands<B>::kMax; ++b) {
        f.bsdf[b] = 1.0f; f.ex[b] = 1.0f;
        if (B == 0 && b >= NB) continue;
        if (has) f.bsdf[b] = over_pi ? coeff[b] / kPi : coeff[b];      // ARTS.cpp:382-386
        f.ex[b] = expf(-kp.air[b] * nd);           // ARTS.cpp:395-397
    }
}

// ---------------------------------------------------------------------------------------------------
// the walk, shared by both kernel variants
// ---------------------------------------------------------------------------------------------------
struct Walker {          // ARTS.cpp:287-291 state + bookkeeping
    uint32_t g, slot, side, li, pair;   // subpath index, launch slot (where its records go), side, pair of the frame, RNG pair
    int k;
    float px, py, pz, nx, ny, nz;
    double dpx, dpy, dpz;   // FS_FLAG_DOUBLE_POSITIONS: the node position as the reference's FVector holds it (px.. = its float rounding)
    bool has_normal;
    bool arrived;      // the current vertex was reached by a hit (lobes are picked only then)
    uint32_t mat;
    uint32_t lobe;     // lobe picked at the current vertex << kLobeShift (FS_FLAG_MATERIAL_LOBES), else 0
    float prob, prob_new;
    uint32_t ign;      // the actor this walk ignores (EXT instantiations; FS_NO_OBJECT: none)
};

// low seed word of item `sid` of a batched frame (grouped frames carry one seed per item; kp.item_seeds <= 4)
__device__ __forceinline__ uint32_t item_seed_lo(const KParams& kp, uint32_t sid) {
    uint32_t s = kp.seed_lo;
    if (kp.item_seeds > 0) {   // (wave-uniform)
        s = kp.item_seed[0];
        s = sid == 1u ? kp.item_seed[1] : s;
        s = sid == 2u ? kp.item_seed[2] : s;
        s = sid == 3u ? kp.item_seed[3] : s;
    }
    return s;
}

// ---- segment records: [step][slot] in the main tier, (step - main_levels, slot) in the overflow tier ------------
__device__ __forceinline__ bool rec_in_main(const SubpathState& st, int k) { return k < st.main_levels; }
__device__ __forceinline__ size_t rec_main(uint32_t total, int k, uint32_t slot) { return (size_t)k * total + slot; }
__device__ __forceinline__ size_t rec_over(const SubpathState& st, int k, uint32_t slot) {
    return (size_t)(k - st.main_levels) * st.over_cap + slot;
}
// does step k of the walk in `slot` have a place?  (always, for a capped depth)
__device__ __forceinline__ bool rec_fits(const SubpathState& st, int k, uint32_t slot) {
    return k < st.main_levels || (slot < st.over_cap && k - st.main_levels < st.over_levels);
}
__device__ __forceinline__ float2 load_np(const SubpathState& st, uint32_t total, int k, uint32_t slot) {
    return rec_in_main(st, k) ? st.seg_np[rec_main(total, k, slot)] : st.over_np[rec_over(st, k, slot)];
}
__device__ __forceinline__ uint32_t load_mat(const SubpathState& st, uint32_t total, int k, uint32_t slot) {
    return rec_in_main(st, k) ? st.seg_mat[rec_main(total, k, slot)] : st.over_mat[rec_over(st, k, slot)];
}
__device__ __forceinline__ float4 load_pos(const SubpathState& st, uint32_t total, int k, uint32_t slot) {
    return rec_in_main(st, k) ? st.seg_pos[rec_main(total, k, slot)] : st.over_pos[rec_over(st, k, slot)];
}
__device__ __forceinline__ float4 load_nrm(const SubpathState& st, uint32_t total, int k, uint32_t slot) {
    return rec_in_main(st, k) ? st.seg_nrm[rec_main(total, k, slot)] : st.over_nrm[rec_over(st, k, slot)];
}
__device__ __forceinline__ void store_mat(const SubpathState& st, uint32_t total, int k, uint32_t slot, uint32_t v) {
    if (rec_in_main(st, k)) st.seg_mat[rec_main(total, k, slot)] = v;
    else if (rec_fits(st, k, slot)) st.over_mat[rec_over(st, k, slot)] = v;
}
// slot of subpath g
__device__ __forceinline__ uint32_t slot_of(const SubpathState& st, uint32_t g) { return st.slot_of ? st.slot_of[g] : g; }

__device__ __forceinline__ void walker_start(Walker& w, uint32_t g, uint32_t slot, const KParams& kp, const SubpathState& st,
                                             bool own = true) {   // own = false: a helper lane without a subpath
    const uint32_t n = kp.num_local;
    w.g = g;
    w.slot = slot;
    if (own && st.slot_of) st.slot_of[g] = slot;   // the connect kernels find the walk's records through this
    w.side = g >= n ? 1u : 0u;
    w.li = g - w.side * n;
    w.pair = kp.pair_begin + w.li;
    w.px = w.side ? kp.lis[0] : kp.src[0];
    w.py = w.side ? kp.lis[1] : kp.src[1];
    w.pz = w.side ? kp.lis[2] : kp.src[2];
    if (kp.src_table) {   // batched frame (wave-uniform): several sources' pairs end to end, same RNG pairs for each
        const uint32_t sid = w.li / kp.pairs_per_source;
        w.pair = kp.pair_begin + (w.li - sid * kp.pairs_per_source);
        if (!w.side) { w.px = kp.src_table[4 * sid]; w.py = kp.src_table[4 * sid + 1]; w.pz = kp.src_table[4 * sid + 2]; }
    }
    w.ign = w.side ? kp.lis_object : kp.src_object;   // AddIgnoredActor ARTS.cpp:322-327 (used by the EXT instantiations only)
    if (kp.src_table && !w.side) w.ign = __float_as_uint(kp.src_table[4 * (w.li / kp.pairs_per_source) + 3]);
    w.dpx = (double)w.px; w.dpy = (double)w.py; w.dpz = (double)w.pz;
    w.nx = 0.f; w.ny = 0.f; w.nz = 0.f;
    w.has_normal = false;
    w.arrived = false;
    w.lobe = 0u;
    w.mat = kNoMat;
    w.prob = 1.0f; w.prob_new = 1.0f;
    w.k = 0;
}

// top of GeneratePath's loop (ARTS.cpp:294-319): depth cap, roulette, direction.  false = the walk ends.
// `ray` still holds the previous segment's ray on entry: its direction is the arrival direction at this vertex.
// LOBES: 0 / 1 = FS_FLAG_MATERIAL_LOBES known at compile time, -1 = read kp.lobes.
// pre: the Philox words of this bounce, computed ahead by another lane (cooperative walk: the roulette and the sample of a
// bounce depend on (seed, pair, side, bounce) only) — the same words, so the same walk.  pre_cone: words y, z, w are already the
// diffuse sample in the cone's own frame (cone_local) — only for a vertex with a normal, without lobes
template <int LOBES = -1>
__device__ __forceinline__ bool walker_next_ray(Walker& w, const KParams& kp, const DeviceScene& sc,
                                                const SubpathState& st, Ray& ray, const uint4* pre = nullptr, const bool pre_cone = false) {
    const bool lobes_on = LOBES < 0 ? kp.lobes != 0 : LOBES != 0;
    if (w.k >= kp.depth && st.over_levels == 0) return false;             // the depth cap
    const uint32_t bs = ((uint32_t)w.k << 1) | w.side;
    // (grouped frames: the item's own low seed word — recomputed from the pair index here, once per bounce, rather than
    // carried in a register through the traversal: one more live VGPR cost the 128-register frame kernel 3 %)
    const uint32_t seed = kp.item_seeds > 0 ? item_seed_lo(kp, w.li / kp.pairs_per_source) : kp.seed_lo;
    const uint4 r = pre ? *pre : philox(w.pair, bs, 0, seed, kp.seed_hi);
    if (kp.russian_roulette && !(u01(r.x) < kp.rr_prob)) return false;    // ARTS.cpp:300-301, 349-353
    if (w.k >= kp.depth) { *st.overflow = 1u; return false; }             // depth = 0 and the walk outlives both tiers
    float dx, dy, dz;
    if (!w.has_normal) {                                                  // ARTS.cpp:306-310
        sample_sphere(w.pair, bs, r, seed, kp.seed_hi, dx, dy, dz);
        float pdf = 1.0f / (4.0f * kPi);
        w.prob_new = pdf * kp.rr_prob;
    } else {                                                              // ARTS.cpp:311-318
        // FS_FLAG_MATERIAL_LOBES (row f4, build-owned): one lobe per vertex, picked with the Philox word the diffuse
        // walk leaves unused, probabilities = band means of the lobe gains (table built at commit)
        uint32_t lobe = kLobeDiffuse;
        float plobe = 1.0f;
        const bool pick = lobes_on && w.arrived && w.mat != kNoMat && (int32_t)w.mat < sc.num_materials;
        if (pick) {
            const float* pr = sc.lobe_prob + 3 * (size_t)w.mat;
            const float p0 = pr[0], p1 = pr[1], p2 = pr[2];
            const float u = u01(r.w);
            const float c1 = p0, c2 = p0 + p1;
            lobe = u < c1 ? kLobeDiffuse : (u < c2 ? kLobeSpecular : kLobeTransmit);
            if (lobe == kLobeTransmit && !(p2 > 0.0f)) lobe = p1 > 0.0f ? kLobeSpecular : kLobeDiffuse;
            plobe = lobe == kLobeDiffuse ? p0 : (lobe == kLobeSpecular ? p1 : p2);
            // the listener side pairs a segment with its ARRIVAL vertex: the record of the previous step describes
            // this vertex and learns its lobe now
            if (w.side) store_mat(st, 2u * kp.num_local, w.k - 1, w.slot, w.mat | (lobe << kLobeShift));
        }
        w.lobe = pick ? lobe << kLobeShift : 0u;
        float ox = w.px, oy = w.py, oz = w.pz;
        if (lobe == kLobeDiffuse) {
            // (pre_cone: r.y, r.z, r.w hold cone_local's result for this bounce, computed ahead by another lane — the same operations)
            if (pre_cone) cone_world(w.nx, w.ny, w.nz, __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w), dx, dy, dz);
            else sample_cone(w.nx, w.ny, w.nz, u01(r.y), u01(r.z), kp.cosine, dx, dy, dz);
            float cos_theta = dx * w.nx + dy * w.ny + dz * w.nz;
            float pdf = cos_theta / kPi;
            w.prob_new = pdf * kp.rr_prob;
        } else if (lobe == kLobeSpecular) {                               // mirror direction of the arriving ray
            float dn = ray.dx * w.nx + ray.dy * w.ny + ray.dz * w.nz;
            float k2 = 2.0f * dn;
            dx = fmaf(-k2, w.nx, ray.dx); dy = fmaf(-k2, w.ny, ray.dy); dz = fmaf(-k2, w.nz, ray.dz);
            w.prob_new = kp.rr_prob;
        } else {                                                          // straight on, from the far side of the surface
            dx = ray.dx; dy = ray.dy; dz = ray.dz;
            w.prob_new = kp.rr_prob;
            float back = -2.0f * kp.surface_offset;
            ox = fmaf(back, w.nx, w.px); oy = fmaf(back, w.ny, w.py); oz = fmaf(back, w.nz, w.pz);
        }
        if (pick) w.prob_new = w.prob_new * plobe;
        ray = make_ray(ox, oy, oz, dx, dy, dz);
        return true;
    }
    ray = make_ray(w.px, w.py, w.pz, dx, dy, dz);
    return true;
}

// An end point's collision (the reference's traces query ECC_Pawn too): a sphere; a ray that starts inside
// leaves through the far side.  The legacy tracer's pawn is the same sphere.
__device__ __forceinline__ bool sphere_hit(const Ray& r, const float c[3], float rad, float tmax, float& t_out) {
    float ox = r.ox - c[0], oy = r.oy - c[1], oz = r.oz - c[2];
    float b = fmaf(ox, r.dx, fmaf(oy, r.dy, oz * r.dz));
    float cc = fmaf(ox, ox, fmaf(oy, oy, oz * oz)) - rad * rad;
    float disc = fmaf(b, b, -cc);
    if (!(disc >= 0.0f)) return false;
    float sq = sqrtf(disc);
    float t = -b - sq;
    if (!(t > 0.0f)) t = sq - b;
    if (!(t > 0.0f && t <= tmax)) return false;
    t_out = t;
    return true;
}
// ImpactNormal of a sphere hit: unit (impact - centre), flipped to face the ray origin side like a triangle's
__device__ __forceinline__ void sphere_normal(const Ray& r, float t, const float c[3], float& nx, float& ny, float& nz) {
    float x = fmaf(t, r.dx, r.ox) - c[0], y = fmaf(t, r.dy, r.oy) - c[1], z = fmaf(t, r.dz, r.oz) - c[2];
    float l2 = x * x + y * y + z * z;
    float inv = 1.0f / sqrtf(l2);
    x = x * inv; y = y * inv; z = z * inv;
    float dn = fmaf(x, r.dx, fmaf(y, r.dy, z * r.dz));
    if (dn > 0.0f) { x = -x; y = -y; z = -z; }
    nx = x; ny = y; nz = z;
}

// bottom of the loop (ARTS.cpp:339-347): apply the closest hit (or the miss) and record the segment
// EXT: the instantiation that knows FS_FLAG_DOUBLE_POSITIONS and the end points' collision spheres (both decided at run
// time inside it); the default instantiation carries neither — not a register, not an instruction
// surf (cooperative walk): the hit triangle's unit normal (as stored, not yet flipped) and material bits, handed over by the
// lane that tested it — the loads of hit_surface are saved
template <bool EXT = false>
__device__ __forceinline__ void walker_apply_hit(Walker& w, const KParams& kp, const DeviceScene& sc,
                                                 const SubpathState& st, const Ray& ray, const Trav& T, const float4* surf = nullptr) {
    float qx = w.px, qy = w.py, qz = w.pz;
    double dqx = w.dpx, dqy = w.dpy, dqz = w.dpz;
    uint32_t mat_new = w.mat;
    bool hit = T.leaf_index >= 0;
    float t = T.t;
    if (hit && surf) {
        float x = surf->x, y = surf->y, z = surf->z;
        const float dn = fmaf(x, ray.dx, fmaf(y, ray.dy, z * ray.dz));
        if (dn > 0.0f) { x = -x; y = -y; z = -z; }
        w.nx = x; w.ny = y; w.nz = z;
        mat_new = __float_as_uint(surf->w);
    } else if (hit) hit_surface(sc, T.leaf_index, ray, w.nx, w.ny, w.nz, mat_new);
    // the OTHER end point's collision sphere (the walk's own actor is ignored, ARTS.cpp:322-334); wins ties with a triangle
    const float other_radius = !EXT ? 0.0f : (w.side ? kp.source_radius : kp.listener_radius);
    if (EXT && other_radius > 0.0f) {
        float c[3] = {w.side ? kp.src[0] : kp.lis[0], w.side ? kp.src[1] : kp.lis[1], w.side ? kp.src[2] : kp.lis[2]};
        if (w.side && kp.src_table) {   // batched frame: this pair's source
            const uint32_t sid = w.li / kp.pairs_per_source;
            c[0] = kp.src_table[4 * sid]; c[1] = kp.src_table[4 * sid + 1]; c[2] = kp.src_table[4 * sid + 2];
        }
        float ts;
        if (sphere_hit(ray, c, other_radius, kp.max_trace_dist, ts) && (!hit || ts <= t)) {
            hit = true; t = ts;
            sphere_normal(ray, ts, c, w.nx, w.ny, w.nz);
            mat_new = kNoMat;                                             // a pawn has no UAcousticGeometryComponent
        }
    }
    const bool DPOS = EXT && kp.dpos != 0;
    if (hit) {                                                            // ARTS.cpp:345-347
        if (DPOS) {   // Hit.ImpactPoint + 0.1 * Hit.ImpactNormal in FVector (double) arithmetic; the ray starts at the node's float rounding
            const bool shifted = ray.ox != w.px || ray.oy != w.py || ray.oz != w.pz;   // (transmitted lobe only: never with this flag)
            const double ipx = (shifted ? (double)ray.ox : w.dpx) + (double)t * (double)ray.dx;
            const double ipy = (shifted ? (double)ray.oy : w.dpy) + (double)t * (double)ray.dy;
            const double ipz = (shifted ? (double)ray.oz : w.dpz) + (double)t * (double)ray.dz;
            dqx = ipx + (double)kp.surface_offset * (double)w.nx;
            dqy = ipy + (double)kp.surface_offset * (double)w.ny;
            dqz = ipz + (double)kp.surface_offset * (double)w.nz;
            qx = (float)dqx; qy = (float)dqy; qz = (float)dqz;
        } else {
            qx = fmaf(kp.surface_offset, w.nx, fmaf(t, ray.dx, ray.ox));   // ray origin = node position, except behind
            qy = fmaf(kp.surface_offset, w.ny, fmaf(t, ray.dy, ray.oy));   // the surface for a transmitted segment
            qz = fmaf(kp.surface_offset, w.nz, fmaf(t, ray.dz, ray.oz));
        }
        w.has_normal = true;
    }
    w.arrived = hit;
    // the segment just added (zero length on a miss: the duplicate node of ARTS.cpp:296)
    float nd;
    if (DPOS) {   // FVector::Dist(...) / 1000.f: a double, narrowed by the assignment to float NodeDistance (ARTS.cpp:372-373)
        const double ex = dqx - w.dpx, ey = dqy - w.dpy, ez = dqz - w.dpz;
        nd = (float)(sqrt(ex * ex + ey * ey + ez * ez) / (double)kp.dist_divisor);
    } else {
        float ddx = qx - w.px, ddy = qy - w.py, ddz = qz - w.pz;
        float dist = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);            // ARTS.cpp:372
        nd = dist / kp.dist_divisor;                                      // ARTS.cpp:373
    }
    // Record for EvaluatePath (done by connect_kernel in path order): node i of the reference's loop is
    // the DEPARTURE node on the source side and — the listener subpath being reversed in the connected
    // path — the ARRIVAL node on the listener side.
    const float2 rec_np = w.side == 0 ? make_float2(nd, w.prob) : make_float2(nd, w.prob_new);
    const uint32_t rec_mat = w.side == 0 ? (w.mat | w.lobe) : mat_new;   // w.lobe: 0 unless FS_FLAG_MATERIAL_LOBES picked one here
    if (rec_in_main(st, w.k)) {
        const size_t r = rec_main(2u * kp.num_local, w.k, w.slot);       // consecutive lanes, consecutive words
#ifdef FS_NT_STORES   // experiment: streaming stores, so that the records do not push the scene out of the L2s
        __builtin_nontemporal_store(v2f{rec_np.x, rec_np.y}, reinterpret_cast<v2f*>(&st.seg_np[r]));
        __builtin_nontemporal_store(rec_mat, &st.seg_mat[r]);
#else
        st.seg_np[r] = rec_np;
        st.seg_mat[r] = rec_mat;
#endif
        if (st.seg_pos) st.seg_pos[r] = make_float4(qx, qy, qz, 0.0f);   // all-connections mode (wave-uniform)
        if (st.seg_nrm) st.seg_nrm[r] = make_float4(w.nx, w.ny, w.nz, 0.0f);   // balance-heuristic weights only
    } else if (rec_fits(st, w.k, w.slot)) {                              // depth = 0: step 65.. of one of the longest walks
        const size_t r = rec_over(st, w.k, w.slot);
        st.over_np[r] = rec_np;
        st.over_mat[r] = rec_mat;
        if (st.seg_pos) st.over_pos[r] = make_float4(qx, qy, qz, 0.0f);
        if (st.seg_nrm) st.over_nrm[r] = make_float4(w.nx, w.ny, w.nz, 0.0f);
    } else {
        *st.overflow = 1u;                                               // the host grows the tier and traces again
    }
    w.px = qx; w.py = qy; w.pz = qz;
    if (DPOS) { w.dpx = dqx; w.dpy = dqy; w.dpz = dqz; }
    w.mat = mat_new;
    w.prob = w.prob_new;
    ++w.k;
}

// staged walks: leave / pick up a walk between two stages (SubpathState::cont_a / cont_b)
__device__ __forceinline__ void walker_suspend(const Walker& w, const SubpathState& st) {
    st.cont_a[w.slot] = make_float4(w.px, w.py, w.pz, w.prob);
    st.cont_b[w.slot] = make_float4(w.nx, w.ny, w.nz, __uint_as_float((w.mat & 0xFFFFu) | (w.has_normal ? kContHasNormal : 0u) |
                                                                        (w.arrived ? kContArrived : 0u) | kContAlive));
}
__device__ __forceinline__ bool walker_resume(Walker& w, const SubpathState& st, int step) {   // false: the walk has ended before
    const float4 a = st.cont_a[w.slot], c = st.cont_b[w.slot];
    const uint32_t bits = __float_as_uint(c.w);
    if (!(bits & kContAlive)) return false;
    w.px = a.x; w.py = a.y; w.pz = a.z; w.prob = a.w; w.prob_new = a.w;
    w.nx = c.x; w.ny = c.y; w.nz = c.z;
    w.mat = bits & 0xFFFFu;
    w.has_normal = (bits & kContHasNormal) != 0u;
    w.arrived = (bits & kContArrived) != 0u;
    w.k = step;
    return true;
}
// slots a stage covers: the walks the previous stage suspended at step stage.begin, i.e. those of stage.begin steps or
// more (one of exactly that length ends at its first roulette here) — buckets begin .. FS_MAX_DEPTH of the length-sorted
// schedule (the last bucket holds every walk of FS_MAX_DEPTH steps or more: a stage that starts later than that visits
// them all and the continuation record says which still walk: a walk of that bucket that ENDS — in whatever stage, also one that
// began before FS_MAX_DEPTH — clears its record; round 3 cleared it only in stages that begin at FS_MAX_DEPTH or later, so
// that a walk of 64 .. 69 steps under bounds like 9, 70 kept the record of its suspension at step 9 and walked on from it)
__device__ __forceinline__ uint32_t stage_slots(const WalkStage& sr, const SubpathState& st, uint32_t total, const unsigned* s_cnt) {
    if (sr.begin <= 0) return total;
    uint32_t n = 0;
    for (int L = min(sr.begin, FS_MAX_DEPTH); L <= FS_MAX_DEPTH; ++L) n += s_cnt[L];
    if (n > sr.slots_cap) { *st.overflow = 1u; n = sr.slots_cap; }   // more long walks than the launch has lanes for: the frame is traced again
    return n;
}

// slots of the long-walk lane (WalkLane): the walks of len steps or more, at most cap — the same number in every
// part of every launch of the frame (a function of the plan pass's bucket counts alone)
__device__ __forceinline__ uint32_t lane_slots(const WalkLane& ln, const unsigned* s_cnt) {
    if (ln.len <= 0) return 0u;
    uint32_t n = 0;
    for (int L = min(ln.len, FS_MAX_DEPTH); L <= FS_MAX_DEPTH; ++L) n += s_cnt[L];
    return min(n, ln.cap);
}

template <bool EXT = false>
__device__ __forceinline__ void walker_finish(const Walker& w, const SubpathState& st) {
    if (EXT && st.end_posd) { st.end_posd[3 * (size_t)w.slot] = w.dpx; st.end_posd[3 * (size_t)w.slot + 1] = w.dpy; st.end_posd[3 * (size_t)w.slot + 2] = w.dpz; }
#ifdef FS_NT_STORES
    __builtin_nontemporal_store(v4f{w.px, w.py, w.pz, w.prob}, reinterpret_cast<v4f*>(&st.end_pos[w.slot]));
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    __builtin_nontemporal_store(v2u{w.mat, (uint32_t)w.k}, reinterpret_cast<v2u*>(&st.end_misc[w.slot]));
#else
    st.end_pos[w.slot] = make_float4(w.px, w.py, w.pz, w.prob);
    st.end_misc[w.slot] = make_uint2(w.mat, (uint32_t)w.k);
#endif
}

// ---------------------------------------------------------------------------------------------------
// plan_kernel: the number of segments a subpath takes under Russian roulette depends only on the RNG
// stream (seed, pair, side, bounce) — never on the geometry — so it is known before any ray is traced.
// One pass buckets the subpath indices by length: bucket L owns perm[L * total, L * total + count[L])
// (worst-case capacity, so no prefix pass is needed); workgroups reserve their share of a bucket with one
// atomicAdd per occupied length.  Walk lanes then read the buckets in DESCENDING length order, so every
// wave holds walks of equal length and no lane idles because its neighbours' walks ended earlier.
// (Order never affects results.)  The pass also performs FlushEnergyBuffer (ARTS.cpp:157-161).
//   scratch[0] = subpath queue head (persistent walk), [1, 1 + kPlanBuckets) = bucket counts.
// ---------------------------------------------------------------------------------------------------
constexpr int kPlanBuckets = FS_MAX_DEPTH + 1;
#ifndef FS_PLAN_ITEMS
#define FS_PLAN_ITEMS 4
#endif
constexpr int kPlanItems = FS_PLAN_ITEMS;   // subpaths per plan-kernel thread

__device__ __forceinline__ int planned_length(uint32_t g, const KParams& kp) {
    const uint32_t n = kp.num_local;
    const uint32_t side = g >= n ? 1u : 0u;
    const uint32_t li = g - side * n, sid = li / kp.pairs_per_source;
    const uint32_t pair = kp.pair_begin + (li - sid * kp.pairs_per_source);       // batched frame: per-source pair index
    const uint32_t seed = item_seed_lo(kp, sid);
    int k = 0;
    for (; k < kp.depth; ++k) {
        const uint4 r = philox(pair, ((uint32_t)k << 1) | side, 0, seed, kp.seed_hi);
        if (!(u01(r.x) < kp.rr_prob)) break;   // ARTS.cpp:300-301
    }
    return k;
}

__device__ __forceinline__ void plan_body(const uint32_t bid, const uint32_t nblocks, const KParams& kp,
                                          unsigned* __restrict__ scratch, uint32_t* __restrict__ perm,
                                          float* __restrict__ energy, const int energy_words,
                                          float* const* __restrict__ energy_tab, const int energy_count) {
    __shared__ unsigned s_hist[kPlanBuckets];
    __shared__ unsigned s_base[kPlanBuckets];
    __shared__ unsigned s_seg;
    if (threadIdx.x == 0) s_seg = 0u;
    for (int i = threadIdx.x; i < kPlanBuckets; i += kBlock) s_hist[i] = 0u;
    if (energy_tab) {   // batched frame: every source's buffer (the table was copied on this stream before the launch)
        for (int k = 0; k < energy_count; ++k) {
            float* e = energy_tab[k];
            for (int i = bid * kBlock + threadIdx.x; i < energy_words; i += nblocks * kBlock) e[i] = 0.0f;
        }
    } else {
        for (int i = bid * kBlock + threadIdx.x; i < energy_words; i += nblocks * kBlock) energy[i] = 0.0f;
    }
    __syncthreads();
    const uint32_t total = 2u * kp.num_local;
    // kPlanItems subpaths per thread: the bucket counters are a handful of hot addresses, and every workgroup
    // pays one global atomic per occupied length — fewer, larger workgroup batches mean fewer of them
    int L[kPlanItems];   // bucket = planned length, walks of more than FS_MAX_DEPTH steps (depth = 0 only) share the last one
    unsigned rank[kPlanItems];
    unsigned my_segments = 0;
#pragma unroll
    for (int it = 0; it < kPlanItems; ++it) {
        const uint32_t g = (bid * kPlanItems + it) * kBlock + threadIdx.x;
        L[it] = 0; rank[it] = 0;
        if (g < total) {
            const int len = planned_length(g, kp);
            my_segments += (unsigned)len;
            L[it] = min(len, FS_MAX_DEPTH);
            rank[it] = atomicAdd(&s_hist[L[it]], 1u);
        }
    }
    if (my_segments) atomicAdd(&s_seg, my_segments);
    __syncthreads();
    for (int i = threadIdx.x; i < kPlanBuckets; i += kBlock)
        if (s_hist[i]) s_base[i] = atomicAdd(&scratch[1 + i], s_hist[i]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kPlanItems; ++it) {
        const uint32_t g = (bid * kPlanItems + it) * kBlock + threadIdx.x;
        if (perm && g < total) perm[(size_t)L[it] * total + s_base[L[it]] + rank[it]] = g;
    }
    // work counter: walk segments of this frame (a walk of length L traces L rays), one atomic per workgroup
    if (threadIdx.x == 0 && s_seg) atomicAdd(reinterpret_cast<unsigned long long*>(scratch + kCounterWord) + 7, (unsigned long long)s_seg);   // fs_stats.planned_segments
}



// The same pass for SMALL frames (at most kPlanCoopMax subpaths) and for uncapped walks somebody waits for (at most
// kPlanCoopMaxUncapped: KParams.plan_coop, set by frame_describe): the roulette of one subpath is a serial chain of Philox
// evaluations — up to ~85 for the longest of 2 000 uncapped walks, 42 us with a subpath per thread, a tenth of the reference's
// tick — but its bounces are independent: a wave takes plan_coop_items() subpaths and evaluates 64 bounces of one at a time,
// lane j the roulette of bounce j; the first lane whose draw ends the walk gives its length (ballot + find-first).  (64 draws
// per subpath where the chain makes 10 on average: for capped walks of a chip-filling frame the chain is the cheaper one.)
// subpaths per wave: 8 for the reference's own frame (2 000 subpaths: 250 waves), more for the ticks of many sources — every
// workgroup adds its counts to the ~ 30 occupied length buckets with one global atomic each, and with 32 subpaths per workgroup
// those atomics (40 000 on 65 addresses at 64 000 subpaths) were the pass: 34 us
__host__ __device__ inline int plan_coop_items(uint32_t lanes) { return lanes <= 4096u ? 8 : (lanes <= 32768u ? 16 : 32); }
__device__ __forceinline__ void plan_coop_body(const uint32_t bid, const uint32_t nblocks, const KParams& kp,
                                               unsigned* __restrict__ scratch, uint32_t* __restrict__ perm,
                                               float* __restrict__ energy, const int energy_words,
                                               float* const* __restrict__ energy_tab, const int energy_count) {
    __shared__ unsigned s_hist[kPlanBuckets];
    __shared__ unsigned s_base[kPlanBuckets];
    __shared__ unsigned s_seg;
    if (threadIdx.x == 0) s_seg = 0u;
    for (int i = threadIdx.x; i < kPlanBuckets; i += kBlock) s_hist[i] = 0u;
    if (energy_tab) {
        for (int k = 0; k < energy_count; ++k) {
            float* e = energy_tab[k];
            for (int i = bid * kBlock + threadIdx.x; i < energy_words; i += nblocks * kBlock) e[i] = 0.0f;
        }
    } else {
        for (int i = bid * kBlock + threadIdx.x; i < energy_words; i += nblocks * kBlock) energy[i] = 0.0f;
    }
    __syncthreads();
    const uint32_t total = 2u * kp.num_local, n = kp.num_local;
    const uint32_t lane = threadIdx.x & 63u;
    const int items = plan_coop_items(total);
    const uint32_t first = (bid * (kBlock / 64) + (threadIdx.x >> 6)) * (uint32_t)items;   // this wave's subpaths [first, first + items)
    int my_len = 0;
    for (int it = 0; it < items; ++it) {                     // (wave-uniform)
        const uint32_t g = first + (uint32_t)it;
        if (g >= total) break;
        const uint32_t side = g >= n ? 1u : 0u;
        const uint32_t li = g - side * n, sid = li / kp.pairs_per_source;
        const uint32_t pair = kp.pair_begin + (li - sid * kp.pairs_per_source);
        const uint32_t seed = item_seed_lo(kp, sid);
        int len = kp.depth;
        for (int k0 = 0; k0 < kp.depth; k0 += 64) {
            const int k = k0 + (int)lane;
            bool ends = k >= kp.depth;
            if (!ends) {
                const uint4 r = philox(pair, ((uint32_t)k << 1) | side, 0, seed, kp.seed_hi);
                ends = !(u01(r.x) < kp.rr_prob);              // ARTS.cpp:300-301
            }
            const unsigned long long m = __ballot(ends);
            if (m != 0ull) { len = k0 + __ffsll((long long)m) - 1; break; }
        }
        if (lane == (uint32_t)it) my_len = len;
    }
    const bool mine = lane < (uint32_t)items && first + lane < total;
    const int L = min(my_len, FS_MAX_DEPTH);
    unsigned rank = 0;
    if (mine) {
        rank = atomicAdd(&s_hist[L], 1u);
        if (my_len) atomicAdd(&s_seg, (unsigned)my_len);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kPlanBuckets; i += kBlock)
        if (s_hist[i]) s_base[i] = atomicAdd(&scratch[1 + i], s_hist[i]);
    __syncthreads();
    if (mine && perm) perm[(size_t)L * total + s_base[L] + rank] = first + lane;
    if (threadIdx.x == 0 && s_seg) atomicAdd(reinterpret_cast<unsigned long long*>(scratch + kCounterWord) + 7, (unsigned long long)s_seg);   // fs_stats.planned_segments
}

// launch slot -> subpath index through the buckets, longest walks first.  s_cnt = bucket counts in LDS.
__device__ __forceinline__ uint32_t planned_subpath(uint32_t slot, int depth, uint32_t total, const unsigned* s_cnt,
                                                    const uint32_t* __restrict__ perm) {
    uint32_t acc = 0;
    for (int L = depth; L > 0; --L) {
        const uint32_t c = s_cnt[L];
        if (slot < acc + c) return perm[(size_t)L * total + (slot - acc)];
        acc += c;
    }
    return perm[slot - acc];   // bucket 0
}



// ---------------------------------------------------------------------------------------------------
// Wave work sharing for closest-hit AND any-hit queries (one implementation, three instantiations).
//
// The rays of a wave need very different numbers of traversal steps (median 19, p99 40) and the wave waits for
// its slowest ray in every bounce.  A query parallelises: disjoint subtrees can be searched by different lanes.
// So a lane that has finished its own ray takes the OLDEST pending subtree (bottom of the stack: the one its
// owner would reach last) from a lane that still has pending entries, traverses it with that lane's ray and
// reports into the owner's LDS mailbox.  All of it happens inside one wave (lock-step), without atomics on the stacks.
//   closest hit (ANY = false): the partial answers merge by atomicMin on the 64-bit key (t bits << 32 | triangle
//     id) — exactly the (t, id) order a single traversal applies, so the result does not depend on who searched
//     what; a taken subtree starts from min(donor's bound, owner's mailbox).
//   any hit (ANY = true): most connection rays are blocked and end at their first hit, the unobstructed ones must
//     search every box along the segment; a hit anywhere settles the ray (flag in the owner's mailbox) and lanes
//     still searching for a settled ray drop their work.
//   IGN: FCollisionQueryParams::AddIgnoredActor per ray (legacy tracer).
// LDS rows of kBlock words: rays (closest: origin, direction, reciprocals = 9 rows, the thief recomputes -o*inv;
// any: origin, direction, tmax = 7 rows — the thief recomputes the reciprocals too, 40 B per lane keep two connect
// workgroups on a CU next to the histogram) | [ignored actor] | mailbox (closest: u64 key + leaf; any: blocked flag) | donation boxes ref,
// owner [, bound].  Every lane of the wave must call it (has_ray = false: nothing of its own, helps from the start).
// ---------------------------------------------------------------------------------------------------
template <bool ANY, bool IGN>
struct ShareArea {
    static constexpr int kRayRows = ANY ? 7 : 9;
    static constexpr int kRows = kRayRows + (IGN ? 1 : 0) + (ANY ? 1 : 3) + 2 + (ANY ? 0 : 1);
    static constexpr size_t kBytes = (size_t)kBlock * 4 * kRows;
    float* rs; uint32_t* rign; unsigned long long* rkey; int* rleaf; int* blocked; int* dref; int* down; float* dbound;
    __device__ __forceinline__ explicit ShareArea(int* base) {
        rs = reinterpret_cast<float*>(base);
        int* p = base + kRayRows * kBlock;
        rign = reinterpret_cast<uint32_t*>(p); if (IGN) p += kBlock;
        rkey = reinterpret_cast<unsigned long long*>(p); blocked = p; rleaf = p + 2 * kBlock;
        p += (ANY ? 1 : 3) * kBlock;
        dref = p; down = p + kBlock; dbound = reinterpret_cast<float*>(p + 2 * kBlock);
    }
};
constexpr size_t kShareLdsBytes = ShareArea<false, false>::kBytes;      // 60 B per lane
// The sharing round (ballots, donation boxes, the thieves' ray reload: ~50 VALU + ~35 SALU for the whole wave) runs only
// once this many lanes have nothing to do: feeding the first few idle lanes costs every lane more than it returns.
// Measured at cfg3 (profiles/r02_share_min_idle.log): 1 / 4 / 8 / 16 / 24 / 32 / 48 -> walk 0.304 / 0.303 / 0.298 /
// 0.294 / 0.300 / 0.309 / 0.334 ms, connect 0.075 -> 0.072 ms at 16.  Sparse waves start above it.
// (again on round 5's fused stream, profiles/r05_share_min_idle.log: 8 / 12 / 16 / 20 / 24 -> 971 / 981 / 988 / 981 / 975 M rays/s.)
#ifndef FS_SHARE_MIN_IDLE
#define FS_SHARE_MIN_IDLE 16
#endif
constexpr int kShareMinIdle = FS_SHARE_MIN_IDLE;
constexpr size_t kShareAnyLdsBytes = ShareArea<true, false>::kBytes;    // 40 B per lane
constexpr size_t kShareIgnLdsBytes = ShareArea<false, true>::kBytes;    // 64 B per lane

// returns: ANY — the ray is blocked; closest — a hit was found (T.t, T.id, T.leaf_index describe it)
template <bool ANY, bool IGN, bool COUNT = false>
__device__ __forceinline__ bool trav_shared(const DeviceScene& sc, bool has_ray, const Ray& own, float tmax,
                                            uint32_t ignore, Trav& T, int* stack, int* share) {
    const ShareArea<ANY, IGN> A(share);
    float* rs = A.rs;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wbase = tid & ~63u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    // publish this lane's ray and clear its mailbox
    rs[0 * kBlock + tid] = own.ox;  rs[1 * kBlock + tid] = own.oy;  rs[2 * kBlock + tid] = own.oz;
    rs[3 * kBlock + tid] = own.dx;  rs[4 * kBlock + tid] = own.dy;  rs[5 * kBlock + tid] = own.dz;
    if (ANY) {
        rs[6 * kBlock + tid] = tmax;
        A.blocked[tid] = 0;
    } else {
        rs[6 * kBlock + tid] = own.ix;  rs[7 * kBlock + tid] = own.iy;  rs[8 * kBlock + tid] = own.iz;
        A.rkey[tid] = ~0ull;
        A.rleaf[tid] = -1;
    }
    if (IGN) A.rign[tid] = ignore;
    unsigned owner = tid;   // block-local lane whose ray this lane is working on
    uint32_t wign = ignore;
    Ray wr = own;
    trav_init(T, tmax, has_ray && sc.num_nodes > 0);
    trav_deep_reset(sc, stack);
    // The records of the NEXT step are requested as soon as this step has decided what they are: behind the node part
    // of the step, before its triangle test (trav_advance) and before the work-sharing round below (ballots, donation
    // boxes, mailboxes: half a dozen LDS round trips), which both run in the shadow of the fetch.  A wave in the thin
    // tail of the frame runs alone on its SIMD and nothing else hides that latency.  Lanes that take work in the round
    // request theirs at its end.  The triangle records alternate between two register sets (the loop body is
    // instantiated twice): the one being tested is still needed while the next one is already arriving.
    NodeRegs N;
    TriRegs X, Y;
    trav_settle(sc, T, stack);
    trav_issue(sc, T, N, X);
    // one step of the wave; cur = the triangle registers that arrive with this step, nxt = the ones requested for the
    // next.  true = nothing is left anywhere in the wave.
    auto step = [&](TriRegs& cur, TriRegs& nxt) -> bool {
        trav_wait(N, cur);
#ifdef FS_TRAV_STATS
        {
            const unsigned long long mb = __ballot(trav_busy(T)), mth = __ballot(trav_busy(T) && owner != tid);
            if (lane == 0u) {
                unsigned long long* gs = g_trav_stats + (ANY ? 16 : 0);
                atomicAdd(&gs[8], (unsigned long long)__popcll(mb)); atomicAdd(&gs[9], (unsigned long long)__popcll(mth));
                atomicAdd(&gs[10], 1ull);
            }
        }
#endif
        if (trav_busy(T)) {
            trav_advance<ANY, IGN, COUNT>(sc, wr, T, stack, N, cur, nxt, wign);
            if (ANY) {
                if (T.leaf_index >= 0) { A.blocked[owner] = 1; T.leaf_index = -1; }   // first hit ends the query (T is idle now)
                else if (A.blocked[owner]) { T.cur = kDone; T.sp = 0; T.sb = 0; T.tri_i = 0; T.tri_n = 0; }   // settled by another lane
            } else if (!trav_busy(T) && T.leaf_index >= 0) {   // this (sub)traversal is over: report to the owner of the ray
                const unsigned long long key = ((unsigned long long)__float_as_uint(T.t) << 32) | (unsigned long long)T.id;
                atomicMin(&A.rkey[owner], key);
                if (A.rkey[owner] == key) A.rleaf[owner] = T.leaf_index;
            }
        }
        const bool idle = !trav_busy(T);
        const unsigned long long busy_m = __ballot(!idle);
        if (busy_m == 0ull) return true;                // nothing left anywhere in the wave
        const unsigned long long idle_m = __ballot(idle);
        if (__popcll(idle_m) < kShareMinIdle) return false;
        const bool can_give = !idle && T.sp > T.sb && T.sb >= 0;   // (a lane with entries in the deep store keeps what it has)
        const unsigned long long give_m = __ballot(can_give);
        if (idle_m != 0ull && give_m != 0ull) {
            const int n = min(__popcll(idle_m), __popcll(give_m));
            if (can_give) {
                const int r = __popcll(give_m & lt);
                if (r < n) {
                    A.dref[wbase + r] = stack[T.sb * kBlock];
                    A.down[wbase + r] = (int)owner;
                    if (!ANY) A.dbound[wbase + r] = T.t;
                    ++T.sb;
                    if (T.sb == T.sp) { T.sb = 0; T.sp = 0; }
                }
            }
            if (idle) {
                const int r = __popcll(idle_m & lt);
                if (r < n) {
                    const int e = A.dref[wbase + r];
                    owner = (unsigned)A.down[wbase + r];
                    float bound;
                    if (ANY) {
                        wr = make_ray(rs[0 * kBlock + owner], rs[1 * kBlock + owner], rs[2 * kBlock + owner],
                                      rs[3 * kBlock + owner], rs[4 * kBlock + owner], rs[5 * kBlock + owner]);
                        bound = rs[6 * kBlock + owner];
                    } else {
                        // the owner's mailbox may already hold a closer hit than the donor knew of
                        bound = __uint_as_float(min(__float_as_uint(A.dbound[wbase + r]), (uint32_t)(A.rkey[owner] >> 32)));
                        wr.ox = rs[0 * kBlock + owner];  wr.oy = rs[1 * kBlock + owner];  wr.oz = rs[2 * kBlock + owner];
                        wr.dx = rs[3 * kBlock + owner];  wr.dy = rs[4 * kBlock + owner];  wr.dz = rs[5 * kBlock + owner];
                        wr.ix = rs[6 * kBlock + owner];  wr.iy = rs[7 * kBlock + owner];  wr.iz = rs[8 * kBlock + owner];
                        wr.nox = -(wr.ox * wr.ix); wr.noy = -(wr.oy * wr.iy); wr.noz = -(wr.oz * wr.iz);   // as make_ray
                    }
                    if (IGN) wign = A.rign[owner];
                    T.cur = e; T.sp = 0; T.sb = 0; T.tri_i = 0; T.tri_n = 0;
                    T.t = bound; T.leaf_index = -1; T.id = 0xFFFFFFFFu;
                    trav_deep_reset(sc, stack);   // (an any-hit query that ended early may have left entries there)
                    trav_settle(sc, T, stack);
                    trav_issue(sc, T, N, nxt);
                }
            }
        }
        return false;
    };
    while (true) {
        if (step(X, Y)) break;
        if (step(Y, X)) break;
    }
    trav_drain(N, X, Y);
    if (ANY) return A.blocked[tid] != 0;
    // everything searched: the mailbox holds the closest hit of this lane's own ray
    const unsigned long long key = A.rkey[tid];
    if (key != ~0ull) {
        T.t = __uint_as_float((uint32_t)(key >> 32));
        T.id = (uint32_t)key;
        T.leaf_index = A.rleaf[tid];
        return true;
    }
    T.t = tmax;
    T.leaf_index = -1;
    T.id = 0xFFFFFFFFu;
    return false;
}
// the three uses: BDPT walk, ConnectSubpaths' visibility ray, legacy tracer
template <bool COUNT = false, bool IGN = false>
__device__ __forceinline__ void trav_run_shared(const DeviceScene& sc, const Ray& own, Trav& T, int* stack, int* s_dyn,
                                                float tmax, bool has_ray = true, uint32_t ignore = 0xFFFFFFFFu) {
    trav_shared<false, IGN, COUNT>(sc, has_ray, own, tmax, ignore, T, stack, s_dyn + (size_t)sc.stack_rows * kBlock);
}
template <bool COUNT = false>
__device__ __forceinline__ bool trav_any_shared(const DeviceScene& sc, bool has_ray, const Ray& own, float tmax,
                                                int* stack, int* share, uint32_t* nv = nullptr, uint32_t* nt = nullptr) {
    Trav T;
    const bool blocked = trav_shared<true, false, COUNT>(sc, has_ray, own, tmax, 0xFFFFFFFFu, T, stack, share);
    if (COUNT) { *nv += T.nv; *nt += T.nt; }
    return blocked;
}
// COUNT instantiations: this lane's record fetches -> the frame scratch's work counters (one atomic per lane: the
// counting frames are not timed)
__device__ __forceinline__ void add_fetch_counts(unsigned* scratch, int first_counter, uint32_t nv, uint32_t nt) {
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(scratch + kCounterWord);
    if (nv) atomicAdd(&counters[first_counter], (unsigned long long)nv);
    if (nt) atomicAdd(&counters[first_counter + 1], (unsigned long long)nt);
}


// IGN (= EXT unless said otherwise): the queries skip the triangles of the actor the walk starts from; the fused frame kernel's
// EXT flavour asks for that alone (IGN without EXT: no double positions, no end-point spheres — 23 -> 64 spilled registers with them)
template <int LOBES, bool COUNT, bool EXT = false, bool IGN = EXT>
__device__ __forceinline__ void walk_shared_body(const uint32_t bid, const DeviceScene& sc, const KParams& kp,
                                                 const SubpathState& st, const unsigned* __restrict__ scratch,
                                                 const uint32_t* __restrict__ perm, const WalkStage sr = WalkStage()) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock] | work-sharing area
    int* s_stack = s_dyn;
    __shared__ unsigned s_cnt[kPlanBuckets];
    if (perm) {
        for (int i = threadIdx.x; i <= FS_MAX_DEPTH; i += kBlock) s_cnt[i] = i <= min(kp.depth, FS_MAX_DEPTH) ? scratch[1 + i] : 0u;
    }
    if (perm) __syncthreads();
    const uint32_t li = bid * kBlock + threadIdx.x;
    if (li >= stage_slots(sr, st, 2u * kp.num_local, s_cnt)) return;
    const uint32_t slot = li;
    const uint32_t g = perm ? planned_subpath(slot, min(kp.depth, FS_MAX_DEPTH), 2u * kp.num_local, s_cnt, perm) : slot;
    int* stack = &s_stack[threadIdx.x];
    Walker w;
    walker_start(w, g, slot, kp, st, sr.begin == 0);
    if (sr.begin > 0 && !walker_resume(w, st, sr.begin)) return;
    Ray ray;
    uint32_t cnt_nv = 0u, cnt_nt = 0u, cnt_ni = 0u, cnt_nl = 0u, cnt_nd = 0u;
#ifdef FS_WAVE_TIMELINE
    const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long tl_trav = 0, tl_seg = 0;
#endif
    while (true) {
        if (w.k >= sr.end) { walker_suspend(w, st); break; }          // staged walk: the next stage goes on from here
        if (!walker_next_ray<LOBES>(w, kp, sc, st, ray)) {
            walker_finish<EXT>(w, st);
            if (st.cont_b && w.k >= FS_MAX_DEPTH) st.cont_b[slot] = make_float4(0.f, 0.f, 0.f, 0.f);   // a walk of the last schedule bucket: later stages visit this slot again
            break;
        }
        Trav T;
#ifdef FS_WAVE_TIMELINE
        const unsigned long long tl_a = __builtin_amdgcn_s_memtime();
#endif
        trav_run_shared<COUNT, IGN>(sc, ray, T, stack, s_dyn, kp.max_trace_dist, true, w.ign);   // (IGN: the walk's own actor is ignored)
#ifdef FS_WAVE_TIMELINE
        tl_trav += __builtin_amdgcn_s_memtime() - tl_a;
        ++tl_seg;
#endif
        if (COUNT) { cnt_nv += T.nv; cnt_nt += T.nt; cnt_ni += T.ni; cnt_nl += T.nl; cnt_nd += T.nd; }
        walker_apply_hit<EXT>(w, kp, sc, st, ray, T);
    }
    if (COUNT) {
        add_fetch_counts(const_cast<unsigned*>(scratch), 3, cnt_nv, cnt_nt);
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(const_cast<unsigned*>(scratch) + kCounterWord);
        if (cnt_ni) { atomicAdd(&counters[8], (unsigned long long)cnt_ni); atomicAdd(&counters[9], (unsigned long long)cnt_nl); atomicAdd(&counters[10], (unsigned long long)cnt_nd); }
    }
#ifdef FS_WAVE_TIMELINE
    {
        unsigned long long seg_max = tl_seg, trav_max = tl_trav;   // lanes of a wave leave the loop at different bounces
        for (int o = 32; o > 0; o >>= 1) {
            seg_max = max(seg_max, (unsigned long long)__shfl_xor((long long)seg_max, o));
            trav_max = max(trav_max, (unsigned long long)__shfl_xor((long long)trav_max, o));
        }
        if ((threadIdx.x & 63u) == 0u && g_wave_buf) {
            unsigned long long* o = g_wave_buf + 8ull * (bid * (kBlock / 64) + (threadIdx.x >> 6));
            o[0] = tl_r0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = trav_max;
            o[3] = __builtin_amdgcn_s_memtime() - tl_c0; o[4] = 0; o[5] = seg_max;
            o[6] = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (0 << 6) | 4)            // HW_REG_HW_ID bits [7:0]
                   | ((unsigned long long)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) << 32);   // HW_REG_XCC_ID
            o[7] = slot;
        }
    }
#endif
}



// Small frames on sparse waves: a frame of a few thousand subpaths is a handful of waves and takes the latency of
// its longest chain of closest-hit queries.  Here a wave owns only `rays_per_wave` subpaths (its first lanes) and
// the other lanes help with every query — the legacy tracer's scheme (update_sound_shared_kernel).  The loop is
// wave-uniform: lanes whose walk has ended (or that never had one) keep calling the shared traversal as helpers.
template <int LOBES, bool COUNT, bool EXT = false, bool IGN = EXT>
__device__ __forceinline__ void walk_sparse_body(const uint32_t bid, const DeviceScene& sc, const KParams& kp,
                                                 const SubpathState& st, const unsigned* __restrict__ scratch,
                                                 const uint32_t* __restrict__ perm, const int rays_per_wave,
                                                 const WalkStage sr = WalkStage(), const WalkLane ln = WalkLane()) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock] | work-sharing area
    int* s_stack = s_dyn;
    __shared__ unsigned s_cnt[kPlanBuckets];
    if (perm) {
        for (int i = threadIdx.x; i <= min(kp.depth, FS_MAX_DEPTH); i += kBlock) s_cnt[i] = scratch[1 + i];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = bid * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t slot = wave * (uint32_t)rays_per_wave + lane;
    bool alive = lane < (uint32_t)rays_per_wave && slot < stage_slots(sr, st, 2u * kp.num_local, s_cnt);
    if (ln.len > 0 && ln.mode == kLaneSkip) alive = alive && slot >= lane_slots(ln, s_cnt);   // (the long-walk lane's slots: cooperative waves of the same launch)
    int* stack = &s_stack[threadIdx.x];
    Walker w;
    walker_start(w, alive ? (perm ? planned_subpath(slot, min(kp.depth, FS_MAX_DEPTH), 2u * kp.num_local, s_cnt, perm) : slot) : 0u,
                 slot, kp, st, alive && sr.begin == 0);
    if (alive && sr.begin > 0) alive = walker_resume(w, st, sr.begin);
    Ray ray;
    uint32_t cnt_nv = 0u, cnt_nt = 0u;
    while (true) {
        bool go = false;
        if (alive) {
            if (w.k >= sr.end) { walker_suspend(w, st); alive = false; }   // staged walk: the next stage goes on from here
            else {
                go = walker_next_ray<LOBES>(w, kp, sc, st, ray);
                if (!go) {
                    walker_finish<EXT>(w, st);
                    if (st.cont_b && w.k >= FS_MAX_DEPTH) st.cont_b[slot] = make_float4(0.f, 0.f, 0.f, 0.f);   // a walk of the last schedule bucket: later stages visit this slot again
                    alive = false;
                }
            }
        }
        if (__ballot(go) == 0ull) break;
        Trav T;
        trav_run_shared<COUNT, IGN>(sc, ray, T, stack, s_dyn, kp.max_trace_dist, go, w.ign);
        if (COUNT) { cnt_nv += T.nv; cnt_nt += T.nt; }
        if (go) walker_apply_hit<EXT>(w, kp, sc, st, ray, T);
    }
    if (COUNT) add_fetch_counts(const_cast<unsigned*>(scratch), 3, cnt_nv, cnt_nt);
}



// ---------------------------------------------------------------------------------------------------
// Cooperative traversal (round 4): the frames a game actually issues — the reference's 1000 pairs per source, walks without
// a depth cap (ARTS.h:176, ARTS.cpp:294) — are a CHAIN of ~70 dependent closest-hit queries; their time is the latency of
// one query times the length of the longest walk, and nothing else.  On sparse waves (a wave owns 1, 2 or 4 subpaths) the
// lane-private descent with work stealing above spends ~10 us per query: every step is a 64-B node per LANE, a 4-way
// sort, three stack pushes and a round of ballots / donation boxes / mailboxes, and the parallelism only doubles per step.
// Here the G = 64 / R lanes of a group search ONE ray together, breadth first:
//   * the tree it walks is the 4-wide one folded two levels at a time into 16-WIDE nodes (fs_refit.hip: coop16_kernel) — half
//     the levels, and a query takes about as many steps as the tree has levels;
//   * the group keeps ONE stack of pending inner nodes in LDS; a step pops up to G / 16 of them, lane j takes child j & 15 of
//     node j >> 4 and fetches exactly that child's 16-byte record (CoopChild, fs_internal.hpp: the box as fp16, rounded
//     outwards, + the reference) — with ONE ds_read_b128 if the node is among the first DeviceScene.lds_nodes of the
//     array, which every workgroup stages in its LDS, else with one global_load_dwordx4 — and tests that box;
//   * the children that are hit and inner go back on the stack by a ballot + prefix count (no sort, no donation protocol);
//   * a lane whose child is a hit LEAF requests that leaf's triangles right away and tests them itself in the NEXT step,
//     in the shadow of that step's node fetch; a closer hit goes into the group's mailbox with ds_min_u64 on the
//     (t bits << 32 | triangle id) key — the (t, id) order of the closest-hit rule — and the mailbox's t is the bound every
//     lane prunes with from the next step on.
// Without pruning order this visits more boxes than the sorted descent, with lanes that would idle anyway; a step is one
// record fetch + ~50 instructions, and a query takes about as many steps as the tree has levels.  The closest hit
// is the minimum of the key over ALL triangles the ray hits within tmax (boxes only prune, and these are supersets of the
// quantised ones), so the result is the one of trav_shared and of the oracle's brute-force scan, bit for bit, whatever
// the visiting order.
// The stack cannot overflow: a step pops k nodes and pushes at most 16 k; k is the full G / 16 only while that leaves room
// for the tree's worst-case one-node-at-a-time descent (DeviceScene.stack_need) on top, else the group descends one
// node per step (LIFO: from a stack of n entries such a descent never holds more than n + stack_need).
// LDS of a workgroup: [lds_nodes][16] CoopChild | per wave: kCoopCap pending-node words (divided among the R groups), R rays
// of 8 words, R mailboxes (u64 key, leaf).
// ---------------------------------------------------------------------------------------------------
constexpr int kCoopCap = 1024;
constexpr int kCoopMaxGroups = 4;
constexpr int kCoopRayWords = 12;    // origin, direction, reciprocals, tmax, ignored actor, -
constexpr int kCoopBoxWords = 8;     // mailbox: key (u64), hit leaf, - | unit normal of the hit triangle, its material
constexpr int kCoopRngWords = 64 * 4 + kCoopMaxGroups * 4;   // the walk's Philox words of the next bounces, one uint4 per lane | (pair, side, seed, first bounce) per group
constexpr int kCoopWaveWords = kCoopCap + kCoopMaxGroups * kCoopRayWords + kCoopMaxGroups * kCoopBoxWords + kCoopRngWords;
constexpr size_t kCoopWaveBytes = sizeof(int) * (size_t)kCoopWaveWords;
// may R rays share a wave on this tree?  (a group's share of the node stack must hold the worst-case descent + one wide step)
inline bool coop_fits(const CoopView& cv, int R) {
    const int per = (1 << cv.wshift) - 1, kfull = std::max(1, (64 / R) >> cv.wshift);
    return cv.rec != nullptr && cv.nodes > 0 && (64 / R) >= (1 << cv.wshift) && cv.stack_need + 8 + per * kfull <= kCoopCap / R;
}
// the node array waves of R rays walk (FS_COOP_WIDE: bit r set = 16-wide nodes for 2^r rays per wave; default 1 and 2 rays)
inline const CoopView* coop_view(const DeviceScene& sc, int R) {
    static const int wide_mask = std::getenv("FS_COOP_WIDE") ? std::atoi(std::getenv("FS_COOP_WIDE")) : 3;
    if (!sc.coop_info) return nullptr;
    const int bit = R == 1 ? 1 : (R == 2 ? 2 : 4);
    const CoopView* v = (wide_mask & bit) ? &sc.coop_info->wide16 : &sc.coop_info->wide4;
    if (!coop_fits(*v, R)) v = v == &sc.coop_info->wide16 ? &sc.coop_info->wide4 : &sc.coop_info->wide16;
    return coop_fits(*v, R) ? v : nullptr;
}
// dynamic LDS of a cooperative kernel launched with `waves` waves per workgroup and the view's resident nodes
inline size_t coop_lds_bytes(int waves, const CoopView& cv) { return ((size_t)cv.lds_nodes << cv.wshift) * 16u + kCoopWaveBytes * (size_t)waves; }
// the first words of the workgroup's dynamic LDS: the resident records.  Every thread of the workgroup must call it.
__device__ __forceinline__ void coop_stage_nodes(const CoopView& cv, int* s_dyn) {
    const uint4* src = reinterpret_cast<const uint4*>(cv.rec);
    uint4* dst = reinterpret_cast<uint4*>(s_dyn);
    for (int i = threadIdx.x; i < (cv.lds_nodes << cv.wshift); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}
// this wave's words behind the staged records
__device__ __forceinline__ int* coop_wave_words(const CoopView& cv, int* s_dyn) {
    return s_dyn + (((size_t)cv.lds_nodes << cv.wshift) * 4) + (size_t)(threadIdx.x >> 6) * kCoopWaveWords;
}

// LDS words other lanes of the wave write: typed address-space-3 accesses (ds_read / ds_write; a `volatile` generic pointer
// makes the compiler emit flat loads with system scope and a full s_waitcnt vmcnt(0) behind each — which also waits for
// every triangle record in flight), relaxed wave-scope atomics so that nothing is cached in a register across a step.
typedef __attribute__((address_space(3))) int LdsInt;
typedef __attribute__((address_space(3))) unsigned long long LdsU64;
__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load((const LdsInt*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store((LdsInt*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ unsigned long long lds_ld64(const unsigned long long* p) { return __hip_atomic_load((const LdsU64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void lds_st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store((LdsU64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
__device__ __forceinline__ void lds_min64(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_fetch_min((LdsU64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

// one triangle (leaf-order index `leaf`) against the group's ray; a closer hit updates the lane's own best
// best_surf: the unit normal and the material of the lane's best triangle — computed from the record in hand with the
// builders' own operation sequence (fs_bvh.cpp / fs_build.hip / update_tris_kernel: the stored normal is exactly this), so
// that the walker needs neither the normal array nor the record again
template <bool IGN>
__device__ __forceinline__ bool coop_tri(const float4 a, const float4 b, const float4 c, const Ray& r, const float bound, const uint32_t ign,
                                         const int leaf, unsigned long long& best_key, int& best_leaf, float4& best_surf) {
    float t = 0.0f;
    bool hit = tri_hit(a, b, c, r, bound, t);
    if (IGN) hit = hit & (__float_as_uint(c.w) != ign);
    const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) | __float_as_uint(c.z);
    const bool better = hit & (key < best_key);
    best_key = better ? key : best_key;
    best_leaf = better ? leaf : best_leaf;
    if (better) {
        const float e1x = a.w, e1y = b.x, e1z = b.y, e2x = b.z, e2y = b.w, e2z = c.x;
        const float nx = fmaf(e1y, e2z, -(e1z * e2y));
        const float ny = fmaf(e1z, e2x, -(e1x * e2z));
        const float nz = fmaf(e1x, e2y, -(e1y * e2x));
        const float l2 = nx * nx + ny * ny + nz * nz;
        const float inv = 1.0f / sqrtf(l2);
        best_surf = make_float4(nx * inv, ny * inv, nz * inv, c.y);
    }
    return better;
}

// The records are requested by hand, like the lane-private traversal's (trav_issue: one asm statement executed by every
// lane, the lanes that want a record selected by EXEC inside it, every destination tied in and out).  Left to the
// compiler, the loop's loads are waited for with s_waitcnt vmcnt(0) at the top of every step (its counter bookkeeping
// gives up at the loop's back edge) — i.e. the triangles requested at the end of a step land before the next step's
// node records are even requested.  By hand a step is: pop -> request the records -> s_waitcnt vmcnt(1): the triangle
// records of the previous step have landed (loads return in order and exactly the one record request is younger) -> test
// them while the records are in flight -> s_waitcnt vmcnt(0) -> boxes -> push -> request the hit leaves' triangles.
// tools/check_isa_hazards.py proves on the final ISA that nothing touches a register that is still in flight (it knows
// counted waits inside a basic block).
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4u LdsV4u;
struct CoopTris { v4f a0, b0, c0, a1, b1, c1; };
// m: the lanes that fetch from global memory (not zero).  ONE vector memory instruction, always.
__device__ __forceinline__ void coop_issue_node(const CoopView& cv, const uint32_t rec, const unsigned long long m, v4u& N) {
    const CoopChild* np = cv.rec + rec;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m]\n\t"
                 "global_load_dwordx4 %[q], %[np], off\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [q] "+&v"(N), [sv] "=&s"(sv)
                 : [np] "v"(np), [m] "s"(m)
                 : "memory");
}
// m1: lanes with a hit leaf (its first triangle), m2: those whose leaf has a second one
__device__ __forceinline__ void coop_issue_tris(const DeviceScene& sc, const int first, const unsigned long long m1, const unsigned long long m2,
                                                CoopTris& X) {
    const Tri48* tp = sc.tris + (uint32_t)first;
    unsigned long long sv;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_mov_b64 exec, %[m1]\n\t"
                 "s_cbranch_execz 2f\n\t"
                 "global_load_dwordx4 %[a0], %[tp], off\n\t"
                 "global_load_dwordx4 %[b0], %[tp], off offset:16\n\t"
                 "global_load_dwordx4 %[c0], %[tp], off offset:32\n\t"
                 "s_mov_b64 exec, %[m2]\n\t"
                 "s_cbranch_execz 2f\n\t"
                 "global_load_dwordx4 %[a1], %[tp], off offset:48\n\t"
                 "global_load_dwordx4 %[b1], %[tp], off offset:64\n\t"
                 "global_load_dwordx4 %[c1], %[tp], off offset:80\n"
                 "2:\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [a0] "+&v"(X.a0), [b0] "+&v"(X.b0), [c0] "+&v"(X.c0), [a1] "+&v"(X.a1), [b1] "+&v"(X.b1), [c1] "+&v"(X.c1), [sv] "=&s"(sv)
                 : [tp] "v"(tp), [m1] "s"(m1), [m2] "s"(m2)
                 : "memory");
}
// the triangle records of the previous step have landed: exactly the one record request of coop_issue_node is younger
__device__ __forceinline__ void coop_wait_tris_behind_node(CoopTris& X) {
    asm volatile("s_waitcnt vmcnt(1)" : "+v"(X.a0), "+v"(X.b0), "+v"(X.c0), "+v"(X.a1), "+v"(X.b1), "+v"(X.c1));
}
__device__ __forceinline__ void coop_wait_all(v4u& N, CoopTris& X) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(N), "+v"(X.a0), "+v"(X.b0), "+v"(X.c0), "+v"(X.a1), "+v"(X.b1), "+v"(X.c1));
}
__device__ __forceinline__ float4 f4(const v4f v) { return make_float4(v.x, v.y, v.z, v.w); }
typedef _Float16 h2f __attribute__((ext_vector_type(2)));

// R = 1, 2 or 4 rays per wave: lane r < R owns ray r (has_ray: it has one), group r = lanes [r G, (r + 1) G) searches it.
// Every lane of the wave must call it.  Returns (owner lanes): a hit was found, T.t / T.id / T.leaf_index describe it.
// lds_nodes_base: the workgroup's staged records (coop_stage_nodes), wl: this wave's words behind them.
// surf_out (owner lanes, on a hit): the hit triangle's stored unit normal and material bits (walker_apply_hit).
template <bool IGN, bool COUNT>
__device__ __forceinline__ bool trav_coop(const DeviceScene& sc, const CoopView& cv, const int R, const bool has_ray, const Ray& own, const float tmax,
                                          const uint32_t ignore, Trav& T, const int* lds_nodes_base, int* wl, unsigned* overflow,
                                          float4* surf_out = nullptr) {
    const unsigned lane = threadIdx.x & 63u;
    const int gshift = R == 1 ? 6 : (R == 2 ? 5 : 4);
    const int G = 1 << gshift;
    const int g = (int)(lane >> gshift), j = (int)(lane & (unsigned)(G - 1));
    const int cap = kCoopCap >> (6 - gshift);
    int* stk = wl + g * cap;
    int* rayw = wl + kCoopCap;                                                                 // [group][kCoopRayWords]
    int* boxw = wl + kCoopCap + kCoopMaxGroups * kCoopRayWords;                                // [group][kCoopBoxWords]: key | leaf, - | normal, material
    unsigned long long* keyw = reinterpret_cast<unsigned long long*>(boxw);                    // (group g's key: keyw[4 g])
    T.nv = 0u; T.nt = 0u; T.sp = 0;
    T.t = tmax; T.leaf_index = -1; T.id = 0xFFFFFFFFu;
    if (sc.num_nodes <= 0) return false;                   // empty scene (wave-uniform)
#ifdef FS_WAVE_TIMELINE   // diagnostic build: T.cur = cycles before the loop, T.tri_i = in the triangle sections (wait + tests), T.tri_n = waiting for the records, T.sb = behind the loop
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime();
    T.cur = 0; T.tri_i = 0; T.tri_n = 0; T.sb = 0;
#endif
    if (lane < (unsigned)R) {   // the owners publish their rays, clear their mailboxes and put the root on their group's stack
        // (the ray's reciprocals travel too: the owner has them from make_ray; three v_rcp_f32 and their guards per lane and query saved)
        if (R > 1) {
            int* rw = rayw + lane * kCoopRayWords;
            lds_st(rw + 0, __float_as_int(own.ox)); lds_st(rw + 1, __float_as_int(own.oy)); lds_st(rw + 2, __float_as_int(own.oz));
            lds_st(rw + 3, __float_as_int(own.dx)); lds_st(rw + 4, __float_as_int(own.dy)); lds_st(rw + 5, __float_as_int(own.dz));
            lds_st(rw + 6, __float_as_int(own.ix)); lds_st(rw + 7, __float_as_int(own.iy)); lds_st(rw + 8, __float_as_int(own.iz));
            lds_st(rw + 9, __float_as_int(has_ray ? tmax : -1.0f));
            lds_st(rw + 10, (int)ignore);
        }
        lds_st64(keyw + 4 * lane, ((unsigned long long)__float_as_uint(tmax) << 32) | 0xFFFFFFFFull);
        lds_st(boxw + lane * kCoopBoxWords + 2, -1);
        lds_st(wl + lane * cap, 0);
    }
    __builtin_amdgcn_wave_barrier();
    const int* rw = rayw + g * kCoopRayWords;
    Ray r;
    uint32_t ign;
    int n;                                                 // pending nodes of this group (the same number in all its lanes)
    if (R == 1) {   // (wave-uniform) one ray per wave: lane 0's registers are broadcast as they are — no round trip through LDS
        auto bc = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
        r.ox = bc(own.ox); r.oy = bc(own.oy); r.oz = bc(own.oz);
        r.dx = bc(own.dx); r.dy = bc(own.dy); r.dz = bc(own.dz);
        r.ix = bc(own.ix); r.iy = bc(own.iy); r.iz = bc(own.iz);
        ign = (uint32_t)__builtin_amdgcn_readfirstlane((int)ignore);
        n = __builtin_amdgcn_readfirstlane((has_ray && tmax > 0.0f) ? 1 : 0);
    } else {
        r.ox = __int_as_float(lds_ld(rw + 0)); r.oy = __int_as_float(lds_ld(rw + 1)); r.oz = __int_as_float(lds_ld(rw + 2));
        r.dx = __int_as_float(lds_ld(rw + 3)); r.dy = __int_as_float(lds_ld(rw + 4)); r.dz = __int_as_float(lds_ld(rw + 5));
        r.ix = __int_as_float(lds_ld(rw + 6)); r.iy = __int_as_float(lds_ld(rw + 7)); r.iz = __int_as_float(lds_ld(rw + 8));
        ign = (uint32_t)lds_ld(rw + 10);
        n = __int_as_float(lds_ld(rw + 9)) > 0.0f ? 1 : 0;
    }
    r.nox = -(r.ox * r.ix); r.noy = -(r.oy * r.iy); r.noz = -(r.oz * r.iz);   // as make_ray
    const unsigned long long gmask = R == 1 ? ~0ull : (((1ull << G) - 1ull) << (g * G));
    const unsigned long long below = gmask & ((1ull << lane) - 1ull);
    const int wshift = cv.wshift, per = (1 << wshift) - 1; // 16 (or 4) lanes per node: lane j takes child j & per of node j >> wshift
    const int kfull = G >> wshift;
    const int theta = cap - (cv.stack_need + 8);           // the stack may grow to here by wide steps
    const int wide_to = theta - per * kfull;               // with n <= wide_to a full step cannot pass theta
    const int* bound_w = boxw + g * kCoopBoxWords + 1;      // high word of the mailbox key = closest t so far
    const bool negx = r.ix < 0.0f, negy = r.iy < 0.0f, negz = r.iz < 0.0f;
    unsigned long long best_key = ~0ull;                   // this lane's own closest hit
    int best_leaf = -1;
    float4 best_surf = make_float4(0.f, 0.f, 0.f, 0.f);
    int pfirst = 0, pcnt = 0;                              // the leaf whose triangles this lane requested in the previous step
    v4u N = {0u, 0u, 0u, 0u};
    CoopTris X;
    X.a0 = v4f{0.f, 0.f, 0.f, 0.f}; X.b0 = X.a0; X.c0 = X.a0; X.a1 = X.a0; X.b1 = X.a0; X.c1 = X.a0;
    const int q = j >> wshift, c = j & per;
    const int resident = cv.lds_nodes;
#if defined(FS_WAVE_TIMELINE) && !defined(FS_WAVE_TIMELINE_FINE)
    T.cur = (int)(__builtin_amdgcn_s_memtime() - dbg_t0);
#endif
    while (true) {
#ifdef FS_WAVE_TIMELINE
        ++T.sp;                                            // diagnostic build: steps of this query (T.sp is not used here otherwise)
#endif
#ifdef FS_WAVE_TIMELINE_FINE
        const unsigned long long fine_top = __builtin_amdgcn_s_memtime();
#endif
        const float bound = __int_as_float(lds_ld(bound_w));
        // ---- pop: up to G / 4 nodes, fewer when the stack is close to the room the worst-case descent needs
        int kw = kfull;
        if (n > wide_to) { const int room = theta - n; kw = room >= per ? room / per : 1; }
        const int k = n < kw ? n : kw;
        const bool act = q < k;
        const int ref = act ? lds_ld(stk + (n - 1 - q)) : 0;
#ifdef FS_WAVE_TIMELINE_FINE   // (finer split of a step: T.cur = pop until the stack entry is here, T.sb = boxes + pushes + leaf requests)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long fine_t0 = __builtin_amdgcn_s_memtime();
        T.cur += (int)(fine_t0 - fine_top);
#endif
        const uint32_t rec = ((uint32_t)ref << wshift) + (uint32_t)c;
        const bool in_lds = ref < resident;
        // (exactly one request in every step, whatever the lanes need — the counted wait below relies on it: when every
        // record is resident, or only triangles are left, lane 0 fetches record 0 once more)
        const unsigned long long m_glob = __ballot(act && !in_lds);
        coop_issue_node(cv, rec, m_glob != 0ull ? m_glob : 1ull, N);
        v4u L = {0u, 0u, 0u, 0u};
        if (act && in_lds) {
            L = *reinterpret_cast<const LdsV4u*>((const LdsInt*)lds_nodes_base + 4u * rec);   // ds_read_b128
        }
#ifdef FS_WAVE_TIMELINE
        const unsigned long long dbg_t1 = __builtin_amdgcn_s_memtime();
#endif
        coop_wait_tris_behind_node(X);
        if (COUNT) T.nv += (act && c == 0) ? 1u : 0u;
        // ---- the triangles requested in the previous step, tested while this step's records are in flight
        auto test_pending = [&]() {
            if (pcnt > 0) {
                // (both triangles of the leaf in one straight line — two independent chains the scheduler interleaves; a leaf of one
                // triangle tests the registers' old content with a bound nothing passes)
                bool better = coop_tri<IGN>(f4(X.a0), f4(X.b0), f4(X.c0), r, bound, ign, pfirst, best_key, best_leaf, best_surf);
                better = coop_tri<IGN>(f4(X.a1), f4(X.b1), f4(X.c1), r, pcnt > 1 ? bound : -1.0f, ign, pfirst + 1, best_key, best_leaf, best_surf) | better;
                for (int i = 2; i < pcnt; ++i) {                // leaves of three and four triangles (FS_BVH_LEAF > 2 only)
                    const Tri48 x = sc.tris[pfirst + i];
                    better = coop_tri<IGN>(x.a, x.b, x.c, r, bound, ign, pfirst + i, best_key, best_leaf, best_surf) | better;
                }
                if (COUNT) T.nt += (uint32_t)pcnt;
                if (better) lds_min64(keyw + 4 * g, best_key);  // ds_min_u64: the group's closest hit so far
                pcnt = 0;
            }
        };
        test_pending();
        // ---- this lane's child box: fp16 planes, entry / exit distances as one fma per plane
#ifdef FS_WAVE_TIMELINE
        const unsigned long long dbg_t2 = __builtin_amdgcn_s_memtime();
        T.tri_i += (int)(dbg_t2 - dbg_t1);
#endif
        coop_wait_all(N, X);
#ifdef FS_WAVE_TIMELINE_FINE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
#ifdef FS_WAVE_TIMELINE
        T.tri_n += (int)(__builtin_amdgcn_s_memtime() - dbg_t2);
#endif
#ifdef FS_WAVE_TIMELINE_FINE
        const unsigned long long fine_box = __builtin_amdgcn_s_memtime();
#endif
        const v4u rc = in_lds ? L : N;
        const uint32_t w0 = rc.x, w1 = rc.y, w2 = rc.z;   // (scalars first: __builtin_bit_cast of a vector ELEMENT reads the vector's first word for each of them)
        const h2f lxy = __builtin_bit_cast(h2f, w0), lzhx = __builtin_bit_cast(h2f, w1), hyz = __builtin_bit_cast(h2f, w2);
        const float lox = (float)lxy.x, loy = (float)lxy.y, loz = (float)lzhx.x, hix = (float)lzhx.y, hiy = (float)hyz.x, hiz = (float)hyz.y;
        const float tnx = fmaf(negx ? hix : lox, r.ix, r.nox), tfx = fmaf(negx ? lox : hix, r.ix, r.nox);
        const float tny = fmaf(negy ? hiy : loy, r.iy, r.noy), tfy = fmaf(negy ? loy : hiy, r.iy, r.noy);
        const float tnz = fmaf(negz ? hiz : loz, r.iz, r.noz), tfz = fmaf(negz ? loz : hiz, r.iz, r.noz);
        const float tn = fmaxf(fmaxf(tnx, tny), fmaxf(tnz, 0.0f));
        const float tf = fminf(fminf(tfx, tfy), fminf(tfz, bound));
        const bool h = act & (tn <= tf);
        const int cref = (int)rc.w;
        const bool inner = h & (cref >= 0), leaf = h & (cref < 0);
        // ---- the hit inner children go back on the stack, in lane order
        const unsigned long long m_in = __ballot(inner) & gmask;
        const int pos = (n - k) + (int)__popcll(m_in & below);
        if (inner) {
            if (pos < cap) lds_st(stk + pos, cref);
            else *overflow = 1u;                            // (cannot happen while DeviceScene.stack_need is the tree's; the frame would be traced again)
        }
        n = n - k + (int)__popcll(m_in);
        n = n < cap ? n : cap;
        // ---- a hit leaf: request its triangles now, test them in the next step
        if (leaf) {
            const int code = ~cref;
            pfirst = code >> 2;
            pcnt = (code & 3) + 1;
        }
        const unsigned long long m_leaf = __ballot(leaf);
        if (m_leaf != 0ull) coop_issue_tris(sc, leaf ? pfirst : 0, m_leaf, __ballot(leaf && pcnt > 1), X);
#ifdef FS_WAVE_TIMELINE_FINE
        T.sb += (int)(__builtin_amdgcn_s_memtime() - fine_box);
#endif
        if (__ballot(n > 0) == 0ull) {
            // no group of the wave has a node left: only the triangles just requested are pending — they are tested here and now
            // instead of in another turn of the loop (an empty pop, a record request nobody needs, 64 boxes of zeros: ~ 600 cycles
            // of the ~ 9 000 of a query)
            coop_wait_all(N, X);
            test_pending();
            break;
        }
    }
    coop_wait_all(N, X);                                    // (nothing is in flight here; the compiler must know the registers are free)
#ifdef FS_WAVE_TIMELINE
    const unsigned long long dbg_t3 = __builtin_amdgcn_s_memtime();
#endif
    // ---- the mailbox holds the closest hit of the group's ray; the lane that found it says which triangle it was
    __builtin_amdgcn_wave_barrier();
    const unsigned long long fin = lds_ld64(keyw + 4 * g);
    if (R == 1) {   // (wave-uniform) one ray per wave: the finder's registers are read across, no second trip through LDS
        const unsigned long long who = __ballot(best_leaf >= 0 && best_key == fin);   // (one lane: a triangle is tested once per query)
        bool found1 = false;
        if (who != 0ull) {
            const int f = __ffsll((long long)who) - 1;
            const int leaf1 = __builtin_amdgcn_readlane(best_leaf, f);
            const float sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_surf.x), f));
            const float sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_surf.y), f));
            const float sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_surf.z), f));
            const float sw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(best_surf.w), f));
            if (lane == 0u) {
                T.t = __uint_as_float((uint32_t)(fin >> 32));
                T.id = (uint32_t)fin;
                T.leaf_index = leaf1;
                if (surf_out) *surf_out = make_float4(sx, sy, sz, sw);
                found1 = true;
            }
        }
        __builtin_amdgcn_wave_barrier();                    // (the next query's owner rewrites the mailbox)
#if defined(FS_WAVE_TIMELINE) && !defined(FS_WAVE_TIMELINE_FINE)
        T.sb = (int)(__builtin_amdgcn_s_memtime() - dbg_t3);
#endif
        return found1;
    }
    if (best_leaf >= 0 && best_key == fin) {               // (one lane: a triangle is tested once per query)
        int* bw = boxw + g * kCoopBoxWords;
        lds_st(bw + 2, best_leaf);
        lds_st(bw + 4, __float_as_int(best_surf.x)); lds_st(bw + 5, __float_as_int(best_surf.y));
        lds_st(bw + 6, __float_as_int(best_surf.z)); lds_st(bw + 7, __float_as_int(best_surf.w));
    }
    __builtin_amdgcn_wave_barrier();
    bool found = false;
    if (lane < (unsigned)R) {
        const int* bw = boxw + lane * kCoopBoxWords;
        const unsigned long long key = lds_ld64(keyw + 4 * lane);
        if ((uint32_t)key != 0xFFFFFFFFu) {
            T.t = __uint_as_float((uint32_t)(key >> 32));
            T.id = (uint32_t)key;
            T.leaf_index = lds_ld(bw + 2);
            if (surf_out) *surf_out = make_float4(__int_as_float(lds_ld(bw + 4)), __int_as_float(lds_ld(bw + 5)), __int_as_float(lds_ld(bw + 6)),
                                                  __int_as_float(lds_ld(bw + 7)));
            found = true;
        }
    }
    __builtin_amdgcn_wave_barrier();                        // (the next query's owners rewrite the rays and mailboxes)
#if defined(FS_WAVE_TIMELINE) && !defined(FS_WAVE_TIMELINE_FINE)
    T.sb = (int)(__builtin_amdgcn_s_memtime() - dbg_t3);
#endif
    return found;
}

// The walk on cooperative waves: a wave owns R = 1, 2 or 4 subpaths (its first lanes), every query is searched by the
// whole group of 64 / R lanes (trav_coop).  Same walker, records, stages and schedule as walk_sparse_body; the workgroup
// has blockDim.x / 64 waves (4 or 16: the more waves share the staged records, the more of them fit).
template <int LOBES, bool COUNT, bool EXT = false>
__device__ __forceinline__ void walk_coop_body(const uint32_t bid, const DeviceScene& sc, const CoopView& cv, const KParams& kp,
                                               const SubpathState& st, const unsigned* __restrict__ scratch,
                                               const uint32_t* __restrict__ perm, const int rays_per_wave,
                                               const WalkStage sr_in = WalkStage(), const WalkLane ln = WalkLane()) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [lds_nodes][16] records of 4 words | [waves][kCoopWaveWords]
    __shared__ unsigned s_cnt[kPlanBuckets];
    if (perm) {
        for (int i = threadIdx.x; i <= min(kp.depth, FS_MAX_DEPTH); i += blockDim.x) s_cnt[i] = scratch[1 + i];
    }
    coop_stage_nodes(cv, s_dyn);                            // (with the barrier the bucket counts need)
    const uint32_t lane = threadIdx.x & 63u;
    // (consecutive waves, consecutive slots: the eight longest walks of a frame share a CU, two to a SIMD.  Strided over the workgroups
    // instead — wave w of workgroup b = wave number w * gridDim.x + b — they have a CU each; measured: one-source tick's walk kernel
    // 243.8 -> 241.2 us, and frames of more waves than the chip holds LOSE — a workgroup then lives as long as its longest walk with seven
    // dead waves: 8-source tick 0.42 -> 0.55 ms.  Not kept)
    const uint32_t wave = bid * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint32_t slot = wave * (uint32_t)rays_per_wave + lane;
    int* wl = coop_wave_words(cv, s_dyn);
    // the stage of this lane's walk: the launch's, or the long-walk lane's own (WalkLane; the lane's slots come first)
    WalkStage sr = sr_in;
    bool mine = true;
    if (ln.len > 0) {
        const bool in_lane = slot < lane_slots(ln, s_cnt);
        if (in_lane) { sr.begin = ln.begin; sr.end = ln.end; }
        mine = ln.mode == kLaneBoth || (ln.mode == kLaneOnly) == in_lane;
    }
    bool alive = mine && lane < (uint32_t)rays_per_wave && slot < stage_slots(sr, st, 2u * kp.num_local, s_cnt);
    Walker w;
    walker_start(w, alive ? (perm ? planned_subpath(slot, min(kp.depth, FS_MAX_DEPTH), 2u * kp.num_local, s_cnt, perm) : slot) : 0u,
                 slot, kp, st, alive && sr.begin == 0);
    if (alive && sr.begin > 0) alive = walker_resume(w, st, sr.begin);
    Ray ray = make_ray(0.f, 0.f, 0.f, 0.f, 0.f, 1.f);
    uint32_t cnt_nv = 0u, cnt_nt = 0u;
    // The Philox words of a walk's bounces depend on (seed, pair, side, bounce) alone: the 64 / R lanes of the walk's group
    // compute the words of the next 64 / R bounces at once (ten rounds of four quarter-rate multiplies each, per bounce and
    // walk otherwise: a sixth of the time between two queries), the owner picks its bounce's words out of LDS.
    const int rshift = rays_per_wave == 1 ? 6 : (rays_per_wave == 2 ? 5 : 4);
    const int RG = 1 << rshift;
    int* rngw = wl + kCoopCap + kCoopMaxGroups * (kCoopRayWords + kCoopBoxWords);   // [64] uint4 | [group] (pair, side, seed, first bounce)
    int* rngb = rngw + 64 * 4;
    int rng_k0 = -(1 << 20);                               // owner lanes: the first bounce their group's cache holds
#ifdef FS_WAVE_TIMELINE
    const unsigned long long tl_r0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long tl_trav = 0, tl_seg = 0, tl_next = 0, tl_steps = 0, tl_pro = 0, tl_tri = 0, tl_nodewait = 0, tl_epi = 0;
#endif
    while (true) {
        bool go = false;
#ifdef FS_WAVE_TIMELINE
        const unsigned long long tl_n = __builtin_amdgcn_s_memtime();
#endif
        {   // refill the Philox cache of the groups whose walk has left it (every group recomputes: the others get the words they had)
            const bool need = alive && w.k < sr.end && (w.k < rng_k0 || w.k >= rng_k0 + RG);
            if (__ballot(need) != 0ull) {
                if (lane < (uint32_t)rays_per_wave) {
                    if (need) rng_k0 = w.k;
                    int* b = rngb + lane * 4;
                    lds_st(b + 0, (int)w.pair); lds_st(b + 1, (int)w.side);
                    lds_st(b + 2, (int)(kp.item_seeds > 0 ? item_seed_lo(kp, w.li / kp.pairs_per_source) : kp.seed_lo));
                    lds_st(b + 3, rng_k0);
                }
                __builtin_amdgcn_wave_barrier();
                const int* b = rngb + (lane >> rshift) * 4;
                const uint32_t bounce = (uint32_t)(lds_ld(b + 3) + (int)(lane & (uint32_t)(RG - 1)));
                uint4 pr = philox((uint32_t)lds_ld(b + 0), (bounce << 1) | (uint32_t)lds_ld(b + 1), 0, (uint32_t)lds_ld(b + 2), kp.seed_hi);
                if (LOBES == 0 && bounce > 0u) {   // every bounce but the first leaves a surface (unless every ray so far missed: the owner
                    float lx, ly, cphi;            //   then draws its words again): the cone sample's hit-independent half, here
                    cone_local(u01(pr.y), u01(pr.z), kp.cosine, lx, ly, cphi);
                    pr.y = __float_as_uint(lx); pr.z = __float_as_uint(ly); pr.w = __float_as_uint(cphi);
                }
                lds_st(rngw + 4 * lane + 0, (int)pr.x); lds_st(rngw + 4 * lane + 1, (int)pr.y);
                lds_st(rngw + 4 * lane + 2, (int)pr.z); lds_st(rngw + 4 * lane + 3, (int)pr.w);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (alive) {
            if (w.k >= sr.end) { walker_suspend(w, st); alive = false; }   // staged walk: the next stage goes on from here
            else {
                const int ri = 4 * (((int)lane << rshift) + (w.k - rng_k0));   // owner lane g: its group's lanes start at g * RG
                const uint4 pre = make_uint4((uint32_t)lds_ld(rngw + ri), (uint32_t)lds_ld(rngw + ri + 1), (uint32_t)lds_ld(rngw + ri + 2),
                                             (uint32_t)lds_ld(rngw + ri + 3));
                const bool cone_form = LOBES == 0 && w.k > 0;          // what the cache holds for this bounce
                go = (cone_form && !w.has_normal) ? walker_next_ray<LOBES>(w, kp, sc, st, ray)   // (all misses so far: a sphere sample from the raw words)
                                                  : walker_next_ray<LOBES>(w, kp, sc, st, ray, &pre, cone_form);
                if (!go) {
                    walker_finish<EXT>(w, st);
                    if (st.cont_b && w.k >= FS_MAX_DEPTH) st.cont_b[slot] = make_float4(0.f, 0.f, 0.f, 0.f);   // a walk of the last schedule bucket: later stages visit this slot again
                    alive = false;
                }
            }
        }
        if (__ballot(go) == 0ull) break;
        Trav T;
#ifdef FS_WAVE_TIMELINE
        const unsigned long long tl_a = __builtin_amdgcn_s_memtime();
        tl_next += tl_a - tl_n;
#endif
        float4 surf = make_float4(0.f, 0.f, 0.f, 0.f);
        trav_coop<EXT, COUNT>(sc, cv, rays_per_wave, go, ray, kp.max_trace_dist, w.ign, T, s_dyn, wl, st.overflow, &surf);
#ifdef FS_WAVE_TIMELINE
        tl_trav += __builtin_amdgcn_s_memtime() - tl_a;
        tl_steps += (unsigned long long)T.sp;
        tl_pro += (unsigned long long)T.cur; tl_tri += (unsigned long long)T.tri_i; tl_nodewait += (unsigned long long)T.tri_n; tl_epi += (unsigned long long)T.sb;
        ++tl_seg;
#endif
        if (COUNT) { cnt_nv += T.nv; cnt_nt += T.nt; }
        if (go) walker_apply_hit<EXT>(w, kp, sc, st, ray, T, &surf);
    }
    if (COUNT) add_fetch_counts(const_cast<unsigned*>(scratch), 3, cnt_nv, cnt_nt);
#ifdef FS_WAVE_TIMELINE
    if (lane == 0u && g_wave_buf) {   // [0] start, [1] end (100 MHz) | cycles: [2] in queries, [3] in all, [6] in the loop head | [4] traversal steps, [5] queries
        unsigned long long* o = g_wave_buf + 8ull * wave;
        o[0] = tl_r0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = tl_trav;
        o[3] = __builtin_amdgcn_s_memtime() - tl_c0; o[4] = tl_steps; o[6] = tl_next;
        o[5] = tl_seg | ((unsigned long long)__builtin_amdgcn_s_getreg(((16 - 1) << 11) | (0 << 6) | 4) << 32)             // HW_REG_HW_ID bits [15:0]: wave, simd, pipe, cu, sh, se
               | ((unsigned long long)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu) << 48);           // HW_REG_XCC_ID
        o[7] = (tl_pro & 0xFFFFull) | ((tl_tri / 16) & 0xFFFFull) << 16 | ((tl_nodewait / 16) & 0xFFFFull) << 32 | ((tl_epi / 16) & 0xFFFFull) << 48;   // (/16, 16 bits each)
        o[7] = (tl_pro / 16 & 0xFFFFull) | (o[7] & ~0xFFFFull);
    }
#endif
}



// ---------------------------------------------------------------------------------------------------
// connect_kernel: ConnectSubpaths + EvaluatePath + clamp/gain + deposit
// ---------------------------------------------------------------------------------------------------
// pairs_per_wave < 64: sparse waves for small frames — a wave owns that many pairs (its first lanes), the other
// lanes only help with the shared visibility queries (a frame of a few thousand pairs is otherwise a few waves
// waiting for their longest traversal).
// BATCH: a batched frame (fs_compute_energy_response_batch): the pairs of several sources lie end to end
// (kp.pairs_per_source each) and every source has its own energy buffer (energy_tab / fixed_tab); a workgroup
// takes (source, chunk) items and flushes its LDS histogram whenever the source changes.
// AHEAD (the connect kernels of uncapped walks that are waited for): a lane that evaluates its pair's path alone requests the
// records of AHEAD segments at once and applies them in path order — the same operations in the same order.  The records of a
// walk lie a whole level apart ([step][slot]): one at a time, every segment of a 100-segment path waited for its own miss.
// (Also measured: the paths of 40 segments or more evaluated by the whole wave, as a sparse wave does for every path — no gain
// on top of this: 88 -> 92 us at cfg3's size.  With the records ahead the longest path is no longer what the pass waits for.)
template <int B, int LOBES, bool BATCH, bool COUNT, bool EXT = false, int AHEAD = 1>
__device__ __forceinline__ void connect_body(const uint32_t bid, const uint32_t nblocks, const DeviceScene& sc,
                                             const KParams& kp, const SubpathState& st, float* __restrict__ energy,
                                             unsigned long long* __restrict__ fixed, unsigned* queue_head,
                                             const int pairs_per_wave, float* const* __restrict__ energy_tab,
                                             unsigned long long* const* __restrict__ fixed_tab) {
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];   // [stack_rows][kBlock] stack | [B][hist_window] histogram
    int* s_stack = s_dyn;
    float* s_hist = reinterpret_cast<float*>(s_dyn + (size_t)sc.stack_rows * kBlock);
    const int nb = kp.num_bins, W = kp.hist_window, NB = band_count<B>(kp);   // LDS histogram = the first W bins of every band (see KParams)
    int* s_share = reinterpret_cast<int*>(s_hist + (size_t)NB * W);   // work-sharing area of trav_any_shared
    __shared__ int s_lo, s_hi;
    __shared__ unsigned s_dep, s_tst, s_sgs;
#ifdef FS_WAVE_TIMELINE
    unsigned long long tl[6] = {__builtin_amdgcn_s_memrealtime(), 0, 0, 0, 0, 0};
#endif
    for (int i = threadIdx.x; i < NB * W; i += kBlock) s_hist[i] = 0.0f;
    if (threadIdx.x == 0) { s_lo = nb; s_hi = -1; s_dep = 0u; s_tst = 0u; s_sgs = 0u; }
    // this frame's walk is over: rearm the frame scratch (queue head, plan counts and cursors) for the next one
    if (bid == 0u)
        for (int i = threadIdx.x; i < 1 + 2 * kPlanBuckets; i += kBlock) queue_head[i] = 0u;
    __syncthreads();

    const uint32_t n = kp.num_local;
    const uint32_t total = 2u * n;
    unsigned my_deposits = 0, my_tested = 0, my_segments = 0;
    uint32_t cnt_nv = 0u, cnt_nt = 0u;
    // whole workgroups step through the pairs: every lane of a wave takes part in the shared visibility queries,
    // also the ones without a pair or without a segment to test
    const uint32_t ppw = (uint32_t)pairs_per_wave, per_block = ppw * (kBlock / 64);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;

    // one chunk of per_block pairs [first, first + per_block) clipped to `end`, deposits into s_hist / fixed_dst
    auto chunk = [&](uint32_t first, uint32_t end, unsigned long long* fixed_dst, float* far_dst) {
        const uint32_t li = first + wave * ppw + lane;
        const bool active = lane < ppw && li < end;
        const uint32_t lc = active ? li : 0u;
        const uint32_t sf = slot_of(st, lc), sl = slot_of(st, n + lc);   // where the walk left the two subpaths of the pair
        const float4 F = st.end_pos[sf];
        const uint2 Fm = st.end_misc[sf];
        const float4 L = st.end_pos[sl];
        const uint2 Lm = st.end_misc[sl];
        // visibility F_k -> B_m - 0.1 * unit(B_m - F_k) (ARTS.cpp:252-254); visible iff NO hit
        float dx = L.x - F.x, dy = L.y - F.y, dz = L.z - F.z;
        float l2 = dx * dx + dy * dy + dz * dz;
        float len = sqrtf(l2);
        float inv = 1.0f / len;
        float tmax = len - kp.connect_pullback;
        float ux = dx * inv, uy = dy * inv, uz = dz * inv;
        float conn_nd = 0.0f;       // FS_FLAG_DOUBLE_POSITIONS: the connection segment's scaled length, from the double end points
        if (EXT && kp.dpos) {       // (wave-uniform) FVector end points: difference, length and unit direction in double
            const double* Fd = st.end_posd + 3 * (size_t)sf;
            const double* Ld = st.end_posd + 3 * (size_t)sl;
            const double ex = Ld[0] - Fd[0], ey = Ld[1] - Fd[1], ez = Ld[2] - Fd[2];
            const double e2 = ex * ex + ey * ey + ez * ez;
            const double elen = sqrt(e2), einv = 1.0 / elen;
            l2 = e2 > 1e-8f ? 1.0f : 0.0f;                       // only its comparison with 1e-8 is used below
            ux = (float)(ex * einv); uy = (float)(ey * einv); uz = (float)(ez * einv);
            tmax = (float)(elen - kp.connect_pullback);
            conn_nd = (float)(elen / (double)kp.dist_divisor);
        }
        bool has_ray = active && (l2 > 1e-8f) && (tmax > 0.0f);
        Ray ray = make_ray(F.x, F.y, F.z, ux, uy, uz);
        // ConnectSubpaths ignores no actor (ARTS.cpp:252-254): the end points' collision spheres block
        bool sphere_blocked = false;
        if (EXT && has_ray && (kp.listener_radius > 0.0f || kp.source_radius > 0.0f)) {
            float ts;
            if (kp.listener_radius > 0.0f && sphere_hit(ray, kp.lis, kp.listener_radius, tmax, ts)) sphere_blocked = true;
            if (kp.source_radius > 0.0f) {
                float c[3] = {kp.src[0], kp.src[1], kp.src[2]};
                if (kp.src_table) { const uint32_t sid = lc / kp.pairs_per_source; c[0] = kp.src_table[4 * sid]; c[1] = kp.src_table[4 * sid + 1]; c[2] = kp.src_table[4 * sid + 2]; }
                if (sphere_hit(ray, c, kp.source_radius, tmax, ts)) sphere_blocked = true;
            }
            if (sphere_blocked) has_ray = false;   // settled without a traversal
        }
#ifdef FS_WAVE_TIMELINE
        if (!tl[1]) tl[1] = __builtin_amdgcn_s_memrealtime();   // first chunk: set-up and end-state loads done
#endif
        my_tested += active ? 1u : 0u;                                // one ConnectSubpaths per pair (ARTS.cpp:232 counts the connected ones)
        my_segments += active ? Fm.y + Lm.y : 0u;                     // the steps the two walks TOOK (each wrote its own count with its end state)
        const bool hit = trav_any_shared<COUNT>(sc, has_ray, ray, tmax, &s_stack[threadIdx.x], s_share, &cnt_nv, &cnt_nt);
#ifdef FS_WAVE_TIMELINE
        if (!tl[2]) tl[2] = __builtin_amdgcn_s_memrealtime();   // first chunk: visibility queries done
#endif
        float E[Bands<B>::kMax];
#pragma unroll
        for (int b = 0; b < Bands<B>::kMax; ++b) E[b] = 1.0f;
        float sd = 0.0f;
        // one lane evaluates its pair's connected path alone: EvaluatePath over F0..Fk, Bm..B0 (ARTS.cpp:262-267, 360-420), in path order
        auto eval_alone = [&]() {
        const int kf = (int)Fm.y, kl = (int)Lm.y;
        if (AHEAD > 1) {
            for (int j0 = 0; j0 < kf; j0 += AHEAD) {                      // source-side segments F_j -> F_j+1, AHEAD records in flight
                float2 np[AHEAD];
                uint32_t mt[AHEAD];
#pragma unroll
                for (int u = 0; u < AHEAD; ++u) { const int j = min(j0 + u, kf - 1); np[u] = load_np(st, total, j, sf); mt[u] = load_mat(st, total, j, sf); }
#pragma unroll
                for (int u = 0; u < AHEAD; ++u)
                    if (j0 + u < kf) { sd += np[u].x; apply_segment<B, LOBES>(E, np[u].x, mt[u], np[u].y, kp, sc); }
            }
        } else
        for (int j = 0; j < kf; ++j) {                                // source-side segments F_j -> F_j+1
            const float2 np = load_np(st, total, j, sf);
            sd += np.x;                                               // ARTS.cpp:374
            apply_segment<B, LOBES>(E, np.x, load_mat(st, total, j, sf), np.y, kp, sc);
        }
        {                                                             // connection segment: F_k's material/prob
            float dist = sqrtf(l2);
            float nd = (EXT && kp.dpos) ? conn_nd : dist / kp.dist_divisor;
            sd += nd;
            apply_segment<B, LOBES>(E, nd, Fm.x, F.w, kp, sc);
        }
        if (AHEAD > 1) {
            for (int j0 = kl - 1; j0 >= 0; j0 -= AHEAD) {                 // listener-side segments B_j+1 -> B_j
                float2 np[AHEAD];
                uint32_t mt[AHEAD];
#pragma unroll
                for (int u = 0; u < AHEAD; ++u) { const int j = max(j0 - u, 0); np[u] = load_np(st, total, j, sl); mt[u] = load_mat(st, total, j, sl); }
#pragma unroll
                for (int u = 0; u < AHEAD; ++u)
                    if (j0 - u >= 0) { sd += np[u].x; apply_segment<B, LOBES>(E, np[u].x, mt[u], np[u].y, kp, sc); }
            }
        } else
        for (int j = kl - 1; j >= 0; --j) {                           // listener-side segments B_j+1 -> B_j
            const float2 np = load_np(st, total, j, sl);
            sd += np.x;
            apply_segment<B, LOBES>(E, np.x, load_mat(st, total, j, sl), np.y, kp, sc);
        }
        };
        if (ppw == 1u || (ppw <= 8u && st.over_levels != 0)) {
            // Few pairs per wave (the reference's own frames: one; ticks of several sources with uncapped walks: up to eight, of
            // which a fifth connect — and a connected path of uncapped walks has up to a few hundred segments, 40 us of ONE
            // lane's time at 160): the wave's 64 lanes evaluate a connected path together — lane i the factors of segment i
            // (64 segments per round), then every lane runs the same product over them in path order, the factors read across
            // with v_readlane — one connected pair of the wave after the other; the pair's own lane keeps the result and deposits.
            bool go = active && !hit && !sphere_blocked;
            if (go && st.over_levels && !(rec_fits(st, (int)Fm.y - 1, sf) && rec_fits(st, (int)Lm.y - 1, sl))) go = false;
            unsigned long long todo = __ballot(go);
            if (todo == 0ull) return;
            const float my_nd = (EXT && kp.dpos) ? conn_nd : sqrtf(l2) / kp.dist_divisor;
            while (todo != 0ull) {                                       // (wave-uniform)
                const int o = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const int kf = __builtin_amdgcn_readlane((int)Fm.y, o), kl = __builtin_amdgcn_readlane((int)Lm.y, o);
                const uint32_t usf = (uint32_t)__builtin_amdgcn_readlane((int)sf, o), usl = (uint32_t)__builtin_amdgcn_readlane((int)sl, o);
                const float c_nd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_nd), o));
                const uint32_t c_mat = (uint32_t)__builtin_amdgcn_readlane((int)Fm.x, o);
                const float c_prob = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(F.w), o));
                const int segs = kf + 1 + kl;
                float Et[Bands<B>::kMax];
#pragma unroll
                for (int b = 0; b < Bands<B>::kMax; ++b) Et[b] = 1.0f;
                float sdt = 0.0f;
                for (int base = 0; base < segs; base += 64) {
                    const int i = base + (int)lane;
                    float nd = 0.0f, prob = 1.0f;
                    uint32_t mat = kNoMat;
                    if (i < kf) {                                            // source-side segment F_i -> F_i+1
                        const float2 np = load_np(st, total, i, usf);
                        nd = np.x; prob = np.y; mat = load_mat(st, total, i, usf);
                    } else if (i == kf) {                                    // connection segment: F_k's material / prob
                        nd = c_nd; prob = c_prob; mat = c_mat;
                    } else if (i < segs) {                                   // listener-side segment B_j+1 -> B_j, j = kl - 1 .. 0
                        const int j = kl - 1 - (i - kf - 1);
                        const float2 np = load_np(st, total, j, usl);
                        nd = np.x; prob = np.y; mat = load_mat(st, total, j, usl);
                    }
                    SegFactors<B> f;
                    segment_factors<B, LOBES>(f, nd, mat, prob, kp, sc);
                    const int cnt = min(64, segs - base);
                    for (int q = 0; q < cnt; ++q) {                          // (wave-uniform: the product in path order, in every lane alike)
                        sdt += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nd), q));                 // ARTS.cpp:374
                        if (!__builtin_amdgcn_readlane((int)f.live, q)) continue;
                        const float geo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.geo), q));
                        const float pw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.pw), q));
#pragma unroll
                        for (int b = 0; b < Bands<B>::kMax; ++b) {
                            if (B == 0 && b >= NB) break;
                            float e = Et[b];
                            e *= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.bsdf[b]), q));
                            e *= geo;
                            e *= __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.ex[b]), q));
                            e /= pw;
                            Et[b] = e;
                        }
                    }
                }
                if ((int)lane == o) {
#pragma unroll
                    for (int b = 0; b < Bands<B>::kMax; ++b) E[b] = Et[b];
                    sd = sdt;
                }
            }
            if (!go) return;
            ++my_deposits;
        } else {
        if (!active || hit || sphere_blocked) return;
        // depth = 0 only: a walk that outlived the record store has raised the overflow word — the frame is void and will
        // be traced again (FS_ERR_OVERFLOW); its pair must not be evaluated, the records it would read do not exist
        if (st.over_levels && !(rec_fits(st, (int)Fm.y - 1, sf) && rec_fits(st, (int)Lm.y - 1, sl))) return;
        ++my_deposits;
        eval_alone();
        }
        float delay = sd / kp.sound_speed;                            // ARTS.cpp:419
        float x = (delay * 1000.f) / 1.0f;                            // FSAC.h:89, BinSizeMs = 1
        float fl = floorf(x);
        int bin = !(fl > 0.0f) ? 0 : (fl >= (float)(nb - 1) ? nb - 1 : (int)fl);
        const bool near = bin < W;
        if (!fixed_dst && near) {
            atomicMin(&s_lo, bin);
            atomicMax(&s_hi, bin);
        }
#pragma unroll
        for (int b = 0; b < Bands<B>::kMax; ++b) {
            if (B == 0 && b >= NB) break;
            float e = E[b];
            e = (e < kp.energy_clamp) ? e : kp.energy_clamp;          // FMath::Min ARTS.cpp:410
            e *= kp.energy_gain;                                      // ARTS.cpp:413
            e *= kp.norm;                                             // ARTS.cpp:164-170
            if (fixed_dst)   // deterministic mode: integer sum of 2^-40 quanta — exact, so order- and shard-independent
                atomicAdd(&fixed_dst[b * nb + bin], (unsigned long long)__double2ll_rn((double)e * kFixedScale));
            else if (near)
                atomicAdd(&s_hist[b * W + bin], e);                   // ds_add_f32
            else
                atomicAdd(&far_dst[b * nb + bin], e);                 // beyond the LDS window: global_atomic_add_f32
        }
    };
    // LDS histogram -> one source's energy buffer (touched bin range only); clear = rearm it for the next source
    auto flush = [&](float* dst, bool clear) {
        __syncthreads();
        const int lo = s_lo, hi = s_hi;
        if (hi >= lo) {
            const int span = hi - lo + 1;
            for (int i = threadIdx.x; i < NB * span; i += kBlock) {
                int b = i / span, bin = lo + (i - b * span);
                float v = s_hist[b * W + bin];
                if (v != 0.0f) atomicAdd(&dst[b * nb + bin], v);      // global_atomic_add_f32
                if (clear) s_hist[b * W + bin] = 0.0f;
            }
        }
        if (clear) {
            __syncthreads();
            if (threadIdx.x == 0) { s_lo = nb; s_hi = -1; }
            __syncthreads();
        }
    };

    if (BATCH) {
        const uint32_t nps = kp.pairs_per_source, sources = n / nps;
        const uint32_t chunks = (nps + per_block - 1) / per_block;   // per source
        int cur = -1;
        for (uint32_t it = bid; it < chunks * sources; it += nblocks) {
            const uint32_t sid = it / chunks;
            if ((int)sid != cur) {
                if (cur >= 0 && !fixed_tab) flush(energy_tab[cur], true);
                cur = (int)sid;
            }
            chunk(sid * nps + (it - sid * chunks) * per_block, (sid + 1) * nps, fixed_tab ? fixed_tab[sid] : nullptr,
                  energy_tab[sid]);
        }
        if (cur >= 0 && !fixed_tab) flush(energy_tab[cur], false);
    } else {
        for (uint32_t base = bid * per_block; base < n; base += nblocks * per_block) chunk(base, n, fixed, energy);
    }
#ifdef FS_WAVE_TIMELINE
    tl[3] = __builtin_amdgcn_s_memrealtime();   // all chunks evaluated and deposited into LDS
#endif
    if (COUNT) add_fetch_counts(queue_head, 5, cnt_nv, cnt_nt);
    {   // work counters: summed per wave, then per workgroup in LDS — one global atomic per workgroup (thousands of
        // atomics on one address cost the kernel ~10 %)
        unsigned d = my_deposits, t = my_tested, g = my_segments;
        for (int o = 32; o > 0; o >>= 1) { d += __shfl_down(d, o); t += __shfl_down(t, o); g += __shfl_down(g, o); }
        if ((threadIdx.x & 63u) == 0u && d) atomicAdd(&s_dep, d);
        if ((threadIdx.x & 63u) == 0u && t) atomicAdd(&s_tst, t);
        if ((threadIdx.x & 63u) == 0u && g) atomicAdd(&s_sgs, g);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long* counters = reinterpret_cast<unsigned long long*>(queue_head + kCounterWord);
        if (s_dep) atomicAdd(&counters[2], (unsigned long long)s_dep);
        if (s_tst) atomicAdd(&counters[1], (unsigned long long)s_tst);   // the pairs this workgroup's lanes tested
        if (s_sgs) atomicAdd(&counters[0], (unsigned long long)s_sgs);   // fs_stats.segments: observed (planned: counters[7], by the plan pass)
    }
    if (!BATCH) {
        const int lo = s_lo, hi = s_hi;
        if (hi >= lo) {
            const int span = hi - lo + 1;
            for (int i = threadIdx.x; i < NB * span; i += kBlock) {
                int b = i / span, bin = lo + (i - b * span);
                float v = s_hist[b * W + bin];
                if (v != 0.0f) atomicAdd(&energy[b * nb + bin], v);       // global_atomic_add_f32
            }
        }
    }
#ifdef FS_WAVE_TIMELINE
    if ((threadIdx.x & 63u) == 0u && g_conn_buf) {
        unsigned long long* o = g_conn_buf + 8ull * (bid * (kBlock / 64) + (threadIdx.x >> 6));
        o[0] = tl[0]; o[1] = tl[1]; o[2] = tl[2]; o[3] = tl[3]; o[4] = __builtin_amdgcn_s_memrealtime();
        o[5] = my_deposits; o[6] = 0; o[7] = 0;
    }
#endif
}





// ReconstructImpulseResponse (FSAC.cpp:320-380) for one row (band, or row B = the band mean = the channel view) and one block
// of kBlock chunks of kChunk samples; s_amp: [nb] floats of LDS.  Shared by reconstruct_kernel (tail stream) and the
// reconstruct part of the fused frame kernel (fs_frame.hip).
#ifndef FS_RECON_CHUNK
#define FS_RECON_CHUNK 16
#endif
constexpr int kChunk = FS_RECON_CHUNK;
constexpr int kWarm = 96;
// A publish without the host's help: the reconstruct workgroups of a launch write the channel views straight into the sources'
// pinned host ring slots; every one of them, once its stores have been acknowledged, takes a ticket, and the workgroup that takes
// the last one stores the launch's id into the context's pinned host word — fs_get_impulse_response* and the ring's back-pressure
// read that word: no event, no copy command, no second stream (fs_capi_publish.cpp: owed_publish).  The ticket cell re-arms itself.
// The samples go to the host with SYSTEM-scope stores (store_sys: sc0 sc1 — written through to the host before they are
// acknowledged), so a wave whose store counter has run out (s_waitcnt vmcnt(0): on gfx9 stores count there too) knows that its
// samples are where the host reads them; the barrier collects the workgroup's waves, the tickets the launch's workgroups, and the
// word — a system-scope store as well — is issued only then: it can never overtake the samples.  Two things that do NOT work:
// plain stores + the counter (the word overtook the samples: tests/test_round3.py's stream of grouped frames read 6 of 7
// publishes too early — plain stores to fine-grained memory are acknowledged by the L2, not by the host), and a system-scope
// FENCE per wave (__threadfence_system(): correct, but it also writes back every dirty L2 line of the chip each time: the 3 072
// reconstruct workgroups of a 128-source tick paid 0.27 ms for it, the fused frame kernel 3 %).
__device__ __forceinline__ void store_sys(float* p, const float4 v) {   // 16 bytes, system scope (p 16-byte aligned)
    typedef float sys_v4f __attribute__((ext_vector_type(4)));
    const sys_v4f x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(x) : "memory");
}
__device__ __forceinline__ void store_sys(float* p, const float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void publish_arrive(unsigned* __restrict__ tickets, unsigned total, unsigned long long* __restrict__ host_word,
                                               unsigned long long id) {
    if (tickets == nullptr) return;                       // (uniform: this launch is published through an event)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t + 1u == total) {
            __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next launch of the stream starts behind this one)
            __hip_atomic_store(host_word, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// host_out (optional, row B only): the block's samples also go to that pinned host buffer, staged in `s_stage`
// (kBlock x (kChunk + 1) floats of LDS) and written with one 16-byte store per lane and instruction.
// ir_bands == nullptr (a frame whose IR is superseded within its own launch): only the channel row is produced, for the host.
// The zero-block rule of the host slot (slot_mask: one device word per ring slot, bit b = block b of the slot may hold non-zero
// samples).  A block none of whose reachable amplitudes (its bins, the bin before, the kWarm run-in) is non-zero produces exact
// zeros: it is written across the bus only if the slot still holds something else there.  A room's IR ends after 60 - 230 of the
// 1000 bins: 9 - 10 of a slot's 12 blocks stay on the device side of the bus (the 128-source tick wrote 24.6 MB per tick).
// Returns whether this workgroup must write its block to the host; every thread of the workgroup must call it.
__device__ __forceinline__ bool host_block_wanted(const float* s_amp, int nb, int spb, int base, int block, uint32_t* __restrict__ slot_mask) {
    if (slot_mask == nullptr) return true;
    const int b0 = max((base - kWarm) / spb - 1, 0), b1 = min((base + kBlock * kChunk - 1) / spb, nb - 1);
    bool nz = false;
    for (int b = b0 + (int)threadIdx.x; b <= b1; b += kBlock) nz = nz || s_amp[b] != 0.0f;
    const bool any = __syncthreads_or(nz ? 1 : 0) != 0;
    const uint32_t bit = 1u << block;
    const bool dirty = (__hip_atomic_load(slot_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit) != 0u;   // (only this workgroup touches this bit)
    __syncthreads();                                        // (everybody has read the word before thread 0 rewrites it)
    if (threadIdx.x == 0) {
        if (any && !dirty) __hip_atomic_fetch_or(slot_mask, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!any && dirty) __hip_atomic_fetch_and(slot_mask, ~bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return any || dirty;
}

__device__ __forceinline__ void reconstruct_body(const int row, const int chunk_block, const float* __restrict__ energy, int B, int nb,
                                                 int num_samples, int spb, float* __restrict__ ir_bands,
                                                 float* __restrict__ ir_mono, float* s_amp, float* host_out = nullptr,
                                                 float* s_stage = nullptr, uint32_t* __restrict__ slot_mask = nullptr) {
    const float Pi4 = sqrtf(4.0f * kPi);                           // FSAC.cpp:323
    if (ir_bands == nullptr && (row < B || host_out == nullptr)) return;   // (uniform) nobody wants this row
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
    const int chunk = chunk_block * kBlock + threadIdx.x;
    const int s0 = chunk * kChunk;
    bool to_host = host_out != nullptr && row == B;         // (uniform for the workgroup)
    if (to_host) to_host = host_block_wanted(s_amp, nb, spb, chunk_block * kBlock * kChunk, chunk_block, slot_mask);
    const bool staged = s_stage != nullptr;                 // (uniform) the block's samples leave through LDS: 16-byte stores of consecutive lanes
    if (s0 >= num_samples && !staged) return;
    float* out = ir_bands == nullptr ? nullptr : (row < B ? ir_bands + (size_t)row * num_samples : ir_mono);   // (nullptr: staged, host only)
    const int s1 = min(s0 + kChunk, num_samples);        // (a thread beyond the end: an empty range, it only joins the barrier below)
    const int i0 = s0 < num_samples ? max(s0 - kWarm, 0) : s1;
    int bin = i0 / spb;
    int bs = i0 - bin * spb;
    float cur = bin < nb ? s_amp[bin] : 0.0f;
    float prev = bin == 0 ? cur : (bin - 1 < nb ? s_amp[bin - 1] : 0.0f);   // FSAC.cpp:347-355
    const float fspb = (float)spb;
    float y = 0.0f;
    for (int i = i0; i < s1; ++i) {
        float x = 0.0f;
        if (bin < nb) {
            float wgt = (float)bs / fspb;                           // FSAC.cpp:359
            float a = (1.0f - wgt) * prev;
            float b = wgt * cur;
            x = a + b;                                              // FSAC.cpp:360
        }
        if (i == 0) {
            y = x;                                                  // Filtered[0] = IR[0] FSAC.cpp:371
        } else {
            float a = 0.25f * x;
            float b = (1.0f - 0.25f) * y;
            y = a + b;                                              // FSAC.cpp:374
        }
        if (i >= s0) {
            // (a thread's own 16 samples lie 64 bytes from its neighbour's: stored one by one, every store instruction of a wave
            // touches 64 lines — 20 of the 26 us of a one-source reconstruct, the same again for the host copy)
            if (staged) s_stage[threadIdx.x * (kChunk + 1) + (i - s0)] = y;   // (+ 1: conflict-free rows)
            else if (out) out[i] = y;
        }
        if (++bs == spb) {
            bs = 0;
            ++bin;
            prev = cur;
            cur = bin < nb ? s_amp[bin] : 0.0f;
        }
    }
    if (staged) {   // the block's kBlock * kChunk consecutive samples, 16 bytes per lane: to the device array, and the channel row to the host slot
        __syncthreads();
        const int base = chunk_block * kBlock * kChunk;
        for (int v = threadIdx.x; v < kBlock * kChunk / 4; v += kBlock) {
            const int s = 4 * v;
            if (base + s + 3 < num_samples) {
                float4 o;   // (sample s of the block lives in row s / kChunk of kChunk + 1 words)
                o.x = s_stage[s + s / kChunk]; o.y = s_stage[s + 1 + (s + 1) / kChunk];
                o.z = s_stage[s + 2 + (s + 2) / kChunk]; o.w = s_stage[s + 3 + (s + 3) / kChunk];
                if (out) *reinterpret_cast<float4*>(out + base + s) = o;
                if (to_host) store_sys(host_out + base + s, o);
            } else {
                for (int e = 0; e < 4; ++e)
                    if (base + s + e < num_samples) {
                        const float y1 = s_stage[s + e + (s + e) / kChunk];
                        if (out) out[base + s + e] = y1;
                        if (to_host) store_sys(host_out + base + s + e, y1);
                    }
            }
        }
    }
}

// The same reconstruct for the kernels that only reconstruct (reconstruct_kernel, reconstruct_batch_kernel), in two phases.  In
// reconstruct_body a thread's 16 samples cost it a chain of 16 + kWarm interpolated samples — a division, four branches and
// their bookkeeping each, ~ 60 instructions a sample on a wave that is alone on its SIMD: 20 of the 26 us of a one-source
// reconstruct (the stores, scattered or not, to the device or to the host, were 2 of them: profiles/r04 notes in DESIGN.md).
// Here every interpolated sample of the block (and of the kWarm before it) is computed ONCE, by the thread that owns it, into
// LDS; the filter chain then reads them: three arithmetic instructions a sample.  The same operations on the same operands in
// the same order as reconstruct_body: the same bits.
// LDS: s_amp [nb] | s_x [kBlock * kChunk + kWarm] | s_stage [kBlock][kChunk + 1].
__device__ __forceinline__ void reconstruct_body_fast(const int row, const int chunk_block, const float* __restrict__ energy, int B, int nb,
                                                      int num_samples, int spb, float* __restrict__ ir_bands, float* __restrict__ ir_mono,
                                                      float* s_amp, float* host_out, uint32_t* __restrict__ slot_mask = nullptr) {
    float* s_x = s_amp + nb;
    float* s_stage = s_x + kBlock * kChunk + kWarm;
    const float Pi4 = sqrtf(4.0f * kPi);                           // FSAC.cpp:323
    if (ir_bands == nullptr && (row < B || host_out == nullptr)) return;   // (uniform) a superseded frame: only its channel row, for the host
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
    bool to_host = host_out != nullptr && row == B;         // (uniform for the workgroup)
    float* out = ir_bands == nullptr ? nullptr : (row < B ? ir_bands + (size_t)row * num_samples : ir_mono);
    const int base = chunk_block * kBlock * kChunk;          // the block's first sample
    if (to_host) to_host = host_block_wanted(s_amp, nb, spb, base, chunk_block, slot_mask);
    const float fspb = (float)spb;
    // ---- phase 1: the interpolated samples x[base - kWarm .. base + kBlock * kChunk) -> s_x[0 ..): thread t its own kChunk, and the
    // first kWarm threads one sample each of the run-in (samples before 0 do not exist: never read)
    auto interp = [&](int i) {
        const int bin = i / spb, bs = i - bin * spb;
        float x = 0.0f;
        if (bin < nb) {
            const float cur = s_amp[bin];
            const float prev = bin == 0 ? cur : s_amp[bin - 1];         // FSAC.cpp:347-355
            const float wgt = (float)bs / fspb;                          // FSAC.cpp:359
            const float a = (1.0f - wgt) * prev;
            const float b = wgt * cur;
            x = a + b;                                                   // FSAC.cpp:360
        }
        return x;
    };
    {
        const int s0 = base + (int)threadIdx.x * kChunk;
        int bin = s0 / spb, bs = s0 - bin * spb;                     // (incrementally within the thread's own samples: no division by spb per sample)
        float cur = bin < nb ? s_amp[bin] : 0.0f;
        float prev = bin == 0 ? cur : (bin - 1 < nb ? s_amp[bin - 1] : 0.0f);
#pragma unroll 4
        for (int e = 0; e < kChunk; ++e) {
            float x = 0.0f;
            if (bin < nb) {
                const float wgt = (float)bs / fspb;
                const float a = (1.0f - wgt) * prev;
                const float b = wgt * cur;
                x = a + b;
            }
            s_x[kWarm + (int)threadIdx.x * kChunk + e] = x;
            if (++bs == spb) { bs = 0; ++bin; prev = cur; cur = bin < nb ? s_amp[bin] : 0.0f; }
        }
        if ((int)threadIdx.x < kWarm) {
            const int i = base - kWarm + (int)threadIdx.x;
            s_x[threadIdx.x] = i >= 0 ? interp(i) : 0.0f;
        }
    }
    __syncthreads();
    // ---- phase 2: the one-pole filter over this thread's kChunk samples behind a run-in of kWarm (0.75^96 ~ 1e-12)
    {
        const int s0 = base + (int)threadIdx.x * kChunk;
        const int i0 = max(s0 - kWarm, 0);                           // global index of the first sample of the chain
        const float* xs = s_x + (i0 - (base - kWarm));               // x[i0] in LDS
        float y = 0.0f;
        const int run = s0 - i0;                                     // kWarm, less at the very beginning of the IR
        float* my = s_stage + threadIdx.x * (kChunk + 1);            // (+ 1: conflict-free rows)
        int e = 0;
        if (i0 == 0) {                                               // Filtered[0] = IR[0] FSAC.cpp:371 (the first threads of the first block)
            y = xs[0];
            if (run == 0) my[0] = y;
            e = 1;
        }
#pragma unroll 8
        for (; e < run; ++e) { const float a = 0.25f * xs[e]; const float b = (1.0f - 0.25f) * y; y = a + b; }   // FSAC.cpp:374
#pragma unroll 4
        for (; e < run + kChunk; ++e) {
            const float a = 0.25f * xs[e]; const float b = (1.0f - 0.25f) * y; y = a + b;
            my[e - run] = y;
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < kBlock * kChunk / 4; v += kBlock) {   // 16 bytes per lane: the device array, and the channel row to the host slot
        const int sidx = 4 * v;
        if (base + sidx + 3 < num_samples) {
            float4 o;
            o.x = s_stage[sidx + sidx / kChunk]; o.y = s_stage[sidx + 1 + (sidx + 1) / kChunk];
            o.z = s_stage[sidx + 2 + (sidx + 2) / kChunk]; o.w = s_stage[sidx + 3 + (sidx + 3) / kChunk];
            if (out) *reinterpret_cast<float4*>(out + base + sidx) = o;
            if (to_host) store_sys(host_out + base + sidx, o);
        } else {
            for (int e = 0; e < 4; ++e)
                if (base + sidx + e < num_samples) {
                    const float y1 = s_stage[sidx + e + (sidx + e) / kChunk];
                    if (out) out[base + sidx + e] = y1;
                    if (to_host) store_sys(host_out + base + sidx + e, y1);
                }
        }
    }
}

// dynamic LDS of a traversal kernel: the scene's stack rows (+ extra bytes behind them).  Sizes above the default
// 48 KB limit are announced to the runtime once per (kernel instantiation, device); the host side has already
// checked the worst case against the device's LDS (fs_capi.cpp: lds_budget_ok), so a failure here is unexpected
// and is left for the launch's own error to report.
#ifdef FS_EXPERIMENTS
#define FS_SHARED_WALK(wl) ((wl).variant == 2)
#else
#define FS_SHARED_WALK(wl) true
#endif
inline size_t stack_bytes(const DeviceScene& sc) { return sizeof(int) * (size_t)sc.stack_rows * (size_t)kBlock; }
// The deep store must have a column for every lane of the grid about to be launched (DeviceScene.deep).  Grows it if
// not — a new buffer; the old one stays allocated for the launches already in the stream (DeepStore.retired).
// false: the allocation failed, the launch must be skipped (DeepStore.failed is set; the host reports it).
inline bool attach_deep(DeviceScene& sc, uint32_t blocks) {
    DeepStore* d = sc.deep_owner;
    if (d == nullptr || d->rows <= 0) { sc.deep = nullptr; sc.deep_lanes = 0; return true; }
    const size_t lanes = (size_t)blocks * kBlock;
    if (lanes > d->lanes) {
        size_t want = std::max<size_t>(d->lanes * 2, 2048 * (size_t)kBlock);
        while (want < lanes) want *= 2;
        int32_t* nb = nullptr;
        if (hipMalloc((void**)&nb, sizeof(int32_t) * want * (size_t)d->rows) != hipSuccess) { (void)hipGetLastError(); d->failed = true; return false; }
        if (d->buf) d->retired.push_back(d->buf);
        d->buf = nb; d->lanes = want;
    }
    sc.deep = d->buf; sc.deep_lanes = (uint32_t)d->lanes;
    return true;
}
constexpr int kMaxDevices = 64;
// LDS one workgroup may have on the current device (MI355X: all 160 KB of its CU)
inline size_t device_lds_per_block() {
    static std::atomic<int> cached[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
        cached[dev].store(v, std::memory_order_relaxed);
    }
    return (size_t)v;
}
// How many nodes of the breadth-first array a cooperative kernel stages in every workgroup's LDS: all of the tree if it
// fits, else its top.  A launch whose workgroups all fit the chip at once (one per CU) may take the CU's whole LDS; one
// that comes in rounds leaves room for a second workgroup per CU.
inline int coop_resident_nodes(const CoopView& cv, int waves_per_block, uint32_t blocks, int num_cus) {
    // (three workgroups of four waves per CU instead of one of eight, or LDS sized for three: 0.624 / 0.603 / 0.603 ms per 32-source tick — no setting)
    const size_t cu_lds = 160 * 1024, per_block = std::min(device_lds_per_block(), blocks <= (uint32_t)std::max(num_cus, 1) ? cu_lds : cu_lds / 2);
    const size_t fixed = kCoopWaveBytes * (size_t)waves_per_block + 1024;   // + the kernels' small static arrays
    if (per_block <= fixed || !cv.rec) return 0;
    return (int)std::min<size_t>((size_t)std::max(cv.nodes, 0), (per_block - fixed) / ((size_t)16 << cv.wshift));
}
template <typename K>
inline void allow_lds(K kernel, size_t bytes) {
    static std::atomic<size_t> allowed[kMaxDevices];   // per kernel instantiation; 0 = the default limit
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    const size_t have = std::max<size_t>(allowed[dev].load(std::memory_order_relaxed), 48 * 1024);
    if (bytes > have &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess)
        allowed[dev].store(bytes, std::memory_order_relaxed);
}

}  // namespace
}  // namespace fs
