// fs_refit.hip — moving geometry without a rebuild (row f4: dynamic props).
//
// The reference's line traces run against the live physics scene and include ECC_WorldDynamic objects
// (AudioRayTracingSubsystem.cpp:333-336, FrequenSeeAudioComponent.cpp:229-232), so a moved prop is seen by
// the next frame.  The host SAH build takes ~180 ms for 100 000 triangles; a moved subset is instead written
// straight into the leaf-order triangle records and the 4-wide tree is refitted bottom-up on the device:
// same topology, new boxes.  Results stay a function of ray and triangles only (boxes are conservative:
// padded and quantised outwards exactly as in fs_bvh.cpp), so they equal a fresh build's bit for bit.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr int kRefitBlock = 256;

// new vertex positions -> Tri64 {v0, e1, e2, unit normal}; material, input index and actor id are kept.
// Same fp32 operation order as the host build (fs_bvh.cpp) — the normal is part of the hit-normal spec.
__global__ __launch_bounds__(kRefitBlock) void update_tris_kernel(Tri64* __restrict__ tris, Tri48* __restrict__ packed,
                                                                  float4* __restrict__ nrm,
                                                                  const uint32_t* __restrict__ leaf_pos, int first,
                                                                  int count, const float* __restrict__ xyz) {
    const int i = blockIdx.x * kRefitBlock + threadIdx.x;
    if (i >= count) return;
    const float* p = xyz + 9 * (size_t)i;
    const uint32_t pos = leaf_pos[first + i];
    Tri64& r = tris[pos];
    const float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
    const float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
    r.a = make_float4(p[0], p[1], p[2], e1x);
    r.b = make_float4(e1y, e1z, e2x, e2y);
    r.c.x = e2z;
    const float nx = fmaf(e1y, e2z, -(e1z * e2y));
    const float ny = fmaf(e1z, e2x, -(e1x * e2z));
    const float nz = fmaf(e1x, e2y, -(e1y * e2x));
    const float l2 = nx * nx + ny * ny + nz * nz;
    const float inv = 1.0f / sqrtf(l2);
    r.d = make_float4(nx * inv, ny * inv, nz * inv, 0.f);
    packed[pos] = Tri48{r.a, r.b, r.c};
    nrm[pos] = r.d;
}

// the kernels' view of the records (fs_internal.hpp: Tri48 + normals) from the authoring records
__global__ __launch_bounds__(kRefitBlock) void pack_tris_kernel(const Tri64* __restrict__ tris, int count,
                                                                Tri48* __restrict__ packed, float4* __restrict__ nrm) {
    const int i = blockIdx.x * kRefitBlock + threadIdx.x;
    if (i >= count) return;
    const Tri64 r = tris[i];
    packed[i] = Tri48{r.a, r.b, r.c};
    nrm[i] = r.d;
}

struct Box3 {
    float lo[3], hi[3];
};

__device__ __forceinline__ void box_point(Box3& b, float x, float y, float z) {
    b.lo[0] = fminf(b.lo[0], x); b.hi[0] = fmaxf(b.hi[0], x);
    b.lo[1] = fminf(b.lo[1], y); b.hi[1] = fmaxf(b.hi[1], y);
    b.lo[2] = fminf(b.lo[2], z); b.hi[2] = fmaxf(b.hi[2], z);
}

// one thread per node of one tree level: child boxes (leaves from their triangles, inner children from the
// level below), own bounds, outward 8-bit quantisation on the node's power-of-two grid (as fs_bvh.cpp)
__global__ __launch_bounds__(kRefitBlock) void refit_level_kernel(NodeQ4* __restrict__ nodes,
                                                                  const Tri64* __restrict__ tris,
                                                                  float4* __restrict__ node_box, int begin, int end,
                                                                  float pad) {
    const int i = begin + blockIdx.x * kRefitBlock + threadIdx.x;
    if (i >= end) return;
    NodeQ4 q = nodes[i];
    const float inf = __builtin_inff();
    Box3 cb[4];
    bool used[4];
    Box3 nb;
    for (int k = 0; k < 3; ++k) { nb.lo[k] = inf; nb.hi[k] = -inf; }
    for (int c = 0; c < 4; ++c) {
        // an empty slot carries lo = 255 > hi = 0 on every axis
        used[c] = ((q.lox >> (8 * c)) & 0xFFu) <= ((q.hix >> (8 * c)) & 0xFFu);
        for (int k = 0; k < 3; ++k) { cb[c].lo[k] = inf; cb[c].hi[k] = -inf; }
        if (!used[c]) continue;
        const int32_t link = q.child[c];
        if (link >= 0) {
            const float4 lo = node_box[2 * (size_t)link], hi = node_box[2 * (size_t)link + 1];
            cb[c].lo[0] = lo.x; cb[c].lo[1] = lo.y; cb[c].lo[2] = lo.z;
            cb[c].hi[0] = hi.x; cb[c].hi[1] = hi.y; cb[c].hi[2] = hi.z;
        } else {
            const int code = ~link;
            const int first = code >> 2, n = (code & 3) + 1;
            for (int t = first; t < first + n; ++t) {
                const float4 a = tris[t].a, b = tris[t].b;
                const float e2z = tris[t].c.x;
                box_point(cb[c], a.x, a.y, a.z);
                box_point(cb[c], a.x + a.w, a.y + b.x, a.z + b.y);      // v0 + e1 (half an ulp off the input
                box_point(cb[c], a.x + b.z, a.y + b.w, a.z + e2z);      // vertex at most; pad >= 0.01 cm)
            }
        }
        for (int k = 0; k < 3; ++k) { nb.lo[k] = fminf(nb.lo[k], cb[c].lo[k]); nb.hi[k] = fmaxf(nb.hi[k], cb[c].hi[k]); }
    }
    node_box[2 * (size_t)i] = make_float4(nb.lo[0], nb.lo[1], nb.lo[2], 0.f);
    node_box[2 * (size_t)i + 1] = make_float4(nb.hi[0], nb.hi[1], nb.hi[2], 0.f);

    double origin[3], scale[3];
    float originf[3];
    for (int k = 0; k < 3; ++k) {
        origin[k] = (double)(nb.lo[k] - pad);
        const double ext = (double)(nb.hi[k] + pad) - origin[k];
        int e = (int)ceil(log2(fmax(ext, 1e-30) / 255.0));
        e = max(-100, min(100, e));
        while (ldexp(255.0, e) < ext) ++e;     // guard the log2 rounding
        scale[k] = ldexp(1.0, e);   // exact as a float: |e| <= 100
        originf[k] = (float)origin[k];
    }
    uint32_t lo4[3] = {0, 0, 0}, hi4[3] = {0, 0, 0};
    for (int c = 0; c < 4; ++c)
        for (int k = 0; k < 3; ++k) {
            uint32_t ql = 255, qh = 0;
            if (used[c]) {
                const double l = ((double)(cb[c].lo[k] - pad) - (double)originf[k]) / scale[k];
                const double h = ((double)(cb[c].hi[k] + pad) - (double)originf[k]) / scale[k];
                ql = (uint32_t)fmax(0.0, fmin(255.0, floor(l)));
                qh = (uint32_t)fmax(0.0, fmin(255.0, ceil(h)));
            }
            lo4[k] |= ql << (8 * c);
            hi4[k] |= qh << (8 * c);
        }
    q.ox = originf[0]; q.oy = originf[1]; q.oz = originf[2];
    q.sx = (float)scale[0]; q.sy = (float)scale[1]; q.sz = (float)scale[2];
    q.lox = lo4[0]; q.loy = lo4[1]; q.loz = lo4[2];
    q.hix = hi4[0]; q.hiy = hi4[1]; q.hiz = hi4[2];
    nodes[i] = q;
}

// ---- the cooperative traversal's view of the nodes (fs_internal.hpp: CoopChild) --------------------------------
// fp16 next below / next above by bit pattern (0x7C00 = +inf, 0xFC00 = -inf)
__device__ __forceinline__ uint16_t h_dec(uint16_t b) {   // toward -inf
    if ((b & 0x7FFFu) == 0u) return 0x8001u;
    return (b & 0x8000u) ? (uint16_t)(b + 1u) : (uint16_t)(b - 1u);
}
__device__ __forceinline__ uint16_t h_inc(uint16_t b) {   // toward +inf
    if ((b & 0x7FFFu) == 0u) return 0x0001u;
    return (b & 0x8000u) ? (uint16_t)(b - 1u) : (uint16_t)(b + 1u);
}
__device__ __forceinline__ float h_val(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
// an fp16 strictly below x (two steps below the nearest one at most: the box only has to be conservative)
__device__ __forceinline__ uint16_t h_below(float x) {
    if (!(x == x)) return 0xFC00u;
    uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)x);
    if (b == 0xFC00u) return b;
    if (h_val(b) > x) b = h_dec(b);
    return b == 0xFC00u ? b : h_dec(b);
}
__device__ __forceinline__ uint16_t h_above(float x) {
    if (!(x == x)) return 0x7C00u;
    uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)x);
    if (b == 0x7C00u) return b;
    if (h_val(b) < x) b = h_inc(b);
    return b == 0x7C00u ? b : h_inc(b);
}
// one thread per (node, child): the child's box off the node's 8-bit grid, rounded outwards to fp16, + its reference.
// An empty slot (lo = 255 > hi = 0) becomes the inverted box (+inf, -inf), which no ray enters.
__global__ __launch_bounds__(kRefitBlock) void coop_nodes_kernel(const NodeQ4* __restrict__ nodes, int n, CoopChild* __restrict__ out) {
    const int i = blockIdx.x * kRefitBlock + threadIdx.x;
    if (i >= 4 * n) return;
    const NodeQ4 q = nodes[i >> 2];
    const int c = i & 3, sh = 8 * c;
    const uint32_t ql[3] = {(q.lox >> sh) & 0xFFu, (q.loy >> sh) & 0xFFu, (q.loz >> sh) & 0xFFu};
    const uint32_t qh[3] = {(q.hix >> sh) & 0xFFu, (q.hiy >> sh) & 0xFFu, (q.hiz >> sh) & 0xFFu};
    const float o[3] = {q.ox, q.oy, q.oz}, st[3] = {q.sx, q.sy, q.sz};
    uint16_t lo[3], hi[3];
    const bool used = ql[0] <= qh[0] && ql[1] <= qh[1] && ql[2] <= qh[2];
    for (int k = 0; k < 3; ++k) {
        lo[k] = used ? h_below(fmaf((float)ql[k], st[k], o[k])) : (uint16_t)0x7C00u;
        hi[k] = used ? h_above(fmaf((float)qh[k], st[k], o[k])) : (uint16_t)0xFC00u;
    }
    CoopChild r;
    r.lo_xy = (uint32_t)lo[0] | ((uint32_t)lo[1] << 16);
    r.loz_hix = (uint32_t)lo[2] | ((uint32_t)hi[0] << 16);
    r.hi_yz = (uint32_t)hi[1] | ((uint32_t)hi[2] << 16);
    r.ref = used ? q.child[c] : 0;
    out[i] = r;
}

// Two levels of the 4-wide tree folded into one 16-wide node (what the cooperative traversal walks: half the steps per
// query).  Node X of an EVEN level becomes 16 records: for each child c of X — a leaf: its own record (and three empty
// ones); an inner node Y: the records of Y's four children.  The inner references among them (nodes of level + 2) are
// renumbered into the DENSE order of the even levels (dense[l] = first dense index of level l, -1 for odd levels), so that
// the array — and its LDS-resident prefix — holds nothing but nodes the traversal can reach.  One thread per (dense node, slot).
__global__ __launch_bounds__(kRefitBlock) void coop16_kernel(const CoopChild* __restrict__ in, const int32_t* __restrict__ level_begin,
                                                             const int32_t* __restrict__ dense, int levels, int n16,
                                                             CoopChild* __restrict__ out) {
    const int i = blockIdx.x * kRefitBlock + threadIdx.x;
    if (i >= 16 * n16) return;
    const int d = i >> 4, slot = i & 15, c = slot >> 2, g = slot & 3;
    int l = 0;
    for (int k = 0; k < levels; k += 2)                      // the even level that holds dense node d
        if (dense[k] >= 0 && dense[k] <= d) l = k;
    const int X = level_begin[l] + (d - dense[l]);
    CoopChild r;
    r.lo_xy = 0x7C007C00u; r.loz_hix = 0xFC007C00u; r.hi_yz = 0xFC00FC00u; r.ref = 0;     // empty: the inverted box
    const CoopChild xc = in[4 * X + c];
    const bool x_empty = xc.lo_xy == 0x7C007C00u && xc.loz_hix == 0xFC007C00u;
    if (!x_empty) {
        if (xc.ref < 0) {                                     // a leaf child of X: its own record in the group's first slot
            if (g == 0) r = xc;
        } else {                                              // an inner child Y: Y's child g
            const CoopChild yc = in[4 * xc.ref + g];
            const bool y_empty = yc.lo_xy == 0x7C007C00u && yc.loz_hix == 0xFC007C00u;
            if (!y_empty) {
                r = yc;
                if (yc.ref >= 0) r.ref = dense[l + 2] + (yc.ref - level_begin[l + 2]);   // a node of level l + 2, renumbered
            }
        }
    }
    out[i] = r;
}

}  // namespace

void launch_coop_nodes(const NodeQ4* nodes, int n, CoopChild* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(coop_nodes_kernel, dim3((unsigned)((4 * n + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, s, nodes, n, out);
}
void launch_coop16(const CoopChild* in, const int32_t* level_begin_dev, const int32_t* dense_dev, int levels, int n16, CoopChild* out, hipStream_t s) {
    if (n16 <= 0) return;
    hipLaunchKernelGGL(coop16_kernel, dim3((unsigned)((16 * n16 + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, s, in, level_begin_dev, dense_dev,
                       levels, n16, out);
}

void launch_update_triangles(Tri64* tris, Tri48* packed, float4* nrm, const uint32_t* leaf_pos, int first, int count,
                             const float* xyz, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(update_tris_kernel, dim3((unsigned)((count + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock),
                       0, s, tris, packed, nrm, leaf_pos, first, count, xyz);
}

void launch_pack_triangles(const Tri64* tris, int count, Tri48* packed, float4* nrm, hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(pack_tris_kernel, dim3((unsigned)((count + kRefitBlock - 1) / kRefitBlock)), dim3(kRefitBlock), 0, s,
                       tris, count, packed, nrm);
}

void launch_refit(NodeQ4* nodes, const Tri64* tris, float4* node_box, const int32_t* level_begin, int levels, float pad,
                  hipStream_t s) {
    for (int l = levels - 1; l >= 0; --l) {   // deepest level first: a node reads its children's bounds
        const int begin = level_begin[l], end = level_begin[l + 1];
        if (end <= begin) continue;
        hipLaunchKernelGGL(refit_level_kernel, dim3((unsigned)((end - begin + kRefitBlock - 1) / kRefitBlock)),
                           dim3(kRefitBlock), 0, s, nodes, tris, node_box, begin, end, pad);
    }
}

}  // namespace fs
