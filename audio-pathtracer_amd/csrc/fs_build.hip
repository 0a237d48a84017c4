// fs_build.hip — the acceleration structure built ON THE DEVICE, for topology changes.
//
// Actors register and unregister at run time (RegisterGeometry / UnregisterGeometry, AudioRayTracingSubsystem.h:99-100)
// and the reference's traces see the new physics scene in the next frame.  Moving geometry is refitted (fs_refit.hip);
// a changed triangle SET needs a new tree, and the host SAH build takes 18 ms for 100 000 triangles — forty frames.
// This builder produces the same data structures in a fraction of a millisecond:
//   1. morton_kernel     30-bit Morton code of every triangle's centroid (10 bits per axis over the scene bounds)
//                        with the triangle index below it: unique 64-bit keys
//   2. radix sort        hipcub::DeviceRadixSort (rocPRIM) on the keys — sorted order = leaf order
//   3. karras_kernel     the binary radix tree over the sorted keys (Karras 2012): every internal node finds its key
//                        range and split from common-prefix lengths, all nodes in parallel
//   4. collapse_kernel   4-wide nodes level by level, breadth-first (ONE workgroup walks the levels: the tree has
//                        ~N/4 wide nodes): a wide node opens the inner child that spans the most triangles until it
//                        has four (the binary tree needs no boxes: a bottom-up fit pass with its per-node fences cost
//                        more than the rest of the build); subtrees of <= 2 triangles become leaves (their triangles are
//                        adjacent in sorted order); records every level's node range and the worst-case stack need
//   5. records_kernel    leaf-order triangle records {v0, e1, e2, material, index, actor, unit normal} — the same fp32
//                        sequence as the host build — and the input-index -> leaf-position table
//   6. the existing refit pass (fs_refit.hip) derives every node's quantised child boxes bottom-up, exactly as the
//      host builder quantises (outwards, padded): boxes stay conservative, so closest hits are again a function of
//      ray and triangles only — bit-identical to the host-built tree's and to the brute-force oracle's.
// A Morton tree is a worse tree than the host's binned-SAH one (more node visits per ray; measured in DESIGN.md): it is
// the fast path for a frame that must not wait, fs_scene_commit remains the quality path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>

#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr int kBuildBlock = 256;
constexpr int kCollapseBlock = 1024;

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {   // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

__global__ __launch_bounds__(kBuildBlock) void morton_kernel(const float* __restrict__ xyz, int T, float3 lo, float3 inv_ext,
                                                             unsigned long long* __restrict__ keys) {
    const int t = blockIdx.x * kBuildBlock + threadIdx.x;
    if (t >= T) return;
    const float* p = xyz + 9 * (size_t)t;
    float c[3];
    for (int k = 0; k < 3; ++k) {
        const float mn = fminf(fminf(p[k], p[3 + k]), p[6 + k]), mx = fmaxf(fmaxf(p[k], p[3 + k]), p[6 + k]);
        c[k] = 0.5f * (mn + mx);
    }
    const uint32_t x = (uint32_t)fminf(fmaxf((c[0] - lo.x) * inv_ext.x * 1024.0f, 0.0f), 1023.0f);
    const uint32_t y = (uint32_t)fminf(fmaxf((c[1] - lo.y) * inv_ext.y * 1024.0f, 0.0f), 1023.0f);
    const uint32_t z = (uint32_t)fminf(fmaxf((c[2] - lo.z) * inv_ext.z * 1024.0f, 0.0f), 1023.0f);
    const uint32_t code = (expand_bits10(x) << 2) | (expand_bits10(y) << 1) | expand_bits10(z);
    keys[t] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)t;   // unique
}

// binary radix tree: internal nodes [0, N-1), leaves are the sorted keys; child >= 0 internal, ~leaf otherwise
struct Bvh2 {
    int* left; int* right; int* parent;   // parent of internal node i: parent[i]; of leaf j: parent[(N-1) + j]
    int* first; int* last;                // key range of an internal node
};

__device__ __forceinline__ int delta(const unsigned long long* __restrict__ k, int N, int i, int j) {
    if (j < 0 || j >= N) return -1;
    return __clzll((long long)(k[i] ^ k[j]));   // keys are unique: never 64
}

__global__ __launch_bounds__(kBuildBlock) void karras_kernel(const unsigned long long* __restrict__ keys, int N, Bvh2 b) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= N - 1) return;
    const int d = delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, N, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, N, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    const int lc = lo == gamma ? ~gamma : gamma;             // leaf if the split leaves one key on that side
    const int rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
    b.left[i] = lc; b.right[i] = rc;
    b.first[i] = lo; b.last[i] = hi;
    b.parent[lc >= 0 ? lc : (N - 1) + ~lc] = i;
    b.parent[rc >= 0 ? rc : (N - 1) + ~rc] = i;
    if (i == 0) b.parent[0] = -1;
}

using BuildInfo = DeviceBuildInfo;   // build outputs the host reads back in one copy (fs_internal.hpp)

// One workgroup, breadth-first.  wide_src[w] = binary node the wide node w spans, need[w] = pending stack entries
// when a traversal arrives at w having hit every child on the way (the bound the LDS stack is sized with).
__global__ __launch_bounds__(kCollapseBlock) void collapse_kernel(int N, Bvh2 b, NodeQ4* __restrict__ nodes,
                                                                  int* __restrict__ wide_src, int* __restrict__ need,
                                                                  BuildInfo* __restrict__ info, int max_nodes) {
    __shared__ int s_begin, s_end, s_next, s_need, s_fail;
    if (threadIdx.x == 0) {
        s_begin = 0; s_end = 1; s_next = 1; s_need = 0; s_fail = 0;
        wide_src[0] = N > 1 ? 0 : -1;     // a one-triangle scene: the root's only child is the leaf
        need[0] = 0;
        info->level_begin[0] = 0;
    }
    __syncthreads();
    int level = 0;
    while (true) {
        const int begin = s_begin, end = s_end;
        for (int w = begin + (int)threadIdx.x; w < end; w += kCollapseBlock) {
            int child[4];
            int n = 0;
            const int src = wide_src[w];
            if (src < 0) {
                child[n++] = ~0;                                   // leaf 0
            } else {
                child[n++] = b.left[src];
                child[n++] = b.right[src];
                while (n < 4) {                                    // open the inner child that spans the most triangles
                    int best = -1, bc = 2;                         // (subtrees of <= 2 triangles stay closed: leaves)
                    for (int i = 0; i < n; ++i) {
                        const int c = child[i];
                        if (c >= 0) {
                            const int cnt = b.last[c] - b.first[c] + 1;
                            if (cnt > bc) { bc = cnt; best = i; }
                        }
                    }
                    if (best < 0) break;
                    const int c = child[best];
                    child[best] = b.left[c];
                    child[n++] = b.right[c];
                }
            }
            NodeQ4 q{};
            const int my_need = need[w] + (n - 1);
            atomicMax(&s_need, my_need);
            uint32_t lo4 = 0, hi4 = 0;
            for (int c = 0; c < 4; ++c) {
                uint32_t ql = 255, qh = 0;                         // empty slot: lo > hi (refit keeps it empty)
                q.child[c] = -1;
                if (c < n) {
                    ql = 0; qh = 255;                              // placeholder box: "used"; the refit pass computes it
                    const int cn = child[c];
                    if (cn < 0) {
                        q.child[c] = ~(int32_t)((~cn) * 4 + 0);    // one triangle at sorted position ~cn
                    } else if (b.last[cn] - b.first[cn] + 1 <= 2) {
                        q.child[c] = ~(int32_t)(b.first[cn] * 4 + (b.last[cn] - b.first[cn]));
                    } else {
                        const int slot = atomicAdd(&s_next, 1);
                        if (slot < max_nodes) { wide_src[slot] = cn; need[slot] = my_need; q.child[c] = slot; }
                        else { s_fail = 1; q.child[c] = ~0; }
                    }
                }
                lo4 |= ql << (8 * c);
                hi4 |= qh << (8 * c);
            }
            q.lox = q.loy = q.loz = lo4;
            q.hix = q.hiy = q.hiz = hi4;
            nodes[w] = q;
        }
        __syncthreads();
        ++level;
        if (threadIdx.x == 0) {
            s_begin = end;
            s_end = min(s_next, max_nodes);
            if (level <= kMaxBuildLevels) info->level_begin[level] = end;
        }
        __syncthreads();
        if (s_end <= s_begin || level >= kMaxBuildLevels) break;
    }
    if (threadIdx.x == 0) {
        info->num_nodes = s_begin;
        info->levels = (s_end > s_begin || s_fail) ? -1 : level;   // -1: too deep / out of space -> host build
        info->stack_need = s_need;
    }
}

__global__ __launch_bounds__(kBuildBlock) void records_kernel(const float* __restrict__ xyz, const uint16_t* __restrict__ mat,
                                                              const uint32_t* __restrict__ object_id,
                                                              const unsigned long long* __restrict__ keys, int N,
                                                              Tri64* __restrict__ tris, uint32_t* __restrict__ leaf_pos) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= N) return;
    const uint32_t t = (uint32_t)keys[i];
    const float* p = xyz + 9 * (size_t)t;
    Tri64 r;
    const float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
    const float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
    r.a = make_float4(p[0], p[1], p[2], e1x);
    r.b = make_float4(e1y, e1z, e2x, e2y);
    const uint32_t m = mat ? (uint32_t)mat[t] : (uint32_t)FS_NO_MATERIAL;
    const uint32_t obj = object_id ? object_id[t] : t;            // default: every triangle its own actor
    r.c = make_float4(e2z, __uint_as_float(m), __uint_as_float(t), __uint_as_float(obj));
    const float nx = fmaf(e1y, e2z, -(e1z * e2y));                 // as fs_bvh.cpp / update_tris_kernel: part of the hit-normal spec
    const float ny = fmaf(e1z, e2x, -(e1x * e2z));
    const float nz = fmaf(e1x, e2y, -(e1y * e2x));
    const float l2 = nx * nx + ny * ny + nz * nz;
    const float inv = 1.0f / sqrtf(l2);
    r.d = make_float4(nx * inv, ny * inv, nz * inv, 0.f);
    tris[i] = r;
    leaf_pos[t] = (uint32_t)i;
}

}  // namespace

size_t device_build_scratch_bytes(int T) {
    size_t sort_tmp = 0;
    unsigned long long* k = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, k, k, T, 0, 62);
    const size_t n = (size_t)std::max(T, 1);
    size_t bytes = 0;
    bytes += 2 * sizeof(unsigned long long) * n;            // keys, sorted keys
    bytes += 5 * sizeof(int) * 2 * n;                       // left, right, first, last (N) + parent (2N) — rounded up
    bytes += 2 * sizeof(int) * n;                           // wide_src, need
    bytes += sizeof(BuildInfo) + 256;
    return bytes + sort_tmp + 4096;
}

// xyz / mat / object_id: device copies of the inputs (object_id may be null).  nodes [>= max(T-1, 1)], tris [T],
// leaf_pos [T]: outputs.  scratch: device_build_scratch_bytes(T).  info_host: pinned or pageable, filled after the
// stream has been synchronised by the caller.  Returns false if T < 1.
bool launch_device_build(const float* xyz, const uint16_t* mat, const uint32_t* object_id, int T, const float lo[3],
                         const float hi[3], NodeQ4* nodes, Tri64* tris, uint32_t* leaf_pos, void* scratch, size_t scratch_bytes,
                         DeviceBuildInfo* info_dev, hipStream_t s) {
    if (T < 1) return false;
    const size_t n = (size_t)T;
    char* p = static_cast<char*>(scratch);
    auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~(size_t)255; return r; };
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * n));
    unsigned long long* sorted = reinterpret_cast<unsigned long long*>(take(sizeof(unsigned long long) * n));
    Bvh2 b;
    b.left = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.right = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.first = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.last = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.parent = reinterpret_cast<int*>(take(sizeof(int) * 2 * n));
    int* wide_src = reinterpret_cast<int*>(take(sizeof(int) * n));
    int* need = reinterpret_cast<int*>(take(sizeof(int) * n));
    size_t sort_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, keys, sorted, T, 0, 62);
    void* sort_buf = take(sort_tmp);
    if ((size_t)(p - static_cast<char*>(scratch)) > scratch_bytes) return false;

    float3 l3 = make_float3(lo[0], lo[1], lo[2]);
    float3 ie = make_float3(1.0f / fmaxf(hi[0] - lo[0], 1e-20f), 1.0f / fmaxf(hi[1] - lo[1], 1e-20f), 1.0f / fmaxf(hi[2] - lo[2], 1e-20f));
    const unsigned blocks = (unsigned)((T + kBuildBlock - 1) / kBuildBlock);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, xyz, T, l3, ie, keys);
    (void)hipcub::DeviceRadixSort::SortKeys(sort_buf, sort_tmp, keys, sorted, T, 0, 62, s);
    if (T > 1) hipLaunchKernelGGL(karras_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, sorted, T, b);
    hipLaunchKernelGGL(collapse_kernel, dim3(1), dim3(kCollapseBlock), 0, s, T, b, nodes, wide_src, need,
                       reinterpret_cast<BuildInfo*>(info_dev), std::max(T - 1, 1));
    hipLaunchKernelGGL(records_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, xyz, mat, object_id, sorted, T, tris, leaf_pos);
    return true;
}

}  // namespace fs
