// fs_build.hip — the acceleration structure built ON THE DEVICE, for topology changes.
//
// Actors register and unregister at run time (RegisterGeometry / UnregisterGeometry, AudioRayTracingSubsystem.h:99-100)
// and the reference's traces see the new physics scene in the next frame.  Moving geometry is refitted (fs_refit.hip);
// a changed triangle SET needs a new tree, and the host SAH build takes 18 ms for 100 000 triangles — forty frames.
// This builder produces the same data structures in about two milliseconds:
//   1. morton_kernel     30-bit Morton code of every triangle's centroid (10 bits per axis over the scene bounds)
//                        with the triangle index below it: unique 64-bit keys
//   2. radix sort        hipcub::DeviceRadixSort (rocPRIM) on the keys
//   3. PLOC              parallel locally-ordered clustering (Meister & Bittner 2018) over the sorted triangles: every
//                        cluster looks at its kPlocRadius neighbours on either side in the (Morton) order for the one
//                        whose union with it has the smallest surface area; mutual nearest neighbours merge into a new
//                        binary node; the survivors are compacted (prefix sum) and the round repeats until one cluster
//                        is left (~30 rounds).  Bottom-up agglomeration by surface area gives a tree close to the host's
//                        binned-SAH one — round 2's Karras radix tree (spatial-median splits) traced 1.6x slower.
//                        (FS_BUILD_LBVH=1 still selects it, for comparison.)
//   4. collapse_kernel   4-wide nodes level by level, breadth-first (ONE workgroup walks the levels: the tree has
//                        ~N/4 wide nodes): a wide node opens the inner child of largest surface area until it has four;
//                        subtrees of <= 2 triangles become leaves; every subtree owns a contiguous range of leaf-order
//                        positions (start of the node + the sizes of its left siblings), which also fixes the order of
//                        the triangle records; records every level's node range and the worst-case stack need
//   5. records_kernel    leaf-order triangle records {v0, e1, e2, material, index, actor, unit normal} — the same fp32
//                        sequence as the host build — and the input-index -> leaf-position table
//   6. the existing refit pass (fs_refit.hip) derives every node's quantised child boxes bottom-up, exactly as the
//      host builder quantises (outwards, padded): boxes stay conservative, so closest hits are again a function of
//      ray and triangles only — bit-identical to the host-built tree's and to the brute-force oracle's.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

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

// binary tree over the sorted triangles: internal nodes [0, N-1), leaves are sorted positions; child >= 0 internal,
// ~leaf otherwise.  count = triangles below an internal node, area = surface area of its box (PLOC; 0 for the radix tree)
struct Bvh2 {
    int* left; int* right; int* parent;   // (radix tree only) parent of internal node i: parent[i]; of leaf j: parent[(N-1) + j]
    int* first; int* last;                // (radix tree only) key range of an internal node
    int* count; float* area;
    int root;
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
    b.count[i] = hi - lo + 1; b.area[i] = (float)(hi - lo + 1);   // no boxes here: "area" = size, the round-2 opening rule
    b.parent[lc >= 0 ? lc : (N - 1) + ~lc] = i;
    b.parent[rc >= 0 ? rc : (N - 1) + ~rc] = i;
    if (i == 0) b.parent[0] = -1;
}

// ---- PLOC ------------------------------------------------------------------------------------------------------
// Cluster c of a round is a node: id < N a triangle (sorted position id), else internal node id - N.  Boxes by node id.
constexpr int kPlocRadius = 10;
struct Ploc {
    float4* bmin; float4* bmax;    // [2N] node boxes (xyz)
    int* cluster[2];               // [N] the round's clusters, ping-pong
    int* nn; int* valid; int* pos; int* merged;   // [N] nearest neighbour, survives, its place in the next round, its node next round
    int* n;                        // [2] device-side cluster counts, ping-pong
    int* next_node;                // internal nodes made so far
};
__device__ __forceinline__ float half_area(float4 lo, float4 hi) {
    const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
    return dx * dy + dy * dz + dz * dx;
}
__global__ __launch_bounds__(kBuildBlock) void ploc_init_kernel(const float* __restrict__ xyz, const unsigned long long* __restrict__ keys,
                                                                int N, Ploc P) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i == 0) { P.n[0] = N; P.n[1] = 0; *P.next_node = 0; }
    if (i >= N) return;
    const float* p = xyz + 9 * (size_t)(uint32_t)keys[i];
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = fminf(fminf(p[k], p[3 + k]), p[6 + k]);
        hi[k] = fmaxf(fmaxf(p[k], p[3 + k]), p[6 + k]);
    }
    P.bmin[i] = make_float4(lo[0], lo[1], lo[2], 0.f);
    P.bmax[i] = make_float4(hi[0], hi[1], hi[2], 0.f);
    P.cluster[0][i] = i;
}
// nearest neighbour by merged surface area within kPlocRadius places on either side
__global__ __launch_bounds__(kBuildBlock) void ploc_nn_kernel(Ploc P, int cur) {
    const int n = P.n[cur];
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= n) return;
    const int* cl = P.cluster[cur];
    const float4 lo = P.bmin[cl[i]], hi = P.bmax[cl[i]];
    float best = INFINITY;
    int bj = -1;
    for (int j = max(0, i - kPlocRadius); j <= min(n - 1, i + kPlocRadius); ++j) {
        if (j == i) continue;
        const float4 l2 = P.bmin[cl[j]], h2 = P.bmax[cl[j]];
        const float a = half_area(make_float4(fminf(lo.x, l2.x), fminf(lo.y, l2.y), fminf(lo.z, l2.z), 0.f),
                                  make_float4(fmaxf(hi.x, h2.x), fmaxf(hi.y, h2.y), fmaxf(hi.z, h2.z), 0.f));
        if (a < best) { best = a; bj = j; }
    }
    P.nn[i] = bj;
}
// mutual nearest neighbours merge (the lower place makes the node and survives)
__global__ __launch_bounds__(kBuildBlock) void ploc_merge_kernel(Ploc P, int cur, int N, Bvh2 b, int upper) {
    const int n = P.n[cur];
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= upper) return;
    if (i >= n) { P.valid[i] = 0; return; }
    const int* cl = P.cluster[cur];
    const int j = P.nn[i];
    int node = cl[i], keep = 1;
    if (j >= 0 && P.nn[j] == i) {
        if (i < j) {
            const int id = atomicAdd(P.next_node, 1);     // internal node id (tree index), node id N + id
            const int a = cl[i], c = cl[j];
            const float4 la = P.bmin[a], ha = P.bmax[a], lc = P.bmin[c], hc = P.bmax[c];
            const float4 lo = make_float4(fminf(la.x, lc.x), fminf(la.y, lc.y), fminf(la.z, lc.z), 0.f);
            const float4 hi = make_float4(fmaxf(ha.x, hc.x), fmaxf(ha.y, hc.y), fmaxf(ha.z, hc.z), 0.f);
            P.bmin[N + id] = lo; P.bmax[N + id] = hi;
            b.left[id] = a < N ? ~a : a - N;
            b.right[id] = c < N ? ~c : c - N;
            b.count[id] = (a < N ? 1 : b.count[a - N]) + (c < N ? 1 : b.count[c - N]);
            b.area[id] = half_area(lo, hi);
            node = N + id;
        } else {
            keep = 0;
        }
    }
    P.merged[i] = node;
    P.valid[i] = keep;
}
__global__ __launch_bounds__(kBuildBlock) void ploc_compact_kernel(Ploc P, int cur, int upper) {
    const int n = P.n[cur];
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= upper) return;
    if (i < n && P.valid[i]) P.cluster[cur ^ 1][P.pos[i]] = P.merged[i];
    if (i == upper - 1) P.n[cur ^ 1] = P.pos[i] + P.valid[i];
}
// The last rounds in ONE workgroup: once kCollapseBlock clusters or fewer are left a round is three short passes with
// workgroup barriers between them instead of four launches (the thirty rounds of a 5 000-triangle scene cost 1.3 ms as
// launches, most of them on a few hundred clusters).  Ends with the root.
__global__ __launch_bounds__(kCollapseBlock) void ploc_tail_kernel(Ploc P, int cur, int N, Bvh2 b, int* root_out) {
    __shared__ int s_cl[kCollapseBlock], s_nn[kCollapseBlock], s_n;
    typedef hipcub::BlockScan<int, kCollapseBlock> Scan;
    __shared__ typename Scan::TempStorage s_scan;
    const int i = (int)threadIdx.x;
    int n = P.n[cur];
    if (i < n) s_cl[i] = P.cluster[cur][i];
    __syncthreads();
    while (n > 1) {
        int bj = -1;
        if (i < n) {
            const float4 lo = P.bmin[s_cl[i]], hi = P.bmax[s_cl[i]];
            float best = INFINITY;
            for (int j = max(0, i - kPlocRadius); j <= min(n - 1, i + kPlocRadius); ++j) {
                if (j == i) continue;
                const float4 l2 = P.bmin[s_cl[j]], h2 = P.bmax[s_cl[j]];
                const float a = half_area(make_float4(fminf(lo.x, l2.x), fminf(lo.y, l2.y), fminf(lo.z, l2.z), 0.f),
                                          make_float4(fmaxf(hi.x, h2.x), fmaxf(hi.y, h2.y), fmaxf(hi.z, h2.z), 0.f));
                if (a < best) { best = a; bj = j; }
            }
            s_nn[i] = bj;
        }
        __syncthreads();
        int node = i < n ? s_cl[i] : 0, keep = i < n ? 1 : 0;
        if (i < n && bj >= 0 && s_nn[bj] == i) {
            if (i < bj) {
                const int id = atomicAdd(P.next_node, 1);
                const int a = s_cl[i], c = s_cl[bj];
                const float4 la = P.bmin[a], ha = P.bmax[a], lc = P.bmin[c], hc = P.bmax[c];
                const float4 lo = make_float4(fminf(la.x, lc.x), fminf(la.y, lc.y), fminf(la.z, lc.z), 0.f);
                const float4 hi = make_float4(fmaxf(ha.x, hc.x), fmaxf(ha.y, hc.y), fmaxf(ha.z, hc.z), 0.f);
                P.bmin[N + id] = lo; P.bmax[N + id] = hi;
                b.left[id] = a < N ? ~a : a - N;
                b.right[id] = c < N ? ~c : c - N;
                b.count[id] = (a < N ? 1 : b.count[a - N]) + (c < N ? 1 : b.count[c - N]);
                b.area[id] = half_area(lo, hi);
                node = N + id;
            } else {
                keep = 0;
            }
        }
        int pos = 0, total = 0;
        Scan(s_scan).ExclusiveSum(keep, pos, total);
        __syncthreads();                       // everyone has read s_cl / s_nn of this round (and the scan storage is free again)
        if (keep) s_cl[pos] = node;
        if (i == 0) s_n = total;
        __threadfence_block();                 // the new nodes' boxes and counts are read by other threads next round
        __syncthreads();
        n = s_n;
    }
    if (i == 0) { const int c = s_cl[0]; *root_out = c < N ? -1 : c - N; }      // (N == 1: no internal node)
}

using BuildInfo = DeviceBuildInfo;   // build outputs the host reads back in one copy (fs_internal.hpp)

// One workgroup, breadth-first.  wide_src[w] = binary node the wide node w spans, wide_start[w] = first leaf-order
// position of its triangles, need[w] = pending stack entries when a traversal arrives at w having hit every child on
// the way (the bound the LDS stack is sized with).  pos_of[j] = leaf-order position of sorted triangle j.
__global__ __launch_bounds__(kCollapseBlock) void collapse_kernel(int N, Bvh2 b, const int* __restrict__ root_ptr,
                                                                  NodeQ4* __restrict__ nodes, int* __restrict__ wide_src,
                                                                  int* __restrict__ wide_start, int* __restrict__ need,
                                                                  int* __restrict__ pos_of, BuildInfo* __restrict__ info,
                                                                  int max_nodes) {
    __shared__ int s_begin, s_end, s_next, s_need, s_fail;
    if (threadIdx.x == 0) {
        s_begin = 0; s_end = 1; s_next = 1; s_need = 0; s_fail = 0;
        wide_src[0] = N > 1 ? (root_ptr ? *root_ptr : b.root) : -1;     // a one-triangle scene: the root's only child is the leaf
        wide_start[0] = 0;
        need[0] = 0;
        info->level_begin[0] = 0;
    }
    __syncthreads();
    auto count_of = [&](int c) { return c < 0 ? 1 : b.count[c]; };
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
                while (n < 4) {                                    // open the inner child of largest surface area
                    int best = -1;                                 // (subtrees of <= 2 triangles stay closed: leaves)
                    float ba = -1.0f;
                    for (int i = 0; i < n; ++i) {
                        const int c = child[i];
                        if (c >= 0 && b.count[c] > 2 && b.area[c] > ba) { ba = b.area[c]; best = i; }
                    }
                    if (best < 0) break;
                    // keep the children in tree order (left before right): positions stay contiguous per subtree
                    const int c = child[best];
                    for (int i = n; i > best + 1; --i) child[i] = child[i - 1];
                    child[best] = b.left[c];
                    child[best + 1] = b.right[c];
                    ++n;
                }
            }
            NodeQ4 q{};
            const int my_need = need[w] + (n - 1);
            atomicMax(&s_need, my_need);
            uint32_t lo4 = 0, hi4 = 0;
            int start = wide_start[w];
            for (int c = 0; c < 4; ++c) {
                uint32_t ql = 255, qh = 0;                         // empty slot: lo > hi (refit keeps it empty)
                q.child[c] = -1;
                if (c < n) {
                    ql = 0; qh = 255;                              // placeholder box: "used"; the refit pass computes it
                    const int cn = child[c];
                    const int cnt = count_of(cn);
                    if (cn < 0) {
                        q.child[c] = ~(int32_t)(start * 4 + 0);    // one triangle: sorted triangle ~cn at position start
                        pos_of[~cn] = start;
                    } else if (cnt <= 2) {                         // an inner node over two triangles
                        q.child[c] = ~(int32_t)(start * 4 + 1);
                        pos_of[~b.left[cn]] = start;
                        pos_of[~b.right[cn]] = start + 1;
                    } else {
                        const int slot = atomicAdd(&s_next, 1);
                        if (slot < max_nodes) { wide_src[slot] = cn; wide_start[slot] = start; need[slot] = my_need; q.child[c] = slot; }
                        else { s_fail = 1; q.child[c] = ~0; }
                    }
                    start += cnt;
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
                                                              const int* __restrict__ pos_of, Tri64* __restrict__ tris,
                                                              uint32_t* __restrict__ leaf_pos) {
    const int i = blockIdx.x * kBuildBlock + threadIdx.x;
    if (i >= N) return;
    const uint32_t t = (uint32_t)keys[i];
    const int at = pos_of[i];                                      // leaf-order position of sorted triangle i
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
    tris[at] = r;
    leaf_pos[t] = (uint32_t)at;
}

}  // namespace

size_t device_build_scratch_bytes(int T) {
    size_t sort_tmp = 0, scan_tmp = 0;
    unsigned long long* k = nullptr;
    int* q = nullptr;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, k, k, T, 0, 62);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, q, q, T);
    const size_t n = (size_t)std::max(T, 1);
    size_t bytes = 0;
    bytes += 2 * sizeof(unsigned long long) * n;            // keys, sorted keys
    bytes += 6 * sizeof(int) * n + sizeof(int) * 2 * n;     // left, right, first, last, count, area (N) + parent (2N)
    bytes += 4 * sizeof(int) * n;                           // wide_src, wide_start, need, pos_of
    bytes += 2 * sizeof(float4) * 2 * n;                    // PLOC node boxes
    bytes += 6 * sizeof(int) * n;                           // PLOC clusters (2), nn, valid, pos, merged
    bytes += sizeof(BuildInfo) + 1024;
    return bytes + sort_tmp + scan_tmp + 64 * 256;          // + alignment of every piece
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
    Bvh2 b{};
    b.left = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.right = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.first = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.last = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.count = reinterpret_cast<int*>(take(sizeof(int) * n));
    b.area = reinterpret_cast<float*>(take(sizeof(float) * n));
    b.parent = reinterpret_cast<int*>(take(sizeof(int) * 2 * n));
    b.root = 0;
    int* wide_src = reinterpret_cast<int*>(take(sizeof(int) * n));
    int* wide_start = reinterpret_cast<int*>(take(sizeof(int) * n));
    int* need = reinterpret_cast<int*>(take(sizeof(int) * n));
    int* pos_of = reinterpret_cast<int*>(take(sizeof(int) * n));
    Ploc P{};
    P.bmin = reinterpret_cast<float4*>(take(sizeof(float4) * 2 * n));
    P.bmax = reinterpret_cast<float4*>(take(sizeof(float4) * 2 * n));
    P.cluster[0] = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.cluster[1] = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.nn = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.valid = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.pos = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.merged = reinterpret_cast<int*>(take(sizeof(int) * n));
    P.n = reinterpret_cast<int*>(take(sizeof(int) * 4));
    P.next_node = P.n + 2;
    int* root_dev = P.n + 3;
    size_t sort_tmp = 0, scan_tmp = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_tmp, keys, sorted, T, 0, 62);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, P.valid, P.pos, T);
    void* sort_buf = take(sort_tmp);
    void* scan_buf = take(scan_tmp);
    if ((size_t)(p - static_cast<char*>(scratch)) > scratch_bytes) return false;

    float3 l3 = make_float3(lo[0], lo[1], lo[2]);
    float3 ie = make_float3(1.0f / fmaxf(hi[0] - lo[0], 1e-20f), 1.0f / fmaxf(hi[1] - lo[1], 1e-20f), 1.0f / fmaxf(hi[2] - lo[2], 1e-20f));
    const unsigned blocks = (unsigned)((T + kBuildBlock - 1) / kBuildBlock);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, xyz, T, l3, ie, keys);
    (void)hipcub::DeviceRadixSort::SortKeys(sort_buf, sort_tmp, keys, sorted, T, 0, 62, s);
    static const bool lbvh = [] { const char* v = std::getenv("FS_BUILD_LBVH"); return v && std::atoi(v) != 0; }();
    const int* root_ptr = nullptr;
    if (T > 1 && lbvh) {
        hipLaunchKernelGGL(karras_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, sorted, T, b);
    } else if (T > 1) {
        // PLOC rounds.  The cluster count lives on the device; the host only knows an upper bound (every round with more
        // than one cluster merges at least the globally closest pair) and reads the true count back every few rounds.
        hipLaunchKernelGGL(ploc_init_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, xyz, sorted, T, P);
        int upper = T, cur = 0;
        for (int round = 0; round < 4096 && upper > kCollapseBlock; ++round) {
            const unsigned gb = (unsigned)((upper + kBuildBlock - 1) / kBuildBlock);
            hipLaunchKernelGGL(ploc_nn_kernel, dim3(gb), dim3(kBuildBlock), 0, s, P, cur);
            hipLaunchKernelGGL(ploc_merge_kernel, dim3(gb), dim3(kBuildBlock), 0, s, P, cur, T, b, upper);
            (void)hipcub::DeviceScan::ExclusiveSum(scan_buf, scan_tmp, P.valid, P.pos, upper, s);
            hipLaunchKernelGGL(ploc_compact_kernel, dim3(gb), dim3(kBuildBlock), 0, s, P, cur, upper);
            cur ^= 1;
            upper -= 1;
            if ((round & 3) == 3) {   // the true count: rounds usually merge a third of the clusters
                int n_now = 0;
                if (hipMemcpyAsync(&n_now, P.n + cur, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
                    hipStreamSynchronize(s) != hipSuccess) return false;
                upper = std::max(1, std::min(upper, n_now));
            }
        }
        hipLaunchKernelGGL(ploc_tail_kernel, dim3(1), dim3(kCollapseBlock), 0, s, P, cur, T, b, root_dev);   // the remaining rounds
        root_ptr = root_dev;
    }
    hipLaunchKernelGGL(collapse_kernel, dim3(1), dim3(kCollapseBlock), 0, s, T, b, root_ptr, nodes, wide_src, wide_start, need, pos_of,
                       reinterpret_cast<BuildInfo*>(info_dev), std::max(T - 1, 1));
    hipLaunchKernelGGL(records_kernel, dim3(blocks), dim3(kBuildBlock), 0, s, xyz, mat, object_id, sorted, T, pos_of, tris, leaf_pos);
    return true;
}

}  // namespace fs
