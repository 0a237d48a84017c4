// fs_bvh.cpp — host-side BVH2 construction for the HIP traversal kernels.
//
// Replaces the acceleration structure behind the engine call the reference delegates to
// (UWorld::LineTraceSingleByObjectType, call sites AudioRayTracingSubsystem.cpp:252-254, 340-342).
// Top-down binned SAH over triangle centroids, <= 4 triangles per leaf, tree depth capped at
// kStackDepth so the per-lane LDS stack of the kernels can never overflow.  Flattened breadth-first
// into 64-byte nodes that carry both children's boxes (one fetch per traversal step); the top of the
// tree is therefore a contiguous prefix of the node array (staged into LDS by the kernels).
//
// Boxes are padded far beyond the float error of the slab and triangle tests, so the closest hit
// found through the BVH equals the brute-force closest hit: results do not depend on the tree.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>

#include "fs_internal.hpp"

namespace fs {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::numeric_limits<float>::infinity();
            hi[k] = -std::numeric_limits<float>::infinity();
        }
    }
    void grow(const Box& o) {
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], o.lo[k]);
            hi[k] = std::max(hi[k], o.hi[k]);
        }
    }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BuildNode {
    Box box;
    int left = -1, right = -1;  // children (build-node indices) or -1
    int first = 0, count = 0;   // range in the permuted primitive order
    int depth = 0;
};

struct Builder {
    std::vector<Box> pbox;       // per input triangle
    std::vector<float> cen;      // centroids [T][3]
    std::vector<int> order;      // permutation of triangle indices
    std::vector<BuildNode> nodes;
    int max_depth = 0;

    static constexpr int kBins = 32;
    static constexpr int kLeaf = 4;

    int make(int first, int count, int depth) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        BuildNode n;
        n.first = first; n.count = count; n.depth = depth;
        n.box.reset();
        Box cb; cb.reset();
        for (int i = first; i < first + count; ++i) {
            int t = order[i];
            n.box.grow(pbox[t]);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], cen[3 * t + k]);
                cb.hi[k] = std::max(cb.hi[k], cen[3 * t + k]);
            }
        }
        max_depth = std::max(max_depth, depth);
        if (count > kLeaf) {
            int mid = -1;
            // levels still available below this node; a balanced split needs ceil(log2(count/4)) of them
            int need = 0;
            for (int c = count; c > kLeaf; c = (c + 1) / 2) ++need;
            bool force_median = depth + need + 1 >= kStackDepth;
            if (!force_median) mid = sah_split(first, count, cb);
            if (mid < 0) mid = median_split(first, count, cb);
            int l = make(first, mid - first, depth + 1);
            int r = make(mid, first + count - mid, depth + 1);
            n.left = l; n.right = r;
        }
        nodes[id] = n;
        return id;
    }

    int sah_split(int first, int count, const Box& cb) {
        int best_axis = -1, best_bin = -1;
        float best = std::numeric_limits<float>::infinity();
        for (int ax = 0; ax < 3; ++ax) {
            float ext = cb.hi[ax] - cb.lo[ax];
            if (!(ext > 0.f)) continue;
            Box bb[kBins]; int bc[kBins];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
            float scale = kBins / ext;
            for (int i = first; i < first + count; ++i) {
                int t = order[i];
                int b = std::min(kBins - 1, std::max(0, (int)((cen[3 * t + ax] - cb.lo[ax]) * scale)));
                bb[b].grow(pbox[t]); bc[b]++;
            }
            float ra[kBins]; int rc[kBins];
            Box acc; acc.reset(); int c = 0;
            for (int b = kBins - 1; b >= 1; --b) {
                acc.grow(bb[b]); c += bc[b];
                ra[b] = c ? acc.half_area() : 0.f; rc[b] = c;
            }
            acc.reset(); c = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]); c += bc[b];
                if (c == 0 || rc[b + 1] == 0) continue;
                float cost = acc.half_area() * c + ra[b + 1] * rc[b + 1];
                if (cost < best) { best = cost; best_axis = ax; best_bin = b; }
            }
        }
        if (best_axis < 0) return -1;
        float ext = cb.hi[best_axis] - cb.lo[best_axis];
        float scale = kBins / ext;
        auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int t) {
            int b = std::min(kBins - 1, std::max(0, (int)((cen[3 * t + best_axis] - cb.lo[best_axis]) * scale)));
            return b <= best_bin;
        });
        int mid = (int)(it - order.begin());
        if (mid == first || mid == first + count) return -1;
        // keep the tree shallow enough: reject hopelessly lopsided splits near the depth cap
        return mid;
    }

    int median_split(int first, int count, const Box& cb) {
        int ax = 0;
        float e = -1.f;
        for (int k = 0; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > e) { e = cb.hi[k] - cb.lo[k]; ax = k; }
        int mid = first + count / 2;
        std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                         [&](int a, int b) {
                             float ca = cen[3 * a + ax], cb2 = cen[3 * b + ax];
                             return ca < cb2 || (ca == cb2 && a < b);
                         });
        return mid;
    }
};

inline int32_t leaf_code(int first, int count) { return ~(int32_t)(first * 4 + (count - 1)); }

}  // namespace

void build_bvh(const float* xyz, const uint16_t* mat, int32_t T, HostBVH& out) {
    out.nodes.clear();
    out.tris.clear();
    out.max_depth = 0;
    if (T <= 0) return;

    Builder b;
    b.pbox.resize(T);
    b.cen.resize(3 * (size_t)T);
    b.order.resize(T);
    float amax = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* p = xyz + 9 * (size_t)t;
        Box bx; bx.reset();
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                bx.lo[k] = std::min(bx.lo[k], p[3 * v + k]);
                bx.hi[k] = std::max(bx.hi[k], p[3 * v + k]);
                amax = std::max(amax, std::fabs(p[3 * v + k]));
            }
        b.pbox[t] = bx;
        for (int k = 0; k < 3; ++k) b.cen[3 * t + k] = 0.5f * (bx.lo[k] + bx.hi[k]);
        b.order[t] = t;
    }
    // conservative padding (cm): >> float error of the tests at this coordinate magnitude
    const float pad = std::max(0.01f, amax * 3.8146973e-06f);

    b.nodes.reserve(2 * (size_t)T / 3 + 16);
    int root = b.make(0, T, 0);
    out.max_depth = b.max_depth;

    // triangles in leaf order
    out.tris.resize(T);
    for (int i = 0; i < T; ++i) {
        int t = b.order[i];
        const float* p = xyz + 9 * (size_t)t;
        Tri64 r;
        const float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
        const float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
        r.a = make_float4(p[0], p[1], p[2], e1x);
        r.b = make_float4(e1y, e1z, e2x, e2y);
        uint32_t m = mat ? (uint32_t)mat[t] : (uint32_t)FS_NO_MATERIAL;
        float mf, idf;
        uint32_t id = (uint32_t)t;
        std::memcpy(&mf, &m, 4);
        std::memcpy(&idf, &id, 4);
        r.c = make_float4(e2z, mf, idf, 0.f);
        // unit geometric normal, fixed operation order (part of the hit-normal spec; built with
        // -ffp-contract=off, so this is the same fp32 sequence the oracle evaluates at hit time)
        float nx = std::fmaf(e1y, e2z, -(e1z * e2y));
        float ny = std::fmaf(e1z, e2x, -(e1x * e2z));
        float nz = std::fmaf(e1x, e2y, -(e1y * e2x));
        float l2 = nx * nx + ny * ny + nz * nz;
        float inv = 1.0f / std::sqrt(l2);
        r.d = make_float4(nx * inv, ny * inv, nz * inv, 0.f);
        out.tris[i] = r;
    }

    // flatten inner nodes breadth-first; a leaf root becomes an inner node with one empty child
    auto child_box = [&](const BuildNode& n, float lo[3], float hi[3]) {
        for (int k = 0; k < 3; ++k) { lo[k] = n.box.lo[k] - pad; hi[k] = n.box.hi[k] + pad; }
    };
    const float inf = std::numeric_limits<float>::infinity();
    std::vector<int> flat_of(b.nodes.size(), -1);
    std::vector<int> bfs;
    if (b.nodes[root].left < 0) {
        Node64 n{};
        float lo[3], hi[3];
        child_box(b.nodes[root], lo, hi);
        // empty second child: a degenerate far-away box no ray segment can reach (an inverted infinite
        // box would pass the slab test: min/max of +-inf)
        const float far = 3.0e38f;
        (void)inf;
        n.q0 = make_float4(lo[0], lo[1], lo[2], hi[0]);
        n.q1 = make_float4(hi[1], hi[2], far, far);
        n.q2 = make_float4(far, far, far, far);
        n.c0 = leaf_code(b.nodes[root].first, b.nodes[root].count);
        n.c1 = -1;
        out.nodes.push_back(n);
        return;
    }
    bfs.push_back(root);
    flat_of[root] = 0;
    for (size_t h = 0; h < bfs.size(); ++h) {
        const BuildNode& n = b.nodes[bfs[h]];
        for (int c : {n.left, n.right})
            if (b.nodes[c].left >= 0) { flat_of[c] = (int)bfs.size(); bfs.push_back(c); }
    }
    out.nodes.resize(bfs.size());
    for (size_t h = 0; h < bfs.size(); ++h) {
        const BuildNode& n = b.nodes[bfs[h]];
        const BuildNode& l = b.nodes[n.left];
        const BuildNode& r = b.nodes[n.right];
        float l0[3], h0[3], l1[3], h1[3];
        child_box(l, l0, h0);
        child_box(r, l1, h1);
        Node64 o{};
        o.q0 = make_float4(l0[0], l0[1], l0[2], h0[0]);
        o.q1 = make_float4(h0[1], h0[2], l1[0], l1[1]);
        o.q2 = make_float4(l1[2], h1[0], h1[1], h1[2]);
        o.c0 = l.left >= 0 ? flat_of[n.left] : leaf_code(l.first, l.count);
        o.c1 = r.left >= 0 ? flat_of[n.right] : leaf_code(r.first, r.count);
        out.nodes[h] = o;
    }
}

}  // namespace fs
