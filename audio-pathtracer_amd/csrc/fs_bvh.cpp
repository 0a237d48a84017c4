// fs_bvh.cpp — host-side acceleration-structure build for the HIP traversal kernels.
//
// Replaces the structure behind the engine call the reference delegates to
// (UWorld::LineTraceSingleByObjectType, call sites AudioRayTracingSubsystem.cpp:252-254, 340-342).
//
//   1. top-down binned SAH BVH2 over triangle centroids, <= 2 triangles per leaf (a triangle test costs about
//      as much as 2.5 child boxes on this kernel, so small leaves win: DESIGN.md section 5);
//   2. collapse to a 4-wide tree: the set of BVH2 nodes that become 4-wide nodes is chosen by dynamic programming so
//      that their summed surface area — the expected number of node visits — is least (CollapsePlan);
//   3. flatten breadth-first into 64-byte nodes: child boxes quantised to 8 bits per plane on a
//      per-node power-of-two grid (rounded outwards), so one node = 4 child boxes = 4 x 16-byte loads.
//
// Leaf boxes are padded far beyond the float error of the slab and triangle tests and quantisation only
// grows boxes, so the closest hit found through the tree equals the brute-force closest hit: results do
// not depend on the tree.  The builder reports the traversal stack the tree needs (worst-case number of pending
// entries along any root-to-leaf path; the kernels size their LDS stack and its deep store from it) and rebuilds the
// BVH2 shallower only if that exceeds kStackDepth.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>

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

// One record per triangle, permuted in place by the splits, so every pass over a node streams through
// contiguous memory (gathering boxes through an index array made the build cache-bound and kept worker
// threads from scaling).  Worker threads own disjoint ranges.
struct Prim {
    Box box;
    float cen[3];
    int idx;      // input triangle
};

struct BuildInput {
    std::vector<Prim> prims;
};

struct Builder {
    BuildInput& in;
    std::vector<Prim>& prims;
    std::vector<BuildNode> nodes;
    int max_depth = 0;
    int depth_cap = 48;
    // parallel build: nodes with at most `defer_below` triangles (below the root) are left unsplit and listed in
    // `deferred`; worker threads build their subtrees with private builders, spliced in afterwards
    int defer_below = 0;
    std::vector<int> deferred;
    const std::atomic<bool>* cancel = nullptr;   // set: the result is no longer wanted — stop splitting (the caller discards the tree)

    explicit Builder(BuildInput& i) : in(i), prims(i.prims) {}

    static constexpr int kBinsMax = 256;
    int kBins = std::getenv("FS_BVH_BINS") ? std::max(2, std::min(kBinsMax, std::atoi(std::getenv("FS_BVH_BINS")))) : 64;   // 32 -> 64 bins: -0.8 % node visits (tools/tree_cost.cpp), +0.6 % rays/s (profiles/r03_ab_tree.log)
    int kLeaf = 2;   // measured: a triangle test costs about as much as 2.5 child boxes, small leaves win

    int make(int first, int count, int depth) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        BuildNode n;
        n.first = first; n.count = count; n.depth = depth;
        n.box.reset();
        Box cb; cb.reset();
        for (int i = first; i < first + count; ++i) {
            const Prim& p = prims[(size_t)i];
            n.box.grow(p.box);
            for (int k = 0; k < 3; ++k) {
                cb.lo[k] = std::min(cb.lo[k], p.cen[k]);
                cb.hi[k] = std::max(cb.hi[k], p.cen[k]);
            }
        }
        max_depth = std::max(max_depth, depth);
        if (count > kLeaf && defer_below > 0 && depth > 0 && count <= defer_below) {
            deferred.push_back(id);
            nodes[id] = n;
            return id;
        }
        if (count > 2048 && cancel && cancel->load(std::memory_order_relaxed)) { nodes[id] = n; return id; }
        if (count > kLeaf) {
            int mid = -1;
            // levels a balanced split still needs below this node
            int need = 0;
            for (int c = count; c > kLeaf; c = (c + 1) / 2) ++need;
            bool force_median = depth + need + 1 >= depth_cap;
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
            Box bb[kBinsMax]; int bc[kBinsMax];
            for (int b = 0; b < kBins; ++b) { bb[b].reset(); bc[b] = 0; }
            float scale = kBins / ext;
            for (int i = first; i < first + count; ++i) {
                const Prim& p = prims[(size_t)i];
                int b = std::min(kBins - 1, std::max(0, (int)((p.cen[ax] - cb.lo[ax]) * scale)));
                bb[b].grow(p.box); bc[b]++;
            }
            float ra[kBinsMax]; int rc[kBinsMax];
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
        const float lo = cb.lo[best_axis];
        auto it = std::partition(prims.begin() + first, prims.begin() + first + count, [&](const Prim& p) {
            int b = std::min(kBins - 1, std::max(0, (int)((p.cen[best_axis] - lo) * scale)));
            return b <= best_bin;
        });
        int mid = (int)(it - prims.begin());
        if (mid == first || mid == first + count) return -1;
        return mid;
    }

    int median_split(int first, int count, const Box& cb) {
        int ax = 0;
        float e = -1.f;
        for (int k = 0; k < 3; ++k)
            if (cb.hi[k] - cb.lo[k] > e) { e = cb.hi[k] - cb.lo[k]; ax = k; }
        int mid = first + count / 2;
        std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                         [&](const Prim& a, const Prim& b) {
                             return a.cen[ax] < b.cen[ax] || (a.cen[ax] == b.cen[ax] && a.idx < b.idx);
                         });
        return mid;
    }
};

inline int32_t leaf_code(int first, int count) { return ~(int32_t)(first * 4 + (count - 1)); }

// 4-wide node before quantisation: children are build-node indices
struct Wide {
    int src;            // build node this wide node spans
    int child[4];
    int n;
};

// Which BVH2 nodes become 4-wide nodes.  Every 4-wide node visit costs the kernel the same (four boxes are tested
// whatever the slots hold) and the leaves are given, so the expected cost of a collapse is the summed surface area of
// the BVH2 nodes that end up as 4-wide nodes.  That has an exact minimum by dynamic programming over the BVH2
// (as the wide-BVH construction of Ylitie, Karras, Laine 2017 does for 8-wide nodes):
//   F(n, k) = least cost of covering n's subtree with at most k child slots
//           = min( slot(n), min_i F(left, i) + F(right, k - i) ),   slot(leaf) = 0, slot(inner) = W(inner)
//   W(n)    = area(n) + min_i F(left, i) + F(right, 4 - i)            (n is a 4-wide node: its 4 slots cover both children)
// cut[n][k] remembers the arg min: 0 = n takes one slot itself, i > 0 = left gets i slots, right k - i.
struct CollapsePlan {
    std::vector<float> F;            // [node][k-1], k = 1..4
    std::vector<unsigned char> cut;  // [node][k-1]
    // pad: what the flattening adds around every box — the cost of a node is the area of the box the traversal tests, never
    // zero (colinear or coincident triangles have flat boxes: with exact areas every collapse of them costs the same 0)
    void solve(const std::vector<BuildNode>& bn, int root, float pad) {
        F.assign(bn.size() * 4, 0.f);
        cut.assign(bn.size() * 4, 0);
        std::vector<int> order;   // parents before children
        order.reserve(bn.size());
        order.push_back(root);
        for (size_t h = 0; h < order.size(); ++h) {
            const BuildNode& n = bn[(size_t)order[h]];
            if (n.left >= 0) { order.push_back(n.left); order.push_back(n.right); }
        }
        for (size_t h = order.size(); h-- > 0;) {
            const int id = order[h];
            const BuildNode& n = bn[(size_t)id];
            float* f = &F[(size_t)id * 4];
            unsigned char* c = &cut[(size_t)id * 4];
            if (n.left < 0) continue;   // leaf: all zero
            const float* fl = &F[(size_t)n.left * 4];
            const float* fr = &F[(size_t)n.right * 4];
            float split[5];             // split[k]: best of giving the two children k slots together
            unsigned char arg[5];
            for (int k = 2; k <= 4; ++k) {
                split[k] = std::numeric_limits<float>::infinity(); arg[k] = 1;
                for (int i = 1; i < k; ++i) {
                    const float v = fl[i - 1] + fr[k - i - 1];
                    if (v < split[k]) { split[k] = v; arg[k] = (unsigned char)i; }
                }
            }
            const float dx = n.box.hi[0] - n.box.lo[0] + 2.f * pad, dy = n.box.hi[1] - n.box.lo[1] + 2.f * pad,
                        dz = n.box.hi[2] - n.box.lo[2] + 2.f * pad;
            const float w = (dx * dy + dy * dz + dz * dx) + split[4];
            f[0] = w; c[0] = 0;
            for (int k = 2; k <= 4; ++k) {
                if (split[k] < w) { f[k - 1] = split[k]; c[k - 1] = arg[k]; }
                else { f[k - 1] = w; c[k - 1] = 0; }
            }
            // a wide node always opens itself: remember its own split in slot 0's neighbour (k = 4 forced)
            own.resize(bn.size(), 1);
            own[(size_t)id] = arg[4];
        }
    }
    std::vector<unsigned char> own;   // [node]: slots its left child gets when the node is a 4-wide node
    // the children (BVH2 nodes) of the 4-wide node rooted at `id`, in tree order
    void children(const std::vector<BuildNode>& bn, int id, int* out, int& n) const {
        n = 0;
        const BuildNode& r = bn[(size_t)id];
        const int i = own.empty() ? 1 : own[(size_t)id];
        cover(bn, r.left, i, out, n);
        cover(bn, r.right, 4 - i, out, n);
    }
    void cover(const std::vector<BuildNode>& bn, int id, int k, int* out, int& n) const {
        const BuildNode& b = bn[(size_t)id];
        const int c = b.left < 0 ? 0 : cut[(size_t)id * 4 + (size_t)(k - 1)];
        if (c == 0) { out[n++] = id; return; }
        cover(bn, b.left, c, out, n);
        cover(bn, b.right, k - c, out, n);
    }
};

void collapse(const std::vector<BuildNode>& bn, int root, std::vector<Wide>& wide, std::vector<int>& wide_of,
              std::vector<int>& level, float pad) {
    // default: the exact minimum (16 % fewer nodes and 1.9 % fewer node visits per ray on old_mine, tools/tree_cost.cpp;
    // +0.7 % rays/s, profiles/r03_ab_tree.log).  Its fuller nodes raise the worst-case stack need by three rows, which
    // the bounded LDS stack makes harmless.  FS_BVH_GREEDY_COLLAPSE=1: open the child of largest area instead.
    static const bool greedy = std::getenv("FS_BVH_GREEDY_COLLAPSE") != nullptr;
    CollapsePlan plan;
    if (!greedy) plan.solve(bn, root, pad);
    // breadth-first so the top of the tree is a contiguous prefix of the node array
    wide.clear();
    wide_of.assign(bn.size(), -1);
    std::vector<int> queue;
    queue.push_back(root);
    wide_of[root] = 0;
    level.assign(1, 0);   // depth of every wide node; breadth-first order keeps each level contiguous
    for (size_t h = 0; h < queue.size(); ++h) {
        const BuildNode& n = bn[queue[h]];
        Wide w;
        w.src = queue[h];
        w.n = 0;
        if (n.left < 0) {           // a leaf root: one child
            w.child[w.n++] = queue[h];
        } else if (!greedy) {
            plan.children(bn, queue[h], w.child, w.n);
        } else {
            w.child[w.n++] = n.left;
            w.child[w.n++] = n.right;
            while (w.n < 4) {       // open the inner child with the largest surface area
                int best = -1;
                float ba = -1.f;
                for (int i = 0; i < w.n; ++i) {
                    const BuildNode& c = bn[w.child[i]];
                    if (c.left >= 0 && c.box.half_area() > ba) { ba = c.box.half_area(); best = i; }
                }
                if (best < 0) break;
                const BuildNode& c = bn[w.child[best]];
                w.child[best] = c.left;
                w.child[w.n++] = c.right;
            }
        }
        for (int i = 0; i < w.n; ++i) {
            int c = w.child[i];
            if (bn[c].left >= 0) { wide_of[c] = (int)queue.size(); queue.push_back(c); level.push_back(level[h] + 1); }
        }
        wide.push_back(w);
    }
}

// worst-case pending stack entries along any root-to-leaf path (every child hit at every level)
int stack_need(const std::vector<BuildNode>& bn, const std::vector<Wide>& wide, const std::vector<int>& wide_of) {
    std::vector<int> need(wide.size(), 0);
    int worst = 0;
    for (size_t i = 0; i < wide.size(); ++i) {   // BFS order: parents before children
        const Wide& w = wide[i];
        int here = need[i] + (w.n - 1);
        worst = std::max(worst, here);
        for (int c = 0; c < w.n; ++c)
            if (bn[w.child[c]].left >= 0) need[wide_of[w.child[c]]] = here;
    }
    return worst;
}

}  // namespace

void build_bvh(const float* xyz, const uint16_t* mat, const uint32_t* object_id, int32_t T, HostBVH& out,
               const std::atomic<bool>* cancel) {
    out.nodes.clear();
    out.tris.clear();
    out.max_depth = 0;
    out.stack_need = 0;
    out.leaf_pos.clear();
    out.level_begin.clear();
    out.pad = 0.01f;
    if (T <= 0) return;

    BuildInput input;
    Builder b(input);
    b.cancel = cancel;
    if (const char* v = std::getenv("FS_BVH_LEAF")) b.kLeaf = std::max(1, std::min(4, std::atoi(v)));
    int threads = (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (const char* v = std::getenv("FS_BVH_THREADS")) threads = std::max(1, std::min(64, std::atoi(v)));
    std::vector<Prim> prims0((size_t)T);   // input order, copied into the working array before every (re)build
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
        Prim& q = prims0[(size_t)t];
        q.box = bx;
        for (int k = 0; k < 3; ++k) q.cen[k] = 0.5f * (bx.lo[k] + bx.hi[k]);
        q.idx = t;
    }
    // conservative padding (cm): >> float error of the tests at this coordinate magnitude
    const float pad = std::max(0.01f, amax * 3.8146973e-06f);

    std::vector<Wide> wide;
    std::vector<int> wide_of;
    std::vector<int> level;
    int root = 0;
    for (int cap = 48; cap >= 8; cap -= 4) {   // rebuild shallower until the traversal stack bound holds
        b.depth_cap = cap;
        b.nodes.clear();
        b.nodes.reserve(2 * (size_t)T / 3 + 16);
        b.max_depth = 0;
        input.prims = prims0;
        b.deferred.clear();
        b.defer_below = (threads > 1 && T >= 20000) ? std::max(1024, T / (4 * threads)) : 0;
        root = b.make(0, T, 0);
        if (!b.deferred.empty()) {
            // Worker threads split the deferred nodes.  A subtree depends only on its own triangle range, so the
            // tree is the one a serial build produces; only the order of the build-node array differs, and the
            // flattened output (breadth-first from the root) does not depend on that.
            const std::vector<int> tasks = b.deferred;
            std::vector<std::vector<BuildNode>> sub(tasks.size());
            std::vector<int> sub_depth(tasks.size(), 0);
            std::atomic<size_t> next{0};
            auto work = [&]() {
                for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
                    if (cancel && cancel->load(std::memory_order_relaxed)) continue;
                    const BuildNode& ph = b.nodes[(size_t)tasks[k]];
                    Builder lb(input);
                    lb.cancel = cancel;
                    lb.kLeaf = b.kLeaf;
                    lb.depth_cap = b.depth_cap;
                    lb.nodes.reserve(2 * (size_t)ph.count / 3 + 16);
                    lb.make(ph.first, ph.count, ph.depth);
                    sub[k] = std::move(lb.nodes);
                    sub_depth[k] = lb.max_depth;
                }
            };
            std::vector<std::thread> pool;
            for (int w = 1; w < threads; ++w) pool.emplace_back(work);
            work();
            for (std::thread& th : pool) th.join();
            if (cancel && cancel->load(std::memory_order_relaxed)) { out = HostBVH{}; return; }
            for (size_t k = 0; k < tasks.size(); ++k) {   // splice: local node 0 replaces the placeholder
                const int ph = tasks[k];
                const int base = (int)b.nodes.size() - 1;   // local j >= 1 -> base + j
                auto remap = [&](int c) { return c < 0 ? c : (c == 0 ? ph : base + c); };
                for (size_t j = 0; j < sub[k].size(); ++j) {
                    BuildNode n = sub[k][j];
                    n.left = remap(n.left);
                    n.right = remap(n.right);
                    if (j == 0) b.nodes[(size_t)ph] = n; else b.nodes.push_back(n);
                }
                b.max_depth = std::max(b.max_depth, sub_depth[k]);
            }
        }
        if (cancel && cancel->load(std::memory_order_relaxed)) { out = HostBVH{}; return; }
        collapse(b.nodes, root, wide, wide_of, level, pad);
        out.stack_need = stack_need(b.nodes, wide, wide_of);
        if (out.stack_need <= kStackDepth) break;
    }
    out.max_depth = b.max_depth;
    out.pad = pad;
    // refit support: where each input triangle sits in leaf order, and the node range of every tree level
    out.leaf_pos.resize(T);
    for (int i = 0; i < T; ++i) out.leaf_pos[(size_t)input.prims[(size_t)i].idx] = (uint32_t)i;
    out.level_begin.clear();
    for (size_t i = 0; i < level.size(); ++i)
        if (i == 0 || level[i] != level[i - 1]) out.level_begin.push_back((int32_t)i);
    out.level_begin.push_back((int32_t)level.size());

    // triangles in leaf order
    out.tris.resize(T);
    for (int i = 0; i < T; ++i) {
        int t = input.prims[(size_t)i].idx;
        const float* p = xyz + 9 * (size_t)t;
        Tri64 r;
        const float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
        const float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
        r.a = make_float4(p[0], p[1], p[2], e1x);
        r.b = make_float4(e1y, e1z, e2x, e2y);
        uint32_t m = mat ? (uint32_t)mat[t] : (uint32_t)FS_NO_MATERIAL;
        float mf, idf, of;
        uint32_t id = (uint32_t)t;
        uint32_t obj = object_id ? object_id[t] : (uint32_t)t;   // default: every triangle its own actor
        std::memcpy(&mf, &m, 4);
        std::memcpy(&idf, &id, 4);
        std::memcpy(&of, &obj, 4);
        r.c = make_float4(e2z, mf, idf, of);
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

    // quantise + flatten
    out.nodes.resize(wide.size());
    double dbg_exact[2] = {0, 0}, dbg_quant[2] = {0, 0};   // FS_BVH_DEBUG: half-areas of the child boxes, [inner, leaf]
    const bool dbg = std::getenv("FS_BVH_DEBUG") != nullptr;
    for (size_t i = 0; i < wide.size(); ++i) {
        const Wide& w = wide[i];
        Box nb; nb.reset();
        for (int c = 0; c < w.n; ++c) nb.grow(b.nodes[w.child[c]].box);
        NodeQ4 q{};
        double origin[3], scale[3];
        for (int k = 0; k < 3; ++k) {
            origin[k] = (double)(nb.lo[k] - pad);
            double ext = (double)(nb.hi[k] + pad) - origin[k];
            int e = (int)std::ceil(std::log2(std::max(ext, 1e-30) / 255.0));
            e = std::max(-100, std::min(100, e));
            while (std::ldexp(255.0, e) < ext) ++e;     // guard the log2 rounding
            scale[k] = std::ldexp(1.0, e);   // exact as a float: |e| <= 100
        }
        q.ox = (float)origin[0]; q.oy = (float)origin[1]; q.oz = (float)origin[2];
        // the float origin may round up by half an ulp; the >= 0.01 cm padding dwarfs it
        q.sx = (float)scale[0]; q.sy = (float)scale[1]; q.sz = (float)scale[2];
        uint32_t lo4[3] = {0, 0, 0}, hi4[3] = {0, 0, 0};
        for (int c = 0; c < 4; ++c) {
            for (int k = 0; k < 3; ++k) {
                uint32_t ql = 255, qh = 0;   // empty slot: lo > hi on every axis -> can never be hit
                if (c < w.n) {
                    const Box& cbx = b.nodes[w.child[c]].box;
                    double l = ((double)(cbx.lo[k] - pad) - (double)(float)origin[k]) / scale[k];
                    double h = ((double)(cbx.hi[k] + pad) - (double)(float)origin[k]) / scale[k];
                    ql = (uint32_t)std::max(0.0, std::min(255.0, std::floor(l)));
                    qh = (uint32_t)std::max(0.0, std::min(255.0, std::ceil(h)));
                }
                lo4[k] |= ql << (8 * c);
                hi4[k] |= qh << (8 * c);
            }
            if (dbg && c < w.n) {
                const Box& cbx = b.nodes[w.child[c]].box;
                double e[3], g[3];
                for (int k = 0; k < 3; ++k) {
                    e[k] = (double)cbx.hi[k] - (double)cbx.lo[k] + 2.0 * pad;
                    g[k] = (double)(((hi4[k] >> (8 * c)) & 255u) - ((lo4[k] >> (8 * c)) & 255u)) * scale[k];
                }
                const int leaf = b.nodes[w.child[c]].left >= 0 ? 0 : 1;
                dbg_exact[leaf] += e[0] * e[1] + e[1] * e[2] + e[2] * e[0];
                dbg_quant[leaf] += g[0] * g[1] + g[1] * g[2] + g[2] * g[0];
            }
            if (c < w.n) {
                const BuildNode& cn = b.nodes[w.child[c]];
                q.child[c] = cn.left >= 0 ? wide_of[w.child[c]] : leaf_code(cn.first, cn.count);
            } else {
                q.child[c] = -1;
            }
        }
        q.lox = lo4[0]; q.loy = lo4[1]; q.loz = lo4[2];
        q.hix = hi4[0]; q.hiy = hi4[1]; q.hiz = hi4[2];
        out.nodes[i] = q;
    }
    if (dbg)
        std::fprintf(stderr, "[fs_bvh] child boxes on the 8-bit grids: inner area x%.3f, leaf area x%.3f (%zu wide nodes)\n",
                     dbg_quant[0] / std::max(dbg_exact[0], 1e-30), dbg_quant[1] / std::max(dbg_exact[1], 1e-30), wide.size());
}

}  // namespace fs
