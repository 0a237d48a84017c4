// tree_width.cpp — host-only experiment (round 5): what would an 8-wide tree buy the dense walk?
// Builds the product's binary SAH tree (the Builder of audio-pathtracer_amd/csrc/fs_bvh.cpp, included as source so that its
// internal types are visible), collapses it to W-wide nodes with the same exact-minimum dynamic programme the product uses
// for W = 4 (Ylitie, Karras, Laine 2017), puts every node's child boxes on the node's own 8-bit grid like NodeQ4, and walks
// seeded diffuse rays through it with the kernels' visiting rule (all W child boxes tested per visit, hits pushed far to
// near, nearest first; a popped entry whose entry distance lies behind the closest hit is skipped without a visit —
// `cull_popped`, the kernels keep the entry distance with the stack entry).  Counts per ray: node visits, triangle tests,
// children hit per visit, and the predicted 16-byte L1 lookups per ray
//     visits x records(W) + triangle tests x 3          records(4) = 4 (64 B), records(8) = 5 (80 B, CWBVH layout) or 6 (96 B)
// — the unit `roofline.l1_lookups_per_launch` counts and the bound the frame kernel sits on (0.52 of the measured peak).
//
//   python -c "import __graft_entry__ as g, numpy as np; np.asarray(g.load_package().scenes.old_mine(8).triangles, np.float32).tofile('/tmp/mine.f32')"
//   g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -o /tmp/tree_width tools/tree_width.cpp && /tmp/tree_width /tmp/mine.f32 400000
#include "../audio-pathtracer_amd/csrc/fs_bvh.cpp"

#include <chrono>
#include <cstdio>
#include <random>

using namespace fs;

namespace {

constexpr int kMaxW = 16;

struct WideN {
    int n = 0;
    int child[kMaxW];        // >= 0: wide node index; < 0: ~(first * 4 + count - 1) leaf code
    float lo[kMaxW][3], hi[kMaxW][3];   // the boxes the traversal tests: padded, on the node's 8-bit grid
};

// F(n, k) = least summed area of the build nodes that become wide nodes when n's subtree is covered by at most k slots
struct Plan {
    int W;
    std::vector<float> F;
    std::vector<unsigned char> cut, own;
    void solve(const std::vector<BuildNode>& bn, int root, float pad) {
        F.assign(bn.size() * (size_t)W, 0.f);
        cut.assign(bn.size() * (size_t)W, 0);
        own.assign(bn.size(), 1);
        std::vector<int> order{root};
        for (size_t h = 0; h < order.size(); ++h) {
            const BuildNode& n = bn[(size_t)order[h]];
            if (n.left >= 0) { order.push_back(n.left); order.push_back(n.right); }
        }
        std::vector<float> split((size_t)W + 1);
        std::vector<unsigned char> arg((size_t)W + 1);
        for (size_t h = order.size(); h-- > 0;) {
            const int id = order[h];
            const BuildNode& n = bn[(size_t)id];
            if (n.left < 0) continue;
            float* f = &F[(size_t)id * W];
            unsigned char* c = &cut[(size_t)id * W];
            const float* fl = &F[(size_t)n.left * W];
            const float* fr = &F[(size_t)n.right * W];
            for (int k = 2; k <= W; ++k) {
                split[k] = std::numeric_limits<float>::infinity(); arg[k] = 1;
                for (int i = 1; i < k; ++i) {
                    const float v = fl[i - 1] + fr[k - i - 1];
                    if (v < split[k]) { split[k] = v; arg[k] = (unsigned char)i; }
                }
            }
            const float dx = n.box.hi[0] - n.box.lo[0] + 2.f * pad, dy = n.box.hi[1] - n.box.lo[1] + 2.f * pad, dz = n.box.hi[2] - n.box.lo[2] + 2.f * pad;
            const float w = (dx * dy + dy * dz + dz * dx) + split[W];
            f[0] = w; c[0] = 0;
            for (int k = 2; k <= W; ++k) {
                if (split[k] < w) { f[k - 1] = split[k]; c[k - 1] = arg[k]; } else { f[k - 1] = w; c[k - 1] = 0; }
            }
            own[(size_t)id] = arg[W];
        }
    }
    void cover(const std::vector<BuildNode>& bn, int id, int k, int* out, int& n) const {
        const BuildNode& b = bn[(size_t)id];
        const int c = b.left < 0 ? 0 : cut[(size_t)id * W + (size_t)(k - 1)];
        if (c == 0) { out[n++] = id; return; }
        cover(bn, b.left, c, out, n);
        cover(bn, b.right, k - c, out, n);
    }
    void children(const std::vector<BuildNode>& bn, int id, int* out, int& n) const {
        n = 0;
        const BuildNode& r = bn[(size_t)id];
        const int i = own[(size_t)id];
        cover(bn, r.left, i, out, n);
        cover(bn, r.right, W - i, out, n);
    }
};

void collapse_w(const std::vector<BuildNode>& bn, int root, int W, float pad, std::vector<WideN>& out) {
    Plan plan; plan.W = W;
    plan.solve(bn, root, pad);
    std::vector<int> queue{root}, wide_of(bn.size(), -1);
    wide_of[(size_t)root] = 0;
    std::vector<std::vector<int>> kids;
    for (size_t h = 0; h < queue.size(); ++h) {
        const BuildNode& n = bn[(size_t)queue[h]];
        int ch[kMaxW], cn = 0;
        if (n.left < 0) ch[cn++] = queue[h]; else plan.children(bn, queue[h], ch, cn);
        for (int i = 0; i < cn; ++i)
            if (bn[(size_t)ch[i]].left >= 0) { wide_of[(size_t)ch[i]] = (int)queue.size(); queue.push_back(ch[i]); }
        kids.emplace_back(ch, ch + cn);
    }
    out.resize(queue.size());
    for (size_t i = 0; i < queue.size(); ++i) {
        WideN& w = out[i];
        w.n = (int)kids[i].size();
        Box nb; nb.reset();
        for (int c : kids[i]) nb.grow(bn[(size_t)c].box);
        double origin[3], scale[3];
        for (int k = 0; k < 3; ++k) {
            origin[k] = (double)(float)(nb.lo[k] - pad);
            const double ext = (double)(nb.hi[k] + pad) - origin[k];
            int e = (int)std::ceil(std::log2(std::max(ext, 1e-30) / 255.0));
            while (std::ldexp(255.0, e) < ext) ++e;
            scale[k] = std::ldexp(1.0, e);
        }
        for (int c = 0; c < w.n; ++c) {
            const BuildNode& cn = bn[(size_t)kids[i][(size_t)c]];
            w.child[c] = cn.left >= 0 ? wide_of[(size_t)kids[i][(size_t)c]] : ~(cn.first * 4 + (cn.count - 1));
            for (int k = 0; k < 3; ++k) {
                const double l = std::max(0.0, std::min(255.0, std::floor(((double)(cn.box.lo[k] - pad) - origin[k]) / scale[k])));
                const double hgh = std::max(0.0, std::min(255.0, std::ceil(((double)(cn.box.hi[k] + pad) - origin[k]) / scale[k])));
                w.lo[c][k] = (float)(origin[k] + l * scale[k]);
                w.hi[c][k] = (float)(origin[k] + hgh * scale[k]);
            }
        }
    }
}

struct Ray { float o[3], d[3], inv[3]; };

bool tri_hit(const float* p, const Ray& r, float tmax, float& tt) {
    const float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
    const float pv[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
    const float det = e1[0] * pv[0] + e1[1] * pv[1] + e1[2] * pv[2];
    if (std::fabs(det) < 1e-12f) return false;
    const float id = 1.0f / det;
    const float s[3] = {r.o[0] - p[0], r.o[1] - p[1], r.o[2] - p[2]};
    const float u = (s[0] * pv[0] + s[1] * pv[1] + s[2] * pv[2]) * id;
    if (u < 0.f || u > 1.f) return false;
    const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * id;
    if (v < 0.f || u + v > 1.f) return false;
    const float th = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * id;
    if (th <= 1e-3f || th >= tmax) return false;
    tt = th;
    return true;
}

struct Counts {
    double nodes = 0, tris = 0, culled = 0, hist[kMaxW + 1] = {};
};

// order: 0 = sorted by entry distance (what the 4-wide kernel's 5-comparator network does), 1 = fixed octant order (children
// sorted once per node along the ray's sign octant by box centre — no per-visit sort: what an 8-wide step could afford)
int closest(const std::vector<WideN>& nodes, const std::vector<float>& tri, const Ray& r, float& tbest, Counts& c, int order, bool cull_popped) {
    int stack[512], sp = 0, cur = 0, hit = -1;
    float stack_t[512];
    float tmax = 1e30f;
    while (true) {
        if (cur >= 0) {
            c.nodes += 1;
            const WideN& n = nodes[(size_t)cur];
            float te[kMaxW], key[kMaxW]; int ord[kMaxW], nh = 0;
            for (int k = 0; k < n.n; ++k) {
                float t0 = 0.f, t1 = tmax;
                for (int a = 0; a < 3; ++a) {
                    float ta = (n.lo[k][a] - r.o[a]) * r.inv[a], tb = (n.hi[k][a] - r.o[a]) * r.inv[a];
                    if (ta > tb) std::swap(ta, tb);
                    t0 = std::max(t0, ta); t1 = std::min(t1, tb);
                }
                if (t0 <= t1) {
                    te[nh] = t0; ord[nh] = k;
                    key[nh] = order == 0 ? t0 : ((n.lo[k][0] + n.hi[k][0]) * r.d[0] + (n.lo[k][1] + n.hi[k][1]) * r.d[1] + (n.lo[k][2] + n.hi[k][2]) * r.d[2]);
                    ++nh;
                }
            }
            c.hist[nh] += 1;
            for (int i = 1; i < nh; ++i)
                for (int j = i; j > 0 && key[j] < key[j - 1]; --j) { std::swap(key[j], key[j - 1]); std::swap(te[j], te[j - 1]); std::swap(ord[j], ord[j - 1]); }
            for (int i = nh - 1; i >= 1; --i) { stack_t[sp] = te[i]; stack[sp++] = n.child[ord[i]]; }
            if (nh) { cur = n.child[ord[0]]; continue; }
        } else {
            const int code = ~cur;
            for (int i = code >> 2, e = (code >> 2) + (code & 3) + 1; i < e; ++i) {
                c.tris += 1;
                float tt;
                if (tri_hit(&tri[(size_t)i * 9], r, tmax, tt)) { tmax = tt; hit = i; }
            }
        }
        cur = INT32_MIN;
        while (sp) {
            --sp;
            if (cull_popped && stack_t[sp] >= tmax) { c.culled += 1; continue; }
            cur = stack[sp];
            break;
        }
        if (cur == INT32_MIN) break;
    }
    tbest = tmax;
    return hit;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: tree_width triangles.f32 [rays]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const int T = (int)(bytes / 36);
    std::vector<float> xyz((size_t)T * 9);
    if (std::fread(xyz.data(), 36, (size_t)T, f) != (size_t)T) return 2;
    std::fclose(f);
    const int rays = argc > 2 ? std::atoi(argv[2]) : 200000;

    BuildInput input;
    input.prims.resize((size_t)T);
    float amax = 0.f;
    for (int t = 0; t < T; ++t) {
        const float* p = &xyz[9 * (size_t)t];
        Prim& q = input.prims[(size_t)t];
        q.box.reset();
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                q.box.lo[k] = std::min(q.box.lo[k], p[3 * v + k]); q.box.hi[k] = std::max(q.box.hi[k], p[3 * v + k]);
                amax = std::max(amax, std::fabs(p[3 * v + k]));
            }
        for (int k = 0; k < 3; ++k) q.cen[k] = 0.5f * (q.box.lo[k] + q.box.hi[k]);
        q.idx = t;
    }
    const float pad = std::max(0.01f, amax * 3.8146973e-06f);
    Builder b(input);
    const int root = b.make(0, T, 0);
    std::vector<float> tri((size_t)T * 9);    // leaf order
    for (int i = 0; i < T; ++i) std::copy_n(&xyz[9 * (size_t)input.prims[(size_t)i].idx], 9, &tri[9 * (size_t)i]);

    // the same seeded rays for every tree: diffuse bounces from random surface points (as tools/tree_cost.cpp)
    std::vector<Ray> rs;
    {
        std::vector<WideN> ref;
        collapse_w(b.nodes, root, 4, pad, ref);
        std::mt19937 g(12345);
        std::uniform_real_distribution<float> U(0.f, 1.f);
        Counts dummy;
        while ((int)rs.size() < rays) {
            const int t = (int)(U(g) * (float)T) % T;
            const float* p9 = &tri[9 * (size_t)t];
            float u = U(g), v = U(g);
            if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
            float p[3], e1[3], e2[3], n[3];
            for (int k = 0; k < 3; ++k) { e1[k] = p9[3 + k] - p9[k]; e2[k] = p9[6 + k] - p9[k]; p[k] = p9[k] + u * e1[k] + v * e2[k]; }
            n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
            const float nl = 1.f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (float& x : n) x *= (U(g) < 0.5f ? -nl : nl);
            for (int bounce = 0; bounce < 8 && (int)rs.size() < rays; ++bounce) {
                const float r1 = U(g), r2 = U(g), phi = 6.2831853f * r1, sr = std::sqrt(r2), cz = std::sqrt(1.f - r2);
                const float a[3] = {std::fabs(n[0]) < 0.9f ? 1.f : 0.f, std::fabs(n[0]) < 0.9f ? 0.f : 1.f, 0.f};
                float tx[3] = {a[1] * n[2] - a[2] * n[1], a[2] * n[0] - a[0] * n[2], a[0] * n[1] - a[1] * n[0]};
                const float tl = 1.f / std::sqrt(tx[0] * tx[0] + tx[1] * tx[1] + tx[2] * tx[2]);
                for (float& x : tx) x *= tl;
                const float ty[3] = {n[1] * tx[2] - n[2] * tx[1], n[2] * tx[0] - n[0] * tx[2], n[0] * tx[1] - n[1] * tx[0]};
                Ray r;
                for (int k = 0; k < 3; ++k) {
                    r.d[k] = sr * std::cos(phi) * tx[k] + sr * std::sin(phi) * ty[k] + cz * n[k];
                    r.o[k] = p[k] + 0.01f * n[k];
                    r.inv[k] = 1.0f / r.d[k];
                }
                rs.push_back(r);
                float tb;
                const int h = closest(ref, tri, r, tb, dummy, 0, true);
                if (h < 0) break;
                const float* q9 = &tri[9 * (size_t)h];
                for (int k = 0; k < 3; ++k) { p[k] = r.o[k] + tb * r.d[k]; e1[k] = q9[3 + k] - q9[k]; e2[k] = q9[6 + k] - q9[k]; }
                n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
                const float hl = 1.f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                for (float& x : n) x *= hl;
                if (n[0] * r.d[0] + n[1] * r.d[1] + n[2] * r.d[2] > 0.f) for (float& x : n) x = -x;
            }
        }
    }
    std::printf("{\"triangles\": %d, \"rays\": %d, \"binary_nodes\": %zu, \"trees\": [\n", T, rays, b.nodes.size());
    const int widths[] = {4, 6, 8, 16};
    bool first = true;
    for (int W : widths) {
        std::vector<WideN> wide;
        collapse_w(b.nodes, root, W, pad, wide);
        double fill = 0;
        for (const WideN& w : wide) fill += w.n;
        for (int order = 0; order < 2; ++order)
            for (int cull = 1; cull >= 0; --cull) {
                Counts c;
                long checksum = 0;
                for (const Ray& r : rs) { float tb; checksum += closest(wide, tri, r, tb, c, order, cull != 0); }
                const double v = c.nodes / rays, t = c.tris / rays;
                const int rec_lo = W == 4 ? 4 : W <= 6 ? 4 : W == 8 ? 5 : 9, rec_hi = W == 4 ? 4 : W <= 6 ? 5 : W == 8 ? 6 : 10;
                double pass = 0, tot = 0;
                for (int k = 0; k <= W; ++k) { tot += c.hist[k]; if (k <= 1) pass += c.hist[k]; }
                std::printf("%s {\"width\": %d, \"order\": \"%s\", \"cull_popped\": %s, \"wide_nodes\": %zu, \"children_per_node\": %.2f, \"node_visits_per_ray\": %.3f, "
                            "\"tri_tests_per_ray\": %.3f, \"visits_with_at_most_one_child_hit\": %.3f, \"popped_entries_culled_per_ray\": %.3f, "
                            "\"lookups16_per_ray\": [%.1f, %.1f], \"records_per_visit\": [%d, %d], \"hit_checksum\": %ld}",
                            first ? "" : ",\n", W, order == 0 ? "entry distance" : "centre along the ray", cull ? "true" : "false", wide.size(), fill / (double)wide.size(), v, t,
                            pass / tot, c.culled / rays, v * rec_lo + 3 * t, v * rec_hi + 3 * t, rec_lo, rec_hi, checksum);
                first = false;
            }
    }
    std::printf("\n]}\n");
    return 0;
}
