// tree_cost.cpp — host-only estimate of what a tree costs the traversal kernels: builds the product's BVH
// (audio-pathtracer_amd/csrc/fs_bvh.cpp) over a triangle file and walks seeded diffuse rays through it with the
// kernels' visiting rule (4 quantised child boxes per node visit, hits sorted by entry distance, nearest first, the
// others pushed; popped entries are visited unconditionally; closest hit shrinks the interval), counting node
// visits and triangle tests per ray.  Used to compare builder variants without a GPU:
//
//   python -c "import __graft_entry__ as g, numpy as np; np.asarray(g.load_package().scenes.old_mine(8).triangles, np.float32).tofile('/tmp/mine.f32')"
//   g++ -O2 -std=c++17 -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -o /tmp/tree_cost tools/tree_cost.cpp \
//       audio-pathtracer_amd/csrc/fs_bvh.cpp && /tmp/tree_cost /tmp/mine.f32 200000
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../audio-pathtracer_amd/csrc/fs_internal.hpp"

using namespace fs;

struct Ray { float o[3], d[3], inv[3]; };

static bool tri_hit(const Tri64& t, const Ray& r, float tmax, float& tt) {
    const float v0[3] = {t.a.x, t.a.y, t.a.z}, e1[3] = {t.a.w, t.b.x, t.b.y}, e2[3] = {t.b.z, t.b.w, t.c.x};
    const float p[3] = {r.d[1] * e2[2] - r.d[2] * e2[1], r.d[2] * e2[0] - r.d[0] * e2[2], r.d[0] * e2[1] - r.d[1] * e2[0]};
    const float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (std::fabs(det) < 1e-12f) return false;
    const float id = 1.0f / det;
    const float s[3] = {r.o[0] - v0[0], r.o[1] - v0[1], r.o[2] - v0[2]};
    const float u = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * id;
    if (u < 0.f || u > 1.f) return false;
    const float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    const float v = (r.d[0] * q[0] + r.d[1] * q[1] + r.d[2] * q[2]) * id;
    if (v < 0.f || u + v > 1.f) return false;
    const float th = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * id;
    if (th <= 1e-3f || th >= tmax) return false;
    tt = th;
    return true;
}

struct Counts { double nodes = 0, tris = 0, steps = 0, pops = 0, late = 0, late_leaf = 0; };

static int closest(const HostBVH& b, const Ray& r, float& tbest, Counts& c) {
    int stack[128], sp = 0, cur = 0, hit = -1;
    float stack_t[128], cur_t = 0.f;
    float tmax = 1e30f;
    int tri_i = 0, tri_n = 0;
    while (true) {
        // the kernel advances a node and a triangle per step; count steps as max(node visits, triangle tests) per lane
        if (cur >= 0) {
            c.nodes += 1;
            const NodeQ4& n = b.nodes[(size_t)cur];
            const uint32_t lo4[3] = {n.lox, n.loy, n.loz}, hi4[3] = {n.hix, n.hiy, n.hiz};
            const float org[3] = {n.ox, n.oy, n.oz}, st[3] = {n.sx, n.sy, n.sz};
            float te[4]; int ord[4], nh = 0;
            for (int k = 0; k < 4; ++k) {
                float t0 = 0.f, t1 = tmax;
                bool empty = false;
                for (int a = 0; a < 3; ++a) {
                    const float lo = org[a] + (float)((lo4[a] >> (8 * k)) & 0xFF) * st[a];
                    const float hi = org[a] + (float)((hi4[a] >> (8 * k)) & 0xFF) * st[a];
                    if (lo > hi) empty = true;
                    float ta = (lo - r.o[a]) * r.inv[a], tb = (hi - r.o[a]) * r.inv[a];
                    if (ta > tb) std::swap(ta, tb);
                    t0 = std::max(t0, ta); t1 = std::min(t1, tb);
                }
                if (!empty && t0 <= t1) { te[nh] = t0; ord[nh] = k; ++nh; }
            }
            for (int i = 1; i < nh; ++i)
                for (int j = i; j > 0 && te[j] < te[j - 1]; --j) { std::swap(te[j], te[j - 1]); std::swap(ord[j], ord[j - 1]); }
            for (int i = nh - 1; i >= 1; --i) { stack_t[sp] = te[i]; stack[sp++] = n.child[ord[i]]; }
            if (nh) { cur = n.child[ord[0]]; cur_t = te[0]; }
            else if (sp) { c.pops += 1; cur = stack[--sp]; cur_t = stack_t[sp]; if (cur_t >= tmax) { (cur >= 0 ? c.late : c.late_leaf) += 1; } }
            else cur = INT32_MIN;
        } else if (cur != INT32_MIN) {   // a leaf
            const int code = ~cur;
            tri_i = code >> 2; tri_n = tri_i + (code & 3) + 1;
            for (; tri_i < tri_n; ++tri_i) {
                c.tris += 1;
                float tt;
                if (tri_hit(b.tris[(size_t)tri_i], r, tmax, tt)) { tmax = tt; hit = tri_i; }
            }
            if (sp) { c.pops += 1; cur = stack[--sp]; cur_t = stack_t[sp]; if (cur_t >= tmax) { (cur >= 0 ? c.late : c.late_leaf) += 1; } }
            else cur = INT32_MIN;
        } else break;
    }
    tbest = tmax;
    return hit;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::printf("usage: tree_cost triangles.f32 [rays]\n"); return 2; }
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    const int T = (int)(bytes / 36);
    std::vector<float> xyz((size_t)T * 9);
    if (std::fread(xyz.data(), 36, (size_t)T, f) != (size_t)T) return 2;
    std::fclose(f);
    const int rays = argc > 2 ? std::atoi(argv[2]) : 100000;
    HostBVH bvh;
    const auto t0 = std::chrono::steady_clock::now();
    build_bvh(xyz.data(), nullptr, nullptr, T, bvh);
    const double build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::mt19937 g(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    Counts c;
    double hits = 0;
    // walks like the kernel's: start on a random triangle, bounce diffusely (cosine lobe about the facing normal)
    for (int i = 0; i < rays; ) {
        int tri = (int)(U(g) * (float)bvh.tris.size()) % (int)bvh.tris.size();
        float u = U(g), v = U(g);
        if (u + v > 1.f) { u = 1.f - u; v = 1.f - v; }
        const Tri64* t = &bvh.tris[(size_t)tri];
        float p[3] = {t->a.x + u * t->a.w + v * t->b.z, t->a.y + u * t->b.x + v * t->b.w, t->a.z + u * t->b.y + v * t->c.x};
        float n[3] = {t->d.x, t->d.y, t->d.z};
        if (U(g) < 0.5f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        for (int bounce = 0; bounce < 8 && i < rays; ++bounce, ++i) {
            // cosine-weighted direction about n
            const float r1 = U(g), r2 = U(g), phi = 6.2831853f * r1, sr = std::sqrt(r2), cz = std::sqrt(1.f - r2);
            float a[3] = {std::fabs(n[0]) < 0.9f ? 1.f : 0.f, std::fabs(n[0]) < 0.9f ? 0.f : 1.f, 0.f};
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
            float tb;
            const int h = closest(bvh, r, tb, c);
            if (h < 0) break;
            hits += 1;
            const Tri64& ht = bvh.tris[(size_t)h];
            for (int k = 0; k < 3; ++k) p[k] = r.o[k] + tb * r.d[k];
            n[0] = ht.d.x; n[1] = ht.d.y; n[2] = ht.d.z;
            if (n[0] * r.d[0] + n[1] * r.d[1] + n[2] * r.d[2] > 0.f) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        }
        if (i < rays && hits == 0 && i > 1000) break;
        ++i;
    }
    std::printf("tris %d  nodes %zu  stack_need %d  depth %d  build %.0f ms | per ray: node visits %.3f  tri tests %.3f  "
                "cost (124 v per node + 52 v per tri) %.0f | hit fraction %.3f | popped behind the hit: %.3f nodes, %.3f leaves per ray\n",
                T, bvh.nodes.size(), bvh.stack_need, bvh.max_depth, build_ms, c.nodes / rays, c.tris / rays,
                (124.0 * c.nodes + 52.0 * c.tris) / rays, hits / rays, c.late / rays, c.late_leaf / rays);
    return 0;
}
