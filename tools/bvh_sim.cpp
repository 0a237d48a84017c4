// bvh_sim.cpp — offline proxy for tree quality (no GPU): builds the host BVH (csrc/fs_bvh.cpp) over a raw float32 triangle
// file and replays the kernels' closest-hit traversal order on the CPU for rays leaving random surface points, counting node
// visits and triangle tests per ray.  old_mine, 100 000 triangles: 14.4 node visits + 5.0 triangle tests per ray with the
// default builder (<= 2 triangles per leaf, 32 SAH bins); <= 1 per leaf 15.8 + 4.1; 16 / 64 / 128 bins 14.7 / 14.3 / 14.3;
// Kensler tree rotations on the binary tree before the collapse (up to 8 bottom-up passes): 14.40 + 4.98, i.e. nothing.
// FS_BVH_DEBUG=1 prints how much the 8-bit child grids inflate the boxes (x1.02 inner, x1.016 leaf area).
//   python -c "import __graft_entry__ as g, numpy as np; np.ascontiguousarray(g.load_package().scenes.old_mine(8).triangles, np.float32).tofile('/tmp/tri.f32')"
//   g++ -O2 -std=c++17 -Iaudio-pathtracer_amd/csrc -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ tools/bvh_sim.cpp audio-pathtracer_amd/csrc/fs_bvh.cpp -o /tmp/bvh_sim -lpthread && /tmp/bvh_sim /tmp/tri.f32
#include "fs_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
using namespace fs;
struct R { float o[3], d[3], inv[3]; };
static void trace(const HostBVH& b, const R& r, float tmax, long& nv, long& nt, float& tout) {
    int stack[256]; int sp = 0; int cur = b.nodes.empty() ? -1 : 0; float T = tmax;
    const int kDone = INT32_MIN;
    if (b.nodes.empty()) return;
    while (true) {
        if (cur == kDone) break;
        if (cur < 0) {   // leaf
            int code = ~cur; int first = code >> 2, n = (code & 3) + 1;
            for (int i = first; i < first + n; ++i) {
                ++nt;
                const Tri64& t = b.tris[i];
                float v0[3] = {t.a.x, t.a.y, t.a.z}, e1[3] = {t.a.w, t.b.x, t.b.y}, e2[3] = {t.b.z, t.b.w, t.c.x};
                float p[3] = {r.d[1]*e2[2]-r.d[2]*e2[1], r.d[2]*e2[0]-r.d[0]*e2[2], r.d[0]*e2[1]-r.d[1]*e2[0]};
                float det = e1[0]*p[0]+e1[1]*p[1]+e1[2]*p[2];
                if (det == 0) continue;
                float s[3] = {r.o[0]-v0[0], r.o[1]-v0[1], r.o[2]-v0[2]};
                float u = (s[0]*p[0]+s[1]*p[1]+s[2]*p[2]) / det; if (u < 0 || u > 1) continue;
                float q[3] = {s[1]*e1[2]-s[2]*e1[1], s[2]*e1[0]-s[0]*e1[2], s[0]*e1[1]-s[1]*e1[0]};
                float v = (r.d[0]*q[0]+r.d[1]*q[1]+r.d[2]*q[2]) / det; if (v < 0 || u + v > 1) continue;
                float tt = (e2[0]*q[0]+e2[1]*q[1]+e2[2]*q[2]) / det;
                if (tt > 0 && tt < T) T = tt;
            }
            cur = sp ? stack[--sp] : kDone;
            continue;
        }
        ++nv;
        const NodeQ4& q = b.nodes[cur];
        float org[3] = {q.ox, q.oy, q.oz}, sc[3] = {q.sx, q.sy, q.sz};
        uint32_t lo[3] = {q.lox, q.loy, q.loz}, hi[3] = {q.hix, q.hiy, q.hiz};
        float key[4]; int ref[4]; int hits = 0;
        for (int c = 0; c < 4; ++c) {
            float tn = 0, tf = T; bool empty = false;
            for (int k = 0; k < 3; ++k) {
                float l = org[k] + (float)((lo[k] >> (8*c)) & 255) * sc[k], h = org[k] + (float)((hi[k] >> (8*c)) & 255) * sc[k];
                if (((lo[k] >> (8*c)) & 255) > ((hi[k] >> (8*c)) & 255)) empty = true;
                float t0 = (l - r.o[k]) * r.inv[k], t1 = (h - r.o[k]) * r.inv[k];
                if (t0 > t1) std::swap(t0, t1);
                tn = std::max(tn, t0); tf = std::min(tf, t1);
            }
            if (!empty && tn <= tf) { key[hits] = tn; ref[hits] = q.child[c]; ++hits; }
        }
        for (int i = 1; i < hits; ++i) for (int j = i; j > 0 && key[j] < key[j-1]; --j) { std::swap(key[j], key[j-1]); std::swap(ref[j], ref[j-1]); }
        for (int i = hits - 1; i >= 1; --i) stack[sp++] = ref[i];
        if (hits) cur = ref[0]; else cur = sp ? stack[--sp] : kDone;
    }
    tout = T;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<float> xyz(n / 4); if (fread(xyz.data(), 1, n, f) != (size_t)n) return 1; fclose(f);
    int T = (int)(xyz.size() / 9);
    std::vector<uint16_t> mat(T, 0);
    HostBVH out;
    build_bvh(xyz.data(), mat.data(), nullptr, T, out);
    printf("T %d nodes %zu depth %d stack %d\n", T, out.nodes.size(), out.max_depth, out.stack_need);
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0, 1);
    long nv = 0, nt = 0; int rays = argc > 2 ? atoi(argv[2]) : 200000; double tsum = 0; int hitc = 0;
    for (int i = 0; i < rays; ++i) {
        int t = (int)(U(rng) * T) % T; const float* v = &xyz[9 * (size_t)t];
        float a = U(rng), b2 = U(rng); if (a + b2 > 1) { a = 1 - a; b2 = 1 - b2; }
        R r; float nrm[3];
        float e1[3] = {v[3]-v[0], v[4]-v[1], v[5]-v[2]}, e2[3] = {v[6]-v[0], v[7]-v[1], v[8]-v[2]};
        nrm[0] = e1[1]*e2[2]-e1[2]*e2[1]; nrm[1] = e1[2]*e2[0]-e1[0]*e2[2]; nrm[2] = e1[0]*e2[1]-e1[1]*e2[0];
        float l = std::sqrt(nrm[0]*nrm[0]+nrm[1]*nrm[1]+nrm[2]*nrm[2]); for (int k = 0; k < 3; ++k) nrm[k] /= l;
        float d[3]; float dl;
        do { for (int k = 0; k < 3; ++k) d[k] = 2 * U(rng) - 1; dl = d[0]*d[0]+d[1]*d[1]+d[2]*d[2]; } while (dl > 1 || dl < 1e-4f);
        dl = std::sqrt(dl); for (int k = 0; k < 3; ++k) d[k] /= dl;
        float dn = d[0]*nrm[0]+d[1]*nrm[1]+d[2]*nrm[2];
        float sgn = (i & 1) ? 1.f : -1.f;   // both sides of the surface (normals are not oriented)
        if (dn * sgn < 0) for (int k = 0; k < 3; ++k) d[k] = -d[k];
        for (int k = 0; k < 3; ++k) { r.o[k] = v[k] + a * e1[k] + b2 * e2[k] + sgn * 0.5f * nrm[k]; r.d[k] = d[k]; r.inv[k] = 1.0f / d[k]; }
        float tt = 0; long a0 = nv;
        trace(out, r, 1e6f, nv, nt, tt);
        if (tt < 1e6f) { ++hitc; tsum += tt; }
        (void)a0;
    }
    printf("rays %d hit %.3f mean_t %.1f node_visits/ray %.2f tri_tests/ray %.2f\n", rays, (double)hitc / rays, tsum / std::max(hitc, 1), (double)nv / rays, (double)nt / rays);
}
