/*
 * fs_oracle.c — CPU restatement of the FrequenSee BDPT hot path.  TEST INFRASTRUCTURE ONLY.
 * See fs_oracle.h for scope, provenance and the "parity unpinned" statement.
 *
 * File:line citations are into /root/reference/Plugins/FrequenSee/Source/FrequenSee/:
 *   ARTS.cpp = Private/AudioRayTracingSubsystem.cpp     ARTS.h = Public/AudioRayTracingSubsystem.h
 *   FSAC.cpp = Private/FrequenSeeAudioComponent.cpp      FSAC.h = Public/FrequenSeeAudioComponent.h
 *   MAT.h    = Public/AcousticMaterial.h
 *
 * Build with -ffp-contract=off: every float op below is a single IEEE-754 binary32 operation
 * (+ - * / sqrtf fmaf), so the path geometry is reproducible bit-for-bit on any conforming target.
 */
#include "fs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define FSO_PI 3.1415926535897932f /* UE_PI */

/* ------------------------------------------------------------------------------------------- */
/* defaults = the constants compiled into the reference                        */
/* ------------------------------------------------------------------------------------------- */
void fso_params_default(fso_params* p) {
    memset(p, 0, sizeof(*p));
    p->seed = 0x5EEDull;
    p->num_pairs = 1000;        /* USED_RAY_COUNT ARTS.h:176 */
    p->depth = 0;               /* no cap in GeneratePath ARTS.cpp:294 */
    p->russian_roulette = 1;
    p->rr_prob = 0.9f;          /* ARTS.cpp:282 */
    p->max_trace_dist = 1000000.f; /* ARTS.cpp:284 */
    p->surface_offset = 0.1f;   /* ARTS.cpp:345 */
    p->connect_pullback = 0.1f; /* ARTS.cpp:253 */
    p->dist_divisor = 1000.f;   /* ARTS.cpp:373 */
    p->min_seg = 1.0f;          /* ARTS.cpp:375 */
    p->prob_exponent = 0.1f;    /* ARTS.cpp:398 */
    p->energy_clamp = 1.0f;     /* ARTS.cpp:410 */
    p->energy_gain = 10.f;      /* ARTS.cpp:413 */
    p->sound_speed = 343.0f;    /* ARTS.cpp:362 */
    for (int b = 0; b < FSO_MAX_BANDS; ++b) p->air_absorption[b] = 0.05f; /* ARTS.cpp:395 */
    p->flags = 0;
    p->source_object = FSO_NO_OBJECT; p->listener_object = FSO_NO_OBJECT;
}

/* ------------------------------------------------------------------------------------------- */
/* RNG: Philox4x32-10 (Salmon et al. 2011), replaces FMath::FRand's global rand()              */
/* ------------------------------------------------------------------------------------------- */
void fso_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* [0,1) with 24 mantissa bits (FRand contract: uniform float) */
float fso_u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

/* draw block `block` of the stream (seed, pair, side, bounce) */
static void fso_draw(uint64_t seed, uint32_t pair, uint32_t side, uint32_t bounce, uint32_t block, uint32_t out[4]) {
    uint32_t ctr[4] = {pair, (bounce << 1) | (side & 1u), block, 0x46533031u /* 'FS01' */};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    fso_philox4x32_10(ctr, key, out);
}

/* sin/cos(2*pi*u), u in [0,1): quadrant reduction + Taylor polynomials evaluated with fmaf only,
 * so CPU and GPU agree bit-for-bit (libm sinf/cosf are not reproducible across targets). */
void fso_sincos2pi(float u, float* s_out, float* c_out) {
    float q = floorf(fmaf(u, 4.0f, 0.5f));         /* nearest quarter turn, 0..4 */
    float a = fmaf(q, -0.25f, u);                   /* [-1/8, 1/8], exact */
    float x = a * 6.283185307179586f;               /* [-pi/4, pi/4] */
    float x2 = x * x;
    float sp = 2.7557319e-06f;                      /* 1/9! */
    sp = fmaf(sp, x2, -1.9841270e-04f);             /* -1/7! */
    sp = fmaf(sp, x2, 8.3333333e-03f);              /* 1/5! */
    sp = fmaf(sp, x2, -1.6666667e-01f);             /* -1/3! */
    float s = fmaf(sp * x2, x, x);
    float cp = 2.4801587e-05f;                      /* 1/8! */
    cp = fmaf(cp, x2, -1.3888889e-03f);             /* -1/6! */
    cp = fmaf(cp, x2, 4.1666667e-02f);              /* 1/4! */
    cp = fmaf(cp, x2, -0.5f);
    float c = fmaf(cp, x2, 1.0f);
    int k = ((int)q) & 3;
    float ss, cc;
    if (k == 0) { ss = s; cc = c; }
    else if (k == 1) { ss = c; cc = -s; }
    else if (k == 2) { ss = -s; cc = -c; }
    else { ss = -c; cc = s; }
    *s_out = ss; *c_out = cc;
}

/* FMath::VRand (ARTS.cpp:308): rejection in the cube until 1e-4 < |v|^2 <= 1, then normalise.
 * r0 = block 0 of the bounce's stream (r0[0] was the roulette draw); retries use blocks 1.. */
void fso_sample_sphere(uint64_t seed, uint32_t pair, uint32_t side, uint32_t bounce, const uint32_t r0[4],
                       float dir[3]) {
    uint32_t r[4] = {r0[1], r0[2], r0[3], 0};
    for (uint32_t attempt = 0; attempt < 16; ++attempt) {
        if (attempt > 0) fso_draw(seed, pair, side, bounce, attempt, r);
        float x = fmaf(fso_u01(r[0]), 2.0f, -1.0f);
        float y = fmaf(fso_u01(r[1]), 2.0f, -1.0f);
        float z = fmaf(fso_u01(r[2]), 2.0f, -1.0f);
        float l2 = x * x + y * y + z * z;
        if (l2 > 1e-4f && l2 <= 1.0f) {
            float inv = 1.0f / sqrtf(l2);
            dir[0] = x * inv; dir[1] = y * inv; dir[2] = z * inv;
            return;
        }
    }
    dir[0] = 0.f; dir[1] = 0.f; dir[2] = 1.f;
}

/* FMath::VRandCone(n, pi/2) (ARTS.cpp:313): theta = 2 pi U; phi = fmod(acos(2V-1), pi/2);
 * rotate n by phi about a perpendicular axis, then by theta about n; GetSafeNormal.
 * With x = 2V-1: x > 0 -> (cos phi, sin phi) = (x, sqrt(1-x^2)); x <= 0 -> phi = acos(x) - pi/2 ->
 * (cos phi, sin phi) = (sqrt(1-x^2), -x).  No acos/fmod needed.
 * cosine != 0: cosine-weighted hemisphere (compat flag "cosine_sampling", quirk A.6-g). */
void fso_sample_cone(const float n[3], float U, float V, int32_t cosine, float dir[3]) {
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
    fso_sincos2pi(U, &st, &ct);
    /* orthonormal basis (Duff et al. 2017, branchless) */
    float sg = copysignf(1.0f, n[2]);
    float a = -1.0f / (sg + n[2]);
    float b = n[0] * n[1] * a;
    float t0 = fmaf(sg * n[0] * n[0], a, 1.0f), t1 = sg * b, t2 = -sg * n[0];
    float b0 = b, b1 = fmaf(n[1] * n[1], a, sg), b2 = -n[1];
    float lx = sphi * ct, ly = sphi * st;
    float d0 = fmaf(lx, t0, fmaf(ly, b0, cphi * n[0]));
    float d1 = fmaf(lx, t1, fmaf(ly, b1, cphi * n[1]));
    float d2 = fmaf(lx, t2, fmaf(ly, b2, cphi * n[2]));
    float l2 = d0 * d0 + d1 * d1 + d2 * d2;
    float inv = 1.0f / sqrtf(l2);
    dir[0] = d0 * inv; dir[1] = d1 * inv; dir[2] = d2 * inv;
}

/* ------------------------------------------------------------------------------------------- */
/* scene: triangles + materials + reference BVH2 (binned SAH, <= 4 tris/leaf)                  */
/* ------------------------------------------------------------------------------------------- */
typedef struct fso_tri { /* 48 B record: v0, e1, e2, material */
    float v0[3]; float e1[3]; float e2[3];
    uint32_t material; uint32_t id; uint32_t object; /* actor the triangle belongs to */
} fso_tri;

typedef struct fso_bnode { /* 32 B */
    float lo[3]; float hi[3];
    uint32_t left_first; /* inner: index of left child (right = left+1); leaf: first triangle */
    uint32_t count;      /* 0 = inner */
} fso_bnode;

struct fso_scene {
    int32_t T, M, B;
    fso_tri* tris;       /* leaf order */
    fso_tri* tris_orig;  /* input order (brute force scans ids ascending) */
    float* absorption;   /* [M][B] */
    float* lobe_gain;    /* [M][3][B] diffuse, specular, transmitted (FSO_FLAG_MATERIAL_LOBES) */
    float* lobe_prob;    /* [M][3] probability of picking each lobe (band means of the gains, normalised) */
    fso_bnode* nodes;
    int32_t num_nodes;
    float pad;
};

static void tri_bounds(const fso_tri* t, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) {
        float a = t->v0[k], b = t->v0[k] + t->e1[k], c = t->v0[k] + t->e2[k];
        lo[k] = fminf(a, fminf(b, c));
        hi[k] = fmaxf(a, fmaxf(b, c));
    }
}

typedef struct build_ctx {
    fso_scene* s;
    float* clo; float* chi; float* cen; /* per-triangle bounds/centroid, permuted with tris */
} build_ctx;

static void swap_tri(build_ctx* bc, int i, int j) {
    if (i == j) return;
    fso_tri tt = bc->s->tris[i]; bc->s->tris[i] = bc->s->tris[j]; bc->s->tris[j] = tt;
    for (int k = 0; k < 3; ++k) {
        float x;
        x = bc->clo[3 * i + k]; bc->clo[3 * i + k] = bc->clo[3 * j + k]; bc->clo[3 * j + k] = x;
        x = bc->chi[3 * i + k]; bc->chi[3 * i + k] = bc->chi[3 * j + k]; bc->chi[3 * j + k] = x;
        x = bc->cen[3 * i + k]; bc->cen[3 * i + k] = bc->cen[3 * j + k]; bc->cen[3 * j + k] = x;
    }
}

static float half_area(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}

#define FSO_BINS 16
static void build_node(build_ctx* bc, uint32_t ni, int first, int count) {
    fso_scene* s = bc->s;
    fso_bnode* nd = &s->nodes[ni];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = first; i < first + count; ++i)
        for (int k = 0; k < 3; ++k) {
            lo[k] = fminf(lo[k], bc->clo[3 * i + k]); hi[k] = fmaxf(hi[k], bc->chi[3 * i + k]);
            clo[k] = fminf(clo[k], bc->cen[3 * i + k]); chi[k] = fmaxf(chi[k], bc->cen[3 * i + k]);
        }
    for (int k = 0; k < 3; ++k) { nd->lo[k] = lo[k] - s->pad; nd->hi[k] = hi[k] + s->pad; }
    if (count <= 4) { nd->left_first = (uint32_t)first; nd->count = (uint32_t)count; return; }

    int best_axis = -1, best_split = -1; float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
        float ext = chi[ax] - clo[ax];
        if (!(ext > 0.0f)) continue;
        float blo[FSO_BINS][3], bhi[FSO_BINS][3]; int bcnt[FSO_BINS];
        for (int b = 0; b < FSO_BINS; ++b) {
            bcnt[b] = 0;
            for (int k = 0; k < 3; ++k) { blo[b][k] = INFINITY; bhi[b][k] = -INFINITY; }
        }
        float scale = (float)FSO_BINS / ext;
        for (int i = first; i < first + count; ++i) {
            int b = (int)((bc->cen[3 * i + ax] - clo[ax]) * scale);
            if (b < 0) b = 0; if (b >= FSO_BINS) b = FSO_BINS - 1;
            bcnt[b]++;
            for (int k = 0; k < 3; ++k) {
                blo[b][k] = fminf(blo[b][k], bc->clo[3 * i + k]); bhi[b][k] = fmaxf(bhi[b][k], bc->chi[3 * i + k]);
            }
        }
        float rarea[FSO_BINS]; int rcnt[FSO_BINS];
        float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY}; int rc = 0;
        for (int b = FSO_BINS - 1; b >= 1; --b) {
            for (int k = 0; k < 3; ++k) { rl[k] = fminf(rl[k], blo[b][k]); rh[k] = fmaxf(rh[k], bhi[b][k]); }
            rc += bcnt[b]; rcnt[b] = rc; rarea[b] = rc ? half_area(rl, rh) : 0.0f;
        }
        float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY}; int lc = 0;
        for (int b = 0; b < FSO_BINS - 1; ++b) {
            for (int k = 0; k < 3; ++k) { ll[k] = fminf(ll[k], blo[b][k]); lh[k] = fmaxf(lh[k], bhi[b][k]); }
            lc += bcnt[b];
            if (lc == 0 || rcnt[b + 1] == 0) continue;
            float cost = half_area(ll, lh) * (float)lc + rarea[b + 1] * (float)rcnt[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = b; }
        }
    }
    int mid;
    if (best_axis < 0) {
        mid = first + count / 2; /* all centroids coincide: split by index */
    } else {
        float ext = chi[best_axis] - clo[best_axis];
        float scale = (float)FSO_BINS / ext;
        int i = first, j = first + count - 1;
        while (i <= j) {
            int b = (int)((bc->cen[3 * i + best_axis] - clo[best_axis]) * scale);
            if (b < 0) b = 0; if (b >= FSO_BINS) b = FSO_BINS - 1;
            if (b <= best_split) ++i; else { swap_tri(bc, i, j); --j; }
        }
        mid = i;
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    uint32_t left = (uint32_t)s->num_nodes;
    s->num_nodes += 2;
    nd->left_first = left; nd->count = 0;
    build_node(bc, left, first, mid - first);
    build_node(bc, left + 1, mid, first + count - mid);
}

fso_scene* fso_scene_create(const float* xyz, const uint16_t* mat_id, int32_t T, const float* absorption,
                            int32_t M, int32_t B) {
    if (T < 0 || B < 1 || B > FSO_MAX_BANDS) return NULL;
    fso_scene* s = (fso_scene*)calloc(1, sizeof(fso_scene));
    s->T = T; s->M = M; s->B = B;
    s->tris = (fso_tri*)calloc((size_t)(T > 0 ? T : 1), sizeof(fso_tri));
    s->tris_orig = (fso_tri*)calloc((size_t)(T > 0 ? T : 1), sizeof(fso_tri));
    s->absorption = (float*)calloc((size_t)(M > 0 ? M * B : 1), sizeof(float));
    if (M > 0) memcpy(s->absorption, absorption, sizeof(float) * (size_t)M * (size_t)B);
    s->lobe_gain = (float*)calloc((size_t)(M > 0 ? M * 3 * B : 1), sizeof(float));
    s->lobe_prob = (float*)calloc((size_t)(M > 0 ? M * 3 : 1), sizeof(float));
    fso_scene_set_lobes(s, NULL, NULL);
    float amax = 0.0f;
    for (int i = 0; i < T; ++i) {
        fso_tri* t = &s->tris[i];
        const float* p = xyz + 9 * (size_t)i;
        for (int k = 0; k < 3; ++k) {
            t->v0[k] = p[k]; t->e1[k] = p[3 + k] - p[k]; t->e2[k] = p[6 + k] - p[k];
            amax = fmaxf(amax, fmaxf(fabsf(p[k]), fmaxf(fabsf(p[3 + k]), fabsf(p[6 + k]))));
        }
        t->material = mat_id ? mat_id[i] : FSO_NO_MATERIAL;
        t->id = (uint32_t)i;
        t->object = (uint32_t)i; /* default: every triangle is its own actor */
        s->tris_orig[i] = *t;
    }
    /* conservative box padding: far above float error of the slab/triangle tests, so the BVH result
     * equals the brute-force result (the closest hit is a function of ray and triangles only) */
    s->pad = fmaxf(0.01f, amax * 3.8146973e-06f /* 2^-18 */);
    s->nodes = (fso_bnode*)calloc((size_t)(2 * (T > 0 ? T : 1)), sizeof(fso_bnode));
    s->num_nodes = 1;
    build_ctx bc;
    bc.s = s;
    bc.clo = (float*)malloc(sizeof(float) * 3 * (size_t)(T > 0 ? T : 1));
    bc.chi = (float*)malloc(sizeof(float) * 3 * (size_t)(T > 0 ? T : 1));
    bc.cen = (float*)malloc(sizeof(float) * 3 * (size_t)(T > 0 ? T : 1));
    for (int i = 0; i < T; ++i) {
        tri_bounds(&s->tris[i], &bc.clo[3 * i], &bc.chi[3 * i]);
        for (int k = 0; k < 3; ++k) bc.cen[3 * i + k] = 0.5f * (bc.clo[3 * i + k] + bc.chi[3 * i + k]);
    }
    if (T > 0) build_node(&bc, 0, 0, T);
    else { s->nodes[0].count = 0; s->nodes[0].left_first = 0; s->num_nodes = 0; }
    free(bc.clo); free(bc.chi); free(bc.cen);
    return s;
}

/* Lobe tables of FSO_FLAG_MATERIAL_LOBES.  The split is the per-bin rule of UMaterialAcousticProcessor::ApplyMaterialFD
 * (MaterialAcousticProcessor.cpp:51-72) applied per band: Refl = 1 - alpha; tau clamped so that Refl + tau <= 1;
 * specular Refl (1 - sigma), diffuse Refl sigma, transmitted tau.  The walk picks ONE lobe per surface vertex with
 * probabilities proportional to the band means of the three gains (build-owned: the reference's walk is diffuse
 * only, "FIXME assuming diffuse", ARTS.cpp:304).  Without Transmission / Scattering arrays: tau = 0, sigma = 1. */
void fso_scene_set_lobes(fso_scene* s, const float* transmission, const float* scattering) {
    const int B = s->B;
    for (int m = 0; m < s->M; ++m) {
        float sum[3] = {0.f, 0.f, 0.f};
        for (int b = 0; b < B; ++b) {
            float alpha = s->absorption[(size_t)m * B + b];
            float tau = transmission ? transmission[(size_t)m * B + b] : 0.0f;
            float sigma = scattering ? scattering[(size_t)m * B + b] : 1.0f;
            float refl = 1.0f - alpha;
            if (refl + tau > 1.0f) tau = 1.0f - refl;
            float g[3];
            g[FSO_LOBE_DIFFUSE] = refl * sigma;
            g[FSO_LOBE_SPECULAR] = refl * (1.0f - sigma);
            g[FSO_LOBE_TRANSMIT] = tau;
            for (int l = 0; l < 3; ++l) {
                if (!(g[l] > 0.0f)) g[l] = 0.0f;       /* also NaN; curves outside [0, 1] give no negative energy */
                s->lobe_gain[((size_t)m * 3 + l) * B + b] = g[l];
                sum[l] += g[l];
            }
        }
        float mean[3], tot = 0.0f;
        for (int l = 0; l < 3; ++l) { mean[l] = sum[l] / (float)B; tot += mean[l]; }
        /* a material that reflects and transmits nothing keeps the diffuse lobe (gain 0): walk lengths stay a
         * function of the random stream alone */
        for (int l = 0; l < 3; ++l) s->lobe_prob[(size_t)m * 3 + l] = tot > 0.0f ? mean[l] / tot : (l == 0 ? 1.0f : 0.0f);
    }
}
void fso_scene_lobe_table(const fso_scene* s, int32_t m, float* gains, float prob[3]) {
    memcpy(gains, s->lobe_gain + (size_t)m * 3 * s->B, sizeof(float) * 3 * (size_t)s->B);
    memcpy(prob, s->lobe_prob + (size_t)m * 3, sizeof(float) * 3);
}

void fso_scene_destroy(fso_scene* s) {
    if (!s) return;
    free(s->tris); free(s->tris_orig); free(s->absorption); free(s->lobe_gain); free(s->lobe_prob); free(s->nodes); free(s);
}
int32_t fso_scene_num_nodes(const fso_scene* s) { return s->num_nodes; }

/* ------------------------------------------------------------------------------------------- */
/* ray queries: the engine contract of UWorld::LineTraceSingleByObjectType     */
/* ------------------------------------------------------------------------------------------- */
/* Moeller-Trumbore, two-sided, t in (0, tmax]; fixed operation order shared with the HIP kernel spec */
static int tri_hit(const fso_tri* tr, const float o[3], const float d[3], float tmax, float* t_out) {
    const float* e1 = tr->e1; const float* e2 = tr->e2;
    float px = fmaf(d[1], e2[2], -(d[2] * e2[1]));
    float py = fmaf(d[2], e2[0], -(d[0] * e2[2]));
    float pz = fmaf(d[0], e2[1], -(d[1] * e2[0]));
    float det = fmaf(e1[0], px, fmaf(e1[1], py, e1[2] * pz));
    if (det == 0.0f) return 0;
    /* barycentric tests on the un-normalised values, sign-normalised by det (exact), so the one
     * division is only paid by rays that are inside the triangle */
    float ad = fabsf(det);
    float sx = o[0] - tr->v0[0], sy = o[1] - tr->v0[1], sz = o[2] - tr->v0[2];
    float U = fmaf(sx, px, fmaf(sy, py, sz * pz));
    float us = det < 0.0f ? -U : U;
    if (!(us >= 0.0f && us <= ad)) return 0;
    float qx = fmaf(sy, e1[2], -(sz * e1[1]));
    float qy = fmaf(sz, e1[0], -(sx * e1[2]));
    float qz = fmaf(sx, e1[1], -(sy * e1[0]));
    float V = fmaf(d[0], qx, fmaf(d[1], qy, d[2] * qz));
    float vs = det < 0.0f ? -V : V;
    if (!(vs >= 0.0f && (us + vs) <= ad)) return 0;
    float t = fmaf(e2[0], qx, fmaf(e2[1], qy, e2[2] * qz)) / det;
    if (!(t > 0.0f && t <= tmax)) return 0;
    *t_out = t;
    return 1;
}

static void ray_inv(const float d[3], float inv[3]) {
    for (int k = 0; k < 3; ++k) {
        float x = d[k];
        if (fabsf(x) < 1e-20f) x = copysignf(1e-20f, x);
        inv[k] = 1.0f / x;
    }
}

/* slab test against a padded box; returns entry distance or INFINITY on miss */
static float box_hit(const fso_bnode* n, const float o[3], const float inv[3], float tmax) {
    float t0x = (n->lo[0] - o[0]) * inv[0], t1x = (n->hi[0] - o[0]) * inv[0];
    float t0y = (n->lo[1] - o[1]) * inv[1], t1y = (n->hi[1] - o[1]) * inv[1];
    float t0z = (n->lo[2] - o[2]) * inv[2], t1z = (n->hi[2] - o[2]) * inv[2];
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return (tn <= tf) ? tn : INFINITY;
}

static void hit_normal(const fso_tri* tr, const float d[3], float n[3]) {
    const float* e1 = tr->e1; const float* e2 = tr->e2;
    float nx = fmaf(e1[1], e2[2], -(e1[2] * e2[1]));
    float ny = fmaf(e1[2], e2[0], -(e1[0] * e2[2]));
    float nz = fmaf(e1[0], e2[1], -(e1[1] * e2[0]));
    float l2 = nx * nx + ny * ny + nz * nz;
    float inv = 1.0f / sqrtf(l2);
    nx *= inv; ny *= inv; nz *= inv;
    float dn = fmaf(nx, d[0], fmaf(ny, d[1], nz * d[2]));
    if (dn > 0.0f) { nx = -nx; ny = -ny; nz = -nz; } /* ImpactNormal faces the ray origin side */
    n[0] = nx; n[1] = ny; n[2] = nz;
}

static int32_t closest_impl(const fso_scene* s, const float o[3], const float d[3], float tmax, int32_t brute,
                            uint32_t ignore_object, float* t_out, int32_t* tri_out, float n_out[3],
                            fso_counters* c) {
    float best_t = tmax; uint32_t best_id = 0xFFFFFFFFu; const fso_tri* best = NULL;
    if (c) c->closest_rays++;
    if (brute || s->num_nodes == 0) {
        for (int i = 0; i < s->T; ++i) {
            float t;
            if (c) c->tri_tests++;
            if (s->tris_orig[i].object == ignore_object) continue;   /* AddIgnoredActor */
            if (tri_hit(&s->tris_orig[i], o, d, best_t, &t)) {
                if (t < best_t || best == NULL) { best_t = t; best = &s->tris_orig[i]; best_id = (uint32_t)i; }
            }
        }
    } else {
        float inv[3]; ray_inv(d, inv);
        uint32_t stack[64]; int sp = 0;
        if (c) c->node_visits++;
        if (box_hit(&s->nodes[0], o, inv, best_t) != INFINITY) stack[sp++] = 0;
        while (sp > 0) {
            const fso_bnode* n = &s->nodes[stack[--sp]];
            if (n->count) {
                for (uint32_t i = n->left_first; i < n->left_first + n->count; ++i) {
                    float t; const fso_tri* tr = &s->tris[i];
                    if (c) c->tri_tests++;
                    if (tr->object == ignore_object) continue;       /* AddIgnoredActor */
                    if (tri_hit(tr, o, d, best_t, &t)) {
                        if (t < best_t || best == NULL || (t == best_t && tr->id < best_id)) {
                            best_t = t; best = tr; best_id = tr->id;
                        }
                    }
                }
            } else {
                uint32_t l = n->left_first;
                if (c) c->node_visits += 2;
                float tl = box_hit(&s->nodes[l], o, inv, best_t);
                float tr_ = box_hit(&s->nodes[l + 1], o, inv, best_t);
                if (tl != INFINITY && tr_ != INFINITY) {
                    if (tl <= tr_) { stack[sp++] = l + 1; stack[sp++] = l; }
                    else { stack[sp++] = l; stack[sp++] = l + 1; }
                } else if (tl != INFINITY) stack[sp++] = l;
                else if (tr_ != INFINITY) stack[sp++] = l + 1;
            }
        }
    }
    if (!best) return 0;
    if (t_out) *t_out = best_t;
    if (tri_out) *tri_out = (int32_t)best_id;
    if (n_out) hit_normal(best, d, n_out);
    return 1;
}

int32_t fso_trace_closest(const fso_scene* s, const float o[3], const float d[3], float tmax, int32_t brute,
                          float* t_out, int32_t* tri_out, float n_out[3], fso_counters* c) {
    return closest_impl(s, o, d, tmax, brute, FSO_NO_OBJECT, t_out, tri_out, n_out, c);
}

void fso_scene_set_objects(fso_scene* s, const uint32_t* object_id) {
    for (int i = 0; i < s->T; ++i) s->tris_orig[i].object = object_id ? object_id[i] : (uint32_t)i;
    for (int i = 0; i < s->T; ++i) s->tris[i].object = s->tris_orig[s->tris[i].id].object;
}

int32_t fso_trace_any(const fso_scene* s, const float o[3], const float d[3], float tmax, int32_t brute,
                      fso_counters* c) {
    float t;
    if (c) c->any_rays++;
    if (brute || s->num_nodes == 0) {
        for (int i = 0; i < s->T; ++i) {
            if (c) { c->tri_tests++; c->any_tri_tests++; }
            if (tri_hit(&s->tris_orig[i], o, d, tmax, &t)) return 1;
        }
        return 0;
    }
    float inv[3]; ray_inv(d, inv);
    uint32_t stack[64]; int sp = 0;
    if (c) { c->node_visits++; c->any_node_visits++; }
    if (box_hit(&s->nodes[0], o, inv, tmax) != INFINITY) stack[sp++] = 0;
    while (sp > 0) {
        const fso_bnode* n = &s->nodes[stack[--sp]];
        if (n->count) {
            for (uint32_t i = n->left_first; i < n->left_first + n->count; ++i) {
                if (c) { c->tri_tests++; c->any_tri_tests++; }
                if (tri_hit(&s->tris[i], o, d, tmax, &t)) return 1;
            }
        } else {
            uint32_t l = n->left_first;
            if (c) { c->node_visits += 2; c->any_node_visits += 2; }
            float tl = box_hit(&s->nodes[l], o, inv, tmax);
            float tr_ = box_hit(&s->nodes[l + 1], o, inv, tmax);
            if (tl != INFINITY && tr_ != INFINITY) {
                if (tl <= tr_) { stack[sp++] = l + 1; stack[sp++] = l; }
                else { stack[sp++] = l; stack[sp++] = l + 1; }
            } else if (tl != INFINITY) stack[sp++] = l;
            else if (tr_ != INFINITY) stack[sp++] = l + 1;
        }
    }
    return 0;
}

/* the end points' collision: a sphere; a ray that starts inside leaves through the far side (build-owned) */
static int sphere_hit(const float o[3], const float d[3], const float c[3], float r, float tmax, float* t_out) {
    float ox = o[0] - c[0], oy = o[1] - c[1], oz = o[2] - c[2];
    float b = fmaf(ox, d[0], fmaf(oy, d[1], oz * d[2]));
    float cc = fmaf(ox, ox, fmaf(oy, oy, oz * oz)) - r * r;
    float disc = fmaf(b, b, -cc);
    if (!(disc >= 0.0f)) return 0;
    float sq = sqrtf(disc);
    float t = -b - sq;
    if (!(t > 0.0f)) t = sq - b;
    if (!(t > 0.0f && t <= tmax)) return 0;
    *t_out = t;
    return 1;
}


/* ImpactNormal of a sphere hit: unit (impact - centre), flipped to face the ray origin side like a triangle's */
static void sphere_normal(const float o[3], const float d[3], float t, const float c[3], float n[3]) {
    float x = fmaf(t, d[0], o[0]) - c[0], y = fmaf(t, d[1], o[1]) - c[1], z = fmaf(t, d[2], o[2]) - c[2];
    float l2 = x * x + y * y + z * z;
    float inv = 1.0f / sqrtf(l2);
    x = x * inv; y = y * inv; z = z * inv;
    float dn = fmaf(x, d[0], fmaf(y, d[1], z * d[2]));
    if (dn > 0.0f) { x = -x; y = -y; z = -z; }
    n[0] = x; n[1] = y; n[2] = z;
}

/* ------------------------------------------------------------------------------------------- */
/* GeneratePath  ARTS.cpp:279-355                                                              */
/* ------------------------------------------------------------------------------------------- */
/* depth cap of one subpath: the explicit one, or none at all (the reference's while (true), ARTS.cpp:294) — the
 * roulette ends such a walk with probability 1.  Without roulette an uncapped walk would never end: 64 then. */
static int32_t depth_cap(const fso_params* p) {
    if (p->depth > 0) return p->depth;
    if (p->russian_roulette && p->rr_prob < 1.0f) return INT32_MAX;
    return FSO_MAX_DEPTH;
}

/* The walk.  Nodes go to *nodes (capacity *cap); grow != 0: the buffer is realloc'ed as needed, else the walk is cut
 * at the capacity. */
static int32_t generate_path(const fso_scene* s, const fso_params* p, uint32_t pair, uint32_t side,
                             const float start[3], const float* other, float other_radius, fso_node** nodes_io,
                             int32_t* cap_io, int grow, fso_counters* c) {
    fso_node* nodes = *nodes_io;
    int32_t max_nodes = *cap_io;
    const int32_t cap = depth_cap(p);
    /* state variables ARTS.cpp:287-291 */
    fso_pos_t pos[3] = {start[0], start[1], start[2]};
    float nrm[3] = {0.f, 0.f, 0.f};
    uint32_t mat = FSO_NO_MATERIAL;
    float prob = 1.0f;
    int has_normal = 0;
    int brute = (p->flags & FSO_FLAG_BRUTE_FORCE) != 0;
    int cosine = (p->flags & FSO_FLAG_COSINE_SAMPLING) != 0;
    /* FSO_FLAG_MATERIAL_LOBES (row f4, build-owned): a vertex reached by a hit on a material picks one lobe with
     * the Philox word the diffuse walk leaves unused (r[3]); diffuse = the reference's cone sample, specular =
     * mirror direction of the arriving ray, transmitted = the arriving direction continued from the far side of the
     * surface (origin pos - 2 offset n).  The node keeps the lobe in material bits 16-17 for EvaluatePath; its
     * probability multiplies the node probability. */
    int lobes = (p->flags & FSO_FLAG_MATERIAL_LOBES) != 0;
    float din[3] = {0.f, 0.f, 0.f};
    int arrived = 0;
    int32_t n = 0;
    for (uint32_t k = 0;; ++k) {
        /* 0. push node ARTS.cpp:296-297 */
        if (n >= max_nodes) {
            if (!grow) break;
            max_nodes *= 2;
            nodes = (fso_node*)realloc(nodes, sizeof(fso_node) * (size_t)max_nodes);
            *nodes_io = nodes; *cap_io = max_nodes;
        }
        fso_node* nd = &nodes[n++];
        memcpy(nd->pos, pos, sizeof(pos)); memcpy(nd->normal, nrm, sizeof(nrm));
        nd->material = mat; nd->prob = prob;
        if (c) c->path_nodes++;
        /* depth cap (build parameter; the reference loop is unbounded, quirk A.6-i) */
        if ((int32_t)k >= cap) break;
        /* 1. Russian roulette ARTS.cpp:300-301 */
        uint32_t r[4];
        fso_draw(p->seed, pair, side, k, 0, r);
        if (p->russian_roulette && !(fso_u01(r[0]) < p->rr_prob)) break; /* ARTS.cpp:349-353 */
        /* 2. direction + probability ARTS.cpp:304-319 */
        float dir[3], org[3];
        int shifted = 0;
        if (!has_normal) {                               /* CurrentNormal.IsNearlyZero() */
            fso_sample_sphere(p->seed, pair, side, k, r, dir);
            float pdf = 1.0f / (4.0f * FSO_PI);
            prob = pdf * p->rr_prob;
        } else {
            uint32_t lobe = FSO_LOBE_DIFFUSE;
            float plobe = 1.0f;
            int pick = lobes && arrived && mat != FSO_NO_MATERIAL && (int32_t)mat < s->M;
            if (pick) {
                const float* pr = s->lobe_prob + 3 * (size_t)mat;
                float u = fso_u01(r[3]);
                float c1 = pr[0], c2 = pr[0] + pr[1];
                lobe = u < c1 ? FSO_LOBE_DIFFUSE : (u < c2 ? FSO_LOBE_SPECULAR : FSO_LOBE_TRANSMIT);
                if (lobe == FSO_LOBE_TRANSMIT && !(pr[2] > 0.0f)) lobe = pr[1] > 0.0f ? FSO_LOBE_SPECULAR : FSO_LOBE_DIFFUSE;
                plobe = pr[lobe];
                nd->material = mat | (lobe << FSO_LOBE_SHIFT);
            }
            if (lobe == FSO_LOBE_DIFFUSE) {
                fso_sample_cone(nrm, fso_u01(r[1]), fso_u01(r[2]), cosine, dir);
                float cos_theta = dir[0] * nrm[0] + dir[1] * nrm[1] + dir[2] * nrm[2];
                float pdf = cos_theta / FSO_PI;
                prob = pdf * p->rr_prob;
            } else if (lobe == FSO_LOBE_SPECULAR) {
                float dn = din[0] * nrm[0] + din[1] * nrm[1] + din[2] * nrm[2];
                float k2 = 2.0f * dn;
                for (int q = 0; q < 3; ++q) dir[q] = fmaf(-k2, nrm[q], din[q]);
                prob = p->rr_prob;
            } else {
                for (int q = 0; q < 3; ++q) dir[q] = din[q];
                prob = p->rr_prob;
            }
            if (pick) prob = prob * plobe;
            if (lobe == FSO_LOBE_TRANSMIT) {
                float back = -2.0f * p->surface_offset;
                for (int q = 0; q < 3; ++q) org[q] = fmaf(back, nrm[q], (float)pos[q]);
                shifted = 1;
            }
        }
        /* 3. closest hit on [pos, pos + dir * MAX_RAYCAST_DIST] ARTS.cpp:339-342 */
        float t, hn[3]; int32_t tri;
        if (!shifted) for (int q = 0; q < 3; ++q) org[q] = (float)pos[q];
        /* the walk ignores the actor it starts from (AddIgnoredActor ARTS.cpp:322-327) */
        int hit = closest_impl(s, org, dir, p->max_trace_dist, brute, side == 0 ? p->source_object : p->listener_object, &t, &tri, hn, c);
        /* the OTHER end point's collision sphere (ECC_Pawn is queried, the walk's own actor is ignored: ARTS.cpp:322-334);
         * it wins ties with a triangle, like the pawn in the legacy tracer */
        int on_sphere = 0;
        float ts = 0.f;
        if (other && other_radius > 0.0f && sphere_hit(org, dir, other, other_radius, p->max_trace_dist, &ts) && (!hit || ts <= t)) {
            hit = 1; on_sphere = 1; t = ts;
            sphere_normal(org, dir, ts, other, hn);
        }
        if (hit) {
            /* 4. ARTS.cpp:345-347 */
            for (int q = 0; q < 3; ++q) {
#ifdef FSO_DOUBLE_POSITIONS   /* Hit.ImpactPoint + Hit.ImpactNormal * 0.1f in FVector (double) arithmetic */
                double ip = (shifted ? (double)org[q] : pos[q]) + (double)t * (double)dir[q];
                pos[q] = ip + (double)p->surface_offset * (double)hn[q];
#else
                float ip = fmaf(t, dir[q], org[q]);
                pos[q] = fmaf(p->surface_offset, hn[q], ip);
#endif
                nrm[q] = hn[q];
                din[q] = dir[q];
            }
            has_normal = 1;
            arrived = 1;
            mat = on_sphere ? FSO_NO_MATERIAL : s->tris_orig[tri].material;   /* a pawn has no UAcousticGeometryComponent */
        } else {
            arrived = 0;
        }
        /* on miss the state is unchanged and the loop continues (duplicate node, new prob) */
    }
    return n;
}

int32_t fso_generate_path(const fso_scene* s, const fso_params* p, uint32_t pair, uint32_t side,
                          const float start[3], fso_node* nodes, int32_t max_nodes, fso_counters* c) {
    return generate_path(s, p, pair, side, start, NULL, 0.0f, &nodes, &max_nodes, 0, c);
}

/* ------------------------------------------------------------------------------------------- */
/* ConnectSubpaths  ARTS.cpp:235-277  — visible iff the line trace does NOT hit                */
/* ------------------------------------------------------------------------------------------- */
int32_t fso_connect(const fso_scene* s, const fso_params* p, const fso_node* f, const fso_node* b,
                    fso_counters* c) {
    return fso_connect_ep(s, p, f, b, NULL, NULL, c);
}

int32_t fso_connect_ep(const fso_scene* s, const fso_params* p, const fso_node* f, const fso_node* b, const float* src,
                       const float* lis, fso_counters* c) {
    fso_pos_t dx = b->pos[0] - f->pos[0], dy = b->pos[1] - f->pos[1], dz = b->pos[2] - f->pos[2];
    fso_pos_t l2 = dx * dx + dy * dy + dz * dz;
    if (!(l2 > 1e-8f)) { /* GetSafeNormal() == 0: zero-length trace, nothing to hit */
        if (c) c->any_rays++;
        return 1;
    }
#ifdef FSO_DOUBLE_POSITIONS
    double len = sqrt(l2), inv = 1.0 / len;
#else
    float len = sqrtf(l2);
    float inv = 1.0f / len;
#endif
    float d[3] = {(float)(dx * inv), (float)(dy * inv), (float)(dz * inv)};
    float tmax = (float)(len - p->connect_pullback); /* End = B - 0.1 * unit(B - F), ARTS.cpp:253 */
    if (!(tmax > 0.0f)) { if (c) c->any_rays++; return 1; }
    const float o[3] = {(float)f->pos[0], (float)f->pos[1], (float)f->pos[2]};
    /* ConnectSubpaths ignores no actor (ARTS.cpp:252-254): both end points' collision spheres block */
    float ts;
    if ((lis && p->listener_radius > 0.0f && sphere_hit(o, d, lis, p->listener_radius, tmax, &ts)) ||
        (src && p->source_radius > 0.0f && sphere_hit(o, d, src, p->source_radius, tmax, &ts))) {
        if (c) c->any_rays++;
        return 0;
    }
    return !fso_trace_any(s, o, d, tmax, (p->flags & FSO_FLAG_BRUTE_FORCE) != 0, c);
}

/* ------------------------------------------------------------------------------------------- */
/* EvaluatePath  ARTS.cpp:360-420  (one Energy per band; band b uses Absorption[b], A.3)       */
/* ------------------------------------------------------------------------------------------- */
void fso_evaluate_path(const fso_scene* s, const fso_params* p, const fso_node* nodes, int32_t n,
                       float gains[FSO_MAX_BANDS], float* delay_seconds) {
    int B = s->B;
    float scaled = 0.0f;                        /* ScaledDistance ARTS.cpp:364 */
    float E[FSO_MAX_BANDS];
    for (int b = 0; b < B; ++b) E[b] = 1.0f;    /* Energy ARTS.cpp:365 */
    for (int i = 0; i < n - 1; ++i) {
        const fso_node* a = &nodes[i]; const fso_node* q = &nodes[i + 1];
#ifdef FSO_DOUBLE_POSITIONS   /* FVector::Dist(...) / 1000.f: a double, narrowed by the assignment to float NodeDistance */
        double dx = q->pos[0] - a->pos[0], dy = q->pos[1] - a->pos[1], dz = q->pos[2] - a->pos[2];
        float nd = (float)(sqrt(dx * dx + dy * dy + dz * dz) / (double)p->dist_divisor);
#else
        float dx = q->pos[0] - a->pos[0], dy = q->pos[1] - a->pos[1], dz = q->pos[2] - a->pos[2];
        float dist = sqrtf(dx * dx + dy * dy + dz * dz);     /* FVector::Dist ARTS.cpp:372 */
        float nd = dist / p->dist_divisor;                    /* ARTS.cpp:373 */
#endif
        scaled += nd;                                         /* ARTS.cpp:374 */
        if (nd < p->min_seg) continue;                        /* ARTS.cpp:375-378 */
        float nd2 = nd * nd;
        float geo = 1.0f / (4 * FSO_PI * nd2);                /* ARTS.cpp:391 */
        float pw = powf(a->prob, p->prob_exponent);           /* ARTS.cpp:398 */
        const int lobes = (p->flags & FSO_FLAG_MATERIAL_LOBES) != 0;
        const uint32_t mid = lobes && a->material != FSO_NO_MATERIAL ? (a->material & 0xFFFFu) : a->material;
        const uint32_t lobe = lobes && a->material != FSO_NO_MATERIAL ? ((a->material >> FSO_LOBE_SHIFT) & 3u) : 0u;
        for (int b = 0; b < B; ++b) {
            float bsdf = 1.0f;                                /* ARTS.cpp:382-386 */
            if (mid != FSO_NO_MATERIAL && (int32_t)mid < s->M) {
                if (!lobes) {
                    bsdf = s->absorption[(size_t)mid * (size_t)B + (size_t)b] / FSO_PI;
                } else {   /* row f4: gain of the lobe the walk took at this vertex (diffuse at a connection vertex) */
                    float g = s->lobe_gain[((size_t)mid * 3 + lobe) * (size_t)B + (size_t)b];
                    bsdf = lobe == FSO_LOBE_DIFFUSE ? g / FSO_PI : g;
                }
            }
            float e = E[b];
            e *= bsdf;                                        /* ARTS.cpp:392 */
            e *= geo;                                         /* ARTS.cpp:393 */
            float media = expf(-p->air_absorption[b] * nd);   /* ARTS.cpp:396 */
            e *= media;                                       /* ARTS.cpp:397 */
            e /= pw;                                          /* ARTS.cpp:398 */
            E[b] = e;
        }
    }
    for (int b = 0; b < B; ++b) {
        float e = (E[b] < p->energy_clamp) ? E[b] : p->energy_clamp; /* FMath::Min ARTS.cpp:410 */
        gains[b] = e * p->energy_gain;                               /* ARTS.cpp:413 */
    }
    *delay_seconds = scaled / p->sound_speed;                        /* ARTS.cpp:419 */
}

/* ------------------------------------------------------------------------------------------- */
/* Balance-heuristic weights for the all-connections mode (row f3).                            */
/* The reference's draft (getExpectedWeight / MISEnergy, ARTS.cpp:548-597) weights a sample by  */
/* "Cj * prob" — its sampling density relative to the other strategies "that could have         */
/* produced the same path" — but is unfinished (getPdf returns 0.9, integer divisions); this   */
/* is that intent with the densities the walk actually uses (ARTS.cpp:306-318), build-owned:   */
/*   path y_0 (source), y_1..y_t (surface vertices, stored normal n_k), y_{t+1} (listener);    */
/*   d_k = unit(y_{k+1} - y_k), L_k = |y_{k+1} - y_k|, k = 0..t.                               */
/*   forward density of y_{k+1} given y_k   pf_k = Pf(k) |n_{k+1} . d_k| / L_k^2,  k = 0..t-1,  */
/*       Pf(0) = 1/(4 pi) (VRand), Pf(k) = max(0, n_k . d_k) / pi (the walk's CosTheta / PI);   */
/*   backward density of y_k given y_{k+1}  pb_k = Pb(k+1) |n_k . d_k| / L_k^2,    k = 1..t,    */
/*       Pb(t+1) = 1/(4 pi), Pb(k) = max(0, -n_k . d_{k-1}) / pi;                               */
/*   strategy s generates y_1..y_s from the source and y_t..y_{s+1} from the listener:         */
/*       p_s = prod_{k<s} pf_k * prod_{k>s} pb_k;   w_s = p_s / sum_{s'} p_s'.                  */
/* (The roulette factor rr^t is common to all strategies of one path and cancels.)  Evaluated   */
/* in ONE pass over the segments, in double: T_k = T_{k-1} pb_k + [lo <= k <= hi] PF_k gives    */
/* T_t = sum_s p_s, Q the same for s alone; PF_k = prod_{m<k} pf_m.  The HIP kernel follows     */
/* the same sequence of operations.                                                            */
/* ------------------------------------------------------------------------------------------- */
double fso_mis_weight(const fso_node* nodes, int32_t n, int32_t s, int32_t D) {
    const int32_t t = n - 2;
    const int32_t lo = t - D > 0 ? t - D : 0, hi = t < D ? t : D;
    const double uniform = 1.0 / (double)(hi - lo + 1);
    if (t <= 0) return uniform;
    const double inv4pi = 1.0 / (4.0 * 3.14159265358979323846), invpi = 1.0 / 3.14159265358979323846;
    double PF = 1.0, T = 0.0, Q = 0.0;
    for (int32_t k = 0; k <= t; ++k) {
        const fso_node* a = &nodes[k]; const fso_node* b = &nodes[k + 1];
        double dx = (double)b->pos[0] - (double)a->pos[0], dy = (double)b->pos[1] - (double)a->pos[1],
               dz = (double)b->pos[2] - (double)a->pos[2];
        double l2 = dx * dx + dy * dy + dz * dz;
        if (!(l2 > 1e-8)) return uniform;
        double inv = 1.0 / sqrt(l2);
        dx *= inv; dy *= inv; dz *= inv;
        /* cosines of the segment with the normals at its two ends (the end points have none) */
        double ca = k > 0 ? (double)a->normal[0] * dx + (double)a->normal[1] * dy + (double)a->normal[2] * dz : 0.0;
        double cb = k < t ? (double)b->normal[0] * dx + (double)b->normal[1] * dy + (double)b->normal[2] * dz : 0.0;
        if (k >= 1) {
            double Pb = k == t ? inv4pi : (cb < 0.0 ? -cb : 0.0) * invpi;
            double pb = Pb * fabs(ca) / l2;
            T *= pb;
            if (k > s) Q *= pb;
        }
        if (k >= lo && k <= hi) T += PF;
        if (k == s) Q = PF;
        if (k < t) {
            double Pf = k == 0 ? inv4pi : (ca > 0.0 ? ca : 0.0) * invpi;
            PF *= Pf * fabs(cb) / l2;
        }
    }
    double w = Q / T;
    if (!(Q > 0.0) || !(T > 0.0) || !(w <= 1.0)) return uniform;   /* also NaN / inf */
    return w;
}

/* ------------------------------------------------------------------------------------------- */
/* energy buffer  FSAC.h:72-91,133-139                                                         */
/* ------------------------------------------------------------------------------------------- */
int32_t fso_num_bins(float simulated_duration, float bin_duration) {
    return (int32_t)ceilf(simulated_duration / bin_duration); /* FSAC.h:137 */
}
int32_t fso_num_samples(float simulated_duration, int32_t sample_rate) {
    return (int32_t)ceilf(simulated_duration * (float)sample_rate); /* FSAC.h:138 */
}
int32_t fso_samples_per_bin(float bin_duration, int32_t sample_rate) {
    return (int32_t)ceilf(bin_duration * (float)sample_rate); /* FSAC.cpp:324: 48.000004f -> 49 */
}

int32_t fso_add_energy_at_delay(float* energy, int32_t num_bins, int32_t bin_size_ms, float delay_s, float e) {
    float x = (delay_s * 1000.f) / (float)bin_size_ms;      /* FSAC.h:89 */
    float fl = floorf(x);
    int32_t bin;
    if (!(fl > 0.0f)) bin = 0;                              /* also NaN */
    else if (fl >= (float)(num_bins - 1)) bin = num_bins - 1;
    else bin = (int32_t)fl;
    energy[bin] += e;                                       /* FSAC.h:90 */
    return bin;
}

/* ------------------------------------------------------------------------------------------- */
/* UpdateSource  ARTS.cpp:128-173 (pairs [pair_begin, pair_end))                               */
/* ------------------------------------------------------------------------------------------- */
void fso_compute_energy(const fso_scene* s, const fso_params* p, const float src[3], const float lis[3],
                        uint32_t pair_begin, uint32_t pair_end, int32_t num_bins, float* energy_f32,
                        double* energy_f64, fso_counters* c) {
    int B = s->B;
    int32_t cap_f = (p->depth > 0 ? p->depth : FSO_MAX_DEPTH) + 1, cap_b = cap_f, cap_all = 2 * cap_f;
    fso_node* fwd = (fso_node*)malloc(sizeof(fso_node) * (size_t)cap_f);      /* grown as needed: depth 0 has no cap */
    fso_node* bwd = (fso_node*)malloc(sizeof(fso_node) * (size_t)cap_b);
    fso_node* all = (fso_node*)malloc(sizeof(fso_node) * (size_t)cap_all);
    /* FlushEnergyBuffer ARTS.cpp:157-161 (a no-op at HEAD after the first call: FSO_FLAG_ACCUMULATE_ENERGY) */
    if (!(p->flags & FSO_FLAG_ACCUMULATE_ENERGY)) {
        memset(energy_f32, 0, sizeof(float) * (size_t)B * (size_t)num_bins);
        if (energy_f64) memset(energy_f64, 0, sizeof(double) * (size_t)B * (size_t)num_bins);
    }
    /* NormalizationFactor ARTS.cpp:164 (quirk A.6-c: literally 1/1000) */
    float norm = (p->flags & FSO_FLAG_FIXED_NORM_1000) ? 1.0f / 1000.0f : 1.0f / (float)p->num_pairs;
    for (uint32_t i = pair_begin; i < pair_end; ++i) {     /* GenerateFullPaths ARTS.cpp:215-230 */
        int32_t nf = generate_path(s, p, i, 0, src, lis, p->listener_radius, &fwd, &cap_f, 1, c);
        int32_t nb = generate_path(s, p, i, 1, lis, src, p->source_radius, &bwd, &cap_b, 1, c);
        if (nf + nb > cap_all) { cap_all = 2 * (nf + nb); all = (fso_node*)realloc(all, sizeof(fso_node) * (size_t)cap_all); }
        if (nf == 0 || nb == 0) continue;                  /* ARTS.cpp:237 */
        if (p->flags & (FSO_FLAG_ALL_CONNECTIONS | FSO_FLAG_MIS_BALANCE)) {
            /* Row f3 — the reference's unfinished "naive connections" draft (Is_NaiveConnections, ARTS.cpp:518-546:
             * "connect every bounce of every sample in forward dir with every bounce in backward dir, Equation
             * 12"), restated for one pair: forward prefix F0..Fi (i = 0..k) x backward prefix B0..Bj (j = 0..m),
             * each connected by ConnectSubpaths' visibility test and evaluated by EvaluatePath on the path
             * F0..Fi, Bj..B0.  Several (i, j) produce paths with the same number of segments i + j + 1; they are
             * combined with uniform multiple-importance weights 1 / N(i + j), N(t) = number of (i', j') in
             * [0, D]^2 with i' + j' = t (D = depth cap) — the weights of one path length sum to 1. */
            int32_t D = p->depth > 0 ? p->depth : (depth_cap(p) == INT32_MAX ? FSO_UNBOUNDED_DEPTH : FSO_MAX_DEPTH);
            for (int32_t fi = 0; fi < nf; ++fi)
                for (int32_t bj = 0; bj < nb; ++bj) {
                    if (!fso_connect_ep(s, p, &fwd[fi], &bwd[bj], src, lis, c)) continue;
                    if (c) c->connected++;
                    memcpy(all, fwd, sizeof(fso_node) * (size_t)(fi + 1));
                    for (int32_t j = 0; j <= bj; ++j) all[fi + 1 + j] = bwd[bj - j];
                    /* the two connection vertices scatter diffusely whatever lobe the walk took there later */
                    if (all[fi].material != FSO_NO_MATERIAL) all[fi].material &= 0xFFFFu;
                    if (all[fi + 1].material != FSO_NO_MATERIAL) all[fi + 1].material &= 0xFFFFu;
                    float gains[FSO_MAX_BANDS], delay;
                    fso_evaluate_path(s, p, all, fi + bj + 2, gains, &delay);
                    int32_t t = fi + bj;
                    int32_t lo = t - D > 0 ? t - D : 0, hi = t < D ? t : D;
                    float w = 1.0f / (float)(hi - lo + 1);
                    if (p->flags & FSO_FLAG_MIS_BALANCE) w = (float)fso_mis_weight(all, fi + bj + 2, fi, D);
                    int32_t bin = 0;
                    for (int b = 0; b < B; ++b) {
                        float e = gains[b];
                        e *= norm;
                        e *= w;
                        bin = fso_add_energy_at_delay(energy_f32 + (size_t)b * (size_t)num_bins, num_bins, 1, delay, e);
                        if (energy_f64) energy_f64[(size_t)b * (size_t)num_bins + (size_t)bin] += (double)e;
                    }
                    if (c) c->deposits++;
                }
            continue;
        }
        if (!fso_connect_ep(s, p, &fwd[nf - 1], &bwd[nb - 1], src, lis, c)) continue;
        if (c) c->connected++;
        /* node order F0..Fk, Bm..B0 ARTS.cpp:262-267 */
        memcpy(all, fwd, sizeof(fso_node) * (size_t)nf);
        for (int32_t j = 0; j < nb; ++j) all[nf + j] = bwd[nb - 1 - j];
        float gains[FSO_MAX_BANDS], delay;
        fso_evaluate_path(s, p, all, nf + nb, gains, &delay);
        int32_t bin = 0;
        for (int b = 0; b < B; ++b) {
            float e = gains[b];
            e *= norm;                                     /* ARTS.cpp:170 */
            bin = fso_add_energy_at_delay(energy_f32 + (size_t)b * (size_t)num_bins, num_bins, 1, delay, e);
            if (energy_f64) energy_f64[(size_t)b * (size_t)num_bins + (size_t)bin] += (double)e;
        }
        if (c) c->deposits++;
    }
    free(fwd); free(bwd); free(all);
}

/* All-cores CPU baseline: the pair range cut into `threads` static shares, every thread with
 * PRIVATE [B][num_bins] double histograms and counters (nothing shared but the read-only scene), one final sum — plain
 * pthreads, no interpreter between the threads.  energy_f64 [B][num_bins] = the sum of the private histograms in thread
 * order (deterministic for a given thread count); energy_f32 = its rounding (the sequential fp32 accumulation of the
 * reference exists per thread only and is not returned).  Returns the number of threads that ran. */
#include <pthread.h>
typedef struct mt_job {
    const fso_scene* s; const fso_params* p; const float* src; const float* lis;
    uint32_t begin, end; int32_t num_bins;
    float* e32; double* e64; fso_counters c;
} mt_job;
static void* mt_run(void* arg) {
    mt_job* j = (mt_job*)arg;
    fso_params q = *j->p;
    q.flags &= ~(uint32_t)FSO_FLAG_ACCUMULATE_ENERGY;   /* the private histograms start at zero; the caller adds */
    fso_compute_energy(j->s, &q, j->src, j->lis, j->begin, j->end, j->num_bins, j->e32, j->e64, &j->c);
    return NULL;
}
int32_t fso_compute_energy_mt(const fso_scene* s, const fso_params* p, const float src[3], const float lis[3],
                              uint32_t pair_begin, uint32_t pair_end, int32_t num_bins, int32_t threads,
                              float* energy_f32, double* energy_f64, fso_counters* c) {
    if (threads < 1) threads = 1;
    if (pair_end < pair_begin) pair_end = pair_begin;
    const uint32_t n = pair_end - pair_begin;
    if ((uint32_t)threads > n && n > 0) threads = (int32_t)n;
    const size_t words = (size_t)s->B * (size_t)num_bins;
    mt_job* jobs = (mt_job*)calloc((size_t)threads, sizeof(mt_job));
    pthread_t* tid = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
    float* e32 = (float*)calloc(words * (size_t)threads, sizeof(float));
    double* e64 = (double*)calloc(words * (size_t)threads, sizeof(double));
    int32_t started = 0;
    for (int32_t t = 0; t < threads; ++t) {
        mt_job* j = &jobs[t];
        j->s = s; j->p = p; j->src = src; j->lis = lis; j->num_bins = num_bins;
        j->begin = pair_begin + (uint32_t)(((uint64_t)n * (uint64_t)t) / (uint64_t)threads);
        j->end = pair_begin + (uint32_t)(((uint64_t)n * (uint64_t)(t + 1)) / (uint64_t)threads);
        j->e32 = e32 + words * (size_t)t; j->e64 = e64 + words * (size_t)t;
        if (t + 1 == threads || pthread_create(&tid[t], NULL, mt_run, j) != 0) {   /* the last share (and any share whose thread */
            mt_run(j);                                                               /* could not be started) runs here */
            tid[t] = 0;
        } else {
            ++started;
        }
    }
    for (int32_t t = 0; t < threads; ++t) if (tid[t]) pthread_join(tid[t], NULL);
    if (!(p->flags & FSO_FLAG_ACCUMULATE_ENERGY)) {
        if (energy_f64) memset(energy_f64, 0, sizeof(double) * words);
        memset(energy_f32, 0, sizeof(float) * words);
    }
    double* sum = energy_f64 ? energy_f64 : (double*)calloc(words, sizeof(double));
    if (!energy_f64 && (p->flags & FSO_FLAG_ACCUMULATE_ENERGY)) for (size_t i = 0; i < words; ++i) sum[i] = (double)energy_f32[i];
    for (int32_t t = 0; t < threads; ++t) {
        for (size_t i = 0; i < words; ++i) sum[i] += jobs[t].e64[i];
        if (c) {   /* the counters are sums of uint64 fields */
            uint64_t* dst = (uint64_t*)c; const uint64_t* srcw = (const uint64_t*)&jobs[t].c;
            for (size_t k = 0; k < sizeof(fso_counters) / sizeof(uint64_t); ++k) dst[k] += srcw[k];
        }
    }
    for (size_t i = 0; i < words; ++i) energy_f32[i] = (float)sum[i];
    if (!energy_f64) free(sum);
    free(jobs); free(tid); free(e32); free(e64);
    return started + 1;
}

/* ------------------------------------------------------------------------------------------- */
/* ReconstructImpulseResponse  FSAC.cpp:320-380 (one channel / one band row)                   */
/* ------------------------------------------------------------------------------------------- */
void fso_reconstruct(const float* energy, int32_t num_bins, int32_t sample_rate, float bin_duration,
                     int32_t num_samples, int32_t samples_per_bin_override, float* out) {
    const float kEnergyThreshold = 1e-6f;                          /* FSAC.cpp:322 */
    const float Pi4 = sqrtf(4.0f * FSO_PI);                        /* FSAC.cpp:323 */
    int32_t spb = samples_per_bin_override > 0 ? samples_per_bin_override
                                               : fso_samples_per_bin(bin_duration, sample_rate);
    float* ir = (float*)calloc((size_t)num_samples, sizeof(float)); /* Memset 0 FSAC.cpp:335 */
    for (int32_t bin = 0; bin < num_bins; ++bin) {
        int32_t left = num_samples - bin * spb;
        int32_t nbs = spb < left ? spb : left;                     /* FSAC.cpp:340 */
        float e = 0.0f;
        if (fabsf(energy[bin]) >= kEnergyThreshold)                /* FSAC.cpp:343 (Response and Norm alias) */
            e = energy[bin] / sqrtf(energy[bin] * Pi4);            /* FSAC.cpp:345 */
        float prev = 0.0f;
        if (bin == 0) prev = e;                                    /* FSAC.cpp:348-351 */
        else if (fabsf(energy[bin - 1]) >= kEnergyThreshold)
            prev = energy[bin - 1] / sqrtf(energy[bin - 1] * Pi4); /* FSAC.cpp:352-355 */
        for (int32_t bs = 0, smp = bin * spb; bs < nbs; ++bs, ++smp) {
            float w = (float)bs / (float)spb;                      /* FSAC.cpp:359 */
            float a = (1.0f - w) * prev;
            float bb = w * e;
            ir[smp] = a + bb;                                      /* FSAC.cpp:360-362 */
        }
    }
    const float k = 0.25f;                                         /* FSAC.cpp:366 */
    out[0] = ir[0];                                                /* FSAC.cpp:371 */
    for (int32_t i = 1; i < num_samples; ++i) {
        float a = k * ir[i];
        float bb = (1.0f - k) * out[i - 1];
        out[i] = a + bb;                                           /* FSAC.cpp:374 */
    }
    /* NormalizeImpulseResponse (FSAC.cpp:382-406) zeroes the UNFILTERED array, which is then
     * replaced by Filtered (FSAC.cpp:377-378): the net result is the filtered signal. */
    free(ir);
}

/* ------------------------------------------------------------------------------------------- */
/* Legacy per-frame forward tracer: UpdateSound FSAC.cpp:283-306, CastAudioRay :132-207,         */
/* CastDirectAudioRay :209-280.  Engine semantics defined by the build: an "actor" is an object  */
/* id per triangle (fso_scene_set_objects); the player pawn is a sphere of listener_radius at the */
/* listener position; the source's own actor has no geometry.                                    */
/* ------------------------------------------------------------------------------------------- */
#define FSO_PAWN_OBJECT 0xFFFFFFFEu
/* closest blocking hit among the triangles (minus one ignored actor) and the pawn sphere */
static int legacy_trace(const fso_scene* s, const float o[3], const float d[3], float tmax, uint32_t ignore,
                        const float lis[3], float radius, float* t_out, uint32_t* obj_out, float n_out[3],
                        fso_sound_result* res) {
    float tt = 0.f, ts = 0.f; int32_t tri = -1; float n[3] = {0.f, 0.f, 0.f};
    if (res) res->traces++;
    int ht = closest_impl(s, o, d, tmax, 0, ignore, &tt, &tri, n, NULL);
    int hs = sphere_hit(o, d, lis, radius, tmax, &ts);
    if (!ht && !hs) return 0;
    if (hs && (!ht || ts <= tt)) { *t_out = ts; *obj_out = FSO_PAWN_OBJECT; n_out[0] = n_out[1] = n_out[2] = 0.f; return 1; }
    *t_out = tt; *obj_out = s->tris_orig[tri].object;
    n_out[0] = n[0]; n_out[1] = n[1]; n_out[2] = n[2];
    return 1;
}

/* CastDirectAudioRay FSAC.cpp:209-280 (tail recursion written as a loop) */
static float cast_direct(const fso_scene* s, const fso_sound_params* p, const float dir[3], const float start[3],
                         float max_distance, int bounces, float energy, uint32_t direct_hit_actor,
                         const float lis[3], fso_sound_result* res) {
    float pos[3] = {start[0], start[1], start[2]};
    while (1) {
        if (bounces == 0 || energy <= 0.0f) return 0.0f;                   /* FSAC.cpp:212 */
        float o[3];
        for (int k = 0; k < 3; ++k) o[k] = fmaf(dir[k], 0.1f, pos[k]);     /* DirectStart FSAC.cpp:232 */
        float t, n[3]; uint32_t obj;
        if (!legacy_trace(s, o, dir, max_distance, direct_hit_actor, lis, p->listener_radius, &t, &obj, n, res))
            return 0.0f;                                                   /* no hit FSAC.cpp:279 */
        if (obj == FSO_PAWN_OBJECT) {                                      /* FSAC.cpp:253-270 */
            float travel = p->raycast_distance - max_distance + t;
            travel *= 0.01f;
            float time = travel / 343.0f;
            if (time > p->simulated_duration) return 0.0f;
            energy *= expf(-0.0017f * travel);
            return energy;
        }
        /* a different obstacle: continue through it, FSAC.cpp:272-276 */
        for (int k = 0; k < 3; ++k) pos[k] = fmaf(t, dir[k], o[k]);        /* Hit.ImpactPoint */
        max_distance = max_distance - t;
        bounces -= 1;
        direct_hit_actor = obj;
    }
}

/* CastAudioRay FSAC.cpp:132-207 (tail recursion written as a loop) */
static float cast_audio_ray(const fso_scene* s, const fso_sound_params* p, const float dir_in[3],
                            const float start[3], const float lis[3], fso_sound_result* res) {
    float pos[3] = {start[0], start[1], start[2]};
    float dir[3] = {dir_in[0], dir_in[1], dir_in[2]};
    float max_distance = p->raycast_distance;
    int bounces = p->raycast_bounces;
    const float energy = 1.0f;
    while (1) {
        if (bounces == 0 || energy <= 0.0f) return 0.0f;                   /* FSAC.cpp:134 */
        float l2 = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];    /* GetSafeNormal FSAC.cpp:141 */
        float inv = 1.0f / sqrtf(l2);
        float d[3] = {dir[0] * inv, dir[1] * inv, dir[2] * inv};
        float t, n[3]; uint32_t obj;
        if (!legacy_trace(s, pos, d, max_distance, FSO_NO_OBJECT, lis, p->listener_radius, &t, &obj, n, res))
            return 0.0f;                                                   /* FSAC.cpp:192-196 */
        float ip[3];
        for (int k = 0; k < 3; ++k) ip[k] = fmaf(t, d[k], pos[k]);
        float left = max_distance - t;                                     /* DistanceLeft FSAC.cpp:167 */
        float tp[3] = {lis[0] - ip[0], lis[1] - ip[1], lis[2] - ip[2]};
        float dist_to_player = sqrtf(tp[0] * tp[0] + tp[1] * tp[1] + tp[2] * tp[2]);
        float travel_time = (p->raycast_distance - left + dist_to_player) * 0.01f / 343.0f;  /* FSAC.cpp:171 */
        if (travel_time > p->simulated_duration) return 0.0f;
        if (obj == FSO_PAWN_OBJECT) return energy;                         /* FSAC.cpp:177-181 */
        if (dist_to_player > 0.0f) {                                       /* FSAC.cpp:184-185 (result unused) */
            float invp = 1.0f / dist_to_player;
            float dp[3] = {tp[0] * invp, tp[1] * invp, tp[2] * invp};
            float de = cast_direct(s, p, dp, ip, left, 1, energy, FSO_NO_OBJECT, lis, res);
            if (de > 0.0f && res) { res->direct_hits++; res->direct_energy_sum += de; }
        }
        float dn = d[0] * n[0] + d[1] * n[1] + d[2] * n[2];                /* GetReflectionVector FSAC.cpp:186 */
        for (int k = 0; k < 3; ++k) {
            dir[k] = fmaf(-2.0f * dn, n[k], d[k]);
            pos[k] = fmaf(n[k], 0.5f, ip[k]);                              /* FSAC.cpp:187 */
        }
        max_distance = left;
        bounces -= 1;
    }
}

void fso_sound_params_default(fso_sound_params* p) {
    p->seed = 0x5EEDull;
    p->raycasts_per_tick = 1500;      /* FSAC.h:39 */
    p->raycast_bounces = 10;          /* FSAC.h:42 */
    p->raycast_distance = 5000.0f;    /* FSAC.h:45 */
    p->simulated_duration = 1.0f;     /* FSAC.h:136 */
    p->listener_radius = 34.0f;       /* ADefaultPawn collision sphere (engine default, build-owned) */
}

/* the initial direction FMath::VRandCone((0,-1,0), PI, PI) FSAC.cpp:291: with both half angles PI the
 * polar clamp is the identity, i.e. theta = 2 pi U, phi = acos(2V - 1) about the axis (0,-1,0) */
void fso_legacy_direction(uint64_t seed, uint32_t ray, float dir[3]) {
    uint32_t ctr[4] = {ray, 0u, 0u, 0x46533032u /* 'FS02' */};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t r[4];
    fso_philox4x32_10(ctr, key, r);
    float U = fso_u01(r[0]), V = fso_u01(r[1]);
    float x = fmaf(V, 2.0f, -1.0f);
    float sphi = sqrtf(fmaxf(0.0f, fmaf(-x, x, 1.0f)));
    float st, ct;
    fso_sincos2pi(U, &st, &ct);
    const float n[3] = {0.0f, -1.0f, 0.0f};
    float sg = copysignf(1.0f, n[2]);
    float a = -1.0f / (sg + n[2]);
    float b = n[0] * n[1] * a;
    float t0 = fmaf(sg * n[0] * n[0], a, 1.0f), t1 = sg * b, t2 = -sg * n[0];
    float b0 = b, b1 = fmaf(n[1] * n[1], a, sg), b2 = -n[1];
    float lx = sphi * ct, ly = sphi * st;
    float d0 = fmaf(lx, t0, fmaf(ly, b0, x * n[0]));
    float d1 = fmaf(lx, t1, fmaf(ly, b1, x * n[1]));
    float d2 = fmaf(lx, t2, fmaf(ly, b2, x * n[2]));
    float l2 = d0 * d0 + d1 * d1 + d2 * d2;
    float inv = 1.0f / sqrtf(l2);
    dir[0] = d0 * inv; dir[1] = d1 * inv; dir[2] = d2 * inv;
}

/* UpdateSound FSAC.cpp:283-306 */
void fso_update_sound(const fso_scene* s, const fso_sound_params* p, const float src[3], const float lis[3],
                      fso_sound_result* res) {
    memset(res, 0, sizeof(*res));
    float total = 0.0f;
    for (int32_t i = 0; i < p->raycasts_per_tick; ++i) {                   /* FSAC.cpp:289-294 */
        float dir[3];
        fso_legacy_direction(p->seed, (uint32_t)i, dir);
        float e = cast_audio_ray(s, p, dir, src, lis, res);
        total += e;
        if (e > 0.0f) res->rays_reaching_listener++;
    }
    res->total_energy = p->raycasts_per_tick > 0 ? total / (float)p->raycasts_per_tick : 0.0f;  /* FSAC.cpp:294 */
    float dx = lis[0] - src[0], dy = lis[1] - src[1], dz = lis[2] - src[2];  /* FSAC.cpp:296-297 */
    float l2 = dx * dx + dy * dy + dz * dz;
    if (l2 > 0.0f) {
        float inv = 1.0f / sqrtf(l2);
        float d[3] = {dx * inv, dy * inv, dz * inv};
        /* 10 pass-throughs, ignoring the source's own (geometry-less) actor, FSAC.cpp:299 */
        res->occlusion_attenuation = cast_direct(s, p, d, src, p->raycast_distance, 10, 1.0f, FSO_NO_OBJECT, lis, res);
    }
}

/* ------------------------------------------------------------------------------------------- */
/* f1: SaveArrayToFile FSAC.cpp:492-505 / LoadFloatArray FSAC.cpp:454-490                        */
/* ------------------------------------------------------------------------------------------- */
/* FString::SanitizeFloat(v, 1): printf("%f"), trim trailing zeros, keep one fractional digit */
int32_t fso_sanitize_float(double v, char* out, int32_t cap) {
    char t[512];
    if (v == 0.0) v = 0.0;
    snprintf(t, sizeof(t), "%f", v);
    int numeric = t[0] != 0;
    for (int i = 0; t[i]; ++i) {
        char c = t[i];
        if (!((c >= '0' && c <= '9') || c == '.' || ((c == '-' || c == '+') && i == 0))) numeric = 0;
    }
    int len = (int)strlen(t);
    if (numeric) {
        char* dot = strchr(t, '.');
        if (!dot) { strcat(t, ".0"); len += 2; }
        else { while (len > (int)(dot - t) + 2 && t[len - 1] == '0') t[--len] = 0; }
    }
    if (len + 1 > cap) return -1;
    memcpy(out, t, (size_t)len + 1);
    return len;
}

int32_t fso_save_array_to_file(const float* data, int32_t n, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    char buf[512];
    for (int32_t i = 0; i < n; ++i) {
        int32_t len = fso_sanitize_float((double)data[i], buf, (int32_t)sizeof(buf));
        if (i) fputc('\n', f);
        fwrite(buf, 1, (size_t)len, f);
    }
    fclose(f);
    return 0;
}

int32_t fso_load_float_array(const char* path, float* out, int32_t cap) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    int32_t n = 0;
    char line[1024];
    int len = 0, c;
    while (1) {
        c = fgetc(f);
        if (c == '\n' || c == EOF) {
            if (len > 0) {                       /* ParseIntoArray culls empty lines */
                line[len] = 0;
                if (n < cap) out[n] = (float)atof(line);   /* FCString::Atof */
                ++n;
            }
            len = 0;
            if (c == EOF) break;
        } else if (len < (int)sizeof(line) - 1) {
            line[len++] = (char)c;
        }
    }
    fclose(f);
    return n;
}
