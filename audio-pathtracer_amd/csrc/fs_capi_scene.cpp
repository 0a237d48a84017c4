// fs_capi_scene.cpp — geometry and material registration, tree commits (host SAH, device Morton, progressive),
// moving geometry (C ABI: include/frequensee.h; RegisterGeometry / UAcousticMaterial, ARTS.h:99-100, MAT.h:22-33).
#include "fs_context.hpp"

namespace fsi {

// A background build that is no longer wanted: the builder polls the flag and gives up; the thread itself is joined by
// fs_context_destroy (join_refine_threads) — never left running inside a library that may be unloaded.
void cancel_refine(fs_context* ctx) {
    if (ctx->refine) ctx->refine->cancel.store(true, std::memory_order_relaxed);
    ctx->refine.reset();
    ctx->moved_since_refine = false;
}
void wait_refine(const std::shared_ptr<RefineJob>& j) {
    std::unique_lock<std::mutex> l(j->mu);
    j->cv.wait(l, [&] { return j->done; });
}
void join_refine_threads(fs_context* ctx) {
    for (std::thread& t : ctx->refine_threads)
        if (t.joinable()) t.join();
    ctx->refine_threads.clear();
}

// fs_scene_commit_progressive: once the background build has finished, the next call that traces anything swaps its tree
// in — held frames finish first (they were traced through the old tree's arrays), the stream drains, the records are
// uploaded; triangles moved since the snapshot get their current positions and a refit.
int maybe_install_refined(fs_context* ctx) {
    if (!ctx->refine) return FS_OK;
    std::shared_ptr<RefineJob> j = ctx->refine;
    { std::lock_guard<std::mutex> l(j->mu); if (!j->done) return FS_OK; }
    const bool moved = ctx->moved_since_refine;
    ctx->refine.reset();
    ctx->moved_since_refine = false;
    if (j->T != ctx->T || !ctx->committed) return FS_OK;
    ctx->prebuilt = &j->bvh;
    const int rc = fs_scene_commit(ctx);
    ctx->prebuilt = nullptr;
    if (rc) return rc;
    if (moved) {
        const std::vector<float> now = ctx->h_xyz;
        return fs_scene_update_triangles(ctx, 0, ctx->T, now.data());
    }
    return FS_OK;
}

}  // namespace fsi

extern "C" {

// ---- scene -----------------------------------------------------------------------------------------
int fs_scene_set_triangles(fs_context* ctx, const float* xyz, const uint16_t* mat_id, int32_t T) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (T < 0 || (T > 0 && !xyz)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad triangle array");
    if (T > (1 << 28)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "too many triangles");
    for (size_t i = 0; i < 9 * (size_t)T; ++i)
        if (!std::isfinite(xyz[i])) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "non-finite vertex coordinate");
    cancel_refine(ctx);   // a background build of the previous triangle set is of no use any more
    ctx->h_xyz.assign(xyz, xyz + 9 * (size_t)T);
    if (mat_id) ctx->h_mat.assign(mat_id, mat_id + T);
    else ctx->h_mat.assign((size_t)T, (uint16_t)FS_NO_MATERIAL);
    ctx->T = T;
    ctx->committed = false;
    return FS_OK;
}

int fs_scene_set_materials(fs_context* ctx, const float* absorption, const float* transmission,
                           const float* scattering, int32_t M, int32_t B) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (M < 0 || (M > 0 && !absorption) || M >= (int32_t)FS_NO_MATERIAL)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad material table");
    if (B != ctx->cfg.num_bands) return ctx->fail(FS_ERR_SIZE_MISMATCH, "material bands != context num_bands");
    size_t n = (size_t)M * (size_t)B;
    ctx->h_absorption.assign(absorption, absorption + n);
    if (transmission) ctx->h_transmission.assign(transmission, transmission + n); else ctx->h_transmission.clear();
    if (scattering) ctx->h_scattering.assign(scattering, scattering + n); else ctx->h_scattering.clear();
    ctx->M = M;
    ctx->committed = false;
    return FS_OK;
}

// the material table of a committed scene: absorption [M][B] | lobe gains [M][3][B] | lobe probabilities [M][3]
static int upload_materials(fs_context* ctx) {
    const size_t mb = ctx->h_absorption.size() * sizeof(float);
    if (mb) {
        // absorption [M][B] | lobe gains [M][3][B] | lobe probabilities [M][3] (FS_FLAG_MATERIAL_LOBES).  The split is
        // the per-bin rule of ApplyMaterialFD (MaterialAcousticProcessor.cpp:51-72) per band: Refl = 1 - alpha, tau
        // clamped to Refl + tau <= 1, diffuse Refl sigma, specular Refl (1 - sigma), transmitted tau; a lobe is
        // picked with the band mean of its gain over the sum of the three.  No arrays: tau = 0, sigma = 1.
        const int B = ctx->cfg.num_bands, M = ctx->M;
        std::vector<float> table(ctx->h_absorption);
        table.resize((size_t)M * B + (size_t)M * 3 * B + (size_t)M * 3, 0.f);
        float* gain = table.data() + (size_t)M * B;
        float* prob = gain + (size_t)M * 3 * B;
        const bool has_t = ctx->h_transmission.size() == (size_t)M * B, has_s = ctx->h_scattering.size() == (size_t)M * B;
        for (int m = 0; m < M; ++m) {
            float sum[3] = {0.f, 0.f, 0.f};
            for (int b = 0; b < B; ++b) {
                const float alpha = ctx->h_absorption[(size_t)m * B + b];
                float tau = has_t ? ctx->h_transmission[(size_t)m * B + b] : 0.0f;
                const float sigma = has_s ? ctx->h_scattering[(size_t)m * B + b] : 1.0f;
                const float refl = 1.0f - alpha;
                if (refl + tau > 1.0f) tau = 1.0f - refl;
                float g[3] = {refl * sigma, refl * (1.0f - sigma), tau};
                for (int l = 0; l < 3; ++l) {
                    if (!(g[l] > 0.0f)) g[l] = 0.0f;
                    gain[((size_t)m * 3 + l) * B + b] = g[l];
                    sum[l] += g[l];
                }
            }
            float mean[3], tot = 0.0f;
            for (int l = 0; l < 3; ++l) { mean[l] = sum[l] / (float)B; tot += mean[l]; }
            for (int l = 0; l < 3; ++l) prob[(size_t)m * 3 + l] = tot > 0.0f ? mean[l] / tot : (l == 0 ? 1.0f : 0.0f);
        }
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_absorption, table.size() * sizeof(float)));
        FS_HIP(ctx, hipMemcpy(ctx->d_absorption, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return FS_OK;
}

// the kernels' view of the triangle records (Tri48 + normals), derived on the device from the authoring records; `cap`
// = records the arrays must have room for (the fast commit keeps its arrays for the next registration)
static int pack_scene(fs_context* ctx, size_t count, size_t cap) {
    if (!ctx->d_tris48) {
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_tris48, std::max<size_t>(cap, 1) * sizeof(Tri48)));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_tri_nrm, std::max<size_t>(cap, 1) * sizeof(float4)));
    }
    launch_pack_triangles(ctx->d_tris, (int)count, ctx->d_tris48, ctx->d_tri_nrm, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    return FS_OK;
}

// what every kind of commit ends with: the kernels' view of the scene and the stats
// the cooperative traversal's records, derived from the node array as it stands in the stream (behind an upload, a device
// build's refit, a refit of moved triangles)
static int refresh_coop_nodes(fs_context* ctx, bool topology_changed) {
    const size_t n = ctx->bvh.nodes.size();
    const int levels = (int)ctx->bvh.level_begin.size() - 1;
    ctx->coop_info = CoopInfo{};
    ctx->scene.coop_info = &ctx->coop_info;
    if (n == 0 || levels < 1 || levels > kMaxBuildLevels) return FS_OK;
    if (n > ctx->coop_cap) {
        if (ctx->d_coop) { FS_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_coop); }
        ctx->d_coop = nullptr; ctx->coop_cap = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_coop, sizeof(CoopChild) * 4 * n));
        ctx->coop_cap = n;
    }
    if (topology_changed || !ctx->d_coop_levels) {   // the level table and the dense numbering of the even levels (a refit keeps both)
        std::vector<int32_t> tab(2 * (size_t)(kMaxBuildLevels + 2), -1);
        int32_t n16 = 0;
        for (int l = 0; l <= levels; ++l) tab[(size_t)l] = ctx->bvh.level_begin[(size_t)l];
        for (int l = 0; l < levels; ++l) {
            if (l & 1) continue;
            tab[(size_t)(kMaxBuildLevels + 2) + (size_t)l] = n16;
            n16 += ctx->bvh.level_begin[(size_t)l + 1] - ctx->bvh.level_begin[(size_t)l];
        }
        if (!ctx->d_coop_levels) FS_HIP(ctx, hipMalloc((void**)&ctx->d_coop_levels, sizeof(int32_t) * tab.size()));
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_coop_levels, tab.data(), sizeof(int32_t) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));      // (tab is a local; commits drain the stream anyway)
        ctx->coop16_nodes = n16; ctx->coop_levels = levels;
    }
    if ((size_t)ctx->coop16_nodes > ctx->coop16_cap) {
        if (ctx->d_coop16) { FS_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_coop16); }
        ctx->d_coop16 = nullptr; ctx->coop16_cap = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_coop16, sizeof(CoopChild) * 16 * (size_t)ctx->coop16_nodes));
        ctx->coop16_cap = (size_t)ctx->coop16_nodes;
    }
    launch_coop_nodes(ctx->d_nodes, (int)n, ctx->d_coop, ctx->stream);
    launch_coop16(ctx->d_coop, ctx->d_coop_levels, ctx->d_coop_levels + (kMaxBuildLevels + 2), ctx->coop_levels, ctx->coop16_nodes, ctx->d_coop16, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    // one-node-at-a-time descents: 15 new entries per 16-wide level at most; the 4-wide tree's bound is the builder's (+ the popped node's four)
    ctx->coop_info.wide16 = CoopView{ctx->d_coop16, 0, ctx->coop16_nodes, 4, (int16_t)(15 * ((ctx->coop_levels + 1) / 2) + 1)};
    ctx->coop_info.wide4 = CoopView{ctx->d_coop, 0, (int32_t)n, 2, (int16_t)(std::max(ctx->bvh.stack_need, 2) + 4)};
    // The cooperative records hold ABSOLUTE world coordinates rounded outwards to fp16 (fs_refit.hip): conservative at any size,
    // but the fp16 ulp is 8 units at 10^4 and 16 - 32 at a few 10^4 (beyond 65 504 the planes are infinite): on a large map the
    // boxes stop culling and the search degenerates silently.  Such scenes take the lane-private traversal (node-relative 8-bit
    // grids) for their small frames instead: coop_view finds no usable record array.
    if (ctx->amax > kCoopMaxCoordinate) { ctx->coop_info.wide16.rec = nullptr; ctx->coop_info.wide4.rec = nullptr; }
    static const bool dbg_info = std::getenv("FS_DEBUG_SCENE_INFO") != nullptr;   // (what DESIGN.md quotes)
    if (dbg_info)
        std::fprintf(stderr, "[frequensee] tree: %zu 4-wide nodes in %d levels -> %d 16-wide nodes\n", n, levels, ctx->coop16_nodes);
    return FS_OK;
}

static int finish_commit(fs_context* ctx, size_t scene_bytes) {
    ctx->amax = 0.f;
    for (float v : ctx->h_xyz) ctx->amax = std::max(ctx->amax, std::fabs(v));
    ctx->scene.nodes = ctx->d_nodes;
    ctx->scene.tris = ctx->d_tris48;
    ctx->scene.tri_nrm = ctx->d_tri_nrm;
    ctx->scene.absorption = ctx->d_absorption;
    ctx->scene.lobe_gain = ctx->d_absorption ? ctx->d_absorption + (size_t)ctx->M * ctx->cfg.num_bands : nullptr;
    ctx->scene.lobe_prob = ctx->d_absorption ? ctx->scene.lobe_gain + (size_t)ctx->M * 3 * ctx->cfg.num_bands : nullptr;
    ctx->scene.num_nodes = (int32_t)ctx->bvh.nodes.size();
    ctx->scene.num_tris = ctx->T;
    ctx->scene.num_materials = ctx->M;
    // LDS rows of the traversal stack: the tree's worst case + 1 if that is at most stack_rows_cap + 1 rows; else
    // stack_rows_cap rows + one row of counters, and a deep store in HBM for the rare lane that needs more
    // (fs_internal.hpp: DeviceScene.deep).  The bounded stack is what lets four workgroups share a CU.
    const int worst = std::max(ctx->bvh.stack_need, 2) + kStackSlack;
    int deep_rows = 0;
    if (worst <= ctx->stack_rows_cap + 1) {
        ctx->scene.stack_rows = worst; ctx->scene.stack_limit = worst; ctx->scene.stack_attn = 0x7FFFFFFFu;
    } else {
        ctx->scene.stack_rows = ctx->stack_rows_cap + 1; ctx->scene.stack_limit = ctx->stack_rows_cap;
        ctx->scene.stack_attn = (uint32_t)(ctx->stack_rows_cap - 4);
        deep_rows = ((worst + kDeepChunk - 1) / kDeepChunk) * kDeepChunk + kDeepChunk;
    }
    if (deep_rows != ctx->deep.rows) {   // (every commit has drained the stream: nothing reads the old store any more)
        if (ctx->deep.buf) (void)hipFree(ctx->deep.buf);
        for (int32_t* b : ctx->deep.retired) (void)hipFree(b);
        ctx->deep = DeepStore{};
        ctx->deep.rows = deep_rows;
    }
    ctx->scene.stack_worst = traversal_lds_bytes(worst, ctx->cfg.num_bands, ctx->num_bins) <= ctx->lds_limit ? worst : 0;
    ctx->scene.deep = nullptr; ctx->scene.deep_lanes = 0;
    ctx->scene.deep_owner = &ctx->deep;
    ctx->stats.bvh_nodes = (uint32_t)ctx->bvh.nodes.size();
    ctx->stats.triangles = (uint32_t)ctx->T;
    ctx->stats.bvh_stack_need = (uint32_t)ctx->bvh.stack_need;
    ctx->stats.bvh_depth = (uint32_t)ctx->bvh.max_depth;
    ctx->stats.scene_bytes = scene_bytes;
    ctx->scene.coop_info = nullptr;
    { const int cr = refresh_coop_nodes(ctx, true); if (cr) return cr; }
    ctx->committed = true;
    return FS_OK;
}

int fs_scene_commit(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (!ctx->prebuilt) cancel_refine(ctx);
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_scene(ctx);
    ctx->fast_cap_tris = 0;
    // With a communicator attached rank 0 alone builds the acceleration structure and broadcasts it; the
    // other ranks only hold the triangles for later fs_scene_update_triangles bookkeeping.
    RcclApi* ra = ctx->comm ? rccl() : nullptr;
    if (ctx->comm && !ra) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
    const bool bcast = ra != nullptr;
    const bool root = !bcast || ctx->cfg.rank == 0;
    size_t n_nodes = 0, n_tris = 0, n_leaf = 0, n_lvl = 0;
    if (root) {
        if (ctx->prebuilt) ctx->bvh = std::move(*ctx->prebuilt);   // the background build of fs_scene_commit_progressive
        else
            build_bvh(ctx->h_xyz.data(), ctx->h_mat.data(),
                      ctx->h_obj.size() == (size_t)ctx->T && ctx->T > 0 ? ctx->h_obj.data() : nullptr, ctx->T, ctx->bvh);
        n_nodes = ctx->bvh.nodes.size(); n_tris = ctx->bvh.tris.size();
        n_leaf = ctx->bvh.leaf_pos.size(); n_lvl = ctx->bvh.level_begin.size();
    }
    if (bcast) {   // header first: sizes, stack bound, box padding
        int32_t hdr[8] = {(int32_t)n_nodes, (int32_t)n_tris, (int32_t)n_leaf, (int32_t)n_lvl, ctx->bvh.stack_need,
                          ctx->bvh.max_depth, 0, ctx->T};
        std::memcpy(&hdr[6], &ctx->bvh.pad, sizeof(float));
        int32_t* d_hdr = nullptr;
        FS_HIP(ctx, hipMalloc((void**)&d_hdr, sizeof(hdr)));
        if (root) FS_HIP(ctx, hipMemcpyAsync(d_hdr, hdr, sizeof(hdr), hipMemcpyHostToDevice, ctx->stream));
        ncclResult_t r = ra->Broadcast(d_hdr, d_hdr, 8, ncclInt32, 0, ctx->comm, ctx->stream);
        hipError_t e = hipMemcpyAsync(hdr, d_hdr, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_hdr);
        if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclBroadcast(scene header)");
        if (e != hipSuccess) return ctx->hip_fail(e, "scene header");
        if (hdr[7] != ctx->T) return ctx->fail(FS_ERR_SIZE_MISMATCH, "fs_scene_commit: rank 0 committed a different number of triangles");
        if (!root) {
            n_nodes = (size_t)hdr[0]; n_tris = (size_t)hdr[1]; n_leaf = (size_t)hdr[2]; n_lvl = (size_t)hdr[3];
            ctx->bvh = HostBVH{};
            ctx->bvh.stack_need = hdr[4]; ctx->bvh.max_depth = hdr[5];
            std::memcpy(&ctx->bvh.pad, &hdr[6], sizeof(float));
            ctx->bvh.nodes.resize(n_nodes); ctx->bvh.tris.resize(n_tris);   // sizes only; the records live on the device
            ctx->bvh.level_begin.assign(n_lvl, 0);
        }
    }
    if (ctx->bvh.stack_need > kStackDepth) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "BVH needs a deeper traversal stack");
    {   // the traversal kernels keep stack + energy histogram window + work-sharing area in LDS: fail here, with a
        // message, rather than with a launch error on every frame
        const size_t need = traversal_lds_bytes(std::min(std::max(ctx->bvh.stack_need, 2) + kStackSlack, ctx->stack_rows_cap + 1), ctx->cfg.num_bands, ctx->num_bins);
        if (need > ctx->lds_limit)
            return ctx->fail(FS_ERR_INVALID_ARGUMENT, "scene + energy histogram need " + std::to_string(need) +
                             " B of LDS per workgroup, the device offers " + std::to_string(ctx->lds_limit));
    }
    size_t nb = n_nodes * sizeof(NodeQ4), tb = n_tris * sizeof(Tri64);
    size_t mb = ctx->h_absorption.size() * sizeof(float);
    if (nb) {
#if defined(FS_NODE_STRIDE) && FS_NODE_STRIDE != 64   // sensitivity build: one node per 128-B line (this commit path only)
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_nodes, n_nodes * FS_NODE_STRIDE));
        FS_HIP(ctx, hipMemcpy2DAsync(ctx->d_nodes, FS_NODE_STRIDE, ctx->bvh.nodes.data(), sizeof(NodeQ4), sizeof(NodeQ4), n_nodes,
                                     hipMemcpyHostToDevice, ctx->stream));
#else
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_nodes, nb));
        if (root) FS_HIP(ctx, hipMemcpyAsync(ctx->d_nodes, ctx->bvh.nodes.data(), nb, hipMemcpyHostToDevice, ctx->stream));
#endif
        if (bcast) FS_NCCL(ctx, ra->Broadcast(ctx->d_nodes, ctx->d_nodes, nb, ncclUint8, 0, ctx->comm, ctx->stream));
    }
    if (tb) {
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_tris, tb));
        if (root) FS_HIP(ctx, hipMemcpyAsync(ctx->d_tris, ctx->bvh.tris.data(), tb, hipMemcpyHostToDevice, ctx->stream));
        if (bcast) FS_NCCL(ctx, ra->Broadcast(ctx->d_tris, ctx->d_tris, tb, ncclUint8, 0, ctx->comm, ctx->stream));
    }
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    { int mr = upload_materials(ctx); if (mr) return mr; }
    if (tb) {   // refit support: leaf positions, level table and the bounds scratch
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_leaf_pos, sizeof(uint32_t) * std::max<size_t>(n_leaf, 1)));
        if (root) FS_HIP(ctx, hipMemcpyAsync(ctx->d_leaf_pos, ctx->bvh.leaf_pos.data(), sizeof(uint32_t) * n_leaf,
                                             hipMemcpyHostToDevice, ctx->stream));
        if (bcast) {
            FS_NCCL(ctx, ra->Broadcast(ctx->d_leaf_pos, ctx->d_leaf_pos, n_leaf, ncclUint32, 0, ctx->comm, ctx->stream));
            int32_t* d_lvl = nullptr;   // the level table is host data (one refit launch per level): through a device bounce buffer
            FS_HIP(ctx, hipMalloc((void**)&d_lvl, sizeof(int32_t) * std::max<size_t>(n_lvl, 1)));
            if (root) FS_HIP(ctx, hipMemcpyAsync(d_lvl, ctx->bvh.level_begin.data(), sizeof(int32_t) * n_lvl, hipMemcpyHostToDevice, ctx->stream));
            ncclResult_t r = ra->Broadcast(d_lvl, d_lvl, n_lvl, ncclInt32, 0, ctx->comm, ctx->stream);
            hipError_t e = hipSuccess;
            if (!root) e = hipMemcpyAsync(ctx->bvh.level_begin.data(), d_lvl, sizeof(int32_t) * n_lvl, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            (void)hipFree(d_lvl);
            if (r != ncclSuccess) return nccl_fail(ctx, r, "ncclBroadcast(level table)");
            if (e != hipSuccess) return ctx->hip_fail(e, "level table");
        }
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_node_box, sizeof(float4) * 2 * std::max<size_t>(n_nodes, 1)));
    }
    if (tb) { const int pr = pack_scene(ctx, n_tris, n_tris); if (pr) return pr; }
    return finish_commit(ctx, nb + n_tris * (sizeof(Tri48) + sizeof(float4)) + mb);
}

// RegisterGeometry / UnregisterGeometry at run time (ARTS.h:99-100): a changed triangle set needs a new tree NOW.
// The whole build runs on the device (fs_build.hip) behind one upload of the triangles; the host only reads back the
// level table and the stack bound.  Falls back to the host build when the Morton tree comes out too deep.
int fs_scene_commit_fast(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    cancel_refine(ctx);
    if (ctx->T < 1 || ctx->comm) return fs_scene_commit(ctx);   // empty scene / sharded run: the one build rank 0 broadcasts
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int T = ctx->T;
    const size_t n = (size_t)T;
    const bool reuse = ctx->fast_cap_tris >= n && ctx->d_nodes && ctx->d_tris && ctx->d_leaf_pos && ctx->d_node_box;
    if (!reuse) { free_scene(ctx); ctx->fast_cap_tris = 0; }
    else {   // keep the four arrays of the previous fast commit, drop the rest of the old scene
        if (ctx->d_absorption) (void)hipFree(ctx->d_absorption);
        ctx->d_absorption = nullptr;
        ctx->refit_pending = false;
        ctx->committed = false;
    }
    // inputs + scratch in one grow-only device block (hipMalloc costs more than the build)
    const size_t in_bytes = ((sizeof(float) * 9 * n + 255) & ~(size_t)255) + ((sizeof(uint16_t) * n + 255) & ~(size_t)255) +
                            ((sizeof(uint32_t) * n + 255) & ~(size_t)255) + 256;
    const size_t scratch_bytes = device_build_scratch_bytes(T);
    if (in_bytes + scratch_bytes > ctx->build_cap) {
        if (ctx->d_build) (void)hipFree(ctx->d_build);
        ctx->d_build = nullptr; ctx->build_cap = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_build, in_bytes + scratch_bytes));
        ctx->build_cap = in_bytes + scratch_bytes;
    }
    char* q = ctx->d_build;
    float* d_xyz = reinterpret_cast<float*>(q); q += (sizeof(float) * 9 * n + 255) & ~(size_t)255;
    uint16_t* d_mat = reinterpret_cast<uint16_t*>(q); q += (sizeof(uint16_t) * n + 255) & ~(size_t)255;
    uint32_t* d_obj = reinterpret_cast<uint32_t*>(q); q += (sizeof(uint32_t) * n + 255) & ~(size_t)255;
    DeviceBuildInfo* d_info = reinterpret_cast<DeviceBuildInfo*>(q); q += 256 + ((sizeof(DeviceBuildInfo) + 255) & ~(size_t)255);
    const bool has_obj = ctx->h_obj.size() == n;
    FS_HIP(ctx, hipMemcpyAsync(d_xyz, ctx->h_xyz.data(), sizeof(float) * 9 * n, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(ctx, hipMemcpyAsync(d_mat, ctx->h_mat.data(), sizeof(uint16_t) * n, hipMemcpyHostToDevice, ctx->stream));
    if (has_obj) FS_HIP(ctx, hipMemcpyAsync(d_obj, ctx->h_obj.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, ctx->stream));
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, amax = 0.f;
    for (size_t i = 0; i < 9 * n; ++i) {
        const float v = ctx->h_xyz[i];
        lo[i % 3] = std::min(lo[i % 3], v); hi[i % 3] = std::max(hi[i % 3], v);
        amax = std::max(amax, std::fabs(v));
    }
    const size_t tb = n * sizeof(Tri64);
    if (!reuse) {
        const size_t cap = n + n / 4 + 64;   // room for the next registration
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_nodes, cap * sizeof(NodeQ4)));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_tris, cap * sizeof(Tri64)));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_leaf_pos, sizeof(uint32_t) * cap));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_node_box, sizeof(float4) * 2 * cap));
        ctx->fast_cap_tris = cap;
    }
    if (!launch_device_build(d_xyz, d_mat, has_obj ? d_obj : nullptr, T, lo, hi, ctx->d_nodes, ctx->d_tris, ctx->d_leaf_pos, q,
                             (size_t)(ctx->d_build + ctx->build_cap - q), d_info, ctx->stream))
        return fs_scene_commit(ctx);
    FS_HIP(ctx, hipGetLastError());
    DeviceBuildInfo info{};
    FS_HIP(ctx, hipMemcpyAsync(&info, d_info, sizeof(info), hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (info.levels < 1 || info.stack_need > kStackDepth ||
        traversal_lds_bytes(std::min(std::max(info.stack_need, 2) + kStackSlack, ctx->stack_rows_cap + 1), ctx->cfg.num_bands, ctx->num_bins) > ctx->lds_limit)
        return fs_scene_commit(ctx);   // a degenerate Morton tree (deeper than the LDS stack allows): take the host's SAH build
    ctx->bvh = HostBVH{};
    ctx->bvh.nodes.resize((size_t)info.num_nodes); ctx->bvh.tris.resize(n);   // sizes only: the records live on the device
    ctx->bvh.stack_need = info.stack_need;
    ctx->bvh.max_depth = info.levels;
    ctx->bvh.pad = std::max(0.01f, amax * 3.8146973e-06f);                     // as fs_bvh.cpp
    ctx->bvh.level_begin.assign(info.level_begin, info.level_begin + info.levels + 1);
    launch_refit(ctx->d_nodes, ctx->d_tris, ctx->d_node_box, ctx->bvh.level_begin.data(), info.levels, ctx->bvh.pad, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    { int mr = upload_materials(ctx); if (mr) return mr; }
    { const int pr = pack_scene(ctx, n, ctx->fast_cap_tris); if (pr) return pr; }
    return finish_commit(ctx, (size_t)info.num_nodes * sizeof(NodeQ4) + tb + ctx->h_absorption.size() * sizeof(float));
}

// The tree of fs_scene_commit_fast now, the host's SAH tree as soon as it is built (header).
int fs_scene_commit_progressive(fs_context* ctx) {
    const int rc = fs_scene_commit_fast(ctx);
    if (rc) return rc;
    if (ctx->fast_cap_tris == 0) return FS_OK;   // the fast commit took the host's build itself (sharded run, degenerate tree): nothing to refine
    std::shared_ptr<RefineJob> job = std::make_shared<RefineJob>();
    job->xyz = ctx->h_xyz; job->mat = ctx->h_mat;
    if (ctx->h_obj.size() == (size_t)ctx->T) job->obj = ctx->h_obj;
    job->T = ctx->T;
    ctx->refine = job;
    ctx->moved_since_refine = false;
    // a host that re-registers every few frames must not collect thread objects without bound (cancelled builders end
    // within one poll interval)
    if (ctx->refine_threads.size() >= 8) join_refine_threads(ctx);
    ctx->refine_threads.emplace_back([job] {
        build_bvh(job->xyz.data(), job->mat.data(), job->obj.empty() ? nullptr : job->obj.data(), job->T, job->bvh, &job->cancel);
        { std::lock_guard<std::mutex> l(job->mu); job->done = true; }
        job->cv.notify_all();
    });
    return FS_OK;
}

int fs_scene_refine_pending(fs_context* ctx, int32_t* pending) {
    if (!ctx || !pending) return FS_ERR_INVALID_ARGUMENT;
    *pending = ctx->refine ? 1 : 0;
    return FS_OK;
}

int fs_scene_refine_wait(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->refine) return FS_OK;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    wait_refine(ctx->refine);
    return maybe_install_refined(ctx);
}

// ---- moving geometry (row f4): ECC_WorldDynamic movers are seen by the next trace (ARTS.cpp:333-336) -------------
int fs_scene_update_triangles(fs_context* ctx, int32_t first, int32_t count, const float* xyz) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    if (first < 0 || count < 0 || (int64_t)first + count > ctx->T || (count > 0 && !xyz))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "triangle range outside the committed scene");
    if (count == 0) return FS_OK;
    for (size_t i = 0; i < 9 * (size_t)count; ++i)
        if (!std::isfinite(xyz[i])) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "non-finite vertex coordinate");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if ((size_t)count > ctx->move_cap) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_move) (void)hipFree(ctx->d_move);
        ctx->d_move = nullptr; ctx->move_cap = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_move, sizeof(float) * 9 * (size_t)count));
        ctx->move_cap = (size_t)count;
    }
    std::memcpy(ctx->h_xyz.data() + 9 * (size_t)first, xyz, sizeof(float) * 9 * (size_t)count);   // a later commit stays consistent
    if (ctx->refine) ctx->moved_since_refine = true;   // the background tree was built from the old positions
    for (size_t i = 0; i < 9 * (size_t)count; ++i) ctx->amax = std::max(ctx->amax, std::fabs(xyz[i]));
    // the staging buffer may still be read by the previous update's kernel: same stream, so ordered
    FS_HIP(ctx, hipMemcpyAsync(ctx->d_move, xyz, sizeof(float) * 9 * (size_t)count, hipMemcpyHostToDevice, ctx->stream));
    launch_update_triangles(ctx->d_tris, ctx->d_tris48, ctx->d_tri_nrm, ctx->d_leaf_pos, first, count, ctx->d_move, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));   // xyz is the caller's memory
    ctx->refit_pending = true;
    return FS_OK;
}

int fs_scene_refit(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    ctx->refit_pending = false;
    if (ctx->bvh.nodes.empty()) return FS_OK;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const float pad = std::max(std::max(0.01f, ctx->amax * 3.8146973e-06f), ctx->bvh.pad);   // as fs_bvh.cpp; never shrinks
    launch_refit(ctx->d_nodes, ctx->d_tris, ctx->d_node_box, ctx->bvh.level_begin.data(),
                 (int)ctx->bvh.level_begin.size() - 1, pad, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    return refresh_coop_nodes(ctx, false);
}

int fs_scene_set_objects(fs_context* ctx, const uint32_t* object_id, int32_t T) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (object_id && T != ctx->T) return ctx->fail(FS_ERR_SIZE_MISMATCH, "object ids: T != number of triangles");
    if (object_id) ctx->h_obj.assign(object_id, object_id + T);
    else ctx->h_obj.clear();
    ctx->committed = false;
    return FS_OK;
}

}  // extern "C"
