// fs_capi_pipeline.cpp — pipelined frames (fs_set_pipelining / fs_set_frames_per_launch): what a held frame still owes once its
// connect pass has been enqueued, the parts a held frame contributes to a fused launch, and the flush — the pipeline drained
// through fused launches (drain_fused) or, where that does not apply, every held frame on kernels of its own.
// Split out of fs_capi_frame.cpp in round 5; the reconstruct / publish machinery is fs_capi_publish.cpp.
#include "fs_context.hpp"

#include <chrono>

namespace fsi {

// depth = 0 frames: did a walk's record miss both tiers?  (Called where the compute stream has just been synchronised.)
// Then the frame's energy is incomplete: the tier is grown for the next attempt and the caller is told.
int check_overflow(fs_context* ctx) {
    if (!ctx->overflow_armed || !ctx->h_overflow) return FS_OK;
    ctx->overflow_armed = false;
    unsigned flag = *reinterpret_cast<volatile unsigned*>(ctx->h_overflow);   // (pinned host word, the stream has been synchronised)
    if (ctx->comm) {
        // sharded frame: the ranks must agree — a rank that traced the frame again alone would issue one all-reduce more
        // than the others.  Every rank armed the word for the same frames, so every rank gets here: MAX over the ranks
        // (through the context's device staging: the communicator sums device memory).
        RcclApi* a = rccl();
        if (!a) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
        unsigned* d = reinterpret_cast<unsigned*>(ctx->d_comm_stage);
        FS_HIP(ctx, hipMemcpyAsync(d, &flag, sizeof(flag), hipMemcpyHostToDevice, ctx->stream));
        FS_NCCL(ctx, a->AllReduce(d, d, 1, ncclUint32, ncclMax, ctx->comm, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(&flag, d, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (!flag) return FS_OK;
    *reinterpret_cast<volatile unsigned*>(ctx->h_overflow) = 0u;
    // two things can overflow: the second record tier (more walks beyond FS_MAX_DEPTH steps than it has slots) and the
    // lanes of a later stage of a staged walk (more survivors than provisioned) — the retry gets more of both
    ctx->stage_margin = std::min(ctx->stage_margin * 2.0f, 64.0f);
    const uint32_t grown = (uint32_t)std::min<uint64_t>((uint64_t)std::max<uint32_t>(ctx->over_cap, 16) * 4, 1u << 28);
    if (ctx->d_over_np) (void)hipFree(ctx->d_over_np);
    if (ctx->d_over_mat) (void)hipFree(ctx->d_over_mat);
    if (ctx->d_over_pos) (void)hipFree(ctx->d_over_pos);
    ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr; ctx->d_over_pos = nullptr; ctx->over_cap_pos = 0;
    ctx->over_cap = grown;   // ensure_state allocates at this size next time
    return ctx->fail(FS_ERR_OVERFLOW, "depth = 0: more walks than expected outlived " + std::to_string(FS_MAX_DEPTH) +
                     " steps; the record tier has been grown — trace the frame again");
}

// What a held frame still owes once its connect pass has been enqueued: the fixed-point -> fp32 rounding, the sum over
// the ranks, and the reconstruct the caller asked for in the meantime.  The source may already have moved on to later
// frames (cur rotated): the per-frame fields are switched back for the duration.
int finish_held_frame(fs_context* ctx, const fs_context::PipeFrame& q, bool may_defer_recon, bool tail_waits_already, bool draining) {
    // the items of a frame were connected by ONE launch: once the tail stream waits behind it (the first item's handoff),
    // the other items' reconstructs are ordered too — no further event pairs on the compute stream (each is a bubble
    // between its launches).  Not in deterministic mode: every item's fixed-point rounding runs on the compute stream first.
    // (tail_waits_already: the publishes of this very launch's reconstruct parts made the tail stream wait behind it)
    bool tail_behind_launch = tail_waits_already;
    const bool flushing = !may_defer_recon || draining;   // called by flush_pending: nothing will be launched behind this frame that its reconstructs could overlap
    // the reconstructs of a frame ride in the next launch all together or not at all (a tail-stream reconstruct in between
    // would have to run the deferred ones first, to keep the IRs in frame order)
    int wanted = 0;
    for (const fs_context::PipeFrame::Item& it : q.items) wanted += it.want_recon ? 1 : 0;
    if (wanted + (int)ctx->recon_owed.size() > kMaxReconParts) may_defer_recon = false;   // (256: one table slot)
    for (const fs_context::PipeFrame::Item& it : q.items) {
        Source* s = it.s;
        const bool moved_on = s->cur != it.cur;
        const int cur = s->cur;
        const bool cur_fixed = s->cur_fixed, reduced = s->reduced, handed_off = s->handed_off, tail_ordered = s->tail_ordered;
        s->cur = it.cur; s->cur_fixed = q.fixed; s->reduced = false; s->handed_off = false;
        s->tail_ordered = tail_behind_launch && !q.fixed;
        int rc = FS_OK;
        if (q.fixed) launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), ctx->cfg.num_bands * ctx->num_bins, ctx->stream);
        if (ctx->comm) { rc = reduce_energy(ctx, s); if (!rc && s->tail_ordered) tail_behind_launch = true; }
        if (!rc && it.want_recon) {
            // plain reconstruct: it rides in the next fused launch (fs_context::recon_owed) — on one GPU; with the library's
            // collective, in the launch after next, behind the all-reduce just enqueued on the tail stream
            const bool single = !ctx->comm && ctx->cfg.world_size == 1;
            const bool summed = ctx->comm != nullptr && ctx->fused_recon_comm && s->reduced && s->red_recorded[it.cur];
            if (may_defer_recon && ctx->fused_recon && (single || summed) && ctx->profiling < 2 &&
                !(it.recon.flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT)) {
                fs_context::ReconOwed o; o.s = s; o.cur = it.cur; o.fixed = q.fixed; o.p = it.recon; o.reduced = summed;
                ctx->recon_owed.push_back(o);
            } else {
                // (what cannot ride in a fused launch: a literal second flush, per-kernel timing, a sharded frame without the library's
                // collective, more than a table slot's 256 reconstructs at once — on the tail stream beside the next launch, or, when
                // nothing will be launched behind it, through flush_reconstruct.  Until round 5 a launch had four reconstruct parts and
                // cfg5's eight sources per frame took this path in steady state.)
                rc = flushing ? flush_reconstruct(ctx, s, &it.recon) : reconstruct_now(ctx, s, &it.recon);
                if (!rc && s->tail_ordered) tail_behind_launch = true;
            }
        }
        if (moved_on) { s->cur = cur; s->cur_fixed = cur_fixed; s->reduced = reduced; s->handed_off = handed_off; s->tail_ordered = tail_ordered; }
        if (rc) return rc;
    }
    return FS_OK;
}

void held_connect_part(const fs_context::PipeFrame& q, FrameParts& f) {
    const fs_context::PipeFrame::Item& it = q.items[0];
    f.has_connect = true; f.kpc = q.kp; f.stc = q.st; f.energy = it.s->d_energy[it.cur];
    f.fixed = q.fixed ? it.s->d_fixed[it.cur] : nullptr; f.scratch_c = q.wl.queue_head; f.ppw = q.ppw;
    f.energy_tab = q.energy_tab; f.fixed_tab = q.fixed_tab;
}

// rays per wave of a walk stage: by the number of walks it still has and the steps they have left at most
WalkLaunch stage_launch(const fs_context* ctx, const fs_context::PipeFrame& q, int stage) {
    WalkLaunch wl = q.wl;
    const WalkStage& sr = q.stages[(size_t)stage];
    // inside a fused launch the chip is full: dense waves for every stage that still has a few thousand walks, sparse
    // waves (the other lanes help with every query) only for the few long walks of the late stages, whose chain of
    // dependent bounces is what matters (profiles/r03_stage_sweep.log: the stand-alone frames' rule — ~4096 sparse waves
    // for mid-size frames — costs 0.87 instead of 0.63 ms per frame here; raising the late stages' wave priority: nothing)
    if (sr.begin > 0 && ctx->walk_rays_per_wave <= 0)
        wl.rays_per_wave = walk_stage_slots(q.kp, sr.begin) >= (uint32_t)ctx->stage_dense_from ? 64 : 16;
    return wl;
}

// the next stage of a held frame's walk as a part of the fused launch; false: the launch has no room for more walk parts
bool held_walk_part(const fs_context* ctx, const fs_context::PipeFrame& q, FrameParts& f) {
    if (f.num_walk >= kMaxWalkParts) return false;
    WalkPart& w = f.walk[f.num_walk++];
    w.kp = q.kp; w.st = q.st; w.wl = stage_launch(ctx, q, q.next_stage); w.perm = q.perm; w.stage = q.stages[(size_t)q.next_stage];
    return true;
}

// A flush on one GPU drains the pipeline through the SAME fused launches the stream of frames uses, only without a newest frame:
// every launch carries the next walk stage of every held frame, the connect pass of the oldest complete one and the reconstructs
// owed by the launch before — the passes keep overlapping each other and the IRs are published by the launches themselves.
// (Until round 5 every held frame finished on kernels of its own, one after the other: walk, connect, reconstruct, walk, ... —
// six kernels in a row behind a stream of cfg3 frames, 0.4 ms of the driver's 5.9 ms timed region.)
static int drain_fused(fs_context* ctx) {
    const int B = ctx->cfg.num_bands;
    bool counted = false;
    for (int guard = 0; (!ctx->held.empty() || !ctx->recon_owed.empty()) && guard < 8 * (kMaxWalkParts + 4); ++guard) {
        if (!counted && !ctx->held.empty()) { ctx->dbg.flushes++; ctx->dbg.flushed_frames += ctx->held.size(); counted = true; }
        FrameParts fp;
        bool connects = false;
        std::vector<size_t> advanced;
        for (size_t k = 0; k < ctx->held.size(); ++k) {
            fs_context::PipeFrame& q = ctx->held[k];
            if (q.next_stage < (int)q.stages.size()) {
                if (held_walk_part(ctx, q, fp)) advanced.push_back(k);
            } else if (k == 0 && !connects) {
                held_connect_part(q, fp);
                connects = true;
            }
        }
        OwedLaunch owed;
        { const int orc = owed_prepare(ctx, fp, owed); if (orc) return orc; }
        if (!(fp.num_walk > 0 || fp.has_connect || fp.num_recon > 0)) {
            // nothing a launch could carry.  With a communicator the reconstructs of the frames just summed become due one round
            // later (ReconOwed::age, raised by owed_prepare): go round again; anything else is left to the loop in flush_pending.
            if (!ctx->recon_owed.empty() && ctx->recon_owed.front().reduced && ctx->recon_owed.front().age >= 1) continue;
            break;
        }
        const bool fused = launch_frame(B, ctx->scene, fp, ctx->stream);
        ctx->dbg.launches++;
        if (!fused) {   // no fused form: the same passes one after the other
            if (fp.has_connect) launch_connect(B, ctx->scene, fp.kpc, fp.stc, fp.energy, fp.fixed, fp.scratch_c, fp.ppw, fp.energy_tab, fp.fixed_tab, ctx->stream);
            for (int i = 0; i < fp.num_walk; ++i)
                launch_walk(ctx->scene, fp.walk[i].kp, fp.walk[i].st, fp.walk[i].wl, fp.walk[i].perm, ctx->stream, fp.walk[i].stage);
        }
        FS_HIP(ctx, hipGetLastError());
        { const int prc = owed_publish(ctx, owed, fused); if (prc) return prc; }
        for (size_t k : advanced) ctx->held[k].next_stage++;
        if (connects) {
            const fs_context::PipeFrame done = ctx->held.front();
            ctx->held.pop_front();
            const int rc = finish_held_frame(ctx, done, /*may_defer_recon=*/true, false, /*draining=*/true);
            if (rc) return rc;
        }
    }
    return FS_OK;
}

int flush_pending(fs_context* ctx) {
    using clk = std::chrono::steady_clock;
    const bool dbg = ctx->debug_stalls;
    clk::time_point t0, t1, t2;
    if (dbg) t0 = clk::now();
    if (!ctx->group.empty()) { const int gr = dispatch_group(ctx); if (gr) return gr; }
    if (dbg) t1 = clk::now();
    // (one GPU, or the pairs of a frame shared between ranks with the LIBRARY's collective: the sums over the ranks go onto the
    // tail stream as before, the reconstructs ride behind them in the drain's launches and publish through the host word)
    const bool drainable = (!ctx->comm && ctx->cfg.world_size == 1) || (ctx->comm != nullptr && ctx->fused_recon_comm);
    if (ctx->fused_drain && ctx->fused_recon && drainable && ctx->profiling < 2 &&
        (!ctx->held.empty() || !ctx->recon_owed.empty())) {
        FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
        const int dr = drain_fused(ctx);
        if (dr) return dr;
    }
    { const int orc = run_owed_reconstructs(ctx); if (orc) return orc; }   // (older than every held frame)
    if (dbg) t2 = clk::now();
    if (ctx->held.empty()) return FS_OK;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ctx->dbg.flushes++; ctx->dbg.flushed_frames += ctx->held.size();
    long us_launch = 0, us_finish = 0;
    while (!ctx->held.empty()) {
        const fs_context::PipeFrame q = ctx->held.front();
        ctx->held.pop_front();
        clk::time_point a, b, c;
        if (dbg) a = clk::now();
        for (int k = q.next_stage; k < (int)q.stages.size(); ++k)
            launch_walk(ctx->scene, q.kp, q.st, stage_launch(ctx, q, k), q.perm, ctx->stream, q.stages[(size_t)k]);
        launch_connect(ctx->cfg.num_bands, ctx->scene, q.kp, q.st, q.items[0].s->d_energy[q.items[0].cur],
                       q.fixed ? q.items[0].s->d_fixed[q.items[0].cur] : nullptr, q.wl.queue_head, q.ppw, q.energy_tab, q.fixed_tab,
                       ctx->stream);
        FS_HIP(ctx, hipGetLastError());
        if (dbg) b = clk::now();
        const int rc = finish_held_frame(ctx, q);
        if (dbg) { c = clk::now(); us_launch += (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); us_finish += (long)std::chrono::duration_cast<std::chrono::microseconds>(c - b).count(); }
        if (rc) return rc;
    }
    if (dbg) {
        const long total = (long)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count();
        if (total > 1000)
            std::fprintf(stderr, "[frequensee] flush %ld us: group %ld, owed reconstructs %ld, launches of held frames %ld, their reconstructs %ld\n", total,
                         (long)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count(), (long)std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count(),
                         us_launch, us_finish);
    }
    return FS_OK;
}

}  // namespace fsi
