// fs_capi_frame.cpp — sources and the traced frame (UpdateSource, ARTS.cpp:128-195): describe / resources / commit / launch,
// grouped frames, and the entry points of the hot call (C ABI: include/frequensee.h).  The pipeline around it (held frames, the
// flush) is fs_capi_pipeline.cpp, reconstruct + publish fs_capi_publish.cpp.
#include "fs_context.hpp"

// ---- one traced frame ----------------------------------------------------------------------------------------------
// UpdateSource up to the deposit (ARTS.cpp:128-173) for `count` sources (count == 1: the plain call).  A batch lays the
// sources' pairs end to end in one plan / walk / connect sequence: every source gets exactly the pairs, random streams
// and therefore results of its own fs_compute_energy_response_async call, but the chip sees one large frame instead
// of `count` small ones.  The work is cut into four steps so that nothing of the context has rotated when a step that
// can fail returns early:
//   frame_describe   host only: the kernels' constants, the frame's mode
//   frame_resources  everything that allocates, waits or copies: state arrays, fixed-point buffers, batch tables —
//                    works on the buffers the frame WILL use, without making them current
//   frame_commit     the rotation: frame index, the sources' current energy buffer, batch table slot (cannot fail)
//   frame_launch     the passes (or the fused launch of pipelined frames) and the bookkeeping
namespace {

// energy buffer b of the source exists (those beyond kEnergyBufsBase are allocated when the rotation first reaches them)
int ensure_energy_buffer(fs_context* ctx, Source* s, int b) {
    if (s->d_energy[b]) return FS_OK;
    const size_t eb = sizeof(float) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    FS_HIP(ctx, hipMalloc((void**)&s->d_energy[b], eb));
    FS_HIP(ctx, hipMemsetAsync(s->d_energy[b], 0, eb, ctx->stream));
    if (!s->ev_rec[b]) FS_HIP(ctx, hipEventCreateWithFlags(&s->ev_rec[b], kDeviceEventFlags));
    if (!s->ev_red[b]) FS_HIP(ctx, hipEventCreateWithFlags(&s->ev_red[b], kDeviceEventFlags));
    return FS_OK;
}

struct Frame {
    Source* const* srcs = nullptr;
    int count = 0;
    const fs_params* p = nullptr;
    const fs_context::GroupEntry* group = nullptr;   // grouped frames: one entry per item (own seed, recorded reconstruct)
    // frame_describe
    KParams kp{};
    int B = 1, levels = 0;
    bool batch = false, unbounded = false, all_conn = false, mis = false, fixed = false, accumulate = false, pipe_ok = false;
    std::vector<WalkStage> stages;      // the walk in one piece, or the stages of a pipelined depth = 0 frame
    int lane_len = 0;                   // a waited-for staged frame: walks of this many steps or more take the long-walk lane (0: none)
    // frame_resources
    unsigned fidx = 0;                  // the frame's index: which state / schedule / scratch set it uses
    SubpathState st{};
    unsigned* scratch = nullptr;
    uint32_t* perm_buf = nullptr;
    int next_cur[64];                   // energy buffer each source deposits into (batches above 64 sources: heap)
    std::vector<int> next_cur_heap;
    int* cur_of = nullptr;
    float* const* energy_tab = nullptr; // batched frame: device tables
    unsigned long long* const* fixed_tab = nullptr;
    bool used_batch_slot = false;
};

// The long-walk lane of a waited-for staged frame and, with the default bound, where its first stage ends (measured:
// tools/lane_sweep.py, profiles/r05_lane_sweep*.jsonl — ticks of 16 ... 128 reference sources and the 262 144-ray frame on both scenes).
//   len: the lane holds about 800 walks (three cooperative waves per CU beside the first stage's): n * rr^len = 800;
//   bound: the first stage ends 18 steps (at rr = 0.9; scaled by log 0.9 / log rr otherwise) before the lane begins, within
//          20 ... 34 — a small frame's survivors are few and cheap, a large frame's are a throughput problem of their own.
// Lengths live in the plan pass's buckets (<= FS_MAX_DEPTH): a roulette so close to 1 that 800 walks outlive them gets no lane.
constexpr int kSyncStageFromWithoutLane = 16384;   // (round 4's threshold: without a lane a 16 000-subpath frame does better in one piece)
static bool sync_lane_plan(const fs_context* ctx, const KParams& kp, int* len, int* bound) {
    *len = 0; *bound = 0;
    if (ctx->sync_lane_len == 0 || !kp.russian_roulette || !(kp.rr_prob > 0.0f && kp.rr_prob < 1.0f)) return false;
    if (ctx->sync_lane_len > 0) { *len = ctx->sync_lane_len; return true; }
    const double n = 2.0 * (double)kp.num_local, lq = std::log(1.0 / (double)kp.rr_prob), s = 0.10536 / lq;   // s: steps per step of rr = 0.9
    if (n < 4096.0) return false;
    const double L = std::log(n / 800.0) / lq;
    if (L > (double)FS_MAX_DEPTH || L < 8.0) return false;
    const int b = (int)std::lround(std::min(std::max(L - 18.0 * s, 20.0 * s), 34.0 * s));
    if (b < 2 || (double)b + 4.0 > L) return false;
    *len = (int)std::lround(L); *bound = b;
    return true;
}

void frame_describe(fs_context* ctx, Frame& f) {
    const fs_params* p = f.p;
    Source* s = f.srcs[0];
    f.batch = f.count > 1;
    f.B = ctx->cfg.num_bands;
    const uint64_t P = p->num_rays / 2;
    uint32_t p0 = 0, pn = 0;
    (void)fs_shard_range(p->num_rays, ctx->cfg.rank, ctx->cfg.world_size, &p0, &pn);   // validated by check_params / create
    KParams& kp = f.kp;
    kp.seed_lo = (uint32_t)p->seed;
    kp.seed_hi = (uint32_t)(p->seed >> 32);
    kp.pair_begin = p0;
    kp.pairs_per_source = pn;
    kp.num_local = kp.pairs_per_source * (uint32_t)f.count;
    kp.src_table = nullptr;
    kp.item_seeds = 0;
    if (f.group) {   // (at most 4 items, all with the same high seed word: dispatch_group)
        kp.item_seeds = f.count;
        for (int i = 0; i < f.count && i < 4; ++i) kp.item_seed[i] = (uint32_t)f.group[i].p.seed;
    }
    // depth = 0: no cap, like the reference's while (true) (ARTS.cpp:294) — the roulette ends every walk; the records of
    // steps beyond FS_MAX_DEPTH go to the second tier.  Without roulette an uncapped walk would never end: FS_MAX_DEPTH.
    f.unbounded = p->depth == 0 && p->russian_roulette && p->rr_prob < 1.0f;
    f.levels = p->depth > 0 ? p->depth : FS_MAX_DEPTH;
    kp.depth = f.unbounded ? FS_MAX_DEPTH + kOverLevels : f.levels;
    kp.mis_depth = f.unbounded ? kUnboundedDepth : f.levels;
    kp.russian_roulette = p->russian_roulette ? 1 : 0;
    kp.cosine = (p->flags & FS_FLAG_COSINE_SAMPLING) ? 1 : 0;
    kp.rr_prob = p->rr_prob;
    kp.stage_margin = ctx->stage_margin;
    kp.max_trace_dist = p->max_trace_dist;
    kp.surface_offset = p->surface_offset;
    kp.connect_pullback = p->connect_pullback;
    kp.dist_divisor = p->dist_divisor;
    kp.min_seg = p->min_seg;
    kp.prob_exponent = p->prob_exponent;
    kp.energy_clamp = p->energy_clamp;
    kp.energy_gain = p->energy_gain;
    kp.sound_speed = p->sound_speed;
    kp.norm = (p->flags & FS_FLAG_FIXED_NORM_1000) ? 1.0f / 1000.0f : (P ? 1.0f / (float)P : 0.f);  // ARTS.cpp:164
    for (int b = 0; b < FS_MAX_BANDS; ++b) kp.air[b] = p->air_absorption[b];
    std::memcpy(kp.src, f.group ? f.group[0].pos : s->pos, sizeof(kp.src));
    std::memcpy(kp.lis, f.group ? f.group[0].lis : ctx->listener, sizeof(kp.lis));   // (grouped frames: the positions at their calls)
    kp.count = ctx->profiling >= 3 ? 1 : 0;
    kp.num_bins = ctx->num_bins;
    kp.num_bands = f.B;
    kp.hist_window = std::min(ctx->num_bins, ctx->hist_window);
    kp.lobes = (p->flags & FS_FLAG_MATERIAL_LOBES) ? 1 : 0;
    kp.dpos = (p->flags & FS_FLAG_DOUBLE_POSITIONS) ? 1 : 0;
    kp.listener_radius = p->listener_radius;
    kp.source_radius = p->source_radius;
    kp.src_object = s->object; kp.lis_object = ctx->listener_object;   // AddIgnoredActor ARTS.cpp:322-327
    kp.ignore_on = ctx->listener_object != FS_NO_OBJECT ? 1 : 0;
    for (int i = 0; i < f.count; ++i) if (f.srcs[i]->object != FS_NO_OBJECT) kp.ignore_on = 1;
    f.mis = (p->flags & FS_FLAG_MIS_BALANCE) != 0;
    f.all_conn = f.mis || (p->flags & FS_FLAG_ALL_CONNECTIONS) != 0;
    kp.mis = f.mis ? 1 : 0;
    f.fixed = (p->flags & FS_FLAG_DETERMINISTIC) != 0;
    f.accumulate = (p->flags & FS_FLAG_ACCUMULATE_ENERGY) != 0;
    // Pipelined frames: this frame's passes are held back (to be launched with the next frames') when the frame has the
    // default shape; any other frame first lets the held-back ones finish on their own.  A depth = 0 frame is held at
    // pipeline depth 2 only, as a STAGED walk: its longest walk is a chain of ~ log(subpaths) / log(1 / rr) dependent
    // bounces (118 at 262 144 subpaths) while 97 % of the walks end within 32 — launch s + 1 of the frame walks steps
    // [bound[s - 1], bound[s]) of the walks still alive, next to the other stages of the frames around it, so that every
    // launch carries one frame's worth of work and no chain longer than a stage.
    const bool plain = !(p->flags & (FS_FLAG_MATERIAL_LOBES | FS_FLAG_MIS_BALANCE | FS_FLAG_ALL_CONNECTIONS | FS_FLAG_ACCUMULATE_ENERGY |
                                     FS_FLAG_DOUBLE_POSITIONS));
    // (a walk that ignores the actor it starts from — what the reference's GeneratePath always does, ARTS.cpp:322-327 — is held
    // like any other: the fused launch has a flavour whose walk parts carry the ignored actor, fs_frame_ext.hip)
    f.pipe_ok = ctx->pipelining > 0 && ctx->profiling < 2 && plain && !(p->listener_radius > 0.0f || p->source_radius > 0.0f) &&
                (p->depth > 0 || (f.unbounded && ctx->pipelining >= 2));
    kp.plan_coop = (2ull * kp.num_local <= kPlanCoopMax || (f.unbounded && !f.pipe_ok && 2ull * kp.num_local <= kPlanCoopMaxUncapped)) ? 1 : 0;
    f.stages.clear();
    f.lane_len = 0;
    if (f.pipe_ok && f.unbounded && !ctx->stage_bounds.empty()) {
        int begin = 0;
        // launches of two or more frames have thicker late stages: fewer, longer stages measure 3 % faster there
        // (profiles/r03_stage_sweep.log, last sweep: 0.595 -> 0.571 ms per frame at two frames per launch)
        // (two per launch: 0.595 -> 0.571 ms per frame; four per launch: 0.559 -> 0.544 ms with one stage fewer still)
        // (end of round 3, streams of 600 frames: 16, 36, 64, 96 also at two per launch — 505 -> 509 M rays/s through bench.py,
        // 481 -> 488 M through tools/stage_sweep.py; 12, 24, 40, 64, 96 until then; profiles/r03_unbounded_600_frames.log)
        static const std::vector<int> kGroupedStageBounds = {16, 36, 64, 96};
        const std::vector<int>& bounds = !(ctx->stage_bounds_default && f.group && f.count >= 2) ? ctx->stage_bounds : kGroupedStageBounds;
        for (int bound : bounds) { WalkStage sr; sr.begin = begin; sr.end = bound; f.stages.push_back(sr); begin = bound; }
        WalkStage last; last.begin = begin; last.end = 1 << 30;
        f.stages.push_back(last);
    } else if (!f.pipe_ok && f.unbounded && plain && ctx->profiling < 3 &&
               // (an ignored actor or end-point spheres stage too: the stage kernels' EXT instantiations — tests/test_round5.py)
               !ctx->sync_stage_bounds.empty() && 2ull * kp.num_local >= (unsigned long long)ctx->sync_stage_from &&
               (2ull * kp.num_local >= (unsigned long long)kSyncStageFromWithoutLane || !ctx->sync_stage_from_default ||
                ((ctx->coop_info.wide16.rec != nullptr || ctx->coop_info.wide4.rec != nullptr) && ctx->sync_lane_len != 0 && ctx->sync_stage_bounds_default))) {
        // A depth = 0 frame that is waited for (not held): the same stages, one launch after the other on the stream.  The
        // frame's time is its longest walk — a chain of ~ log(subpaths) / log(1 / rr) dependent bounces — and what a bounce
        // costs depends on who shares the wave: the first stage walks everybody on dense waves (that is where the work is),
        // the few survivors of every later stage get waves of their own whose idle lanes search with them.
        int begin = 0, lane_bound = 0;
        std::vector<int> own_bound;
        // (no cooperative view of the tree — a map beyond kCoopMaxCoordinate —, no lane: the bound stays where it does best without one)
        const bool lane = (ctx->coop_info.wide16.rec != nullptr || ctx->coop_info.wide4.rec != nullptr) && sync_lane_plan(ctx, kp, &f.lane_len, &lane_bound);
        if (!lane) f.lane_len = 0;
        if (lane && ctx->sync_stage_bounds_default && lane_bound > 0) own_bound.push_back(lane_bound);
        for (int bound : own_bound.empty() ? ctx->sync_stage_bounds : own_bound) { WalkStage sr; sr.begin = begin; sr.end = bound; f.stages.push_back(sr); begin = bound; }
        WalkStage last; last.begin = begin; last.end = 1 << 30;
        f.stages.push_back(last);
    } else {
        f.stages.push_back(WalkStage());
    }
}

int frame_resources(fs_context* ctx, Frame& f) {
    KParams& kp = f.kp;
    const int B = f.B, count = f.count;
    // a frame is in flight for (walk stages + 2) launches; the frames behind it use the other sets (ensure_state lets the
    // held frames finish before it reallocates anything)
    const bool staged = f.stages.size() > 1;
    int rc = ensure_state(ctx, kp.num_local, f.levels, f.unbounded, (double)f.p->rr_prob, f.all_conn, f.mis,
                          (int)f.stages.size() + 2, staged);
    if (rc) return rc;
    f.fidx = ctx->frame_index;   // consecutive frames rotate through the state / schedule / scratch sets
    const size_t set = f.fidx % (unsigned)ctx->state_sets;
    SubpathState& st = f.st;
    st = ctx->st;
    st.end_pos += set * ctx->cap_lanes; st.end_misc += set * ctx->cap_lanes; st.slot_of += set * ctx->cap_lanes;
    st.seg_np += set * ctx->cap_seg; st.seg_mat += set * ctx->cap_seg;
    st.cont_a = staged ? ctx->d_cont + 2 * set * ctx->cap_lanes : nullptr;
    st.cont_b = staged ? st.cont_a + ctx->cap_lanes : nullptr;
    st.end_posd = nullptr;
    if (kp.dpos) {   // (never held: the stream is only drained when the array must grow)
        const size_t lanes = 2 * (size_t)kp.num_local;
        if (lanes > ctx->cap_posd) {
            FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_end_posd) (void)hipFree(ctx->d_end_posd);
            ctx->d_end_posd = nullptr; ctx->cap_posd = 0;
            FS_HIP(ctx, hipMalloc((void**)&ctx->d_end_posd, sizeof(double) * 3 * lanes));
            ctx->cap_posd = lanes;
        }
        st.end_posd = ctx->d_end_posd;
    }
    f.scratch = ctx->walk.queue_head + (size_t)(f.fidx % kScratchSets) * kScratchAllocWords;
    f.perm_buf = ctx->walk.perm ? ctx->walk.perm + set * ctx->perm_words : nullptr;
    st.seg_pos = f.all_conn ? ctx->d_seg_pos : nullptr;
    st.seg_nrm = f.mis ? ctx->d_seg_pos + (size_t)f.levels * 2 * (size_t)kp.num_local : nullptr;
    st.main_levels = f.levels;
    st.over_levels = f.unbounded ? kOverLevels : 0;
    st.over_cap = f.unbounded ? ctx->over_cap : 0;
    st.over_np = ctx->d_over_np ? ctx->d_over_np + set * (size_t)kOverLevels * ctx->over_cap : nullptr;
    st.over_mat = ctx->d_over_mat ? ctx->d_over_mat + set * (size_t)kOverLevels * ctx->over_cap : nullptr;
    st.over_pos = f.all_conn ? ctx->d_over_pos : nullptr;
    st.over_nrm = f.mis && ctx->d_over_pos ? ctx->d_over_pos + (size_t)kOverLevels * ctx->over_cap : nullptr;
    st.overflow = ctx->d_overflow;

    if (count > 64) { f.next_cur_heap.resize((size_t)count); f.cur_of = f.next_cur_heap.data(); } else f.cur_of = f.next_cur;
    // A frame owns its energy buffer from the launch that plans it to the launch that reconstructs it; the rotation must not
    // come round to a buffer whose frame is still on its way (held: planned / walking / waiting for its connect pass;
    // recon_owed: connected, its reconstruct rides in a later launch).  Up to two frames per launch the first
    // kEnergyBufsBase buffers are always enough; three and four per launch rotate through all kEnergyBufs.
    const int ring = ctx->frames_per_launch > 2 ? kEnergyBufs : kEnergyBufsBase;
    auto owned = [&](const Source* s, int b, int upto) {
        for (const fs_context::PipeFrame& q : ctx->held)
            for (const fs_context::PipeFrame::Item& it : q.items) if (it.s == s && it.cur == b) return true;
        for (const fs_context::ReconOwed& o : ctx->recon_owed) if (o.s == s && o.cur == b) return true;
        for (int k = 0; k < upto; ++k) if (f.srcs[k] == s && f.cur_of[k] == b) return true;
        return false;
    };
    for (int i = 0; i < count; ++i) {
        Source* si = f.srcs[i];
        // this frame deposits into the next free buffer of the rotation; the tail may still be busy with it.  (FS_FLAG_ACCUMULATE_ENERGY
        // stays in the buffer of the previous frame — behind its reduce / reconstruct — and adds to what it holds.)
        int b = si->cur;
        if (!f.accumulate) {
            int from = si->cur;   // grouped frames: the same source may own several items — the buffers behind its previous item's
            for (int k = 0; k < i; ++k) if (f.srcs[k] == si) from = f.cur_of[k];
            b = (from + 1) % ring;
            int tries = 0;
            while (tries < ring && owned(si, b, i)) { b = (b + 1) % ring; ++tries; }
            if (tries == ring) {   // every buffer belongs to a frame in flight: those finish on their own kernels first
                FS_FLUSH(ctx);
                b = (from + 1) % ring;
                for (tries = 0; tries < ring && owned(si, b, i); ++tries) b = (b + 1) % ring;
                if (tries == ring) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "more frames of one source in a launch than it has energy buffers");
            }
        }
        { const int er = ensure_energy_buffer(ctx, si, b); if (er) return er; }
        if (f.fixed && !si->d_fixed[b]) {
            FS_HIP(ctx, hipMalloc((void**)&si->d_fixed[b], sizeof(unsigned long long) * (size_t)B * (size_t)ctx->num_bins));
            FS_HIP(ctx, hipMemsetAsync(si->d_fixed[b], 0, sizeof(unsigned long long) * (size_t)B * (size_t)ctx->num_bins, ctx->stream));
        }
        f.cur_of[i] = b;
        FS_HIP(ctx, wait_energy_readers(ctx, si, b));
    }
    if (f.batch) {
        // per-frame tables in one pinned staging block: energy pointers [count] | fixed-point buffer pointers [count] |
        // source positions + actor ids [count][4].  The block is rewritten only after the previous frame's copy has left it.
        const size_t bytes = 2 * (size_t)count * sizeof(void*) + (size_t)count * 4 * sizeof(float);
        if (bytes > ctx->batch_cap) {
            FS_FLUSH(ctx);   // a held batched frame still carries pointers into the block that is about to be freed
            FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_batch) (void)hipFree(ctx->d_batch);
            if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
            ctx->d_batch = nullptr; ctx->h_batch = nullptr; ctx->batch_cap = 0;
            const size_t cap = (bytes + 255) & ~(size_t)255;
            FS_HIP(ctx, hipMalloc((void**)&ctx->d_batch, cap * fs_context::kBatchSlots));
            FS_HIP(ctx, hipHostMalloc((void**)&ctx->h_batch, cap * fs_context::kBatchSlots, hipHostMallocDefault));
            ctx->batch_cap = cap;
            for (int k = 0; k < fs_context::kBatchSlots; ++k) {
                if (!ctx->ev_batch[k]) FS_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_batch[k], hipEventDisableTiming));
                ctx->batch_pending[k] = false;
                ctx->batch_bytes[k] = 0;
            }
        }
        int slot = (int)(ctx->batch_frame % fs_context::kBatchSlots);
        // A stream of grouped frames rotates through the same few tables (energy buffers and table slots both cycle): a slot
        // whose device copy already holds exactly this table is used as it is — no in-stream copy between the launches
        // (an H2D copy on the compute stream is a bubble of ~10 us per launch).
        ctx->batch_build.resize(bytes);
        {
            void** t_en = reinterpret_cast<void**>(ctx->batch_build.data());
            void** t_fx = t_en + count;
            float* t_pos = reinterpret_cast<float*>(t_fx + count);
            for (int i = 0; i < count; ++i) {
                t_en[i] = f.srcs[i]->d_energy[f.cur_of[i]];
                t_fx[i] = f.fixed ? (void*)f.srcs[i]->d_fixed[f.cur_of[i]] : nullptr;
                std::memcpy(t_pos + 4 * i, f.group ? f.group[i].pos : f.srcs[i]->pos, sizeof(float) * 3);
                std::memcpy(t_pos + 4 * i + 3, &f.srcs[i]->object, sizeof(uint32_t));   // the source's actor (its walks ignore it)
            }
        }
        bool found = false;   // any slot that already holds exactly this table will do (it is only read)
        for (int k = 0; k < fs_context::kBatchSlots && !found; ++k)
            if (ctx->batch_bytes[k] == bytes && std::memcmp(ctx->h_batch + (size_t)k * ctx->batch_cap, ctx->batch_build.data(), bytes) == 0) {
                slot = k; found = true;
            }
        char* hb = ctx->h_batch + (size_t)slot * ctx->batch_cap;
        char* db = ctx->d_batch + (size_t)slot * ctx->batch_cap;
        if (!found) {
            // a slot that a held frame still reads (it may have found it by its content long after it was written) is skipped:
            // at most kMaxWalkParts + 1 frames are held, there are kBatchSlots > that many slots
            auto in_use = [&](int k) {
                const char* dk = ctx->d_batch + (size_t)k * ctx->batch_cap;
                for (const fs_context::PipeFrame& q : ctx->held)
                    if (reinterpret_cast<const char*>(q.energy_tab) == dk) return true;
                return false;
            };
            for (int tries = 0; tries < fs_context::kBatchSlots && in_use(slot); ++tries) {
                ctx->batch_frame++;
                slot = (int)(ctx->batch_frame % fs_context::kBatchSlots);
            }
            hb = ctx->h_batch + (size_t)slot * ctx->batch_cap;
            db = ctx->d_batch + (size_t)slot * ctx->batch_cap;
            if (ctx->batch_pending[slot]) FS_HIP(ctx, wait_event_polling(ctx->ev_batch[slot]));   // its last copy has left the block
            std::memcpy(hb, ctx->batch_build.data(), bytes);
            ctx->batch_bytes[slot] = 0;   // (until the copy is enqueued)
            FS_HIP(ctx, hipMemcpyAsync(db, hb, bytes, hipMemcpyHostToDevice, ctx->stream));
            FS_HIP(ctx, hipEventRecord(ctx->ev_batch[slot], ctx->stream));
            ctx->batch_pending[slot] = true;
            ctx->batch_bytes[slot] = bytes;
        }
        f.used_batch_slot = !found;   // (the rotation advances only when a slot was written)
        f.energy_tab = reinterpret_cast<float* const*>(db);
        f.fixed_tab = f.fixed ? reinterpret_cast<unsigned long long* const*>(db + (size_t)count * sizeof(void*)) : nullptr;
        kp.src_table = reinterpret_cast<const float*>(db + 2 * (size_t)count * sizeof(void*));
    }
    return FS_OK;
}

// the rotation — nothing here can fail
void frame_commit(fs_context* ctx, Frame& f) {
    ctx->frame_index = f.fidx + 1;
    if (f.used_batch_slot) ctx->batch_frame++;
    if (f.unbounded) ctx->overflow_armed = true;
    for (int i = 0; i < f.count; ++i) {
        Source* si = f.srcs[i];
        si->cur = f.cur_of[i];
        si->cur_fixed = f.fixed;
        si->reduced = false; si->handed_off = false; si->tail_ordered = false;
    }
}

int frame_launch(fs_context* ctx, Frame& f) {
    KParams& kp = f.kp;
    SubpathState& st = f.st;
    Source* s = f.srcs[0];
    Source* const* srcs = f.srcs;
    const int B = f.B, count = f.count;
    const bool fixed = f.fixed, accumulate = f.accumulate;
    unsigned* const scratch = f.scratch;
    uint32_t* const perm_buf = f.perm_buf;
    float* const* energy_tab = f.energy_tab;
    unsigned long long* const* fixed_tab = f.fixed_tab;

    TimedFrame tf{};
    // level 1 may sample: events around every profile_interval-th frame only (an event pair costs the frame a few
    // microseconds of queue bubbles — bench.py times every 8th frame of its timed region)
    bool timed_frame = ctx->profiling && (ctx->profiling >= 2 || ctx->profile_interval <= 1 ||
                                                (ctx->profile_tick++ % (unsigned)ctx->profile_interval) == 0);
    if (timed_frame) {
        resolve_completed_timings(ctx);
        for (int i = 0; i < 5; ++i) tf.e[i] = nullptr;
        tf.e[0] = take_event(ctx);
        tf.e[1] = take_event(ctx);
        if (ctx->profiling >= 2) tf.e[2] = take_event(ctx);
    }
    // FlushEnergyBuffer ARTS.cpp:157-161 is folded into the plan pass (one launch); plain memset otherwise.
    // Deterministic mode zeroes the fixed-point histogram instead (the fp32 buffer is rewritten from it).
    float* zero_ptr = accumulate ? nullptr : (fixed ? reinterpret_cast<float*>(s->d_fixed[s->cur]) : s->energy());
    const int zero_words = (fixed ? 2 : 1) * B * ctx->num_bins;
    float* const* zero_tab = nullptr;   // batched frame: the table of buffers the plan pass zeroes
    if (f.batch) {
        zero_ptr = nullptr;
        if (!accumulate) zero_tab = fixed ? reinterpret_cast<float* const*>(fixed_tab) : energy_tab;
    }
    WalkLaunch wplan = ctx->walk;
    wplan.queue_head = scratch;
    wplan.perm = perm_buf;
    if (f.unbounded || f.stages.size() > 1) wplan.plan = 1;   // the second record tier relies on the schedule: the longest walks own the lowest slots
    bool sort = false;
    const bool plan_runs = plan_shape(kp, wplan, nullptr, &sort);   // the plan pass (and the flush with it) runs for this frame
    const uint32_t* perm = plan_runs && sort ? perm_buf : nullptr;
    if (!perm) st.slot_of = nullptr;   // no schedule: slot == subpath index
    const bool plan_held = f.pipe_ok && ctx->pipelining >= 2 && plan_runs;   // depth 2: the pass joins the fused launch below
    if (plan_runs && !plan_held)
        (void)launch_plan(kp, wplan, zero_ptr, (zero_ptr || zero_tab) ? zero_words : 0, zero_tab, count, ctx->stream);
    if (!plan_runs) {
        if (zero_ptr) FS_HIP(ctx, hipMemsetAsync(zero_ptr, 0, sizeof(float) * (size_t)zero_words, ctx->stream));
        for (int i = 0; i < count && zero_tab; ++i) {
            float* zp = fixed ? reinterpret_cast<float*>(srcs[i]->d_fixed[f.cur_of[i]]) : srcs[i]->d_energy[f.cur_of[i]];
            FS_HIP(ctx, hipMemsetAsync(zp, 0, sizeof(float) * (size_t)zero_words, ctx->stream));
        }
    }
    if (!kp.russian_roulette) ctx->host_segments += 2ull * kp.num_local * (unsigned long long)kp.depth;   // no plan pass to count them
    if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[0], ctx->stream));
    WalkLaunch wl = ctx->walk;
    wl.queue_head = scratch;
    // (a staged walk's first stage is a frame of walks of at most stages[0].end steps)
    wl.rays_per_wave = ctx->walk_rays_per_wave > 0 ? ctx->walk_rays_per_wave
                                                   : auto_rays_per_wave(2ull * kp.num_local, std::min(kp.depth, f.stages[0].end));
    const int ppw = ctx->connect_pairs_per_wave > 0 ? ctx->connect_pairs_per_wave : auto_pairs_per_wave(kp.num_local);
    if (f.pipe_ok) {   // (anything else has flushed the held frames before)
        fs_context::PipeFrame me;
        me.kp = kp; me.st = st; me.wl = wl; me.perm = perm; me.stages = f.stages; me.next_stage = 0; me.fixed = fixed; me.ppw = ppw;
        me.items.resize((size_t)count);
        for (int i = 0; i < count; ++i) {
            me.items[(size_t)i].s = srcs[i]; me.items[(size_t)i].cur = f.cur_of[i];
            if (f.group && f.group[i].want_recon) { me.items[(size_t)i].want_recon = true; me.items[(size_t)i].recon = f.group[i].recon; }
        }
        me.energy_tab = energy_tab; me.fixed_tab = fixed_tab;
        FrameParts fp;
        // ONE launch: the plan pass of this frame (depth 2), the next walk stage of every held frame — oldest frame, i.e.
        // latest stage, first: the few long walks that are left have the longest way to go — and the connect pass of the
        // frame whose walk is complete.  Depth 1: the plan pass ran above, this frame's walk joins the launch right away.
        if (plan_held) {
            fp.has_plan = true; fp.kpp = kp; fp.wl_p = wplan; fp.scratch_p = scratch; fp.perm_p = sort ? perm_buf : nullptr;
            fp.zero_p = zero_ptr; fp.zero_words_p = (zero_ptr || zero_tab) ? zero_words : 0; fp.zero_tab_p = zero_tab; fp.zero_count_p = count;
        }
        if (!plan_held) ctx->held.push_back(me);   // its first stage goes now
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
        if (timed_frame && !(fp.num_walk > 0 && fp.has_connect)) {
            // a pipeline-fill launch (not every part yet) is not a sample of the frame kernel: take the first event back out
            // of the stream's timing (it was recorded above; both go back to the pool unused)
            for (int i = 0; i < 3; ++i) if (tf.e[i]) { ctx->free_events.push_back(tf.e[i]); tf.e[i] = nullptr; }
            timed_frame = false;
        }
        OwedLaunch owed;   // the reconstructs of the frames the previous launch connected ride in this one
        { const int orc = owed_prepare(ctx, fp, owed); if (orc) return orc; }
        bool fused_launch = false;
        if (fp.num_walk > 0 || fp.has_connect || fp.has_plan || fp.num_recon > 0) {
            fused_launch = launch_frame(B, ctx->scene, fp, ctx->stream);
            ctx->dbg.launches++;
            if (!fused_launch) {   // no fused form: the same passes one after the other
                if (fp.has_connect) launch_connect(B, ctx->scene, fp.kpc, fp.stc, fp.energy, fp.fixed, fp.scratch_c, fp.ppw, fp.energy_tab, fp.fixed_tab, ctx->stream);
                for (int i = 0; i < fp.num_walk; ++i)
                    launch_walk(ctx->scene, fp.walk[i].kp, fp.walk[i].st, fp.walk[i].wl, fp.walk[i].perm, ctx->stream, fp.walk[i].stage);
                if (fp.has_plan) (void)launch_plan(kp, wplan, zero_ptr, (zero_ptr || zero_tab) ? zero_words : 0, zero_tab, count, ctx->stream);
            }
        }
        if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[1], ctx->stream));
        FS_HIP(ctx, hipGetLastError());
        const bool tail_behind = !owed.owed.empty();   // owed_publish makes the tail stream wait for this launch
        { const int prc = owed_publish(ctx, owed, fused_launch); if (prc) return prc; }
        for (size_t k : advanced) ctx->held[k].next_stage++;
        if (connects) {
            const fs_context::PipeFrame done = ctx->held.front();
            ctx->held.pop_front();
            const int rc = finish_held_frame(ctx, done, /*may_defer_recon=*/true, tail_behind);
            if (rc) return rc;
        }
        if (plan_held) ctx->held.push_back(me);   // planned by this launch, walked by the next ones
        if (timed_frame) { tf.has_trace = true; ctx->pending.push_back(tf); }
        ctx->stats.frames += f.group ? (uint64_t)count : 1;
        ctx->stats.pairs += kp.num_local;
        ctx->stats.rays += 2ull * kp.num_local;
        return FS_OK;
    }
    if (f.stages.size() > 1) {   // a staged walk that is waited for: its stages back to back
        auto stage_launch_of = [&](size_t k) {
            WalkLaunch wk = wl;
            if (ctx->walk_rays_per_wave <= 0)
                // (profiles/r04_sync_stage_sweep4.jsonl: the first stage — everybody, at most `bound` steps — does best on waves of 16
                // subpaths up to 64 000 of them and of 32 beyond: 0.80 / 0.94 ms per tick of 32 sources, 1.37 / 1.65 ms of 128,
                // 1.19 / 1.49 ms per 262 144-ray frame; dense waves 0.89 / 1.18, 1.41 / 1.71, 1.20 / 1.64)
                wk.rays_per_wave = k == 0 ? (ctx->sync_first_rays_per_wave > 0 ? ctx->sync_first_rays_per_wave : (2ull * kp.num_local >= 131072ull ? 32 : 16)) :
                                   ctx->sync_late_rays_per_wave > 0 ? ctx->sync_late_rays_per_wave :
                    auto_rays_per_wave(walk_stage_slots(kp, f.stages[k].begin), std::min(kp.depth, f.stages[k].end) - f.stages[k].begin, k > 0 ? 8192ull : 0ull);
            if (ctx->walk_rays_per_wave <= 0 && k < ctx->sync_stage_rpw.size() && ctx->sync_stage_rpw[k] > 0) wk.rays_per_wave = ctx->sync_stage_rpw[k];
            return wk;
        };
        // The long-walk lane: the frame's time is its longest walk's chain of queries, and a query of a walk that shares a sparse
        // wave of the first stage takes three to five times what it takes a cooperative wave.  The walks the plan pass found to be
        // of lane_len steps or more (a few hundred: the first slots of the schedule) take cooperative waves from step 0 on, in the
        // same launch as the first stage (to their end by default; FS_SYNC_LANE's `end`: the rest beside the survivors in the second).
        const int lane_len = f.lane_len, lane_end = ctx->sync_lane_end;
        uint32_t lane_cap = 0;
        if (lane_len > 0 && walk_lane_possible(ctx->scene, kp, stage_launch_of(0), stage_launch_of(1), perm)) {
            const double expect = 2.0 * (double)kp.num_local * std::pow((double)kp.rr_prob, (double)lane_len);
            lane_cap = (uint32_t)std::min<double>(2.0 * (double)kp.num_local, 1.3 * expect + 64.0);
            for (size_t k = 2; k < f.stages.size(); ++k)   // (the stages behind the second only have to SKIP the lane: any kernel but the dense one can)
                if (stage_launch_of(k).rays_per_wave >= 64) lane_cap = 0;
        }
        for (size_t k = 0; k < f.stages.size(); ++k) {
            WalkLaunch wk = stage_launch_of(k);
            WalkStage sr = f.stages[k];
            WalkLane ln;
            if (lane_cap > 0) {
                ln.len = lane_len; ln.cap = lane_cap;
                const int split = std::max(lane_end, f.stages[0].end);   // the lane's steps [0, split) run beside the first stage
                if (k == 0) { ln.begin = 0; ln.end = split; ln.mode = kLaneSplit; ctx->dbg.lane_launches++; }
                else if (k == 1 && split < (1 << 30)) { ln.begin = split; ln.end = 1 << 30; ln.mode = kLaneBoth; }
                else ln.mode = kLaneSkip;
            }
            launch_walk(ctx->scene, kp, st, wk, perm, ctx->stream, sr, ln);
        }
    } else {
        launch_walk(ctx->scene, kp, st, wl, perm, ctx->stream);
    }
    if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[1], ctx->stream));
    if (f.all_conn)
        launch_connect_all(B, ctx->scene, kp, st, s->energy(), fixed ? s->d_fixed[s->cur] : nullptr, scratch, ctx->stream);
    else
        launch_connect(B, ctx->scene, kp, st, s->energy(), fixed ? s->d_fixed[s->cur] : nullptr, scratch, ppw,
                       energy_tab, fixed_tab, ctx->stream);
    if (fixed)
        for (int i = 0; i < count; ++i)
            launch_fixed_to_energy(srcs[i]->d_fixed[srcs[i]->cur], srcs[i]->energy(), B * ctx->num_bins, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    if (timed_frame) {
        if (tf.e[2]) FS_HIP(ctx, hipEventRecord(tf.e[2], ctx->stream));
        tf.has_trace = true;
        ctx->pending.push_back(tf);
    }
    ctx->stats.frames++;
    ctx->stats.pairs += kp.num_local;
    ctx->stats.rays += 2ull * kp.num_local;
    // multi-GPU: the sum over the ranks (ARTS.cpp:164-173 deposits ALL pairs into the one buffer) runs on the tail
    // stream right behind the deposit, concurrently with whatever the compute stream traces next
    if (ctx->comm)
        for (int i = 0; i < count; ++i) { const int rr = reduce_energy(ctx, srcs[i]); if (rr) return rr; }
    return FS_OK;
}

int trace_sources(fs_context* ctx, Source* const* srcs, int count, const fs_params* p, const fs_context::GroupEntry* group = nullptr) {
    if (!ctx->group.empty()) { const int gr = dispatch_group(ctx); if (gr) return gr; }   // collected frames keep their place in the order
    { const int ir = maybe_install_refined(ctx); if (ir) return ir; }                     // fs_scene_commit_progressive: the better tree is ready
    if (ctx->refit_pending) { const int rr = fs_scene_refit(ctx); if (rr) return rr; }   // moved triangles: refit before tracing
    int rc = check_params(ctx, p);
    if (rc) return rc;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    Frame f;
    f.srcs = srcs; f.count = count; f.p = p; f.group = group;
    frame_describe(ctx, f);
    if (group && !f.pipe_ok) return FS_ERR_INVALID_ARGUMENT;   // (dispatch_group only groups what can be held; it falls back itself)
    if (!f.pipe_ok) FS_FLUSH(ctx);
    rc = frame_resources(ctx, f);
    if (rc) return rc;
    frame_commit(ctx, f);
    return frame_launch(ctx, f);
}

// may a frame with these parameters wait in the group (fs_set_frames_per_launch)?  Exactly what frame_describe lets the
// pipeline hold, and small enough for 32-bit subpath indices when frames_per_launch of them share a launch
bool groupable(const fs_context* ctx, const fs_params* p) {
    if (ctx->frames_per_launch < 2 || ctx->pipelining < 1 || ctx->profiling >= 2) return false;
    if (p->flags & (FS_FLAG_MATERIAL_LOBES | FS_FLAG_MIS_BALANCE | FS_FLAG_ALL_CONNECTIONS | FS_FLAG_ACCUMULATE_ENERGY | FS_FLAG_DOUBLE_POSITIONS))
        return false;
    if (p->listener_radius > 0.0f || p->source_radius > 0.0f) return false;
    const bool unbounded = p->depth == 0 && p->russian_roulette && p->rr_prob < 1.0f;
    if (!(p->depth > 0 || (unbounded && ctx->pipelining >= 2))) return false;
    return (uint64_t)ctx->frames_per_launch * (p->num_rays / 2) <= (1ull << 29);
}
// the same frame but for the low seed word?
bool same_but_seed(const fs_params& a, const fs_params& b) {
    fs_params x = a, y = b;
    x.seed &= ~0xFFFFFFFFull; y.seed &= ~0xFFFFFFFFull;
    return std::memcmp(&x, &y, sizeof(fs_params)) == 0;
}

}  // namespace

namespace fsi {
// The collected frames as ONE batched frame (every item its own seed, energy buffer and recorded reconstruct); a group
// of one, or one that cannot be held after all, goes frame by frame.
int dispatch_group(fs_context* ctx) {
    std::vector<fs_context::GroupEntry> g;
    g.swap(ctx->group);
    if (g.empty()) return FS_OK;
    if (g.size() > 1) {
        Source* srcs[4];
        for (size_t i = 0; i < g.size(); ++i) srcs[i] = g[i].s;
        const int rc = trace_sources(ctx, srcs, (int)g.size(), &g[0].p, g.data());
        if (rc != FS_ERR_INVALID_ARGUMENT) return rc;   // (FS_ERR_INVALID_ARGUMENT: not holdable any more — one by one below)
    }
    for (fs_context::GroupEntry& e : g) {
        // traced with the positions of its call: they are swapped in for the duration
        float pos_now[3], lis_now[3];
        std::memcpy(pos_now, e.s->pos, sizeof(pos_now)); std::memcpy(lis_now, ctx->listener, sizeof(lis_now));
        std::memcpy(e.s->pos, e.pos, sizeof(e.pos)); std::memcpy(ctx->listener, e.lis, sizeof(e.lis));
        int rc = trace_sources(ctx, &e.s, 1, &e.p);
        std::memcpy(e.s->pos, pos_now, sizeof(pos_now)); std::memcpy(ctx->listener, lis_now, sizeof(lis_now));
        if (rc) return rc;
        if (e.want_recon) {
            bool recorded = false;
            if (!ctx->held.empty()) {   // the frame is held: the reconstruct goes with it, as fs_reconstruct_impulse_response_async does
                fs_context::PipeFrame& q = ctx->held.back();
                for (fs_context::PipeFrame::Item& c : q.items)
                    if (c.s == e.s && !c.want_recon) { c.want_recon = true; c.recon = e.recon; recorded = true; }
            }
            if (!recorded) { FS_FLUSH(ctx); rc = reconstruct_now(ctx, e.s, &e.recon); if (rc) return rc; }
        }
    }
    return FS_OK;
}
}  // namespace fsi

extern "C" {

// ---- sources / listener ------------------------------------------------------------------------------
int fs_source_create(fs_context* ctx, fs_source* out) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    Source* s = new (std::nothrow) Source();
    if (!s) return ctx->fail(FS_ERR_OUT_OF_MEMORY, "source");
    const size_t eb = sizeof(float) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    const size_t ib = sizeof(float) * (size_t)ctx->num_samples;
    auto bail = [&](hipError_t e, const char* what) {
        int rc = ctx->hip_fail(e, what);
        s->alive = false;
        free_source(ctx, s);
        return rc;
    };
    hipError_t e;
    for (int i = 0; i < kEnergyBufsBase; ++i) {   // (the rest of the rotation: at first use, ensure_energy_buffer)
        if ((e = hipMalloc((void**)&s->d_energy[i], eb)) != hipSuccess) return bail(e, "hipMalloc(energy)");
        if ((e = hipMemsetAsync(s->d_energy[i], 0, eb, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
        if ((e = hipEventCreateWithFlags(&s->ev_rec[i], kDeviceEventFlags)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    // the device-resident IR set: [B + 1][samples] — the bands, then the channel view (the fused launch derives the one from the other)
    if ((e = hipMalloc((void**)&s->d_ir_bands, ib * (size_t)(ctx->cfg.num_bands + 1))) != hipSuccess) return bail(e, "hipMalloc(ir_bands)");
    s->d_ir_mono = s->d_ir_bands + (size_t)ctx->cfg.num_bands * (size_t)ctx->num_samples;
    if ((e = hipMemsetAsync(s->d_ir_bands, 0, ib * (size_t)(ctx->cfg.num_bands + 1), ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    for (int i = 0; i < kIrRing; ++i) {
        // (coherent: reconstruct workgroups write the slot and the host reads it while their kernel is still running — publish_arrive)
        if ((e = hipHostMalloc((void**)&s->h_ir[i], ib, hipHostMallocCoherent)) != hipSuccess) return bail(e, "hipHostMalloc");
        std::memset(s->h_ir[i], 0, ib);  // ImpulseBuffer[ch].Init(0, NumSamples) FSAC.cpp:24-28
        if ((e = hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = hipEventCreateWithFlags(&s->ev_dep, kDeviceEventFlags)) != hipSuccess) return bail(e, "hipEventCreate");
    for (int i = 0; i < kEnergyBufsBase; ++i)
        if ((e = hipEventCreateWithFlags(&s->ev_red[i], kDeviceEventFlags)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&s->ev_rev, kDeviceEventFlags)) != hipSuccess) return bail(e, "hipEventCreate");
    // the initial fills above ran on the compute stream; the first reconstruct runs on the tail stream
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    s->alive = true;
    // RegisterSource: ActiveSources.Add (ARTS.cpp:45-48); reuse a dead slot if any
    size_t idx = ctx->sources.size();
    for (size_t i = 0; i < ctx->sources.size(); ++i)
        if (!ctx->sources[i]) { idx = i; break; }
    if (idx == ctx->sources.size()) ctx->sources.push_back(nullptr);
    ctx->sources[idx] = s;
    // the new ring slots are all zero: so are their zero-block masks (the handle's previous owner may have left bits behind)
    if (idx < (size_t)kMaxMaskSources && ctx->d_slot_masks) {
        s->mask_index = (int)idx;
        if ((e = hipMemsetAsync(ctx->d_slot_masks + idx * kIrRing, 0, sizeof(uint32_t) * kIrRing, ctx->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) { ctx->sources[idx] = nullptr; return bail(e, "hipMemsetAsync(slot masks)"); }
    }
    *out = (fs_source)idx;
    return FS_OK;
}

int fs_source_destroy(fs_context* ctx, fs_source h) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (ctx->device_ok) {
        (void)hipStreamSynchronize(ctx->stream); (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->rev_stream);
    }
    ctx->sources[(size_t)h] = nullptr;  // UnRegisterSource ARTS.cpp:50-53
    free_source(ctx, s);
    return FS_OK;
}

int fs_source_set_position(fs_context* ctx, fs_source h, const float xyz[3]) {
    if (!ctx || !xyz) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    std::memcpy(s->pos, xyz, sizeof(float) * 3);
    return FS_OK;
}

int fs_source_set_object(fs_context* ctx, fs_source h, uint32_t object_id) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (object_id != s->object) FS_FLUSH(ctx);   // frames that wait for their launch were asked for with the old setting
    s->object = object_id;
    return FS_OK;
}

int fs_listener_set_object(fs_context* ctx, uint32_t object_id) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (object_id != ctx->listener_object) FS_FLUSH(ctx);
    ctx->listener_object = object_id;
    return FS_OK;
}

int fs_listener_set_position(fs_context* ctx, const float xyz[3]) {
    if (!ctx || !xyz) return FS_ERR_INVALID_ARGUMENT;
    std::memcpy(ctx->listener, xyz, sizeof(float) * 3);
    return FS_OK;
}

// ---- hot path ------------------------------------------------------------------------------------------
int fs_compute_energy_response_async(fs_context* ctx, fs_source h, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    // fs_set_frames_per_launch: a plain pipelinable frame waits until enough of its kind have come (same parameters but
    // for the low seed word); anything else sends the waiting ones off first (trace_sources does that)
    fs_params def;
    const fs_params* q = p;
    if (!q) { fs_params_default(&def); q = &def; }
    if (q->struct_size == sizeof(fs_params) && groupable(ctx, q)) {
        if (!ctx->group.empty() && (!same_but_seed(ctx->group[0].p, *q) || std::memcmp(ctx->group[0].lis, ctx->listener, sizeof(ctx->listener)) != 0)) {
            const int gr = dispatch_group(ctx);   // (a batched frame has ONE listener position)
            if (gr) return gr;
        }
        if (ctx->group.empty()) { const int cr = check_params(ctx, q); if (cr) return cr; }   // a bad frame fails at its own call
        fs_context::GroupEntry e;
        e.s = s; e.p = *q;
        std::memcpy(e.pos, s->pos, sizeof(e.pos));
        std::memcpy(e.lis, ctx->listener, sizeof(e.lis));
        ctx->group.push_back(e);
        if ((int)ctx->group.size() >= std::min(ctx->frames_per_launch, 4)) return dispatch_group(ctx);
        return FS_OK;
    }
    return trace_sources(ctx, &s, 1, p);
}

int fs_set_frames_per_launch(fs_context* ctx, int32_t n) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (n < 1 || n > 4) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_set_frames_per_launch: 1 .. 4");
    if (n != ctx->frames_per_launch) FS_FLUSH(ctx);
    ctx->frames_per_launch = n;
    return FS_OK;
}

int fs_compute_energy_response_batch_async(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (count < 0 || (count > 0 && !sources)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad source list");
    if (count == 0) return FS_OK;
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    std::vector<Source*> srcs((size_t)count);
    for (int32_t i = 0; i < count; ++i) {
        srcs[(size_t)i] = get_source(ctx, sources[i]);
        if (!srcs[(size_t)i]) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
        for (int32_t k = 0; k < i; ++k)
            if (sources[k] == sources[i]) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "a source appears twice in the batch");
    }
    // One launch sequence for all sources when the frame fits 32-bit subpath indices and the mode is the default
    // connection strategy; the all-connections modes (one wave per pair already) and oversized batches go source by source
    const uint64_t P = p ? p->num_rays / 2 : 0;
    const bool one_launch = p && !(p->flags & (FS_FLAG_ALL_CONNECTIONS | FS_FLAG_MIS_BALANCE)) &&
                            (uint64_t)count * P <= (1ull << 29);
    if (one_launch) return trace_sources(ctx, srcs.data(), count, p);
    for (int32_t i = 0; i < count; ++i) {
        int rc = trace_sources(ctx, &srcs[(size_t)i], 1, p);
        if (rc) return rc;
    }
    return FS_OK;
}

int fs_compute_energy_response(fs_context* ctx, fs_source h, const fs_params* p, float* energy_out) {
    int rc = FS_OK;
    for (int attempt = 0; attempt < 4; ++attempt) {   // depth = 0: a frame whose records overflowed is traced again
        rc = fs_compute_energy_response_async(ctx, h, p);
        if (rc) return rc;
        FS_FLUSH(ctx);   // the caller waits for this frame: no point in holding its connect pass back
        if (!ctx->overflow_armed) break;
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rc = check_overflow(ctx);
        if (rc != FS_ERR_OVERFLOW) break;
    }
    if (rc) return rc;
    Source* s = get_source(ctx, h);
    // sharded frame: the sum over the ranks runs on the tail stream — the caller gets the SUMMED buffer (found by the
    // one-shot reduce's test: the RCCL test double works synchronously and hid the missing wait)
    if (ctx->comm) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        const int oc = oneshot_check(ctx);
        if (oc) return oc;
    }
    if (energy_out) {
        FS_HIP(ctx, hipMemcpyAsync(energy_out, s->energy(),
                                   sizeof(float) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins,
                                   hipMemcpyDeviceToHost, ctx->stream));
    }
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->pending.empty()) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));   // reconstruct timings live on the tail stream
        resolve_timings(ctx);
    }
    return FS_OK;
}

int fs_energy_device_ptr(fs_context* ctx, fs_source h, void** dptr, size_t* bytes) {
    if (!ctx || !dptr) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    *dptr = s->energy();
    if (bytes) *bytes = sizeof(float) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    return FS_OK;
}

int fs_energy_handoff(fs_context* ctx, fs_source h, void** dptr, size_t* bytes, void** tail_stream) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, handoff_energy(ctx, s));
    s->handed_off = true;
    const size_t words = (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    if (dptr) *dptr = s->cur_fixed ? (void*)s->d_fixed[s->cur] : (void*)s->energy();
    if (bytes) *bytes = (s->cur_fixed ? sizeof(unsigned long long) : sizeof(float)) * words;
    if (tail_stream) *tail_stream = (void*)ctx->copy_stream;
    return FS_OK;
}

// ---- frame pipeline --------------------------------------------------------------------------------------
int fs_set_pipelining(fs_context* ctx, int32_t on) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (on < 0 || on > 2) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_set_pipelining: 0 (off), 1 or 2 frames held back");
    if (on != ctx->pipelining) FS_FLUSH(ctx);
    ctx->pipelining = on;
    return FS_OK;
}

int fs_set_walk_stages(fs_context* ctx, const int32_t* bounds, int32_t count) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (count < 0 || count > kMaxWalkParts - 1 || (count > 0 && !bounds))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_set_walk_stages: 0 .. 7 stage bounds");
    for (int32_t i = 0; i < count; ++i)
        if (bounds[i] <= (i ? bounds[i - 1] : 0) || bounds[i] >= FS_MAX_DEPTH + kOverLevels)
            return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_set_walk_stages: bounds must ascend within 1 .. 511");
    FS_FLUSH(ctx);
    ctx->stage_bounds.assign(bounds, bounds + count);
    ctx->stage_bounds_default = false;
    return FS_OK;
}

int fs_submit(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    return flush_pending(ctx);
}

}  // extern "C"
