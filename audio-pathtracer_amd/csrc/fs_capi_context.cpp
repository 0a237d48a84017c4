// fs_capi_context.cpp — context lifetime, the helpers the translation units share, defaults, statistics
// (C ABI: include/frequensee.h; subsystem Initialize / Deinitialize, ARTS.cpp:32-42).
#include "fs_context.hpp"

#include <chrono>

namespace fsi {

Source* get_source(fs_context* ctx, fs_source h) {
    if (h < 0 || (size_t)h >= ctx->sources.size()) return nullptr;
    Source* s = ctx->sources[(size_t)h];
    return (s && s->alive) ? s : nullptr;
}

// Before the compute stream writes the current energy buffer: the tail-stream reconstruct that last read it
// must be done (two frames back in steady state, i.e. long finished).
// The compute stream waits for `ev` — unless the event has completed already, which is the rule for these waits (the
// reader of a buffer that comes round again after 24 frames, the all-reduce enqueued two launches ago): a wait for a
// finished event is still a barrier packet between two launches of the compute stream (tools/rccl_tax.sh: 23 us between the
// launches of a context with a communicator, 4 such waits each).
hipError_t compute_waits_for(fs_context* ctx, hipEvent_t ev) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) { ctx->dbg.waits_skipped++; return hipSuccess; }
    (void)hipGetLastError();   // hipErrorNotReady is not an error
    ctx->dbg.waits_enqueued++;
    return hipStreamWaitEvent(ctx->stream, ev, 0);
}
// ---- batched reconstructs: one event per batch (fs_context::tail_batch_ev) ----------------------------------------
hipEvent_t tail_batch_event(fs_context* ctx, uint64_t id) {
    const uint64_t newest = ctx->tail_batch_newest.load(std::memory_order_acquire);
    const uint64_t probe = newest - id < (uint64_t)fs_context::kTailBatches ? id : newest - fs_context::kTailBatches + 1;
    return ctx->tail_batch_ev[probe % fs_context::kTailBatches];
}
bool tail_batch_done(fs_context* ctx, uint64_t id) {
    if (id <= ctx->tail_batch_done.load(std::memory_order_acquire)) return true;
    const uint64_t newest = ctx->tail_batch_newest.load(std::memory_order_acquire);
    if (id > newest) return false;   // (not enqueued yet: cannot happen for an id read from a source)
    const uint64_t probe = newest - id < (uint64_t)fs_context::kTailBatches ? id : newest - fs_context::kTailBatches + 1;
    if (hipEventQuery(ctx->tail_batch_ev[probe % fs_context::kTailBatches]) != hipSuccess) { (void)hipGetLastError(); return false; }
    // (if the entry was recycled meanwhile it now stands for a younger batch: its completion implies this one's)
    uint64_t cur = ctx->tail_batch_done.load(std::memory_order_relaxed);
    while (cur < probe && !ctx->tail_batch_done.compare_exchange_weak(cur, probe, std::memory_order_release, std::memory_order_relaxed)) {}
    return true;
}
hipError_t stream_waits_for_rec(fs_context* ctx, hipStream_t st, Source* s, int buf) {
    if (!s->rec_recorded[buf]) return hipSuccess;
    if (s->rec_on_compute[buf] && st == ctx->stream) return hipSuccess;   // the same stream, earlier
    if (s->rec_batch[buf]) {
        if (tail_batch_done(ctx, s->rec_batch[buf])) return hipSuccess;
        return hipStreamWaitEvent(st, tail_batch_event(ctx, s->rec_batch[buf]), 0);
    }
    if (st == ctx->stream) return compute_waits_for(ctx, s->ev_rec[buf]);
    return hipStreamWaitEvent(st, s->ev_rec[buf], 0);
}
// The producer's short waits (the IR ring's back-pressure: a launch or two, ~ 1 ms) POLL the event instead of sleeping on it: a
// blocking hipEventSynchronize depends on an interrupt reaching the host thread, and on the pool's boxes one such wait in a few
// hundred came back ~ 6 ms late (the work it waited for long done) — inside a 6 ms timed region that halves the rate
// (tools/repeat_driver_bench.py).  After 20 ms of polling it falls back to the blocking wait.
hipError_t wait_event_polling(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return hipSuccess;
        if (q != hipErrorNotReady) return q;
        (void)hipGetLastError();
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) return hipEventSynchronize(ev);
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
}
// ---- publishes of the compute stream: the launch writes its id into a pinned host word (Source::pub_word, publish_arrive) ----
bool pub_word_done(const fs_context* ctx, uint64_t id) {
    return ctx->h_pub_word != nullptr && __atomic_load_n(ctx->h_pub_word, __ATOMIC_ACQUIRE) >= id;
}
PublishWord next_pub_word(fs_context* ctx) {
    PublishWord w;
    w.tickets = ctx->d_pub_tickets; w.host_word = ctx->h_pub_word; w.id = ctx->pub_issued + 1;
    return w;
}
// The producer's wait for such a publish: the launch is already in the stream, the word arrives by itself — a few hundred
// microseconds of polling a cached host line, no runtime call.  A word that does not arrive within 50 ms while the stream still
// has work is waited for with the stream; one that is still missing when the stream is idle belongs to a launch that never ran.
hipError_t wait_pub_word(fs_context* ctx, uint64_t id) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; !pub_word_done(ctx, id); ++spins) {
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) {
            const hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) return e;
            return pub_word_done(ctx, id) ? hipSuccess : hipErrorLaunchFailure;
        }
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
    }
    return hipSuccess;
}
bool slot_published(fs_context* ctx, Source* s, int slot) {
    const uint64_t w = s->pub_word[slot].load(std::memory_order_acquire);
    if (w) return pub_word_done(ctx, w);
    const uint64_t b = s->pub_batch[slot].load(std::memory_order_acquire);
    if (b) return tail_batch_done(ctx, b);
    if (hipEventQuery(s->ev[slot]) != hipSuccess) { (void)hipGetLastError(); return false; }   // (hipErrorNotReady is not an error)
    return true;
}
// ---- zero-block masks of the host ring slots (fs_context::d_slot_masks) -----------------------------------------------------
static bool slot_masks_usable(const fs_context* ctx, const Source* s) {
    // (a mask word has 32 bits: one per block of kBlock * kChunk = 4 096 samples)
    return ctx->d_slot_masks != nullptr && s->mask_index >= 0 && s->mask_index < kMaxMaskSources && ctx->num_samples <= 32 * 4096;
}
uint32_t* slot_mask_ptr(const fs_context* ctx, const Source* s, int slot) {
    return slot_masks_usable(ctx, s) ? ctx->d_slot_masks + (size_t)s->mask_index * kIrRing + (size_t)slot : nullptr;
}
hipError_t slot_mask_all_dirty(fs_context* ctx, const Source* s, int slot, hipStream_t st) {
    uint32_t* w = slot_mask_ptr(ctx, s, slot);
    return w ? hipMemsetAsync(w, 0xFF, sizeof(uint32_t), st) : hipSuccess;
}
hipError_t sync_publish(fs_context* ctx, Source* s, int slot) {
    const uint64_t w = s->pub_word[slot].load(std::memory_order_acquire);
    const uint64_t b = s->pub_batch[slot].load(std::memory_order_acquire);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = w ? wait_pub_word(ctx, w) : wait_event_polling(b ? tail_batch_event(ctx, b) : s->ev[slot]);
    ctx->dbg.sync_publish++;
    ctx->dbg.sync_publish_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    return e;
}
hipError_t wait_energy_readers(fs_context* ctx, Source* s, int buf) {
    if (s->red_recorded[buf]) {   // the tail stream may still be summing this buffer over the ranks
        hipError_t e = compute_waits_for(ctx, s->ev_red[buf]);
        if (e != hipSuccess) return e;
    }
    return stream_waits_for_rec(ctx, ctx->stream, s, buf);
}
hipError_t wait_energy_readers(fs_context* ctx, Source* s) { return wait_energy_readers(ctx, s, s->cur); }
// Hand the current energy buffer over to the tail stream: what the compute stream has enqueued so far
// completes before anything enqueued on the tail stream from now on.
hipError_t handoff_energy(fs_context* ctx, Source* s) {
    hipError_t e = hipEventRecord(s->ev_dep, ctx->stream);
    if (e != hipSuccess) return e;
    e = hipStreamWaitEvent(ctx->copy_stream, s->ev_dep, 0);
    ctx->dbg.tail_ops++;
    if (e == hipSuccess) s->tail_ordered = true;   // until the compute stream writes the buffer again
    return e;
}

void free_source(fs_context* ctx, Source* s) {
    if (!s) return;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->cfg.device);
        for (int i = 0; i < kEnergyBufs; ++i) {
            if (s->d_energy[i]) (void)hipFree(s->d_energy[i]);
            if (s->d_fixed[i]) (void)hipFree(s->d_fixed[i]);
            if (s->ev_rec[i]) (void)hipEventDestroy(s->ev_rec[i]);
        }
        if (s->ev_dep) (void)hipEventDestroy(s->ev_dep);
        for (int i = 0; i < kEnergyBufs; ++i) if (s->ev_red[i]) (void)hipEventDestroy(s->ev_red[i]);
        if (s->ev_rev) (void)hipEventDestroy(s->ev_rev);
        if (s->d_ir_bands) (void)hipFree(s->d_ir_bands);   // (d_ir_mono is its last row)
        for (int i = 0; i < kIrRing; ++i) {
            if (s->h_ir[i]) (void)hipHostFree(s->h_ir[i]);
            if (s->ev[i]) (void)hipEventDestroy(s->ev[i]);
        }
        if (s->d_ring) (void)hipFree(s->d_ring);
        if (s->d_rev_in) (void)hipFree(s->d_rev_in);
        if (s->d_rev_cur) (void)hipFree(s->d_rev_cur);
        if (s->d_rev_out) (void)hipFree(s->d_rev_out);
    }
    delete s;
}

void free_scene(fs_context* ctx) {
    if (ctx->d_nodes) (void)hipFree(ctx->d_nodes);
    if (ctx->d_coop) (void)hipFree(ctx->d_coop);
    if (ctx->d_coop16) (void)hipFree(ctx->d_coop16);
    if (ctx->d_coop_levels) (void)hipFree(ctx->d_coop_levels);
    ctx->d_coop = nullptr; ctx->coop_cap = 0; ctx->d_coop16 = nullptr; ctx->coop16_cap = 0; ctx->d_coop_levels = nullptr;
    ctx->coop16_nodes = 0; ctx->coop_levels = 0;
    if (ctx->d_tris) (void)hipFree(ctx->d_tris);
    if (ctx->d_tris48) (void)hipFree(ctx->d_tris48);
    if (ctx->d_tri_nrm) (void)hipFree(ctx->d_tri_nrm);
    if (ctx->d_absorption) (void)hipFree(ctx->d_absorption);
    if (ctx->d_leaf_pos) (void)hipFree(ctx->d_leaf_pos);
    if (ctx->d_node_box) (void)hipFree(ctx->d_node_box);
    if (ctx->d_move) (void)hipFree(ctx->d_move);
    ctx->d_leaf_pos = nullptr; ctx->d_node_box = nullptr; ctx->d_move = nullptr;
    ctx->move_cap = 0;
    ctx->refit_pending = false;
    ctx->d_nodes = nullptr; ctx->d_tris = nullptr; ctx->d_absorption = nullptr;
    ctx->d_tris48 = nullptr; ctx->d_tri_nrm = nullptr;
    if (ctx->deep.buf) (void)hipFree(ctx->deep.buf);
    for (int32_t* b : ctx->deep.retired) (void)hipFree(b);
    ctx->deep = DeepStore{};
    ctx->scene = DeviceScene{};
    ctx->committed = false;
}

void free_state(fs_context* ctx) {
    if (ctx->st.end_pos) (void)hipFree(ctx->st.end_pos);
    if (ctx->st.end_misc) (void)hipFree(ctx->st.end_misc);
    if (ctx->st.seg_np) (void)hipFree(ctx->st.seg_np);
    if (ctx->st.seg_mat) (void)hipFree(ctx->st.seg_mat);
    if (ctx->d_seg_pos) (void)hipFree(ctx->d_seg_pos);
    ctx->d_seg_pos = nullptr; ctx->cap_pos = 0;
    if (ctx->st.slot_of) (void)hipFree(ctx->st.slot_of);
    for (void* q : {(void*)ctx->d_over_np, (void*)ctx->d_over_mat, (void*)ctx->d_over_pos, (void*)ctx->d_cont, (void*)ctx->d_end_posd})
        if (q) (void)hipFree(q);
    if (ctx->h_overflow) (void)hipHostFree(ctx->h_overflow);
    ctx->h_overflow = nullptr;
    ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr; ctx->d_over_pos = nullptr; ctx->d_overflow = nullptr; ctx->d_cont = nullptr; ctx->d_end_posd = nullptr;
    ctx->over_cap = ctx->over_cap_pos = 0; ctx->cap_posd = 0;
    if (ctx->walk.perm) (void)hipFree(ctx->walk.perm);
    ctx->walk.perm = nullptr;
    ctx->st = SubpathState{};
    ctx->cap_lanes = ctx->cap_seg = 0;
}

hipEvent_t take_event(fs_context* ctx) {
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventReleaseToDevice);   // (timing events: no system-scope cache flush between the kernels they time)
    return e;
}

// fold one finished timed frame into the stats and recycle its events
void fold_timed_frame(fs_context* ctx, TimedFrame& f) {
    float ms = 0.f, ms2 = 0.f;
    if (f.has_trace && hipEventElapsedTime(&ms, f.e[0], f.e[1]) == hipSuccess) {
        ctx->stats.walk_kernel_ms_sum += ms;
        ctx->stats.walk_kernel_ms_last = ms;
        ctx->stats.timed_frames++;
        if (f.e[2] && hipEventElapsedTime(&ms2, f.e[1], f.e[2]) == hipSuccess) {
            ctx->stats.connect_kernel_ms_sum += ms2;
            ctx->stats.timed_connects++;
        }
    }
    if (f.has_recon && hipEventElapsedTime(&ms, f.e[3], f.e[4]) == hipSuccess) {
        ctx->stats.reconstruct_ms_sum += ms;
        ctx->stats.timed_reconstructs++;
    }
    for (int i = 0; i < 5; ++i)
        if (f.e[i]) ctx->free_events.push_back(f.e[i]);
}

// fold finished timed frames into the stats (call only after the streams have been synchronised)
void resolve_timings(fs_context* ctx) {
    for (TimedFrame& f : ctx->pending) fold_timed_frame(ctx, f);
    ctx->pending.clear();
}

// A caller that leaves profiling on and never asks for the stats must not accumulate events without bound:
// once enough frames are pending, the ones whose last event has completed are folded in as they go.
void resolve_completed_timings(fs_context* ctx) {
    if (ctx->pending.size() < 64) return;
    size_t done = 0;
    for (TimedFrame& f : ctx->pending) {
        hipEvent_t last = f.has_recon ? f.e[4] : (f.e[2] ? f.e[2] : f.e[1]);
        if (!last || hipEventQuery(last) != hipSuccess) break;
        fold_timed_frame(ctx, f);
        ++done;
    }
    (void)hipGetLastError();   // hipErrorNotReady from the query is not an error
    if (done) ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + (long)done);
}

// advance `front` over publishes whose D2H copy has completed.  Any thread: the producer calls it where it enqueues, a
// consumer through fs_get_impulse_response_sequence.  `front` only ever grows (compare-exchange to the maximum): a thread
// that looked at a slot just before the producer recycled it for a later publish can only conclude less, never more.
void poll_published(fs_context* ctx, Source* s) {
    uint64_t f = s->front.load(std::memory_order_acquire);
    const uint64_t enq = s->enqueued.load(std::memory_order_acquire);
    while (f < enq) {
        uint64_t next = f + 1;
        int slot = (int)(next % kIrRing);
        if (s->seq_of[slot].load(std::memory_order_acquire) != next) break;
        if (!slot_published(ctx, s, slot)) break;   // the host word, the batch's event or the slot's own (Source::pub_word)
        if (s->seq_of[slot].load(std::memory_order_acquire) != next) break;   // recycled meanwhile: the answer was about a later publish
        f = next;
    }
    uint64_t cur = s->front.load(std::memory_order_relaxed);
    while (cur < f && !s->front.compare_exchange_weak(cur, f, std::memory_order_release, std::memory_order_relaxed)) {}
}

// levels = walk steps with a record in the main tier (min(depth, FS_MAX_DEPTH)); unbounded: also the second tier.
// sets = copies of the per-frame arrays (frames in flight + 1, fs_context::state_sets); staged: with continuation records.
// Anything that must grow is reallocated at the new size — behind the held frames, which still use the old arrays.
int ensure_state(fs_context* ctx, uint32_t n_local, int levels, bool unbounded, double rr_prob, bool want_positions, bool want_normals,
                 int sets, bool staged) {
    size_t lanes = 2 * (size_t)n_local;
    size_t seg = (size_t)levels * lanes;
    sets = std::max(sets, ctx->state_sets);          // never shrinks: a later frame of the old shape finds its sets
    staged = staged || ctx->state_cont;
    const size_t want = want_positions ? seg * (want_normals ? 2 : 1) : 0;   // positions, then normals
    const double tail = std::pow(std::min(std::max(rr_prob, 0.0), 1.0), (double)FS_MAX_DEPTH);
    // the schedule puts the longest walks first: slots below rr^64 * lanes (x4 for the spread, + 64) own a second tier
    // (rr = 0.9, the reference's roulette: 1.2e-3 of the walks; check_params bounds rr for uncapped walks)
    uint32_t want_cap = ctx->over_cap;
    if (unbounded) {
        want_cap = (uint32_t)std::min<size_t>(lanes, std::max<size_t>(ctx->over_cap, (size_t)(4.0 * tail * (double)lanes) + 64));   // (a slot per lane is all a frame can use)
        if (ctx->over_cap_forced > 0) want_cap = std::max(ctx->over_cap, (uint32_t)ctx->over_cap_forced);   // FS_OVER_CAP (tests: force the regrow path)
    }
    const bool new_sets = sets != ctx->state_sets || staged != ctx->state_cont;
    const bool grow_lanes = lanes > ctx->cap_lanes || new_sets;
    const bool grow_seg = seg > ctx->cap_seg || grow_lanes;
    const bool grow_over = unbounded && (want_cap > ctx->over_cap || !ctx->d_over_np || new_sets);
    const bool grow_over_pos = unbounded && want_positions && (want_cap > ctx->over_cap_pos || !ctx->d_over_pos);
    const bool grow_pos = want > ctx->cap_pos;
    if (grow_lanes || grow_seg || grow_over || grow_over_pos || grow_pos) {
        FS_FLUSH(ctx);                                   // held frames still read the arrays that are about to go
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (grow_pos) {
        if (ctx->d_seg_pos) (void)hipFree(ctx->d_seg_pos);
        ctx->d_seg_pos = nullptr; ctx->cap_pos = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_seg_pos, sizeof(float4) * std::max<size_t>(want, 1)));
        ctx->cap_pos = want;
    }
    if (grow_lanes) {
        lanes = std::max(lanes, ctx->cap_lanes);
        if (ctx->st.end_pos) (void)hipFree(ctx->st.end_pos);
        if (ctx->st.end_misc) (void)hipFree(ctx->st.end_misc);
        if (ctx->st.slot_of) (void)hipFree(ctx->st.slot_of);
        if (ctx->walk.perm) (void)hipFree(ctx->walk.perm);
        if (ctx->d_cont) (void)hipFree(ctx->d_cont);
        ctx->st.end_pos = nullptr; ctx->st.end_misc = nullptr; ctx->st.slot_of = nullptr; ctx->d_cont = nullptr;
        ctx->walk.perm = nullptr;
        ctx->cap_lanes = 0;
        // `sets` sets of everything a frame's walk hands to its connect pass: pipelined frames overlap the walks of the
        // frames behind with the connect pass of the oldest
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.end_pos, sizeof(float4) * (size_t)sets * lanes));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.end_misc, sizeof(uint2) * (size_t)sets * lanes));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.slot_of, sizeof(uint32_t) * (size_t)sets * lanes));
        if (staged) FS_HIP(ctx, hipMalloc((void**)&ctx->d_cont, sizeof(float4) * 2 * (size_t)sets * lanes));
        ctx->cap_seg = 0;   // the bucket array is sized with the segment records below
        ctx->cap_lanes = lanes;
        ctx->state_sets = sets;
        ctx->state_cont = staged;
    }
    if (grow_seg) {
        seg = std::max(seg, (size_t)levels * ctx->cap_lanes);
        if (ctx->st.seg_np) (void)hipFree(ctx->st.seg_np);
        if (ctx->st.seg_mat) (void)hipFree(ctx->st.seg_mat);
        ctx->st.seg_np = nullptr; ctx->st.seg_mat = nullptr;
        ctx->cap_seg = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.seg_np, sizeof(float2) * (size_t)ctx->state_sets * seg));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.seg_mat, sizeof(uint32_t) * (size_t)ctx->state_sets * seg));
        if (ctx->walk.perm) (void)hipFree(ctx->walk.perm);
        ctx->walk.perm = nullptr;
        // [levels + 1][lanes] for every later frame shape that fits the two capacities without a reallocation:
        // levels' * lanes' <= cap_seg and lanes' <= cap_lanes  =>  (levels' + 1) * lanes' <= seg + cap_lanes
        ctx->perm_words = seg + ctx->cap_lanes;
        FS_HIP(ctx, hipMalloc((void**)&ctx->walk.perm, sizeof(uint32_t) * (size_t)ctx->state_sets * ctx->perm_words));
        ctx->cap_seg = seg;
    }
    if (!ctx->d_overflow) {
        // the overflow word lives in pinned HOST memory, written by the kernels across the bus in the (rare) event: the host reads it
        // behind a synchronize without a copy — a hipMemcpy of four bytes per fs_synchronize cost an uncapped frame that is waited
        // for 25 us (a copy kernel, its launch gap and a second wait: profiles/r04_tick_trace_*.json)
        FS_HIP(ctx, hipHostMalloc((void**)&ctx->h_overflow, 64, hipHostMallocDefault));
        *ctx->h_overflow = 0u;
        FS_HIP(ctx, hipHostGetDevicePointer((void**)&ctx->d_overflow, ctx->h_overflow, 0));
    }
    if (grow_over) {
        if (ctx->d_over_np) (void)hipFree(ctx->d_over_np);
        if (ctx->d_over_mat) (void)hipFree(ctx->d_over_mat);
        ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_np, sizeof(float2) * (size_t)ctx->state_sets * kOverLevels * want_cap));
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_mat, sizeof(uint32_t) * (size_t)ctx->state_sets * kOverLevels * want_cap));
        ctx->over_cap = want_cap;
    }
    if (grow_over_pos) {   // all-connections frames are never held: one tier
        if (ctx->d_over_pos) (void)hipFree(ctx->d_over_pos);
        ctx->d_over_pos = nullptr;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_pos, sizeof(float4) * 2 * (size_t)kOverLevels * ctx->over_cap));   // positions | normals
        ctx->over_cap_pos = ctx->over_cap;
    }
    return FS_OK;
}

// Subpaths per wave of the walk kernel.  A large frame fills the chip with dense waves; a small one is a few waves
// and takes the latency of its longest chain of closest-hit queries, which shrinks when the idle lanes of sparse
// waves help with every query (walk_kernel_sparse).  Thresholds measured on MI355X
// (tools/sparse_sweep.py, tools/sparse_check.py, profiles/r01_sparse_waves.json).  Frames of shallow walks do best at
// about 2048 waves (16 384 subpaths at depth 8: 0.33 -> 0.19 ms with 8 per wave; from 262 144 subpaths on dense
// waves win).  Walks deeper than 16 segments (the reference's unbounded default) leave long chains of a few
// survivors and do best at about 16 384 waves at every size measured (262 144 subpaths, unbounded depth, 5 000
// triangles: 1.29 -> 0.82 ms with 16 per wave; 1 048 576: dense again).
int auto_rays_per_wave(unsigned long long lanes, int depth, unsigned long long waves) {
    // round 2 (kShareMinIdle, pipelined frames; tools/pipelined_rpw_sweep.py, profiles/r02_pipelined_rpw_sweep.json): mid-size
    // frames do better at ~4096 waves — 65 536 subpaths 16 per wave, 131 072 subpaths 32 per wave (0.244 -> 0.233 ms at
    // depth 8, 0.355 -> 0.309 ms at depth 12; unpipelined 0.300 -> 0.267 ms); 262 144 and more stay dense
    // (waves > 0: the caller's target — the later stages of a staged walk that is waited for do best at ~8192 cooperative waves:
    // 16 400 / 62 000 / 64 000 survivors 0.83 / 1.42 / 1.20 ms per frame with 2 / 4 / 4 per wave, profiles/r04_sync_stage_sweep2.jsonl)
    const unsigned long long target_waves = waves ? waves : (depth > 16 ? 16384ull : (lanes >= 65536ull ? 4096ull : 2048ull));
    // round 3 (tools/ref_defaults_breakdown.py, profiles/r03_ref_defaults_rpw.log): the reference's own update — 2 000
    // subpaths, uncapped — is fastest with ONE subpath per wave (0.88 -> 0.71 ms starter_room, 1.11 -> 0.92 ms old_mine; 2: 0.79,
    // 4: 0.85, 16: 0.90): the floor of 4 per wave went
    int rpw = 1;
    while (rpw < 64 && (unsigned long long)rpw * 2 * target_waves <= lanes) rpw *= 2;   // largest power of two <= lanes / target
    return rpw;
}

// Pairs per wave of the connect kernel, same idea: a small frame's visibility queries are shared by sparse waves.
// About 2048 waves (tools/connect_sparse_sweep.py): 8 192 pairs 0.065 -> 0.031 ms with 4 per wave, 32 768 pairs
// 0.067 -> 0.043 ms with 16, dense waves from 131 072 pairs on.
int auto_pairs_per_wave(unsigned long long pairs) {
    int ppw = 1;   // (round 3: 1 000 pairs, one per wave: 0.056 -> 0.044 ms; profiles/r03_ref_defaults_rpw.log)
    while (ppw < 64 && (unsigned long long)ppw * 2 * 2048ull <= pairs) ppw *= 2;
    return ppw;
}

int check_params(fs_context* ctx, const fs_params* p) {
    if (!p) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    if (p->depth < 0 || p->depth > FS_MAX_DEPTH) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "depth out of range");
    if (p->samples_per_bin < 0 || p->samples_per_bin > 32767) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "samples_per_bin out of range (0 = from the configuration, else 1 .. 32767)");
    if (p->num_rays & 1u) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "num_rays must be even (source + listener subpaths)");
    if (p->num_rays > (1u << 30)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "num_rays above 2^30 per frame (32-bit subpath indices)");
    if ((p->flags & FS_FLAG_MATERIAL_LOBES) && (p->flags & FS_FLAG_MIS_BALANCE))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "FS_FLAG_MATERIAL_LOBES and FS_FLAG_MIS_BALANCE cannot be combined");
    if ((p->flags & FS_FLAG_ACCUMULATE_ENERGY) && ctx->cfg.world_size > 1)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "FS_FLAG_ACCUMULATE_ENERGY (the reference's accumulation quirk) is single-GPU only: "
                                                  "a sharded frame is summed over the ranks, an accumulated one would be summed again");
    // depth = 0 walks until the roulette ends the walk (ARTS.cpp:294); the record store covers FS_MAX_DEPTH + kOverLevels
    // steps, which rr <= 0.95 leaves with probability < 4e-12 per walk.  A weaker roulette needs an explicit cap.
    if (p->depth == 0 && p->russian_roulette && p->rr_prob < 1.0f && p->rr_prob > kMaxUnboundedRr)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "depth = 0 (uncapped walks) needs rr_prob <= 0.95: give a depth cap for a weaker roulette");
    if ((p->flags & FS_FLAG_DOUBLE_POSITIONS) && (p->flags & (FS_FLAG_MATERIAL_LOBES | FS_FLAG_MIS_BALANCE | FS_FLAG_ALL_CONNECTIONS)))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "FS_FLAG_DOUBLE_POSITIONS cannot be combined with the lobe / all-connections modes");
    if (!(p->listener_radius >= 0.f) || !(p->source_radius >= 0.f) || !std::isfinite(p->listener_radius) || !std::isfinite(p->source_radius))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "listener_radius / source_radius must be finite and >= 0");
    if (!(p->dist_divisor > 0.f) || !(p->sound_speed > 0.f))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "dist_divisor and sound_speed must be positive");
    return FS_OK;
}

}  // namespace fsi

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }

void fs_config_default(fs_config* c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->struct_size = sizeof(fs_config);
    c->device = 0;
    c->num_bands = 1;
    c->sample_rate = 48000;       // FSAC.h:133
    c->num_channels = 2;          // FSAC.h:135
    c->simulated_duration = 1.0f; // FSAC.h:136
    c->bin_duration = 0.001f;     // FSAC.h:137
    c->rank = 0;
    c->world_size = 1;
    c->stream = nullptr;
}

void fs_params_default(fs_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(fs_params);
    p->flags = 0;
    p->seed = 0x5EEDull;
    p->num_rays = 2000;            // USED_RAY_COUNT = 1000 pairs, ARTS.h:176
    p->depth = 0;                  // unbounded, ARTS.cpp:294
    p->russian_roulette = 1;
    p->rr_prob = 0.9f;             // ARTS.cpp:282
    p->max_trace_dist = 1000000.f; // ARTS.cpp:284
    p->surface_offset = 0.1f;      // ARTS.cpp:345
    p->connect_pullback = 0.1f;    // ARTS.cpp:253
    p->dist_divisor = 1000.f;      // ARTS.cpp:373
    p->min_seg = 1.0f;             // ARTS.cpp:375
    p->prob_exponent = 0.1f;       // ARTS.cpp:398
    p->energy_clamp = 1.0f;        // ARTS.cpp:410
    p->energy_gain = 10.f;         // ARTS.cpp:413
    p->sound_speed = 343.0f;       // ARTS.cpp:362
    for (int b = 0; b < FS_MAX_BANDS; ++b) p->air_absorption[b] = 0.05f;  // ARTS.cpp:395
    p->samples_per_bin = 0;
    p->listener_radius = 0.0f;     // the end points are points (34 cm reproduces HEAD's pawn collision)
    p->source_radius = 0.0f;
}

int fs_context_create(const fs_config* cfg, fs_context** out) {
    if (!out) return FS_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    fs_config c;
    fs_config_default(&c);
    if (cfg) {
        if (cfg->struct_size != sizeof(fs_config)) return FS_ERR_INVALID_ARGUMENT;
        c = *cfg;
        if (c.num_bands == 0) c.num_bands = 1;
        if (c.sample_rate == 0) c.sample_rate = 48000;
        if (c.num_channels == 0) c.num_channels = 2;
        if (c.simulated_duration == 0.f) c.simulated_duration = 1.0f;
        if (c.bin_duration == 0.f) c.bin_duration = 0.001f;
        if (c.world_size == 0) c.world_size = 1;
    }
    if (c.num_bands < 1 || c.num_bands > FS_MAX_BANDS || c.world_size < 1 || c.rank < 0 || c.rank >= c.world_size ||
        c.sample_rate < 1 || c.num_channels < 1 || !(c.simulated_duration > 0.f) || !(c.bin_duration > 0.f))
        return FS_ERR_INVALID_ARGUMENT;
    fs_context* ctx = new (std::nothrow) fs_context();
    if (!ctx) return FS_ERR_OUT_OF_MEMORY;
    ctx->cfg = c;
    ctx->num_bins = (int)std::ceil(c.simulated_duration / c.bin_duration);             // FSAC.h:137 -> 1000
    ctx->num_samples = (int)std::ceil(c.simulated_duration * (float)c.sample_rate);    // FSAC.h:138 -> 48000
    *out = ctx;  // returned even on device failure so fs_last_error() can be read
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (c.device < 0 || c.device >= ndev) return ctx->fail(FS_ERR_NO_DEVICE, "device ordinal out of range");
    e = hipSetDevice(c.device);
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    if (c.stream) {
        ctx->stream = (hipStream_t)c.stream;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
        ctx->own_stream = true;
    }
    {
        // The tail stream (all-reduce, reconstructs, publishes) and the reverb callback's stream are created with the highest
        // priority: what they carry is small and somebody waits for it — and a priority stream gets hardware queues of its own.
        // With three ordinary streams the rate of a pipelined stream of frames depended on how many streams the process
        // had created before (which hardware queues the compute and the tail stream were given): 965 – 970 M rays/s or
        // 880 M, 995 or 900 M at four frames per launch, by the count of earlier streams (profiles/r03_stream_queues.log;
        // without publishes 990 / 1 015 M whatever the history).  FS_TAIL_STREAM_PRIORITY=0: ordinary streams again; 2: lowest.
        int prio_lo = 0, prio_hi = 0;   // (numerically lower = higher priority)
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        const int mode = std::getenv("FS_TAIL_STREAM_PRIORITY") ? std::atoi(std::getenv("FS_TAIL_STREAM_PRIORITY")) : 1;
        const int prio = mode == 2 ? prio_lo : prio_hi;
        if (mode != 0) e = hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, prio);
        else e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipStreamCreate(copy): ") + hipGetErrorString(e));
        if (mode != 0) e = hipStreamCreateWithPriority(&ctx->rev_stream, hipStreamNonBlocking, prio_hi);
        else e = hipStreamCreateWithFlags(&ctx->rev_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipStreamCreate(reverb): ") + hipGetErrorString(e));
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device) == hipSuccess && cus > 0)
        ctx->walk.num_cus = cus;
    int lds = 0;
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c.device) == hipSuccess && lds > 0)
        ctx->lds_limit = (size_t)lds;
    // the reconstruct kernel stages one amplitude per bin in LDS; the traversal kernels' need depends on the tree and is
    // checked at fs_scene_commit
    if (sizeof(float) * (size_t)ctx->num_bins > ctx->lds_limit || ctx->num_bins < 1 || ctx->num_samples < 1)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "simulated_duration / bin_duration give " + std::to_string(ctx->num_bins) +
                         " bins: more than the reconstruct kernel can stage in the device's LDS");
#ifdef FS_EXPERIMENTS
    if (const char* v = std::getenv("FS_WALK_VARIANT")) ctx->walk.variant = std::atoi(v) == 0 ? 0 : 2;
#endif
    if (const char* v = std::getenv("FS_WALK_PLAN")) ctx->walk.plan = std::atoi(v) ? 1 : 0;
    if (const char* v = std::getenv("FS_DEBUG_STALLS")) ctx->debug_stalls = std::atoi(v) != 0;
    if (const char* v = std::getenv("FS_FLUSH_RECON_ON_COMPUTE")) ctx->flush_recon_on_compute = std::atoi(v) != 0;
    if (const char* v = std::getenv("FS_OVER_CAP")) ctx->over_cap_forced = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("FS_WALK_COOP")) ctx->walk.coop = std::atoi(v) ? 1 : 0;
    if (const char* v = std::getenv("FS_FUSED_RECON")) ctx->fused_recon = std::atoi(v) != 0;
    if (const char* v = std::getenv("FS_FUSED_RECON_COMM")) ctx->fused_recon_comm = std::atoi(v) != 0;
    if (const char* v = std::getenv("FS_FUSED_DRAIN")) ctx->fused_drain = std::atoi(v) != 0;
    ctx->hist_window = default_hist_window(ctx->cfg.num_bands);
    if (const char* v = std::getenv("FS_STACK_ROWS_CAP")) ctx->stack_rows_cap = std::max(kDeepChunk + 4, std::min(kStackDepth + 1, std::atoi(v)));
    if (const char* v = std::getenv("FS_HIST_WINDOW")) ctx->hist_window = std::max(1, std::min(4096, std::atoi(v)));
    if (const char* v = std::getenv("FS_WALK_RAYS_PER_WAVE")) ctx->walk_rays_per_wave = std::max(0, std::min(64, std::atoi(v)));
    if (const char* v = std::getenv("FS_CONNECT_PAIRS_PER_WAVE")) ctx->connect_pairs_per_wave = std::max(0, std::min(64, std::atoi(v)));
    // staged depth = 0 walks (pipelined frames): the steps at which a walk moves on to the next launch.  12-step stages
    // (a launch carries one frame's worth of work, ~0.6 ms at 262 144 subpaths; a bounce of a wave on the full chip
    // takes ~35 us, so longer stages make their chain the launch's length), longer ones for the few hundred walks beyond
    // the main record tier (tools/stage_sweep.py, profiles/r03_stage_sweep.log)
    ctx->stage_bounds = {8, 18, 30, 46, 64, 96};   // best of the sets tried on 262 144-ray frames at roulette 0.9 (profiles/r03_stage_sweep.log)
    if (const char* v = std::getenv("FS_STAGE_DENSE_FROM")) ctx->stage_dense_from = std::max(1, std::atoi(v));
    if (const char* v = std::getenv("FS_WALK_STAGES")) {
        std::vector<int> b;
        for (const char* q = v; *q;) {
            char* end = nullptr;
            const long x = std::strtol(q, &end, 10);
            if (end == q) break;
            if (x > (b.empty() ? 0 : b.back()) && x < FS_MAX_DEPTH + kOverLevels && (int)b.size() < kMaxWalkParts - 1) b.push_back((int)x);
            q = *end ? end + 1 : end;
        }
        ctx->stage_bounds = b;   // empty: depth = 0 frames are not held
        ctx->stage_bounds_default = false;
    }
    if (const char* v = std::getenv("FS_SYNC_FIRST_RPW")) ctx->sync_first_rays_per_wave = std::max(0, std::min(64, std::atoi(v)));
    if (const char* v = std::getenv("FS_SYNC_LATE_RPW")) ctx->sync_late_rays_per_wave = std::max(0, std::min(64, std::atoi(v)));
    ctx->sync_stage_bounds = {24};   // tools/sync_stage_sweep.py, profiles/r04_sync_stage_sweep*.jsonl
    if (const char* v = std::getenv("FS_SYNC_WALK_STAGES")) {
        std::vector<int> b;
        for (const char* q = v; *q;) {
            char* end = nullptr;
            const long x = std::strtol(q, &end, 10);
            if (end == q) break;
            if (x > (b.empty() ? 0 : b.back()) && x < FS_MAX_DEPTH + kOverLevels && (int)b.size() < kMaxWalkParts - 1) b.push_back((int)x);
            q = *end ? end + 1 : end;
        }
        ctx->sync_stage_bounds = b;   // empty: such frames walk in one piece
        ctx->sync_stage_bounds_default = false;
    }
    if (const char* v = std::getenv("FS_SYNC_STAGE_FROM")) { ctx->sync_stage_from = std::max(1, std::atoi(v)); ctx->sync_stage_from_default = false; }
    if (const char* v = std::getenv("FS_SYNC_LANE")) {
        int len = 0, end = 0;
        if (std::sscanf(v, "%d,%d", &len, &end) >= 1) {
            ctx->sync_lane_len = std::max(-1, std::min(FS_MAX_DEPTH, len));   // (-1: the default rule, 0: no lane)
            ctx->sync_lane_end = end > 0 ? end : (1 << 30);
        }
    }
    if (const char* v = std::getenv("FS_SYNC_STAGE_RPW")) {
        for (const char* q = v; *q;) {
            char* end = nullptr;
            const long x = std::strtol(q, &end, 10);
            if (end == q) break;
            ctx->sync_stage_rpw.push_back((int)std::max(0l, std::min(64l, x)));
            q = *end ? end + 1 : end;
        }
    }
    if (const char* v = std::getenv("FS_SOUND_RAYS_PER_WAVE")) ctx->sound_rays_per_wave = std::max(1, std::min(64, std::atoi(v)));
    e = hipMalloc((void**)&ctx->walk.queue_head, sizeof(unsigned) * kScratchSets * kScratchAllocWords);   // each set with its counters
    if (e == hipSuccess) e = hipMemset(ctx->walk.queue_head, 0, sizeof(unsigned) * kScratchSets * kScratchAllocWords);
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipMalloc(queue): ") + hipGetErrorString(e));
    for (int k = 0; k < fs_context::kTailBatches && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&ctx->tail_batch_ev[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void**)&ctx->d_comm_stage, 2048);
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_recon_tab, sizeof(ReconItem) * fs_context::kReconTabSlots * fs_context::kReconTabItems, hipHostMallocDefault);
    // the publish word: coherent (fine-grained) host memory the device writes with system scope while its kernel is still running
    if (e == hipSuccess) e = hipHostMalloc((void**)&ctx->h_pub_word, 64, hipHostMallocCoherent);
    // ONE device allocation: the ticket cell of publish_arrive (kSlotMaskOffsetWords words), then the zero-block masks of the ring slots
    const size_t pub_bytes = sizeof(uint32_t) * ((size_t)kSlotMaskOffsetWords + (size_t)kMaxMaskSources * kIrRing);
    if (e == hipSuccess) { *ctx->h_pub_word = 0ull; e = hipMalloc((void**)&ctx->d_pub_tickets, pub_bytes); }
    if (e == hipSuccess) e = hipMemset(ctx->d_pub_tickets, 0, pub_bytes);
    if (e == hipSuccess) ctx->d_slot_masks = reinterpret_cast<uint32_t*>(ctx->d_pub_tickets) + kSlotMaskOffsetWords;
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("batched reconstructs: ") + hipGetErrorString(e));
    ctx->device_ok = true;
    // A context overlaps the tail of a frame with the next frame's tracing on two HIP streams.  The runtime multiplexes
    // streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); with other libraries' streams in the process
    // (RCCL, the host's renderer) the two can land on one queue and serialise (measured 0.64 vs 0.57 ms per frame).  The
    // variable is read when the HIP runtime initialises — long before a plugin is loaded into a host that uses HIP
    // itself — so the library does not set it behind the host's back: it says what it found.
    {
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        const int have = q ? std::atoi(q) : 0;
        if (have < 8)
            ctx->advice = std::string("GPU_MAX_HW_QUEUES is ") + (q ? q : "unset (the runtime's default is 4)") +
                          ": export GPU_MAX_HW_QUEUES=16 before the process initialises HIP, or the context's compute and "
                          "tail streams may share a hardware queue and serialise";
    }
    return FS_OK;
}

const char* fs_context_advice(const fs_context* ctx) { return ctx ? ctx->advice.c_str() : ""; }

int fs_context_destroy(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->cfg.device);
        if (ctx->debug_stalls)
            std::fprintf(stderr, "[frequensee] stalls: %llu fused launches, %llu flushes of %llu held frames, %llu host waits for a publish (%llu us), "
                         "%llu waits enqueued on the compute stream (%llu already complete), %llu owed reconstructs on the tail stream\n",
                         (unsigned long long)ctx->dbg.launches, (unsigned long long)ctx->dbg.flushes, (unsigned long long)ctx->dbg.flushed_frames,
                         (unsigned long long)ctx->dbg.sync_publish, (unsigned long long)ctx->dbg.sync_publish_us, (unsigned long long)ctx->dbg.waits_enqueued,
                         (unsigned long long)ctx->dbg.waits_skipped, (unsigned long long)ctx->dbg.owed_on_tail);
        (void)flush_pending(ctx);
        cancel_refine(ctx);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
        if (ctx->rev_stream) (void)hipStreamSynchronize(ctx->rev_stream);
        resolve_timings(ctx);
        (void)fs_comm_detach(ctx);
        (void)fs_peers_detach(ctx);
        if (ctx->d_gather) (void)hipFree(ctx->d_gather);
        for (hipEvent_t e : ctx->free_events) (void)hipEventDestroy(e);
        free_scene(ctx);
        free_state(ctx);
        if (ctx->walk.queue_head) (void)hipFree(ctx->walk.queue_head);
        if (ctx->d_sound) (void)hipFree(ctx->d_sound);
        for (void* p : {(void*)ctx->d_fft_x, (void*)ctx->d_fft_y, (void*)ctx->d_fft_w, (void*)ctx->d_fft_in,
                        (void*)ctx->d_fft_resp, (void*)ctx->d_fft_out})
            if (p) (void)hipFree(p);
        if (ctx->fft_graph) (void)hipGraphExecDestroy(ctx->fft_graph);
        if (ctx->h_fft_stage) (void)hipHostFree(ctx->h_fft_stage);
        if (ctx->d_batch) (void)hipFree(ctx->d_batch);
        if (ctx->d_build) (void)hipFree(ctx->d_build);
        if (ctx->h_batch) (void)hipHostFree(ctx->h_batch);
        for (hipEvent_t e : ctx->ev_batch) if (e) (void)hipEventDestroy(e);
    }
    for (hipEvent_t ev : ctx->tail_batch_ev) if (ev) (void)hipEventDestroy(ev);
    if (ctx->h_recon_tab) (void)hipHostFree(ctx->h_recon_tab);
    if (ctx->h_pub_word) (void)hipHostFree(ctx->h_pub_word);
    if (ctx->d_pub_tickets) (void)hipFree(ctx->d_pub_tickets);   // (d_slot_masks lives in the same allocation)
    if (ctx->d_comm_stage) (void)hipFree(ctx->d_comm_stage);
    join_refine_threads(ctx);   // no background build may outlive the context (the library may be unloaded next)
    for (Source* s : ctx->sources) free_source(ctx, s);
    // streams exist even when a later step of fs_context_create failed (device_ok == false)
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->rev_stream) (void)hipStreamDestroy(ctx->rev_stream);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (!ctx->device_ok && ctx->walk.queue_head) (void)hipFree(ctx->walk.queue_head);
    delete ctx;
    return FS_OK;
}

const char* fs_last_error(const fs_context* ctx) {
    if (!ctx) return "null context";
    static thread_local std::string copy;   // the string may be rewritten by another thread while the caller reads it
    {
        std::lock_guard<std::mutex> g(const_cast<fs_context*>(ctx)->err_mu);
        copy = ctx->err;
    }
    return copy.c_str();
}
int fs_num_bins(const fs_context* ctx) { return ctx ? ctx->num_bins : 0; }
int fs_num_samples(const fs_context* ctx) { return ctx ? ctx->num_samples : 0; }

// ---- measurement ----------------------------------------------------------------------------------------------
int fs_set_profiling(fs_context* ctx, int32_t enabled) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    ctx->profiling = enabled < 0 ? 0 : (enabled > 3 ? 3 : enabled);
    return FS_OK;
}

int fs_set_profiling_interval(fs_context* ctx, int32_t frames) {
    if (!ctx || frames < 1) return FS_ERR_INVALID_ARGUMENT;
    ctx->profile_interval = frames;
    ctx->profile_tick = 0;
    return FS_OK;
}

int fs_get_pipeline_counters(fs_context* ctx, fs_pipeline_counters* out) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (out->struct_size != sizeof(fs_pipeline_counters)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_pipeline_counters.struct_size mismatch");
    out->reserved = 0;
    out->fused_launches = ctx->dbg.launches; out->flushes = ctx->dbg.flushes; out->flushed_frames = ctx->dbg.flushed_frames;
    out->host_waits = ctx->dbg.sync_publish; out->host_wait_us = ctx->dbg.sync_publish_us;
    out->stream_waits_enqueued = ctx->dbg.waits_enqueued; out->stream_waits_skipped = ctx->dbg.waits_skipped;
    out->tail_stream_ops = ctx->dbg.tail_ops; out->owed_on_tail = ctx->dbg.owed_on_tail;
    out->publishes_by_word = ctx->dbg.pub_word; out->publishes_by_event = ctx->dbg.pub_event;
    out->lane_launches = ctx->dbg.lane_launches;
    return FS_OK;
}

int fs_get_streams(fs_context* ctx, void** compute_stream, void** tail_stream) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (compute_stream) *compute_stream = (void*)ctx->stream;
    if (tail_stream) *tail_stream = (void*)ctx->copy_stream;
    return FS_OK;
}

int fs_get_stats(fs_context* ctx, fs_stats* out) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (ctx->device_ok && !ctx->pending.empty()) {
        FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        resolve_timings(ctx);
    }
    if (ctx->device_ok && ctx->walk.queue_head) {   // work counters kept on the device since the last reset
        unsigned long long c[kNumCounters] = {0}, cs[kScratchSets][kNumCounters] = {};   // each scratch set carries its own counters
        FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
        for (int k = 0; k < kScratchSets; ++k)
            FS_HIP(ctx, hipMemcpyAsync(cs[k], ctx->walk.queue_head + (size_t)k * kScratchAllocWords + kCounterWord, sizeof(cs[k]),
                                       hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int k = 0; k < kScratchSets; ++k)
            for (int i = 0; i < kNumCounters; ++i) c[i] += cs[k][i];
        ctx->stats.segments = c[0];                              // the walkers' own step counts, summed by the connect pass
        ctx->stats.planned_segments = c[7] + ctx->host_segments;  // the plan pass's prediction (roulette off: depth x subpaths, on the host)
        ctx->stats.connections_tested = c[1];
        ctx->stats.deposits = c[2];
        ctx->stats.walk_node_fetches = c[3];
        ctx->stats.walk_tri_fetches = c[4];
        ctx->stats.any_node_fetches = c[5];
        ctx->stats.any_tri_fetches = c[6];
        ctx->stats.node_request_insts = c[8]; ctx->stats.node_request_lanes = c[9]; ctx->stats.node_request_distinct = c[10];
    }
    *out = ctx->stats;
    return FS_OK;
}

int fs_reset_stats(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    fs_stats keep = ctx->stats;
    ctx->stats = fs_stats{};
    ctx->host_segments = 0;
    ctx->stats.bvh_nodes = keep.bvh_nodes;
    ctx->stats.triangles = keep.triangles;
    ctx->stats.bvh_stack_need = keep.bvh_stack_need;
    ctx->stats.bvh_depth = keep.bvh_depth;
    ctx->stats.scene_bytes = keep.scene_bytes;
    if (ctx->device_ok && ctx->walk.queue_head) {
        FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
        for (int k = 0; k < kScratchSets; ++k)
            FS_HIP(ctx, hipMemsetAsync(ctx->walk.queue_head + (size_t)k * kScratchAllocWords + kCounterWord, 0,
                                       sizeof(unsigned long long) * kNumCounters, ctx->stream));
    }
    return FS_OK;
}

}  // extern "C"
