// fs_capi_publish.cpp — ReconstructImpulseResponse + the publish of its result (FSAC.cpp:320-380, GetImpulseResponse FSAC.h:113):
// the IR ring and its back-pressure, the three ways a ring slot becomes known as published (the compute stream's host word, a
// tail-stream batch's event, a slot's own event behind a copy), the reconstruct parts of a fused launch (owed_prepare /
// owed_publish), the stand-alone reconstruct on the tail stream (reconstruct_now) and the batched one (reconstruct_batch).
// Split out of fs_capi_frame.cpp in round 5; the frame pipeline that calls these lives in fs_capi_pipeline.cpp.
#include "fs_context.hpp"

namespace fsi {

// never overwrite the front buffer of the source's IR ring: at most kIrRing - 1 publishes in flight (publish seq reuses the
// slot of seq - kIrRing, so publish seq - kIrRing + 1 must have completed before seq is enqueued).  `more`: publishes about
// to be enqueued.  May block the host (the ring is the producer's only throttle) — call it without holding ir_mu.
int ir_ring_backpressure(fs_context* ctx, Source* s, int more) {
    poll_published(ctx, s);
    for (int j = 1; j <= more; ++j) {
        if (s->enqueued + (uint64_t)j < (uint64_t)kIrRing) continue;
        const uint64_t must = s->enqueued + (uint64_t)j + 1 - (uint64_t)kIrRing;
        const int slot = (int)(must % kIrRing);
        if (s->seq_of[slot] == must && s->front.load(std::memory_order_relaxed) < must) {
            FS_HIP(ctx, sync_publish(ctx, s, slot));
            poll_published(ctx, s);
        }
    }
    return FS_OK;
}

// publish number `seq` of the source has been enqueued: through the compute stream's host word (word != 0), a tail-stream
// batch's event (batch != 0) or the slot's own event.  (Readers look at enqueued, then seq_of, then the kind: written in reverse.)
static void note_publish(fs_context* ctx, Source* s, uint64_t seq, int slot, uint64_t batch = 0, uint64_t word = 0) {
    if (word) ctx->dbg.pub_word++; else ctx->dbg.pub_event++;
    s->pub_word[slot] = word; s->pub_batch[slot] = batch; s->seq_of[slot] = seq; s->enqueued = seq;
}

// before the COMPUTE stream writes the source's device IR set: whoever reads or writes it on the tail stream goes first
static hipError_t compute_waits_for_tail_ir(fs_context* ctx, Source* s) {
    if (s->cur_pub_seq) {
        const int slot = (int)(s->cur_pub_seq % kIrRing);
        if (s->seq_of[slot] == s->cur_pub_seq && !s->pub_word[slot]) {   // (a reused slot: that publish completed long ago)
            const uint64_t pb = s->pub_batch[slot];
            hipError_t e = hipSuccess;
            if (!pb) e = compute_waits_for(ctx, s->ev[slot]);
            else if (!tail_batch_done(ctx, pb)) e = hipStreamWaitEvent(ctx->stream, tail_batch_event(ctx, pb), 0);
            if (e != hipSuccess) return e;
        }
    }
    if (s->rev_recorded) return compute_waits_for(ctx, s->ev_rev);   // a reverb callback may be reading d_ir_mono
    return hipSuccess;
}

// a table slot of the batch kernel that its previous reader has certainly left
static int acquire_recon_tab(fs_context* ctx, unsigned* slot_out) {
    const unsigned slot_t = ctx->recon_tab_next++ % fs_context::kReconTabSlots;
    if (ctx->recon_tab_batch[slot_t] && !tail_batch_done(ctx, ctx->recon_tab_batch[slot_t]))
        FS_HIP(ctx, wait_event_polling(tail_batch_event(ctx, ctx->recon_tab_batch[slot_t])));
    if (ctx->recon_tab_word[slot_t] && !pub_word_done(ctx, ctx->recon_tab_word[slot_t])) FS_HIP(ctx, wait_pub_word(ctx, ctx->recon_tab_word[slot_t]));
    ctx->recon_tab_batch[slot_t] = 0; ctx->recon_tab_word[slot_t] = 0;
    *slot_out = slot_t;
    return FS_OK;
}

static int spb_of(const fs_context* ctx, const fs_params& p) {
    return p.samples_per_bin > 0 ? p.samples_per_bin : (int)std::ceil(ctx->cfg.bin_duration * (float)ctx->cfg.sample_rate);   // FSAC.cpp:324
}

int owed_prepare(fs_context* ctx, FrameParts& fp, OwedLaunch& ol) {
    if (ctx->recon_owed.empty()) return FS_OK;
    size_t take = 0;   // the oldest entries that are due (frame order: a prefix); summed entries wait one launch (ReconOwed::reduced)
    while (take < ctx->recon_owed.size() && take < (size_t)kMaxReconParts &&
           (!ctx->recon_owed[take].reduced || ctx->recon_owed[take].age >= 1)) ++take;
    for (size_t k = take; k < ctx->recon_owed.size(); ++k) ctx->recon_owed[k].age++;
    if (take == 0) return FS_OK;
    // (the entries leave recon_owed only when everything that can fail here has succeeded: on an error they are still owed
    // and the next flush reconstructs them)
    ol.owed.assign(ctx->recon_owed.begin(), ctx->recon_owed.begin() + (long)take);
    const int B = ctx->cfg.num_bands;
    std::vector<Source*> distinct;
    for (const fs_context::ReconOwed& o : ol.owed)
        if (std::find(distinct.begin(), distinct.end(), o.s) == distinct.end()) distinct.push_back(o.s);
    std::sort(distinct.begin(), distinct.end());           // one locking order for every thread
    // the ring's back-pressure and the table slot BEFORE the IR mutexes are taken: both may wait for the GPU, and
    // fs_reverb_process on the audio thread must never queue behind such a wait
    for (Source* s : distinct) {
        int more = 0;
        for (const fs_context::ReconOwed& o : ol.owed) more += o.s == s ? 1 : 0;
        const int br = ir_ring_backpressure(ctx, s, more);
        if (br) { ol.owed.clear(); return br; }
    }
    { const int ar = acquire_recon_tab(ctx, &ol.tab_slot); if (ar) { ol.owed.clear(); return ar; } }
    ReconItem* tab = ctx->h_recon_tab + (size_t)ol.tab_slot * fs_context::kReconTabItems;
    auto bail = [&](int rc) { ol.owed.clear(); ol.seq.clear(); ol.newest.clear(); ol.locks.clear(); fp.num_recon = 0; return rc; };
#define FS_OWED_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return bail(ctx->hip_fail(e_, #call)); } while (0)
    for (Source* s : distinct) ol.locks.emplace_back(s->ir_mu);
    fp.num_recon = 0; fp.recon_tab = tab; fp.recon_B = B; fp.recon_nb = ctx->num_bins; fp.recon_samples = ctx->num_samples;
    ol.pub = next_pub_word(ctx);
    fp.pub = ol.pub;
    for (size_t i = 0; i < ol.owed.size(); ++i) {
        const fs_context::ReconOwed& o = ol.owed[i];
        Source* s = o.s;
        bool later = false;                                // a later frame of the same source in this launch?
        for (size_t k = i + 1; k < ol.owed.size(); ++k) later = later || ol.owed[k].s == s;
        uint64_t seq = s->enqueued + 1;                    // (the ring's back-pressure was applied above, before the mutexes)
        for (size_t k = 0; k < i; ++k) seq += ol.owed[k].s == s ? 1 : 0;
        const int slot = (int)(seq % kIrRing);
        // An IR that is superseded within the launch only goes to its ring slot (the channel row); the source's newest IR of the
        // launch also becomes the device-resident set (the reverb's, fs_copy_band_impulse_response's).
        if (!later) FS_OWED_HIP(compute_waits_for_tail_ir(ctx, s));
        if (o.reduced && s->red_recorded[o.cur]) FS_OWED_HIP(compute_waits_for(ctx, s->ev_red[o.cur]));   // the sum over the ranks (done a launch ago)
        ol.seq.push_back(seq); ol.newest.push_back(later ? 0 : 1);
        ReconItem& r = tab[fp.num_recon++];
        r.energy = s->d_energy[o.cur]; r.ir_bands = later ? nullptr : s->d_ir_bands; r.ir_mono = later ? nullptr : s->d_ir_mono;
        r.host = s->h_ir[slot]; r.mask = slot_mask_ptr(ctx, s, slot); r.spb = spb_of(ctx, o.p); r.pad = 0;
    }
#undef FS_OWED_HIP
    ctx->recon_owed.erase(ctx->recon_owed.begin(), ctx->recon_owed.begin() + (long)take);
    return FS_OK;
}
// behind the launch: note the publishes (the launch announces them itself); a source with a reverb also gets an event on the
// compute stream for its callbacks to wait on
int owed_publish(fs_context* ctx, OwedLaunch& ol, bool launched_fused) {
    if (ol.owed.empty()) return FS_OK;
    const int B = ctx->cfg.num_bands;
    if (!launched_fused) {   // no fused form for this launch: the same table through the batch kernel, on the compute stream
        launch_reconstruct_batch(ctx->h_recon_tab + (size_t)ol.tab_slot * fs_context::kReconTabItems, (int)ol.owed.size(), B, ctx->num_bins,
                                 ctx->num_samples, ctx->stream, ol.pub);
        FS_HIP(ctx, hipGetLastError());
    }
    ctx->recon_tab_word[ol.tab_slot] = ol.pub.id;          // the slot's reader: this launch
    ctx->pub_issued = ol.pub.id;
    for (size_t i = 0; i < ol.owed.size(); ++i) {
        const fs_context::ReconOwed& o = ol.owed[i];
        Source* s = o.s;
        s->rec_recorded[o.cur] = true; s->rec_on_compute[o.cur] = true; s->rec_batch[o.cur] = 0;
        if (ol.newest[i]) {   // (ir_mu is held)
            s->last_rec = o.cur; s->cur_pub_seq = 0; s->dev_ir_word = ol.pub.id;
            if (s->d_ring) FS_HIP(ctx, hipEventRecord(s->ev_rec[o.cur], ctx->stream));   // fs_reverb_process reads d_ir_mono behind this
        }
        note_publish(ctx, s, ol.seq[i], (int)(ol.seq[i] % kIrRing), 0, ol.pub.id);
    }
    ol.locks.clear();
    return FS_OK;
}

// the reconstructs that were waiting for the next fused launch, on the tail stream after all (a flush, or a reconstruct
// that must not overtake them)
int run_owed_reconstructs(fs_context* ctx) {
    if (ctx->recon_owed.empty()) return FS_OK;
    std::vector<fs_context::ReconOwed> owed;
    owed.swap(ctx->recon_owed);
    ctx->dbg.owed_on_tail += owed.size();
    for (const fs_context::ReconOwed& o : owed) {
        Source* s = o.s;
        const int cur = s->cur;
        const bool cur_fixed = s->cur_fixed, reduced = s->reduced, handed_off = s->handed_off, tail_ordered = s->tail_ordered;
        s->cur = o.cur; s->cur_fixed = o.fixed; s->reduced = o.reduced; s->handed_off = false;
        s->tail_ordered = o.reduced;   // (the tail stream is behind the all-reduce, which is behind the launch)
        const int rc = flush_reconstruct(ctx, s, &o.p);
        s->cur = cur; s->cur_fixed = cur_fixed; s->reduced = reduced; s->handed_off = handed_off; s->tail_ordered = tail_ordered;
        if (rc) return rc;
    }
    return FS_OK;
}

// A reconstruct that no launch is fused with (a flush: fs_submit, fs_synchronize, an observer): on one GPU it goes onto the
// COMPUTE stream as a batch of one — the kernel writes the published host slot itself, nothing crosses to the tail stream.
// (Through the tail stream — a handoff event, a kernel and a copy on the priority queue while the compute queue is busy — one
// flush in three took 6 ms longer than the others on the pool's boxes: tools/repeat_driver_bench.py, the driver's 20-step
// region read 440 or 880 M rays/s.)  FS_FLUSH_RECON_ON_COMPUTE=0 restores the tail-stream path.
int flush_reconstruct(fs_context* ctx, Source* s, const fs_params* p) {
    if (ctx->flush_recon_on_compute && !ctx->comm && ctx->cfg.world_size == 1) { Source* one = s; return reconstruct_batch(ctx, &one, 1, p, true); }
    return reconstruct_now(ctx, s, p);
}

int reconstruct_now(fs_context* ctx, Source* s, const fs_params* p) {
    { const int orc = run_owed_reconstructs(ctx); if (orc) return orc; }   // IRs are published in frame order
    // ReconstructImpulseResponse is not linear in the energy (a = e / sqrt(e * Pi4)): the IR of a rank's PARTIAL
    // histogram is not a partial IR.  A sharded context only reconstructs a frame that was summed over the ranks — by
    // the library (fs_comm_init / fs_comm_attach) or by the caller's collective on the tail stream (fs_energy_handoff).
    if (ctx->cfg.world_size > 1 && !s->reduced && !s->handed_off)
        return ctx->fail(FS_ERR_COMM, "world_size > 1: the energy buffer holds this rank's partial sums only — attach a "
                                      "communicator (fs_comm_init) or reduce it behind fs_energy_handoff before reconstructing");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int B = ctx->cfg.num_bands;
    int spb = p->samples_per_bin > 0 ? p->samples_per_bin
                                     : (int)std::ceil(ctx->cfg.bin_duration * (float)ctx->cfg.sample_rate);  // FSAC.cpp:324

    { const int br = ir_ring_backpressure(ctx, s, 1); if (br) return br; }
    TimedFrame tf{};
    bool timed = ctx->profiling >= 2;
    if (timed) {
        for (int i = 0; i < 5; ++i) tf.e[i] = nullptr;
        tf.e[3] = take_event(ctx);
        tf.e[4] = take_event(ctx);
    }
    if (p->flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT) {  // ARTS.cpp:191 literally
        FS_HIP(ctx, wait_energy_readers(ctx, s));
        FS_HIP(ctx, hipMemsetAsync(s->energy(), 0, sizeof(float) * (size_t)B * (size_t)ctx->num_bins, ctx->stream));
        // (deterministic mode: the fp32 buffer is re-derived from the fixed-point histogram below — the flush empties that one too)
        if (s->cur_fixed && s->d_fixed[s->cur])
            FS_HIP(ctx, hipMemsetAsync(s->d_fixed[s->cur], 0, sizeof(unsigned long long) * (size_t)B * (size_t)ctx->num_bins, ctx->stream));
    }
    // The tail stream takes over: it waits for this frame's deposit (and runs behind any collective the caller
    // put there after fs_energy_handoff), reconstructs and publishes while the compute stream goes on to the
    // next frame.  Reconstructs and publishes of one source are ordered among themselves by the tail stream.
    // (A frame the library has just summed over the ranks — or that the caller took over with fs_energy_handoff — is
    // already ordered: a second event pair per frame on the compute stream is a second bubble between its launches,
    // 2 % of a cfg3 frame: tools/rccl_tax.sh.)
    if (!s->tail_ordered || (p->flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT)) FS_HIP(ctx, handoff_energy(ctx, s));
    else FS_HIP(ctx, tail_waits_for_compute_ir(ctx, s));   // (a fused reconstruct of an earlier frame may still be writing d_ir_*)
    hipStream_t tail = ctx->copy_stream;
    {
        std::lock_guard<std::mutex> g(s->ir_mu);   // against fs_reverb_process on the audio thread
        if (s->rev_recorded) FS_HIP(ctx, hipStreamWaitEvent(tail, s->ev_rev, 0));   // a reverb callback may be reading d_ir_mono
        if (timed) FS_HIP(ctx, hipEventRecord(tf.e[3], tail));
        // deterministic mode: the collective summed the fixed-point histogram; round it to fp32 once, now
        if (s->cur_fixed && !s->reduced) launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), B * ctx->num_bins, tail);
        launch_reconstruct(s->energy(), B, ctx->num_bins, ctx->cfg.sample_rate, ctx->num_samples, spb, s->d_ir_bands,
                           s->d_ir_mono, tail);
        FS_HIP(ctx, hipGetLastError());
        FS_HIP(ctx, hipEventRecord(s->ev_rec[s->cur], tail));
        s->rec_recorded[s->cur] = true; s->rec_batch[s->cur] = 0; s->rec_on_compute[s->cur] = false;
        s->last_rec = s->cur;
    }
    uint64_t seq = s->enqueued + 1;
    int slot = (int)(seq % kIrRing);
    FS_HIP(ctx, hipMemcpyAsync(s->h_ir[slot], s->d_ir_mono, sizeof(float) * (size_t)ctx->num_samples,
                               hipMemcpyDeviceToHost, tail));
    FS_HIP(ctx, slot_mask_all_dirty(ctx, s, slot, tail));   // (a copy wrote every block: the kernels' zero-block bookkeeping starts over)
    FS_HIP(ctx, hipEventRecord(s->ev[slot], tail));
    note_publish(ctx, s, seq, slot);
    s->cur_pub_seq = seq; s->dev_ir_word = 0;
    ctx->dbg.tail_ops += 4;   // the reconstruct kernel, its event, the copy, the publish event
    if (timed) {
        FS_HIP(ctx, hipEventRecord(tf.e[4], tail));
        tf.has_recon = true;
        ctx->pending.push_back(tf);
    }
    return FS_OK;
}

int ir_ring_backpressure_for(fs_context* ctx, Source* s) { return ir_ring_backpressure(ctx, s, 1); }

// Before the TAIL stream writes the source's device IR set: the compute-stream launch that wrote it last (a fused reconstruct part,
// a batch behind a tick) may still be running — the compute stream hands over (everything it has enqueued so far goes first).
hipError_t tail_waits_for_compute_ir(fs_context* ctx, Source* s) {
    if (!s->dev_ir_word || pub_word_done(ctx, s->dev_ir_word)) return hipSuccess;
    hipError_t e = hipEventRecord(s->ev_dep, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->copy_stream, s->ev_dep, 0);
    ctx->dbg.tail_ops++;
    return e;
}

// ReconstructImpulseResponse + publish of MANY sources' current frames (the subsystem's loop over ActiveSources,
// ARTS.cpp:100-126, every one ending in ReconstructImpulseResponse :192): one handoff, ONE kernel that also writes the
// channel views straight into the sources' pinned host ring slots, ONE event — instead of a wait, a kernel, a copy and
// three event records per source (67 us per source of host and queue time: 128 sources took 8.6 ms, 32 took 2.8).
// Anything unusual about a source's frame (a literal second flush, per-kernel timing) sends the whole batch the ordinary way.
// on_compute (fs_update_sources: the caller waits for the tick anyway): the launch goes onto the COMPUTE stream, right behind
// the frame's connect pass — no event pair, no second stream to wake (12 us between the two kernels of a one-source tick).
int reconstruct_batch(fs_context* ctx, Source* const* srcs, int count, const fs_params* p, bool on_compute) {
    { const int orc = run_owed_reconstructs(ctx); if (orc) return orc; }   // IRs are published in frame order
    bool plain = ctx->profiling < 2 && !(p->flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT);   // (also for ONE source: no copy command, one event)
    for (int i = 0; i < count; ++i)
        if (ctx->cfg.world_size > 1 && !srcs[i]->reduced && !srcs[i]->handed_off) plain = false;   // (reconstruct_now refuses with the message)
    if (!plain) {
        for (int i = 0; i < count; ++i) { const int rc = reconstruct_now(ctx, srcs[i], p); if (rc) return rc; }
        return FS_OK;
    }
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const int B = ctx->cfg.num_bands;
    const int spb = p->samples_per_bin > 0 ? p->samples_per_bin : (int)std::ceil(ctx->cfg.bin_duration * (float)ctx->cfg.sample_rate);  // FSAC.cpp:324
    hipStream_t tail = on_compute ? ctx->stream : ctx->copy_stream;
    for (int first = 0; first < count; first += fs_context::kReconTabItems) {
        const int n = std::min(count - first, (int)fs_context::kReconTabItems);
        Source* const* g = srcs + first;
        for (int i = 0; i < n; ++i) { const int br = ir_ring_backpressure(ctx, g[i], 1); if (br) return br; }   // (before the mutexes: may wait for the GPU)
        // the tail stream takes over behind everything the compute stream has enqueued for these frames: one event pair
        bool ordered = true;   // (on_compute: the compute stream is behind its own kernels)
        for (int i = 0; i < n && !on_compute; ++i)
            ordered = ordered && g[i]->tail_ordered && (!g[i]->dev_ir_word || pub_word_done(ctx, g[i]->dev_ir_word));
        if (!ordered) {
            FS_HIP(ctx, handoff_energy(ctx, g[0]));
            for (int i = 0; i < n; ++i) g[i]->tail_ordered = true;
        }
        unsigned slot_t = 0;
        { const int ar = acquire_recon_tab(ctx, &slot_t); if (ar) return ar; }
        ReconItem* tab = ctx->h_recon_tab + (size_t)slot_t * fs_context::kReconTabItems;
        std::vector<Source*> order(g, g + n);
        std::sort(order.begin(), order.end());                 // one locking order for every thread
        std::vector<std::unique_lock<std::mutex>> locks;
        locks.reserve((size_t)n);
        for (Source* s : order) locks.emplace_back(s->ir_mu);  // against fs_reverb_process on the audio thread
        // on the tail stream: ONE event for the batch (tail_batch_ev; the ids are totally ordered because only the tail stream
        // issues them); on the compute stream: the launch announces itself in the host word (publish_arrive)
        const uint64_t batch = on_compute ? 0 : ctx->tail_batch_newest.load(std::memory_order_relaxed) + 1;
        const PublishWord pub = on_compute ? next_pub_word(ctx) : PublishWord();
        for (int i = 0; i < n; ++i) {
            Source* s = g[i];
            if (on_compute) FS_HIP(ctx, compute_waits_for_tail_ir(ctx, s));   // nor write d_ir_* under a publish the tail stream still copies from
            else if (s->rev_recorded) FS_HIP(ctx, hipStreamWaitEvent(tail, s->ev_rev, 0));   // a reverb callback may be reading d_ir_mono
            // deterministic mode: the collective summed the fixed-point histogram; round it to fp32 once, now
            if (s->cur_fixed && !s->reduced) launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), B * ctx->num_bins, tail);
            const uint64_t seq = s->enqueued + 1;
            const int slot = (int)(seq % kIrRing);
            tab[i].energy = s->energy(); tab[i].ir_bands = s->d_ir_bands; tab[i].ir_mono = s->d_ir_mono; tab[i].host = s->h_ir[slot];
            tab[i].mask = slot_mask_ptr(ctx, s, slot);
            tab[i].spb = spb; tab[i].pad = 0;
        }
        launch_reconstruct_batch(tab, n, B, ctx->num_bins, ctx->num_samples, tail, pub);
        FS_HIP(ctx, hipGetLastError());
        if (on_compute) {
            ctx->pub_issued = pub.id;
            ctx->recon_tab_word[slot_t] = pub.id;
        } else {
            FS_HIP(ctx, hipEventRecord(ctx->tail_batch_ev[batch % fs_context::kTailBatches], tail));
            ctx->dbg.tail_ops += 2;
            ctx->tail_batch_newest.store(batch, std::memory_order_release);
            ctx->recon_tab_batch[slot_t] = batch;
        }
        for (int i = 0; i < n; ++i) {
            Source* s = g[i];
            s->rec_recorded[s->cur] = true; s->rec_batch[s->cur] = batch; s->rec_on_compute[s->cur] = on_compute;
            s->last_rec = s->cur;
            const uint64_t seq = s->enqueued + 1;
            note_publish(ctx, s, seq, (int)(seq % kIrRing), batch, pub.id);
            if (on_compute) {
                s->cur_pub_seq = 0; s->dev_ir_word = pub.id;
                if (s->d_ring) FS_HIP(ctx, hipEventRecord(s->ev_rec[s->cur], ctx->stream));   // fs_reverb_process reads d_ir_mono behind this
            } else {
                s->cur_pub_seq = seq; s->dev_ir_word = 0;
            }
        }
    }
    return FS_OK;
}

}  // namespace fsi
