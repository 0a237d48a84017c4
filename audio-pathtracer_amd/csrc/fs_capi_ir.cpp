// fs_capi_ir.cpp — the reconstruct / publish entry points, the tick as one call, impulse-response and energy-buffer access
// (C ABI: include/frequensee.h).  The machinery behind them — the tail stream, the fused reconstruct parts, the IR ring —
// lives in fs_capi_publish.cpp (and the frame pipeline around it in fs_capi_pipeline.cpp).
#include "fs_context.hpp"

using namespace fsi;

extern "C" {

int fs_reconstruct_impulse_response_async(fs_context* ctx, fs_source h, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    fs_params def;
    if (!p) { fs_params_default(&def); p = &def; }
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    if (p->samples_per_bin < 0 || p->samples_per_bin > 32767) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "samples_per_bin out of range (0 = from the configuration, else 1 .. 32767)");
    // grouped frames: the source's current frame still waits for its launch — the reconstruct is recorded with it
    for (size_t k = ctx->group.size(); k-- > 0;) {
        fs_context::GroupEntry& e = ctx->group[k];
        if (e.s != s) continue;
        if (!e.want_recon && !(p->flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT) && (ctx->cfg.world_size == 1 || ctx->comm)) {
            e.want_recon = true;
            e.recon = *p;
            return FS_OK;
        }
        break;
    }
    if (!ctx->group.empty()) FS_FLUSH(ctx);   // (a second reconstruct of the same frame, or one of another kind: the frames go first)
    // pipelined frames: the source's current frame still waits for its connect pass — the reconstruct goes with it
    for (size_t k = ctx->held.size(); k-- > 0;) {   // the source's CURRENT frame is the newest held one that has it
        fs_context::PipeFrame& q = ctx->held[k];
        fs_context::PipeFrame::Item* it = nullptr;
        for (fs_context::PipeFrame::Item& c : q.items) if (c.s == s) it = &c;
        if (!it) continue;
        if (!it->want_recon && !(p->flags & FS_FLAG_FLUSH_BEFORE_RECONSTRUCT) && (ctx->cfg.world_size == 1 || ctx->comm)) {
            it->want_recon = true;
            it->recon = *p;
            return FS_OK;
        }
        break;
    }
    FS_FLUSH(ctx);
    // A frame nobody holds (pipelining off, or a frame of another shape).  A producer that WAITS for every frame — it has synchronized
    // since this source's last reconstruct: the reference's UpdateSource, FSAC.cpp:377-378 — gets the short way: on one GPU the
    // reconstruct goes onto the COMPUTE stream as a batch of one, like a flush's; the kernel writes the published host slot itself — no
    // hand-over to the tail stream, no copy command, no mask reset (30 us of gaps between four small commands: the reference-sized
    // update 0.32 -> 0.29 ms).  A producer that streams frames without waiting keeps the tail stream's overlap with its next frame
    // (712 against 688 M rays/s for unpipelined cfg3 frames), and a buffer the tail stream owns (fs_energy_handoff: the caller's
    // collective is there) is reconstructed there, behind it.
    const bool waited = s->recon_sync_mark != ctx->syncs;
    s->recon_sync_mark = ctx->syncs;
    return (s->tail_ordered || !waited) ? reconstruct_now(ctx, s, p) : flush_reconstruct(ctx, s, p);
}

int fs_reconstruct_impulse_response_batch_async(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (count < 0 || (count > 0 && !sources)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad source list");
    if (count == 0) return FS_OK;
    fs_params def;
    if (!p) { fs_params_default(&def); p = &def; }
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    if (p->samples_per_bin < 0 || p->samples_per_bin > 32767) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "samples_per_bin out of range (0 = from the configuration, else 1 .. 32767)");
    std::vector<Source*> srcs((size_t)count);
    for (int32_t i = 0; i < count; ++i) {
        srcs[(size_t)i] = get_source(ctx, sources[i]);
        if (!srcs[(size_t)i]) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
        for (int32_t k = 0; k < i; ++k)
            if (sources[k] == sources[i]) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "a source appears twice in the batch");
    }
    // frames that still wait for their launch (pipelined / grouped): every reconstruct is recorded with its frame, as the single call does
    if (!ctx->group.empty() || !ctx->held.empty()) {
        for (int32_t i = 0; i < count; ++i) { const int rc = fs_reconstruct_impulse_response_async(ctx, sources[i], p); if (rc) return rc; }
        return FS_OK;
    }
    FS_FLUSH(ctx);
    return reconstruct_batch(ctx, srcs.data(), count, p, false);
}

// UpdateSources (ARTS.cpp:100-126) as the game thread runs it: every listed source gets its UpdateSource and the call returns
// when every IR is in its published host buffer.
int fs_update_sources(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (count < 0 || (count > 0 && !sources)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad source list");
    if (count == 0) return FS_OK;
    fs_params def;
    if (!p) { fs_params_default(&def); p = &def; }
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    if (p->samples_per_bin < 0 || p->samples_per_bin > 32767) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "samples_per_bin out of range (0 = from the configuration, else 1 .. 32767)");
    std::vector<Source*> srcs((size_t)count);
    int rc = FS_OK;
    // depth = 0 only: a tick whose record store overflowed is traced and reconstructed again (up to 4 attempts; the store has been
    // grown meanwhile).  Its reconstruct rode behind the frame without a wait in between — that wait would cost every tick of the
    // reference's own mode ~ 15 us for an event of the first few ticks at most — so the impulse responses (and sequence numbers)
    // published by an attempt that then reports FS_ERR_OVERFLOW are PROVISIONAL: the next attempt publishes the complete ones
    // behind them, and only after the last failed attempt does an incomplete IR stay in front (the call returns FS_ERR_OVERFLOW).
    for (int attempt = 0; attempt < 4; ++attempt) {
        rc = fs_compute_energy_response_batch_async(ctx, sources, count, p);
        if (rc) return rc;
        FS_FLUSH(ctx);                                 // the caller waits: nothing is held back
        for (int32_t i = 0; i < count; ++i) srcs[(size_t)i] = get_source(ctx, sources[i]);
        // single GPU: the reconstructs ride on the compute stream; a sharded frame's sum lives on the tail stream, so do they then
        rc = reconstruct_batch(ctx, srcs.data(), count, p, /*on_compute=*/ctx->comm == nullptr && ctx->cfg.world_size == 1);
        if (rc) return rc;
        // ONE wait for the tick.  depth = 0: a frame whose records overflowed is found here, after its reconstruct — it is
        // traced and reconstructed again (the IR published in between came from an incomplete frame and is replaced)
        rc = fs_synchronize(ctx);
        if (rc != FS_ERR_OVERFLOW) break;
    }
    return rc;
}

int fs_set_impulse_response(fs_context* ctx, fs_source h, const float* ir, int32_t n) {
    if (!ctx || !ir) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (n != ctx->num_samples) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != num_samples");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    { const int br = fsi::ir_ring_backpressure_for(ctx, s); if (br) return br; }
    hipStream_t tail = ctx->copy_stream;   // ordered with reconstructs and publishes of this source
    const size_t bytes = sizeof(float) * (size_t)n;
    {
        std::lock_guard<std::mutex> g(s->ir_mu);   // against fs_reverb_process on the audio thread
        FS_HIP(ctx, tail_waits_for_compute_ir(ctx, s));   // (a reconstruct on the compute stream may still be writing d_ir_*)
        if (s->rev_recorded) FS_HIP(ctx, hipStreamWaitEvent(tail, s->ev_rev, 0));
        FS_HIP(ctx, hipMemcpyAsync(s->d_ir_mono, ir, bytes, hipMemcpyHostToDevice, tail));
        for (int b = 0; b < ctx->cfg.num_bands; ++b)
            FS_HIP(ctx, hipMemcpyAsync(s->d_ir_bands + (size_t)b * (size_t)n, s->d_ir_mono, bytes, hipMemcpyDeviceToDevice, tail));
        const int cur = s->last_rec >= 0 ? s->last_rec : s->cur;
        FS_HIP(ctx, hipEventRecord(s->ev_rec[cur], tail));   // the reverb waits on this before reading d_ir_mono
        s->rec_recorded[cur] = true; s->rec_batch[cur] = 0; s->rec_on_compute[cur] = false;
        s->last_rec = cur;
    }
    uint64_t seq = s->enqueued + 1;
    int slot = (int)(seq % kIrRing);
    FS_HIP(ctx, hipMemcpyAsync(s->h_ir[slot], s->d_ir_mono, bytes, hipMemcpyDeviceToHost, tail));
    FS_HIP(ctx, slot_mask_all_dirty(ctx, s, slot, tail));
    FS_HIP(ctx, hipEventRecord(s->ev[slot], tail));
    ctx->dbg.tail_ops += 4 + (uint64_t)ctx->cfg.num_bands; ctx->dbg.pub_event++;
    s->pub_word[slot] = 0; s->pub_batch[slot] = 0; s->seq_of[slot] = seq; s->enqueued = seq; s->cur_pub_seq = seq; s->dev_ir_word = 0;
    FS_HIP(ctx, hipStreamSynchronize(tail));   // `ir` is the caller's memory
    poll_published(ctx, s);
    return FS_OK;
}

int fs_synchronize(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    ctx->syncs++;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (Source* s : ctx->sources)
        if (s && s->alive) poll_published(ctx, s);
    resolve_timings(ctx);
    { const int oc = oneshot_check(ctx); if (oc) return oc; }
    return check_overflow(ctx);   // FS_ERR_OVERFLOW: the last depth = 0 frame must be traced again (see the header)
}

int fs_reconstruct_impulse_response(fs_context* ctx, fs_source h, const fs_params* p) {
    int rc = fs_reconstruct_impulse_response_async(ctx, h, p);
    if (rc) return rc;
    return fs_synchronize(ctx);
}

int fs_get_impulse_response(fs_context* ctx, fs_source h, int32_t channel, const float** data, int32_t* n) {
    if (!ctx || !data) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return FS_ERR_BAD_HANDLE;  // no err string write: may be called from the audio thread
    if (channel < 0 || channel >= ctx->cfg.num_channels) return FS_ERR_INVALID_ARGUMENT;
    uint64_t f = s->front.load(std::memory_order_acquire);
    *data = s->h_ir[(int)(f % kIrRing)];  // f == 0: slot 0 still holds the zero-initialised IR
    if (n) *n = ctx->num_samples;
    return FS_OK;
}

int fs_get_impulse_response_sequence(fs_context* ctx, fs_source h, uint64_t* completed) {
    if (!ctx || !completed) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return FS_ERR_BAD_HANDLE;  // no err string write: may be called from the audio thread
    poll_published(ctx, s);   // also notices publishes that completed since the producer's last call
    *completed = s->front.load(std::memory_order_acquire);
    return FS_OK;
}

int fs_copy_impulse_response(fs_context* ctx, fs_source h, int32_t channel, float* out, int32_t n) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (n != ctx->num_samples) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != num_samples");
    const float* p = nullptr;
    int rc = fs_get_impulse_response(ctx, h, channel, &p, nullptr);
    if (rc) return rc;
    std::memcpy(out, p, sizeof(float) * (size_t)n);
    return FS_OK;
}

int fs_copy_band_impulse_response(fs_context* ctx, fs_source h, int32_t band, float* out, int32_t n) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (band < 0 || band >= ctx->cfg.num_bands) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "band out of range");
    if (n != ctx->num_samples) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != num_samples");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));   // the reconstruct runs on the tail stream
    FS_HIP(ctx, hipMemcpyAsync(out, s->d_ir_bands + (size_t)band * (size_t)ctx->num_samples, sizeof(float) * (size_t)n,
                               hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

// ---- energy-buffer helpers (FSAC.h:72-91) ---------------------------------------------------------------
int fs_get_energy_buffer(fs_context* ctx, fs_source h, float* out, int32_t n) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (n != ctx->cfg.num_bands * ctx->num_bins) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != bands * bins");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));   // a collective on the tail stream may still be summing it
    FS_HIP(ctx, hipMemcpyAsync(out, s->energy(), sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

int fs_flush_energy_buffer(fs_context* ctx, fs_source h) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    s->cur_fixed = false;   // the fp32 buffer is the truth again (a deterministic frame left its fixed-point twin behind)
    s->handed_off = true;   // the caller owns the content now
    s->tail_ordered = false;  // ... and writes it on the compute stream below
    FS_HIP(ctx, wait_energy_readers(ctx, s));
    FS_HIP(ctx, hipMemsetAsync(s->energy(), 0, sizeof(float) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins,
                               ctx->stream));
    return FS_OK;
}

int fs_add_energy_at_delay(fs_context* ctx, fs_source h, int32_t band, float delay_seconds, float energy) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (band < 0 || band >= ctx->cfg.num_bands) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "band out of range");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    s->cur_fixed = false;   // the fp32 buffer is the truth again (a deterministic frame left its fixed-point twin behind)
    s->handed_off = true;   // the caller owns the content now
    s->tail_ordered = false;  // ... and writes it on the compute stream below
    FS_HIP(ctx, wait_energy_readers(ctx, s));
    launch_add_energy(s->energy() + (size_t)band * (size_t)ctx->num_bins, ctx->num_bins, delay_seconds, energy,
                      ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    return FS_OK;
}

int fs_update_energy_buffer(fs_context* ctx, fs_source h, const float* values, int32_t n) {
    if (!ctx || !values) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    // check(NewEnergyValues.Num() == NumBins) FSAC.h:83 -> status instead of abort
    if (n != ctx->cfg.num_bands * ctx->num_bins) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != bands * bins");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    s->cur_fixed = false;   // the fp32 buffer is the truth again (a deterministic frame left its fixed-point twin behind)
    s->handed_off = true;   // the caller owns the content now
    s->tail_ordered = false;  // ... and writes it on the compute stream below
    FS_HIP(ctx, wait_energy_readers(ctx, s));
    FS_HIP(ctx, hipMemcpyAsync(s->energy(), values, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

}  // extern "C"
