// fs_capi_aux.cpp — rows beside the core path: legacy forward tracer (a9), the engine line trace (tests / tools),
// text interchange (f1), reverb convolution (f2), frequency-domain material response (f4).
#include "fs_context.hpp"

extern "C" {

// ---- legacy forward tracer (a9) -----------------------------------------------------------------------------
void fs_sound_params_default(fs_sound_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->struct_size = sizeof(fs_sound_params);
    p->seed = 0x5EEDull;
    p->raycasts_per_tick = 1500;   // FSAC.h:39
    p->raycast_bounces = 10;       // FSAC.h:42
    p->raycast_distance = 5000.f;  // FSAC.h:45
    p->simulated_duration = 1.0f;  // FSAC.h:136
    p->listener_radius = 34.0f;    // ADefaultPawn collision sphere (engine default, build-owned)
}

int fs_update_sound(fs_context* ctx, fs_source h, const fs_sound_params* p, fs_sound_result* out) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    { int ir = maybe_install_refined(ctx); if (ir) return ir; }                     // fs_scene_commit_progressive: the better tree is ready
    if (ctx->refit_pending) { int rr = fs_scene_refit(ctx); if (rr) return rr; }   // moved triangles: refit before tracing
    fs_sound_params def;
    if (!p) { fs_sound_params_default(&def); p = &def; }
    if (p->struct_size != sizeof(fs_sound_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_sound_params.struct_size mismatch");
    if (p->raycasts_per_tick < 0 || p->raycast_bounces < 0 || !(p->listener_radius >= 0.f))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad sound params");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (!ctx->d_sound) FS_HIP(ctx, hipMalloc((void**)&ctx->d_sound, sizeof(SoundAccum)));
    SoundKParams sp{};
    sp.seed_lo = (uint32_t)p->seed;
    sp.seed_hi = (uint32_t)(p->seed >> 32);
    sp.raycasts_per_tick = p->raycasts_per_tick;
    sp.raycast_bounces = p->raycast_bounces;
    sp.raycast_distance = p->raycast_distance;
    sp.simulated_duration = p->simulated_duration;
    sp.listener_radius = p->listener_radius;
    std::memcpy(sp.src, s->pos, sizeof(sp.src));
    std::memcpy(sp.lis, ctx->listener, sizeof(sp.lis));
    FS_HIP(ctx, hipMemsetAsync(ctx->d_sound, 0, sizeof(SoundAccum), ctx->stream));
    launch_update_sound(ctx->scene, sp, ctx->d_sound, ctx->sound_rays_per_wave, ctx->stream);
    FS_HIP(ctx, hipGetLastError());
    SoundAccum acc{};
    FS_HIP(ctx, hipMemcpyAsync(&acc, ctx->d_sound, sizeof(acc), hipMemcpyDeviceToHost, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->rays_reaching_listener = acc.reaching;
    out->direct_hits = acc.direct_hits;
    out->direct_energy_sum = acc.direct_energy_sum;
    out->traces = acc.traces;
    out->occlusion_attenuation = acc.occlusion;
    // TotalEnergy /= RaycastsPerTick (FSAC.cpp:294); every reaching ray returns Energy == 1
    out->total_energy = p->raycasts_per_tick > 0 ? (float)acc.reaching / (float)p->raycasts_per_tick : 0.0f;
    s->occlusion = acc.occlusion;
    return FS_OK;
}

int fs_get_occlusion_attenuation(fs_context* ctx, fs_source h, float* out) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return FS_ERR_BAD_HANDLE;
    *out = s->occlusion;
    return FS_OK;
}

// ---- engine line trace ------------------------------------------------------------------------------------
int fs_trace_rays(fs_context* ctx, const float* origins, const float* dirs, const float* tmax, int32_t N,
                  int32_t any_hit, int32_t* hit, float* t, int32_t* tri, float* normal) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    { int ir = maybe_install_refined(ctx); if (ir) return ir; }                     // fs_scene_commit_progressive: the better tree is ready
    if (ctx->refit_pending) { int rr = fs_scene_refit(ctx); if (rr) return rr; }   // moved triangles: refit before tracing
    if (N < 0 || (N > 0 && (!origins || !dirs || !tmax || !hit))) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad ray arrays");
    if (any_hit != 1 && N > 0 && (!t || !tri || !normal)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "closest-hit outputs required");
    if (any_hit < 0 || any_hit > 8) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "any_hit: 0 closest, 1 any, 2 .. 8 closest by the cooperative traversal");
    if (N == 0) return FS_OK;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    float *d_o = nullptr, *d_d = nullptr, *d_tm = nullptr, *d_t = nullptr, *d_n = nullptr;
    int32_t *d_hit = nullptr, *d_tri = nullptr;
    size_t n3 = sizeof(float) * 3 * (size_t)N, n1 = sizeof(float) * (size_t)N;
    int rc = FS_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_tm); (void)hipFree(d_t); (void)hipFree(d_n);
        (void)hipFree(d_hit); (void)hipFree(d_tri);
    };
#define FS_TRY(call)                                                   \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) { rc = ctx->hip_fail(e_, #call); cleanup(); return rc; } \
    } while (0)
    FS_TRY(hipMalloc((void**)&d_o, n3));
    FS_TRY(hipMalloc((void**)&d_d, n3));
    FS_TRY(hipMalloc((void**)&d_tm, n1));
    FS_TRY(hipMalloc((void**)&d_t, n1));
    FS_TRY(hipMalloc((void**)&d_n, n3));
    FS_TRY(hipMalloc((void**)&d_hit, n1 + sizeof(int32_t)));   // (+ the cooperative traversal's overflow word)
    FS_TRY(hipMemsetAsync(d_hit + N, 0, sizeof(int32_t), ctx->stream));
    FS_TRY(hipMalloc((void**)&d_tri, n1));
    FS_TRY(hipMemcpyAsync(d_o, origins, n3, hipMemcpyHostToDevice, ctx->stream));
    FS_TRY(hipMemcpyAsync(d_d, dirs, n3, hipMemcpyHostToDevice, ctx->stream));
    FS_TRY(hipMemcpyAsync(d_tm, tmax, n1, hipMemcpyHostToDevice, ctx->stream));
    launch_trace_rays(ctx->scene, d_o, d_d, d_tm, N, any_hit, d_hit, d_t, d_tri, d_n, ctx->stream);
    FS_TRY(hipGetLastError());
    FS_TRY(hipMemcpyAsync(hit, d_hit, n1, hipMemcpyDeviceToHost, ctx->stream));
    int32_t coop_overflow = 0;
    FS_TRY(hipMemcpyAsync(&coop_overflow, d_hit + N, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (any_hit != 1) {
        FS_TRY(hipMemcpyAsync(t, d_t, n1, hipMemcpyDeviceToHost, ctx->stream));
        FS_TRY(hipMemcpyAsync(tri, d_tri, n1, hipMemcpyDeviceToHost, ctx->stream));
        FS_TRY(hipMemcpyAsync(normal, d_n, n3, hipMemcpyDeviceToHost, ctx->stream));
    }
    FS_TRY(hipStreamSynchronize(ctx->stream));
#undef FS_TRY
    cleanup();
    if (coop_overflow) return ctx->fail(FS_ERR_OVERFLOW, "cooperative traversal: a group's node stack overflowed");
    return FS_OK;
}

// ---- f1: text import / export (FSAC.cpp:454-505) ------------------------------------------------------------
extern "C++" {
namespace {
// FString::SanitizeFloat(double, MinFractionalDigits = 1): "%f", trailing zeros trimmed, one fractional digit kept
std::string sanitize_float(double v) {
    if (v == 0.0) v = 0.0;  // strip negative zero
    char buf[512];
    std::snprintf(buf, sizeof(buf), "%f", v);
    std::string t(buf);
    bool numeric = !t.empty();
    for (size_t i = 0; i < t.size(); ++i) {
        char c = t[i];
        if (!((c >= '0' && c <= '9') || c == '.' || ((c == '-' || c == '+') && i == 0))) numeric = false;
    }
    if (!numeric) return t;  // "nan", "inf": left alone like the engine
    size_t dot = t.find('.');
    if (dot == std::string::npos) return t + ".0";
    size_t end = t.size();
    while (end > dot + 2 && t[end - 1] == '0') --end;
    return t.substr(0, end);
}
}  // namespace
}  // extern "C++"

int fs_save_array_to_file(const float* data, int32_t n, const char* path) {
    if (!path || n < 0 || (n > 0 && !data)) return FS_ERR_INVALID_ARGUMENT;
    FILE* f = std::fopen(path, "wb");
    if (!f) return FS_ERR_INVALID_ARGUMENT;
    for (int32_t i = 0; i < n; ++i) {   // FString::Join(Lines, "\n"): no trailing newline
        std::string s = sanitize_float((double)data[i]);
        if (i) std::fputc('\n', f);
        std::fwrite(s.data(), 1, s.size(), f);
    }
    std::fclose(f);
    return FS_OK;
}

int fs_load_float_array(const char* path, float* out, int32_t cap, int32_t* n_out) {
    if (!path || !n_out || cap < 0) return FS_ERR_INVALID_ARGUMENT;
    *n_out = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return FS_ERR_INVALID_ARGUMENT;   // "Failed to load impulse response file" FSAC.cpp:472
    std::string content;
    char buf[65536];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) content.append(buf, got);
    std::fclose(f);
    int32_t n = 0;
    size_t pos = 0;
    while (pos <= content.size()) {
        size_t nl = content.find('\n', pos);
        if (nl == std::string::npos) nl = content.size();
        if (nl > pos) {                   // ParseIntoArray(..., InCullEmpty = true)
            std::string line = content.substr(pos, nl - pos);
            float v = (float)std::atof(line.c_str());   // FCString::Atof
            if (out && n < cap) out[n] = v;
            ++n;
        }
        pos = nl + 1;
    }
    *n_out = n;
    return FS_OK;
}

int fs_save_impulse_response(fs_context* ctx, fs_source h, int32_t channel, const char* path) {
    if (!ctx || !path) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    const float* p = nullptr;
    int32_t n = 0;
    int rc = fs_get_impulse_response(ctx, h, channel, &p, &n);
    if (rc) return rc;
    return fs_save_array_to_file(p, n, path);
}

// ---- f2: reverb convolution (RVB.cpp:74-213) ---------------------------------------------------------------------
int fs_reverb_init(fs_context* ctx, fs_source h, int32_t frame_size) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (frame_size < 1 || frame_size > 16384 || ctx->num_samples - 1 > kReverbRing)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad reverb frame size / IR longer than the history ring");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // Reconstructs on the compute stream record an event for the callbacks only for a source that has a reverb: the ones
    // already in flight finish before this source gets one.
    FS_FLUSH(ctx);
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->rev_stream));
    if (s->d_ring) { (void)hipFree(s->d_ring); (void)hipFree(s->d_rev_in); (void)hipFree(s->d_rev_cur); (void)hipFree(s->d_rev_out); }
    s->d_ring = s->d_rev_in = s->d_rev_cur = s->d_rev_out = nullptr;
    FS_HIP(ctx, hipMalloc((void**)&s->d_ring, sizeof(float) * 2 * kReverbRing));
    FS_HIP(ctx, hipMalloc((void**)&s->d_rev_in, sizeof(float) * 2 * (size_t)frame_size));
    FS_HIP(ctx, hipMalloc((void**)&s->d_rev_cur, sizeof(float) * 2 * (size_t)frame_size));
    FS_HIP(ctx, hipMalloc((void**)&s->d_rev_out, sizeof(float) * 2 * (size_t)frame_size));
    FS_HIP(ctx, hipMemsetAsync(s->d_ring, 0, sizeof(float) * 2 * kReverbRing, ctx->rev_stream));   // SetNumZeroed
    s->rev_head = 0;
    s->rev_frame = frame_size;
    return FS_OK;
}

int fs_reverb_process(fs_context* ctx, fs_source h, const float* in, float* out, int32_t apply_reverb, uint32_t flags) {
    if (!ctx || !in || !out) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (!s->d_ring) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_reverb_init has not been called for this source");
    const int frame = s->rev_frame;
    if (!apply_reverb) {   // bApplyReverb == false: RVB.cpp:128-132
        std::memcpy(out, in, sizeof(float) * 2 * (size_t)frame);
        return FS_OK;
    }
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // Audio thread.  The callback has its own stream: it is never queued behind a traced frame on the compute stream.
    // The device-resident IR is written by reconstructs on the tail stream: read it behind the newest one and make the
    // next one wait for this read — both through events, exchanged with the game thread under the source's ir_mu.
    hipStream_t rs = ctx->rev_stream;
    {
        std::lock_guard<std::mutex> g(s->ir_mu);
        if (s->last_rec >= 0) FS_HIP(ctx, stream_waits_for_rec(ctx, rs, s, s->last_rec));
        FS_HIP(ctx, hipMemcpyAsync(s->d_rev_in, in, sizeof(float) * 2 * (size_t)frame, hipMemcpyHostToDevice, rs));
        launch_reverb(s->d_ir_mono, ctx->num_samples, s->d_ring, s->rev_head, s->d_rev_in, s->d_rev_cur, s->d_rev_out, frame,
                      (flags & FS_REVERB_LITERAL_TAIL) ? 1 : 0, rs);
        FS_HIP(ctx, hipGetLastError());
        FS_HIP(ctx, hipEventRecord(s->ev_rev, rs));
        s->rev_recorded = true;
    }
    s->rev_head += (unsigned)frame;
    FS_HIP(ctx, hipMemcpyAsync(out, s->d_rev_out, sizeof(float) * 2 * (size_t)frame, hipMemcpyDeviceToHost, rs));
    FS_HIP(ctx, hipStreamSynchronize(rs));
    return FS_OK;
}

int fs_reverb_release(fs_context* ctx, fs_source h) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (s->d_ring && ctx->device_ok) {
        FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
        FS_HIP(ctx, hipMemsetAsync(s->d_ring, 0, sizeof(float) * 2 * kReverbRing, ctx->rev_stream));
        s->rev_head = 0;
    }
    return FS_OK;
}

// ---- row f4: UMaterialAcousticProcessor::ApplyMaterialFD (MaterialAcousticProcessor.cpp:8-107) -----------------
int fs_apply_material_fd(fs_context* ctx, const float* in, int32_t L, const float* absorption, const float* transmission,
                         const float* scattering, int32_t num_responses, float* specular, float* diffuse,
                         float* transmitted) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (L < 0 || L > (1 << 24)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "block length out of range (0 .. 2^24)");
    if (!absorption || !transmission || !scattering) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "null response curve");
    if (L > 0 && (!in || !specular || !diffuse || !transmitted)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "null buffer");
    int n = 0;
    while ((1 << n) < L) ++n;                                      // MAP.cpp:15-16: next power of two
    const int N = 1 << n, bins = N / 2 + 1;
    if (num_responses != bins)                                     // MAP.cpp:20-26
        return ctx->fail(FS_ERR_SIZE_MISMATCH, "all response curves must have length " + std::to_string(bins));
    if (L == 0) return FS_OK;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no device");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    if (n > ctx->fft_cap_n || L > ctx->fft_cap_l) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (void* p : {(void*)ctx->d_fft_x, (void*)ctx->d_fft_y, (void*)ctx->d_fft_w, (void*)ctx->d_fft_in,
                        (void*)ctx->d_fft_resp, (void*)ctx->d_fft_out})
            if (p) (void)hipFree(p);
        ctx->d_fft_x = ctx->d_fft_y = ctx->d_fft_w = nullptr;
        ctx->d_fft_in = ctx->d_fft_resp = ctx->d_fft_out = nullptr;
        ctx->fft_cap_n = -1; ctx->fft_cap_l = 0; ctx->fft_n = -1;
        if (ctx->fft_graph) { (void)hipGraphExecDestroy(ctx->fft_graph); ctx->fft_graph = nullptr; }   // captured the old buffers
        ctx->fft_graph_n = -1;
        const int cn = std::max(n, ctx->fft_cap_n);
        const size_t CN = (size_t)1 << cn;
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_x, sizeof(float2) * CN));
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_y, sizeof(float2) * 3 * CN));
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_w, sizeof(float2) * std::max<size_t>(CN / 2, 1)));
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_in, sizeof(float) * CN));
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_resp, sizeof(float) * 3 * (CN / 2 + 1)));
        FS_HIP(ctx, hipMalloc(&ctx->d_fft_out, sizeof(float) * 3 * CN));
        ctx->fft_cap_n = cn; ctx->fft_cap_l = (int)CN;
    }
    if (ctx->fft_n != n) {   // twiddles in double precision: W[k] = exp(-2 pi i k / N)
        std::vector<float2> w(std::max(N / 2, 1));
        for (int k = 0; k < N / 2; ++k) {
            const double a = -2.0 * 3.14159265358979323846 * (double)k / (double)N;
            w[(size_t)k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        if (N < 2) w[0] = make_float2(1.f, 0.f);
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_fft_w, w.data(), sizeof(float2) * w.size(), hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));   // w is a stack-owned staging buffer
        ctx->fft_n = n;
    }
    if (L > (1 << 17)) {   // large blocks are copy-bound, not launch-bound: straight from / to the caller's memory
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_fft_in, in, sizeof(float) * (size_t)L, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_fft_resp, absorption, sizeof(float) * bins, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_fft_resp + bins, transmission, sizeof(float) * bins, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(ctx->d_fft_resp + 2 * bins, scattering, sizeof(float) * bins, hipMemcpyHostToDevice, ctx->stream));
        launch_apply_material_fd(ctx->d_fft_in, L, n, ctx->d_fft_x, ctx->d_fft_y, ctx->d_fft_w, ctx->d_fft_resp, ctx->d_fft_out,
                                 ctx->stream);
        FS_HIP(ctx, hipGetLastError());
        FS_HIP(ctx, hipMemcpyAsync(specular, ctx->d_fft_out, sizeof(float) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(diffuse, ctx->d_fft_out + L, sizeof(float) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(ctx, hipMemcpyAsync(transmitted, ctx->d_fft_out + 2 * (size_t)L, sizeof(float) * (size_t)L, hipMemcpyDeviceToHost, ctx->stream));
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FS_OK;
    }
    // pinned staging so that the captured copies always use the same addresses
    const size_t stage_floats = (size_t)L + 3 * (size_t)bins + 3 * (size_t)L;
    if (stage_floats > ctx->fft_stage_floats) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->fft_graph) { (void)hipGraphExecDestroy(ctx->fft_graph); ctx->fft_graph = nullptr; ctx->fft_graph_n = -1; }
        if (ctx->h_fft_stage) (void)hipHostFree(ctx->h_fft_stage);
        ctx->h_fft_stage = nullptr; ctx->fft_stage_floats = 0;
        FS_HIP(ctx, hipHostMalloc((void**)&ctx->h_fft_stage, sizeof(float) * stage_floats, hipHostMallocDefault));
        ctx->fft_stage_floats = stage_floats;
    }
    float* h_in = ctx->h_fft_stage;
    float* h_resp = h_in + L;
    float* h_out = h_resp + 3 * (size_t)bins;
    std::memcpy(h_in, in, sizeof(float) * (size_t)L);
    std::memcpy(h_resp, absorption, sizeof(float) * (size_t)bins);
    std::memcpy(h_resp + bins, transmission, sizeof(float) * (size_t)bins);
    std::memcpy(h_resp + 2 * (size_t)bins, scattering, sizeof(float) * (size_t)bins);
    auto enqueue = [&](hipStream_t st) -> hipError_t {
        hipError_t e = hipMemcpyAsync(ctx->d_fft_in, h_in, sizeof(float) * (size_t)L, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(ctx->d_fft_resp, h_resp, sizeof(float) * 3 * (size_t)bins, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) return e;
        launch_apply_material_fd(ctx->d_fft_in, L, n, ctx->d_fft_x, ctx->d_fft_y, ctx->d_fft_w, ctx->d_fft_resp,
                                 ctx->d_fft_out, st);
        return hipMemcpyAsync(h_out, ctx->d_fft_out, sizeof(float) * 3 * (size_t)L, hipMemcpyDeviceToHost, st);
    };
    if (ctx->fft_graph_n != n || ctx->fft_graph_l != L) {   // (re)capture for this block size
        if (ctx->fft_graph) { (void)hipGraphExecDestroy(ctx->fft_graph); ctx->fft_graph = nullptr; }
        ctx->fft_graph_n = -1;
        FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            hipError_t e = enqueue(ctx->stream);
            hipError_t e2 = hipStreamEndCapture(ctx->stream, &g);
            if (e == hipSuccess && e2 == hipSuccess && g &&
                hipGraphInstantiate(&ctx->fft_graph, g, nullptr, nullptr, 0) == hipSuccess) {
                ctx->fft_graph_n = n; ctx->fft_graph_l = L;
            } else {
                ctx->fft_graph = nullptr;
            }
            if (g) (void)hipGraphDestroy(g);
        }
        (void)hipGetLastError();
    }
    if (ctx->fft_graph) FS_HIP(ctx, hipGraphLaunch(ctx->fft_graph, ctx->stream));
    else FS_HIP(ctx, enqueue(ctx->stream));   // capture unavailable: the same sequence, launched one by one
    FS_HIP(ctx, hipGetLastError());
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(specular, h_out, sizeof(float) * (size_t)L);
    std::memcpy(diffuse, h_out + L, sizeof(float) * (size_t)L);
    std::memcpy(transmitted, h_out + 2 * (size_t)L, sizeof(float) * (size_t)L);
    return FS_OK;
}

}  // extern "C"
