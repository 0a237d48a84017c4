// fs_capi.cpp — host side of the C ABI declared in include/frequensee.h.
//
// Mirrors the roles of UAudioRayTracingSubsystem (context lifetime, geometry/source registries,
// per-source update: AudioRayTracingSubsystem.cpp:32-53, 128-195) and of UFrequenSeeAudioComponent's
// buffers (EnergyBuffer, ImpulseBuffer: FrequenSeeAudioComponent.h:69-91, 113, 133-143).  All compute
// is HIP on the context's stream; there is no CPU fallback.
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is opened at run time (fs_comm_*), never linked

#include "fs_internal.hpp"

using namespace fs;

namespace {

constexpr int kIrRing = 3;  // published IR ring: a returned pointer stays valid until the second-next publish

// Energy buffers per source, used in rotation: frame f deposits into one while the tail stream still reduces /
// reconstructs frame f - 1 from another; pipelined frames finish frame f - 2 only in the launch that plans frame f
// (and flushes ITS buffer), so that a fourth one lets frame f + 1 start without waiting for that tail.
constexpr int kEnergyBufs = 4;
constexpr int kScratchSets = 3;   // frame scratch (plan counts, cursors, work counters): plan f, walk f-1, connect f-2 in one launch
constexpr int kPermSets = 2;      // schedules: plan f writes one while walk f-1 reads the other

struct Source {
    bool alive = false;
    float pos[3] = {0, 0, 0};
    // Two energy buffers [B][bins], alternating per frame: while the tail stream still reduces /
    // reconstructs frame f from one of them, the compute stream already traces frame f+1 into the other.
    float* d_energy[kEnergyBufs] = {};
    int cur = 0;                       // buffer of the current frame (rotates in fs_compute_energy_response*)
    hipEvent_t ev_dep = nullptr;       // compute stream: everything that writes the current buffer is enqueued
    hipEvent_t ev_rec[kEnergyBufs] = {};   // tail stream: the reconstruct that read buffer i is done
    bool rec_recorded[kEnergyBufs] = {};
    int last_rec = -1;                 // buffer the newest reconstruct read (its event also guards d_ir_*)
    hipEvent_t ev_rev = nullptr;       // reverb stream: the newest reverb callback has read d_ir_mono
    bool rev_recorded = false;
    // fs_reverb_process runs on the AUDIO thread while the game thread reconstructs: ir_mu guards what both touch —
    // last_rec / rec_recorded / ev_rec (written by the reconstruct, read by the callback) and rev_recorded / ev_rev
    // (the other way round).  Held only while work is ENQUEUED (microseconds), never across a stream wait.
    std::mutex ir_mu;
    float* energy() const { return d_energy[cur]; }
    // multi-GPU: the frame in the current buffer has been summed over the ranks (library collective), or handed to the
    // caller's collective (fs_energy_handoff); a world_size > 1 context refuses to reconstruct a frame that is neither
    bool reduced = false, handed_off = false;
    hipEvent_t ev_red[kEnergyBufs] = {};   // tail stream: the library's all-reduce of buffer i is done
    bool red_recorded[kEnergyBufs] = {};
    // deterministic mode (FS_FLAG_DETERMINISTIC): u64 fixed-point histograms [B][bins], allocated on first use,
    // alternating like the energy buffers; cur_fixed = the current frame deposited into d_fixed[cur]
    unsigned long long* d_fixed[kEnergyBufs] = {};
    bool cur_fixed = false;
    float* d_ir_bands = nullptr;  // [B][samples]
    float* d_ir_mono = nullptr;   // [samples] channel view (all channels identical, FSAC.cpp:331)
    float* h_ir[kIrRing] = {nullptr, nullptr, nullptr};  // pinned host copies of the channel view
    hipEvent_t ev[kIrRing] = {nullptr, nullptr, nullptr};
    uint64_t seq_of[kIrRing] = {0, 0, 0};
    uint64_t enqueued = 0;             // publishes enqueued so far
    std::atomic<uint64_t> front{0};    // newest COMPLETED publish (0 = none yet)
    // reverb (row f2): history rings [2][kReverbRing], staging buffers, write head
    float* d_ring = nullptr; float* d_rev_in = nullptr; float* d_rev_cur = nullptr; float* d_rev_out = nullptr;
    unsigned rev_head = 0; int rev_frame = 0;
    float occlusion = 1.0f;            // OcclusionAttenuation FSAC.h:130 (1.f until the first UpdateSound)
};

struct TimedFrame {
    hipEvent_t e[5];  // walk begin, walk end == connect begin, connect end | reconstruct begin, end
    bool has_trace = false, has_recon = false;
};

}  // namespace

// fs_scene_commit_progressive: the host's SAH build of a snapshot of the registered triangles, on its own thread, while
// the frames already trace through the device-built tree.  The thread touches nothing but this object.
struct RefineJob {
    std::vector<float> xyz; std::vector<uint16_t> mat; std::vector<uint32_t> obj;
    int T = 0;
    fs::HostBVH bvh;
    std::mutex mu; std::condition_variable cv;
    bool done = false;
};

struct fs_context {
    fs_config cfg{};
    int num_bins = 0, num_samples = 0;
    hipStream_t stream = nullptr;
    // "tail" stream: [caller's all-reduce] -> reconstruct -> publish of frame f, concurrent with the tracing of
    // frame f+1 on `stream`
    hipStream_t copy_stream = nullptr;
    hipStream_t rev_stream = nullptr;  // the reverb callbacks' own stream (audio thread): never queued behind a traced frame
    bool own_stream = false;
    bool device_ok = false;
    std::string err;
    std::mutex err_mu;                 // the audio thread may fail too

    // scene (host staging + device)
    std::vector<float> h_xyz;
    std::vector<uint16_t> h_mat;
    std::vector<uint32_t> h_obj;   // actor id per triangle (empty = one actor per triangle)
    std::vector<float> h_absorption, h_transmission, h_scattering;
    int32_t T = 0, M = 0;
    bool committed = false;
    NodeQ4* d_nodes = nullptr;
    Tri64* d_tris = nullptr;
    float* d_absorption = nullptr;
    SoundAccum* d_sound = nullptr;
    // refit support (row f4, fs_refit.hip)
    uint32_t* d_leaf_pos = nullptr;   // input triangle -> leaf-order position
    float4* d_node_box = nullptr;     // [nodes][2] fp32 bounds scratch
    float* d_move = nullptr;          // staging for moved triangles
    size_t move_cap = 0;              // in triangles
    float amax = 0.f;                 // largest |coordinate| seen (sets the box padding)
    char* d_build = nullptr;          // fs_scene_commit_fast: device copies of the inputs + build scratch (grow-only)
    size_t build_cap = 0;
    size_t fast_cap_tris = 0;         // triangles the scene arrays of the last fast commit have room for (0: not reusable)
    bool refit_pending = false;
    DeviceScene scene{};
    // ApplyMaterialFD work buffers (row f4), sized for the largest block seen
    int fft_n = -1;              // log2 of the size the twiddle table was built for
    int fft_cap_n = -1, fft_cap_l = 0;
    float2 *d_fft_x = nullptr, *d_fft_y = nullptr, *d_fft_w = nullptr;
    float *d_fft_in = nullptr, *d_fft_resp = nullptr, *d_fft_out = nullptr;
    // the 2 copies + up to 2 log2(N) - 20 launches of one block size are captured once into a hipGraph and replayed:
    // the sequence is launch-bound (16 launches for a 48 000-sample block)
    float* h_fft_stage = nullptr;      // pinned: in [L] | curves [3][bins] | out [3][L]
    size_t fft_stage_floats = 0;
    hipGraphExec_t fft_graph = nullptr;
    int fft_graph_n = -1, fft_graph_l = -1;
    HostBVH bvh;

    float listener[3] = {0, 0, 0};
    std::vector<Source*> sources;

    // multi-GPU: RCCL communicator over the ranks that share the pairs of every frame
    ncclComm_t comm = nullptr;
    // cfg5 (independent sources, one per GPU): a communicator that never touches a frame — only fs_gather_energy uses it
    ncclComm_t peers = nullptr;
    int peers_size = 0;
    float* d_gather = nullptr; size_t gather_cap = 0;   // [peers][B][bins] fp32
    // Pipelined frames (fs_set_pipelining).  depth 1: the connect pass of frame f is held back and launched together with
    // the walk of frame f + 1 as ONE kernel; depth 2: the walk is held back as well — call f launches {plan of f, walk of
    // f - 1, connect of f - 2} as one kernel.  Anything that needs a held frame's result lets it finish alone (flush_pending).
    struct PipeFrame {
        bool has = false;
        KParams kp; SubpathState st;
        WalkLaunch wl;               // queue_head = the frame's scratch set, rays_per_wave
        const uint32_t* perm = nullptr;   // its schedule (nullptr: none)
        bool walked = false;         // only the connect pass is owed
        bool fixed = false;
        int ppw = 64;
        struct Item {                // one per source of the frame (a batched frame has several)
            Source* s = nullptr;
            int cur = 0;             // which of the source's energy buffers the frame deposits into
            bool want_recon = false; // fs_reconstruct_impulse_response_async arrived while the frame was held
            fs_params recon;
        };
        std::vector<Item> items;
        float* const* energy_tab = nullptr;               // batched frame: the per-frame device tables (kBatchSlots of them
        unsigned long long* const* fixed_tab = nullptr;   // rotate; a held frame is connected two calls later at most)
    } held[2];                       // [0] the older frame, [1] the newer one (depth 2 only)
    std::shared_ptr<RefineJob> refine;   // fs_scene_commit_progressive: the background build whose tree replaces the device-built one
    bool moved_since_refine = false;     // fs_scene_update_triangles since the snapshot: re-apply the positions after the swap
    fs::HostBVH* prebuilt = nullptr;     // fs_scene_commit takes this tree instead of building one (install of a refined tree)
    int pipelining = 0;              // 0 off, 1 / 2 = frames held back
    unsigned frame_index = 0;        // consecutive traced frames rotate through the state / schedule / scratch sets
    size_t perm_words = 0;           // words of ONE schedule set (walk.perm holds kPermSets)
    bool comm_owned = false;           // created by fs_comm_init (destroyed with the context) vs attached by the caller

    // subpath state (sized on demand)
    SubpathState st{};
    size_t cap_lanes = 0, cap_seg = 0;
    float4* d_seg_pos = nullptr;   // node positions per walk step, all-connections mode only (row f3)
    size_t cap_pos = 0;
    // second record tier of depth = 0 frames (walk steps beyond FS_MAX_DEPTH): [kOverLevels][over_cap] each, grown when
    // a frame raises the overflow word; d_overflow = that word
    float2* d_over_np = nullptr; uint32_t* d_over_mat = nullptr; float4* d_over_pos = nullptr;
    uint32_t over_cap = 0, over_cap_pos = 0;
    unsigned* d_overflow = nullptr;
    bool overflow_armed = false;   // an unbounded frame has been enqueued since the word was last read
    // batched frames (fs_compute_energy_response_batch_async): per-frame tables of pointers and source positions
    static constexpr int kBatchSlots = 4;   // frames the host may run ahead of the table copies
    char* d_batch = nullptr; char* h_batch = nullptr; size_t batch_cap = 0;   // kBatchSlots blocks of batch_cap bytes
    hipEvent_t ev_batch[kBatchSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool batch_pending[kBatchSlots] = {false, false, false, false};
    unsigned batch_frame = 0;
    unsigned long long host_segments = 0;   // walk segments of frames without a plan pass (roulette off), since the last reset

    // walk kernel launch shape (tunable through FS_WALK_VARIANT / FS_WALK_BLOCKS_PER_CU / FS_REFILL_THRESHOLD)
    WalkLaunch walk{2, 256, nullptr, 1, nullptr};   // variant 2 = wave work sharing
    int hist_window = kHistWindow; // FS_HIST_WINDOW
    size_t lds_limit = 64 * 1024;  // dynamic LDS a workgroup may ask for on this device (hipDeviceAttributeMaxSharedMemoryPerBlock)
    int walk_rays_per_wave = 0;    // BDPT walk: subpaths per wave, 0 = by frame size (FS_WALK_RAYS_PER_WAVE; 64 = dense waves)
    int connect_pairs_per_wave = 0;   // connect kernel: pairs per wave, 0 = by frame size (FS_CONNECT_PAIRS_PER_WAVE; 64 = dense)
    int sound_rays_per_wave = 2;   // legacy tracer: rays per wave, the other lanes help (FS_SOUND_RAYS_PER_WAVE; 64 = no sharing; 2 vs 4: 0.259 vs 0.267 ms at 100 k triangles, 0.195 vs 0.212 at 5 k)

    // measurement
    int profiling = 0;   // 0 off, 1 = HIP events around the dominant (walk) kernel only, 2 = every kernel
    int profile_interval = 1;   // level 1: every n-th frame carries the events (fs_set_profiling_interval)
    unsigned profile_tick = 0;
    std::vector<TimedFrame> pending;
    std::vector<hipEvent_t> free_events;
    fs_stats stats{};

    int fail(int code, const std::string& m) {
        std::lock_guard<std::mutex> g(err_mu);
        err = m;
        return code;
    }
    int hip_fail(hipError_t e, const char* what) {
        std::lock_guard<std::mutex> g(err_mu);
        err = std::string(what) + ": " + hipGetErrorString(e);
        return FS_ERR_HIP;
    }
};

// pipelined frames: launch a held-back connect pass on its own (defined next to trace_sources)
static int flush_pending(fs_context* ctx);
// fs_scene_commit_progressive: swap the finished background tree in (defined next to the commit functions)
static int maybe_install_refined(fs_context* ctx);
static void cancel_refine(fs_context* ctx) { ctx->refine.reset(); ctx->moved_since_refine = false; }
static void wait_refine(const std::shared_ptr<RefineJob>& j) {
    std::unique_lock<std::mutex> l(j->mu);
    j->cv.wait(l, [&] { return j->done; });
}
#define FS_FLUSH(ctx)                       \
    do {                                    \
        int fr_ = flush_pending(ctx);       \
        if (fr_) return fr_;                \
    } while (0)

#define FS_HIP(ctx, call)                                        \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return (ctx)->hip_fail(e_, #call); \
    } while (0)

namespace {

Source* get_source(fs_context* ctx, fs_source h) {
    if (h < 0 || (size_t)h >= ctx->sources.size()) return nullptr;
    Source* s = ctx->sources[(size_t)h];
    return (s && s->alive) ? s : nullptr;
}

// Before the compute stream writes the current energy buffer: the tail-stream reconstruct that last read it
// must be done (two frames back in steady state, i.e. long finished).
hipError_t wait_energy_readers(fs_context* ctx, Source* s) {
    if (s->red_recorded[s->cur]) {   // the tail stream may still be summing this buffer over the ranks
        hipError_t e = hipStreamWaitEvent(ctx->stream, s->ev_red[s->cur], 0);
        if (e != hipSuccess) return e;
    }
    if (!s->rec_recorded[s->cur]) return hipSuccess;
    return hipStreamWaitEvent(ctx->stream, s->ev_rec[s->cur], 0);
}
// Hand the current energy buffer over to the tail stream: what the compute stream has enqueued so far
// completes before anything enqueued on the tail stream from now on.
hipError_t handoff_energy(fs_context* ctx, Source* s) {
    hipError_t e = hipEventRecord(s->ev_dep, ctx->stream);
    if (e != hipSuccess) return e;
    return hipStreamWaitEvent(ctx->copy_stream, s->ev_dep, 0);
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
        if (s->d_ir_bands) (void)hipFree(s->d_ir_bands);
        if (s->d_ir_mono) (void)hipFree(s->d_ir_mono);
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
    if (ctx->d_tris) (void)hipFree(ctx->d_tris);
    if (ctx->d_absorption) (void)hipFree(ctx->d_absorption);
    if (ctx->d_leaf_pos) (void)hipFree(ctx->d_leaf_pos);
    if (ctx->d_node_box) (void)hipFree(ctx->d_node_box);
    if (ctx->d_move) (void)hipFree(ctx->d_move);
    ctx->d_leaf_pos = nullptr; ctx->d_node_box = nullptr; ctx->d_move = nullptr;
    ctx->move_cap = 0;
    ctx->refit_pending = false;
    ctx->d_nodes = nullptr; ctx->d_tris = nullptr; ctx->d_absorption = nullptr;
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
    for (void* q : {(void*)ctx->d_over_np, (void*)ctx->d_over_mat, (void*)ctx->d_over_pos, (void*)ctx->d_overflow})
        if (q) (void)hipFree(q);
    ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr; ctx->d_over_pos = nullptr; ctx->d_overflow = nullptr;
    ctx->over_cap = ctx->over_cap_pos = 0;
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
    (void)hipEventCreate(&e);
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

// advance `front` over publishes whose D2H copy has completed (producer thread only)
void poll_published(Source* s) {
    uint64_t f = s->front.load(std::memory_order_relaxed);
    while (f < s->enqueued) {
        uint64_t next = f + 1;
        int slot = (int)(next % kIrRing);
        if (s->seq_of[slot] != next) break;
        if (hipEventQuery(s->ev[slot]) != hipSuccess) break;
        f = next;
    }
    s->front.store(f, std::memory_order_release);
}

// levels = walk steps with a record in the main tier (min(depth, FS_MAX_DEPTH)); unbounded: also the second tier
int ensure_state(fs_context* ctx, uint32_t n_local, int levels, bool unbounded, bool want_positions, bool want_normals) {
    size_t lanes = 2 * (size_t)n_local;
    size_t seg = (size_t)levels * lanes;
    const size_t want = want_positions ? seg * (want_normals ? 2 : 1) : 0;   // positions, then normals
    if (want > ctx->cap_pos) {
        if (ctx->d_seg_pos) (void)hipFree(ctx->d_seg_pos);
        ctx->d_seg_pos = nullptr; ctx->cap_pos = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_seg_pos, sizeof(float4) * std::max<size_t>(want, 1)));
        ctx->cap_pos = want;
    }
    if (lanes > ctx->cap_lanes) {
        if (ctx->st.end_pos) (void)hipFree(ctx->st.end_pos);
        if (ctx->st.end_misc) (void)hipFree(ctx->st.end_misc);
        if (ctx->st.slot_of) (void)hipFree(ctx->st.slot_of);
        if (ctx->walk.perm) (void)hipFree(ctx->walk.perm);
        ctx->st.end_pos = nullptr; ctx->st.end_misc = nullptr; ctx->st.slot_of = nullptr;
        ctx->walk.perm = nullptr;
        ctx->cap_lanes = 0;
        // two sets of everything a frame's walk hands to its connect pass: pipelined frames overlap walk f + 1 with connect f
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.end_pos, sizeof(float4) * 2 * lanes));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.end_misc, sizeof(uint2) * 2 * lanes));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.slot_of, sizeof(uint32_t) * 2 * lanes));
        ctx->cap_seg = 0;   // the bucket array is sized with the segment records below
        ctx->cap_lanes = lanes;
    }
    if (seg > ctx->cap_seg) {
        if (ctx->st.seg_np) (void)hipFree(ctx->st.seg_np);
        if (ctx->st.seg_mat) (void)hipFree(ctx->st.seg_mat);
        ctx->st.seg_np = nullptr; ctx->st.seg_mat = nullptr;
        ctx->cap_seg = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.seg_np, sizeof(float2) * 2 * seg));
        FS_HIP(ctx, hipMalloc((void**)&ctx->st.seg_mat, sizeof(uint32_t) * 2 * seg));
        if (ctx->walk.perm) (void)hipFree(ctx->walk.perm);
        ctx->walk.perm = nullptr;
        // [levels + 1][lanes] for every later frame shape that fits the two capacities without a reallocation:
        // levels' * lanes' <= cap_seg and lanes' <= cap_lanes  =>  (levels' + 1) * lanes' <= seg + cap_lanes
        ctx->perm_words = seg + ctx->cap_lanes;
        FS_HIP(ctx, hipMalloc((void**)&ctx->walk.perm, sizeof(uint32_t) * kPermSets * ctx->perm_words));
        ctx->cap_seg = seg;
    }
    if (!ctx->d_overflow) {
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_overflow, sizeof(unsigned)));
        FS_HIP(ctx, hipMemsetAsync(ctx->d_overflow, 0, sizeof(unsigned), ctx->stream));
    }
    if (unbounded) {
        // the schedule puts the longest walks first: slots below 0.9^64 * lanes (x4 for the spread, + 64) own a second tier
        uint32_t want_cap = std::max(ctx->over_cap, (uint32_t)std::min<size_t>(lanes, (size_t)(4.0 * 1.18e-3 * (double)lanes) + 64));
        if (const char* v = std::getenv("FS_OVER_CAP")) want_cap = std::max(ctx->over_cap, (uint32_t)std::max(1, std::atoi(v)));   // tests: force the regrow path
        const bool grow_main = want_cap > ctx->over_cap || !ctx->d_over_np;
        const bool grow_pos = want_positions && (want_cap > ctx->over_cap_pos || !ctx->d_over_pos);
        if (grow_main) {
            FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_over_np) (void)hipFree(ctx->d_over_np);
            if (ctx->d_over_mat) (void)hipFree(ctx->d_over_mat);
            ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr;
            FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_np, sizeof(float2) * (size_t)kOverLevels * want_cap));
            FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_mat, sizeof(uint32_t) * (size_t)kOverLevels * want_cap));
            ctx->over_cap = want_cap;
        }
        if (grow_pos) {
            FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_over_pos) (void)hipFree(ctx->d_over_pos);
            ctx->d_over_pos = nullptr;
            FS_HIP(ctx, hipMalloc((void**)&ctx->d_over_pos, sizeof(float4) * 2 * (size_t)kOverLevels * ctx->over_cap));   // positions | normals
            ctx->over_cap_pos = ctx->over_cap;
        }
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
int auto_rays_per_wave(unsigned long long lanes, int depth) {
    // round 2 (kShareMinIdle, pipelined frames; tools/pipelined_rpw_sweep.py, profiles/r02_pipelined_rpw_sweep.json): mid-size
    // frames do better at ~4096 waves — 65 536 subpaths 16 per wave, 131 072 subpaths 32 per wave (0.244 -> 0.233 ms at
    // depth 8, 0.355 -> 0.309 ms at depth 12; unpipelined 0.300 -> 0.267 ms); 262 144 and more stay dense
    const unsigned long long target_waves = depth > 16 ? 16384ull : (lanes >= 65536ull ? 4096ull : 2048ull);
    int rpw = 4;
    while (rpw < 64 && (unsigned long long)rpw * 2 * target_waves <= lanes) rpw *= 2;   // largest power of two <= lanes / target
    return rpw;
}

// Pairs per wave of the connect kernel, same idea: a small frame's visibility queries are shared by sparse waves.
// About 2048 waves (tools/connect_sparse_sweep.py): 8 192 pairs 0.065 -> 0.031 ms with 4 per wave, 32 768 pairs
// 0.067 -> 0.043 ms with 16, dense waves from 131 072 pairs on.
int auto_pairs_per_wave(unsigned long long pairs) {
    int ppw = 4;
    while (ppw < 64 && (unsigned long long)ppw * 2 * 2048ull <= pairs) ppw *= 2;
    return ppw;
}

int check_params(fs_context* ctx, const fs_params* p) {
    if (!p) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "params is NULL");
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    if (p->depth < 0 || p->depth > FS_MAX_DEPTH) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "depth out of range");
    if (p->num_rays & 1u) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "num_rays must be even (source + listener subpaths)");
    if (p->num_rays > (1u << 30)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "num_rays above 2^30 per frame (32-bit subpath indices)");
    if ((p->flags & FS_FLAG_MATERIAL_LOBES) && (p->flags & FS_FLAG_MIS_BALANCE))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "FS_FLAG_MATERIAL_LOBES and FS_FLAG_MIS_BALANCE cannot be combined");
    if ((p->flags & FS_FLAG_ACCUMULATE_ENERGY) && ctx->cfg.world_size > 1)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "FS_FLAG_ACCUMULATE_ENERGY (the reference's accumulation quirk) is single-GPU only: "
                                                  "a sharded frame is summed over the ranks, an accumulated one would be summed again");
    if (!(p->dist_divisor > 0.f) || !(p->sound_speed > 0.f))
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "dist_divisor and sound_speed must be positive");
    return FS_OK;
}

}  // namespace

// A context overlaps the tail of a frame with the next frame's tracing on two HIP streams.  The runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default); with other libraries' streams in the process (RCCL)
// the two can land on one queue and serialise (measured 0.64 vs 0.57 ms per frame).  Ask for 16 queues unless the
// process environment already says otherwise; it only takes effect if the HIP runtime has not initialised yet
// (INTEGRATION.md), which is why bench.py also sets it itself.
__attribute__((constructor)) static void fs_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", /*overwrite=*/0); }

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
    e = hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipStreamCreate(copy): ") + hipGetErrorString(e));
    e = hipStreamCreateWithFlags(&ctx->rev_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipStreamCreate(reverb): ") + hipGetErrorString(e));
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
    if (const char* v = std::getenv("FS_HIST_WINDOW")) ctx->hist_window = std::max(1, std::min(4096, std::atoi(v)));
    if (const char* v = std::getenv("FS_WALK_RAYS_PER_WAVE")) ctx->walk_rays_per_wave = std::max(0, std::min(64, std::atoi(v)));
    if (const char* v = std::getenv("FS_CONNECT_PAIRS_PER_WAVE")) ctx->connect_pairs_per_wave = std::max(0, std::min(64, std::atoi(v)));
    if (const char* v = std::getenv("FS_SOUND_RAYS_PER_WAVE")) ctx->sound_rays_per_wave = std::max(1, std::min(64, std::atoi(v)));
    e = hipMalloc((void**)&ctx->walk.queue_head, sizeof(unsigned) * kScratchSets * kScratchAllocWords);   // each set with its counters
    if (e == hipSuccess) e = hipMemset(ctx->walk.queue_head, 0, sizeof(unsigned) * kScratchSets * kScratchAllocWords);
    if (e != hipSuccess) return ctx->fail(FS_ERR_NO_DEVICE, std::string("hipMalloc(queue): ") + hipGetErrorString(e));
    ctx->device_ok = true;
    return FS_OK;
}

int fs_context_destroy(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->cfg.device);
        (void)flush_pending(ctx);
        if (ctx->refine) { wait_refine(ctx->refine); cancel_refine(ctx); }   // let the background build end before the process may
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

}  // extern "C" (reopened below)

// ---- multi-GPU: RCCL over xGMI behind the C ABI ---------------------------------------------------
// The pairs of a frame are sharded over world_size ranks (one process per GPU); what the ranks exchange is
//   * one sum all-reduce of the [B][bins] energy histogram per source and frame (fp32, or the u64 fixed-point
//     histogram of FS_FLAG_DETERMINISTIC), on the tail stream, concurrent with the next frame's tracing;
//   * one broadcast of the acceleration structure at fs_scene_commit (rank 0 builds it, the others receive nodes,
//     triangle records and the refit tables).
// librccl is opened at run time — an already loaded copy first (a host process that uses torch.distributed has its
// own), then FS_RCCL_LIB, then the system's — so single-GPU users need no RCCL at all.
namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;   // why loading failed
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = std::getenv("FS_RCCL_LIB");
        const char* names[] = {"librccl.so.1", "librccl.so"};
        if (env && *env) api.handle = dlopen(env, RTLD_NOW | RTLD_GLOBAL);       // an explicit choice wins
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy the process already has
        for (const char* n : names)
            if (!api.handle) api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.handle) { const char* e = dlerror(); api.why = std::string("librccl not found: ") + (e ? e : "?"); return; }
        auto sym = [&](const char* n) { void* p = dlsym(api.handle, n); if (!p && api.why.empty()) api.why = std::string("librccl lacks ") + n; return p; };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
        api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if (!api.why.empty()) api.handle = nullptr;
    });
    return api.handle ? &api : nullptr;
}

int nccl_fail(fs_context* ctx, ncclResult_t r, const char* what) {
    RcclApi* a = rccl();
    return ctx->fail(FS_ERR_COMM, std::string(what) + ": " + (a && a->GetErrorString ? a->GetErrorString(r) : "RCCL error"));
}
#define FS_NCCL(ctx, call)                                          \
    do {                                                            \
        ncclResult_t r_ = (call);                                   \
        if (r_ != ncclSuccess) return nccl_fail((ctx), r_, #call);  \
    } while (0)

// Sum the source's current energy buffer over the ranks, on the tail stream, behind everything the compute stream
// has enqueued so far.  No-op without a communicator or when the frame has been summed already.
int reduce_energy(fs_context* ctx, Source* s) {
    if (!ctx->comm || s->reduced) return FS_OK;
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
    FS_HIP(ctx, handoff_energy(ctx, s));
    const size_t words = (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    if (s->cur_fixed) {   // deterministic mode: integer sum of the fixed-point histogram, rounded to fp32 once, behind it
        FS_NCCL(ctx, a->AllReduce(s->d_fixed[s->cur], s->d_fixed[s->cur], words, ncclUint64, ncclSum, ctx->comm, ctx->copy_stream));
        launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), (int)words, ctx->copy_stream);
    } else
        FS_NCCL(ctx, a->AllReduce(s->energy(), s->energy(), words, ncclFloat32, ncclSum, ctx->comm, ctx->copy_stream));
    FS_HIP(ctx, hipEventRecord(s->ev_red[s->cur], ctx->copy_stream));
    s->red_recorded[s->cur] = true;
    s->reduced = true;
    return FS_OK;
}

}  // namespace

extern "C" {

int fs_shard_range(uint32_t num_rays, int32_t rank, int32_t world_size, uint32_t* pair_begin, uint32_t* pair_count) {
    if (world_size < 1 || rank < 0 || rank >= world_size || (num_rays & 1u)) return FS_ERR_INVALID_ARGUMENT;
    const uint64_t P = num_rays / 2, W = (uint64_t)world_size, R = (uint64_t)rank;
    const uint64_t p0 = P * R / W, p1 = P * (R + 1) / W;
    if (pair_begin) *pair_begin = (uint32_t)p0;
    if (pair_count) *pair_count = (uint32_t)(p1 - p0);
    return FS_OK;
}

int fs_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes != FS_COMM_ID_BYTES) return FS_ERR_INVALID_ARGUMENT;
    RcclApi* a = rccl();
    if (!a) return FS_ERR_COMM;
    ncclUniqueId id;
    if (a->GetUniqueId(&id) != ncclSuccess) return FS_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == FS_COMM_ID_BYTES, "FS_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    std::memcpy(id_out, &id, sizeof(id));
    return FS_OK;
}

int fs_comm_init(fs_context* ctx, const void* unique_id, size_t bytes) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!unique_id || bytes != FS_COMM_ID_BYTES) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "unique id must be FS_COMM_ID_BYTES bytes");
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (ctx->comm) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "a communicator is already attached");
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, rccl() ? "" : "librccl is not loadable (set FS_RCCL_LIB)");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    FS_NCCL(ctx, a->CommInitRank(&ctx->comm, ctx->cfg.world_size, id, ctx->cfg.rank));
    ctx->comm_owned = true;
    return FS_OK;
}

int fs_comm_attach(fs_context* ctx, void* nccl_comm) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!nccl_comm) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "null communicator");
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (ctx->comm) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "a communicator is already attached");
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "librccl is not loadable (set FS_RCCL_LIB)");
    int n = 0, r = -1;
    FS_NCCL(ctx, a->CommCount((ncclComm_t)nccl_comm, &n));
    FS_NCCL(ctx, a->CommUserRank((ncclComm_t)nccl_comm, &r));
    if (n != ctx->cfg.world_size || r != ctx->cfg.rank)
        return ctx->fail(FS_ERR_INVALID_ARGUMENT, "communicator size / rank differ from fs_config.world_size / rank");
    ctx->comm = (ncclComm_t)nccl_comm;
    ctx->comm_owned = false;
    return FS_OK;
}

int fs_comm_detach(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    if (!ctx->comm) return FS_OK;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->cfg.device);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->copy_stream);
    }
    RcclApi* a = rccl();
    if (ctx->comm_owned && a) (void)a->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_owned = false;
    return FS_OK;
}

int fs_peers_init(fs_context* ctx, const void* unique_id, size_t bytes, int32_t rank, int32_t world_size) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!unique_id || bytes != FS_COMM_ID_BYTES) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "unique id must be FS_COMM_ID_BYTES bytes");
    if (world_size < 1 || rank < 0 || rank >= world_size) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "bad rank / world_size");
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (ctx->peers) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "a peer communicator is already attached");
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "librccl is not loadable (set FS_RCCL_LIB)");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    FS_NCCL(ctx, a->CommInitRank(&ctx->peers, world_size, id, rank));
    ctx->peers_size = world_size;
    return FS_OK;
}

int fs_peers_detach(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->peers) return FS_OK;
    if (ctx->device_ok) {
        (void)hipSetDevice(ctx->cfg.device);
        (void)hipStreamSynchronize(ctx->copy_stream);
    }
    if (RcclApi* a = rccl()) (void)a->CommDestroy(ctx->peers);
    ctx->peers = nullptr;
    ctx->peers_size = 0;
    return FS_OK;
}

int fs_gather_energy_async(fs_context* ctx, fs_source h, void** dptr, size_t* bytes) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (!ctx->peers) return ctx->fail(FS_ERR_COMM, "fs_gather_energy needs fs_peers_init first");
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "peer communicator attached but librccl is not loadable");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    const size_t words = (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    const size_t need = words * (size_t)ctx->peers_size;
    if (need > ctx->gather_cap) {
        FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
        if (ctx->d_gather) (void)hipFree(ctx->d_gather);
        ctx->d_gather = nullptr; ctx->gather_cap = 0;
        FS_HIP(ctx, hipMalloc((void**)&ctx->d_gather, sizeof(float) * need));
        ctx->gather_cap = need;
    }
    // behind this frame's deposit (and its all-reduce, on a sharded context), in tail-stream order
    if (ctx->comm) { int rr = reduce_energy(ctx, s); if (rr) return rr; }
    else FS_HIP(ctx, handoff_energy(ctx, s));
    FS_NCCL(ctx, a->AllGather(s->energy(), ctx->d_gather, words, ncclFloat32, ctx->peers, ctx->copy_stream));
    // the buffer is being read on the tail stream: the frame after next must not deposit into it before that
    FS_HIP(ctx, hipEventRecord(s->ev_red[s->cur], ctx->copy_stream));
    s->red_recorded[s->cur] = true;
    if (dptr) *dptr = ctx->d_gather;
    if (bytes) *bytes = sizeof(float) * need;
    return FS_OK;
}

int fs_gather_energy(fs_context* ctx, fs_source h, float* out, int32_t n) {
    if (!ctx || !out) return FS_ERR_INVALID_ARGUMENT;
    if (ctx->peers && n != ctx->peers_size * ctx->cfg.num_bands * ctx->num_bins)
        return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != peers * bands * bins");
    void* d = nullptr; size_t b = 0;
    const int rc = fs_gather_energy_async(ctx, h, &d, &b);
    if (rc) return rc;
    FS_HIP(ctx, hipMemcpyAsync(out, d, b, hipMemcpyDeviceToHost, ctx->copy_stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    return FS_OK;
}

}  // extern "C"

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

// what every kind of commit ends with: the kernels' view of the scene and the stats
static void finish_commit(fs_context* ctx, size_t scene_bytes) {
    ctx->amax = 0.f;
    for (float v : ctx->h_xyz) ctx->amax = std::max(ctx->amax, std::fabs(v));
    ctx->scene.nodes = ctx->d_nodes;
    ctx->scene.tris = ctx->d_tris;
    ctx->scene.absorption = ctx->d_absorption;
    ctx->scene.lobe_gain = ctx->d_absorption ? ctx->d_absorption + (size_t)ctx->M * ctx->cfg.num_bands : nullptr;
    ctx->scene.lobe_prob = ctx->d_absorption ? ctx->scene.lobe_gain + (size_t)ctx->M * 3 * ctx->cfg.num_bands : nullptr;
    ctx->scene.num_nodes = (int32_t)ctx->bvh.nodes.size();
    ctx->scene.num_tris = ctx->T;
    ctx->scene.num_materials = ctx->M;
    ctx->scene.stack_rows = std::max(ctx->bvh.stack_need, 2) + kStackSlack;
#ifdef FS_EXPERIMENTS   // occupancy experiments only: fewer rows than the tree's worst case (an overflowing lane corrupts the share area)
    if (const char* v = std::getenv("FS_UNSAFE_STACK_ROWS")) ctx->scene.stack_rows = std::max(4, std::atoi(v));
#endif
    ctx->stats.bvh_nodes = (uint32_t)ctx->bvh.nodes.size();
    ctx->stats.triangles = (uint32_t)ctx->T;
    ctx->stats.bvh_stack_need = (uint32_t)ctx->bvh.stack_need;
    ctx->stats.bvh_depth = (uint32_t)ctx->bvh.max_depth;
    ctx->stats.scene_bytes = scene_bytes;
    ctx->committed = true;
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
        const size_t need = traversal_lds_bytes(std::max(ctx->bvh.stack_need, 2) + kStackSlack, ctx->cfg.num_bands, ctx->num_bins);
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
    finish_commit(ctx, nb + tb + mb);
    return FS_OK;
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
        traversal_lds_bytes(std::max(info.stack_need, 2) + kStackSlack, ctx->cfg.num_bands, ctx->num_bins) > ctx->lds_limit)
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
    finish_commit(ctx, (size_t)info.num_nodes * sizeof(NodeQ4) + tb + ctx->h_absorption.size() * sizeof(float));
    return FS_OK;
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
    std::thread([job] {
        build_bvh(job->xyz.data(), job->mat.data(), job->obj.empty() ? nullptr : job->obj.data(), job->T, job->bvh);
        { std::lock_guard<std::mutex> l(job->mu); job->done = true; }
        job->cv.notify_all();
    }).detach();
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
    launch_update_triangles(ctx->d_tris, ctx->d_leaf_pos, first, count, ctx->d_move, ctx->stream);
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
    return FS_OK;
}

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
    for (int i = 0; i < kEnergyBufs; ++i) {
        if ((e = hipMalloc((void**)&s->d_energy[i], eb)) != hipSuccess) return bail(e, "hipMalloc(energy)");
        if ((e = hipMemsetAsync(s->d_energy[i], 0, eb, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
        if ((e = hipEventCreateWithFlags(&s->ev_rec[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = hipMalloc((void**)&s->d_ir_bands, ib * (size_t)ctx->cfg.num_bands)) != hipSuccess) return bail(e, "hipMalloc(ir_bands)");
    if ((e = hipMalloc((void**)&s->d_ir_mono, ib)) != hipSuccess) return bail(e, "hipMalloc(ir_mono)");
    if ((e = hipMemsetAsync(s->d_ir_bands, 0, ib * (size_t)ctx->cfg.num_bands, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    if ((e = hipMemsetAsync(s->d_ir_mono, 0, ib, ctx->stream)) != hipSuccess) return bail(e, "hipMemsetAsync");
    for (int i = 0; i < kIrRing; ++i) {
        if ((e = hipHostMalloc((void**)&s->h_ir[i], ib, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
        std::memset(s->h_ir[i], 0, ib);  // ImpulseBuffer[ch].Init(0, NumSamples) FSAC.cpp:24-28
        if ((e = hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    if ((e = hipEventCreateWithFlags(&s->ev_dep, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (int i = 0; i < kEnergyBufs; ++i)
        if ((e = hipEventCreateWithFlags(&s->ev_red[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&s->ev_rev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    // the initial fills above ran on the compute stream; the first reconstruct runs on the tail stream
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return bail(e, "hipStreamSynchronize");
    s->alive = true;
    // RegisterSource: ActiveSources.Add (ARTS.cpp:45-48); reuse a dead slot if any
    for (size_t i = 0; i < ctx->sources.size(); ++i)
        if (!ctx->sources[i]) { ctx->sources[i] = s; *out = (fs_source)i; return FS_OK; }
    ctx->sources.push_back(s);
    *out = (fs_source)(ctx->sources.size() - 1);
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

int fs_listener_set_position(fs_context* ctx, const float xyz[3]) {
    if (!ctx || !xyz) return FS_ERR_INVALID_ARGUMENT;
    std::memcpy(ctx->listener, xyz, sizeof(float) * 3);
    return FS_OK;
}

// ---- hot path ------------------------------------------------------------------------------------------
// One traced frame for `count` sources (count == 1: the plain call).  A batch lays the sources' pairs end to end in one
// plan / walk / connect sequence: every source gets exactly the pairs, random streams and therefore results of its own
// fs_compute_energy_response_async call, but the chip sees one large frame instead of `count` small ones.
// depth = 0 frames: did a walk's record miss both tiers?  (Called where the compute stream has just been synchronised.)
// Then the frame's energy is incomplete: the tier is grown for the next attempt and the caller is told.
static int check_overflow(fs_context* ctx) {
    if (!ctx->overflow_armed || !ctx->d_overflow) return FS_OK;
    ctx->overflow_armed = false;
    unsigned flag = 0;
    FS_HIP(ctx, hipMemcpy(&flag, ctx->d_overflow, sizeof(flag), hipMemcpyDeviceToHost));
    if (!flag) return FS_OK;
    FS_HIP(ctx, hipMemset(ctx->d_overflow, 0, sizeof(flag)));
    const uint32_t grown = std::max<uint32_t>(ctx->over_cap, 16) * 4;
    if (ctx->d_over_np) (void)hipFree(ctx->d_over_np);
    if (ctx->d_over_mat) (void)hipFree(ctx->d_over_mat);
    if (ctx->d_over_pos) (void)hipFree(ctx->d_over_pos);
    ctx->d_over_np = nullptr; ctx->d_over_mat = nullptr; ctx->d_over_pos = nullptr; ctx->over_cap_pos = 0;
    ctx->over_cap = grown;   // ensure_state allocates at this size next time
    return ctx->fail(FS_ERR_OVERFLOW, "depth = 0: more walks than expected outlived " + std::to_string(FS_MAX_DEPTH) +
                     " steps; the record tier has been grown — trace the frame again");
}

static int reconstruct_now(fs_context* ctx, Source* s, const fs_params* p);

// fs_scene_commit_progressive: once the background build has finished, the next call that traces anything swaps its tree
// in — held frames finish first (they were traced through the old tree's arrays), the stream drains, the records are
// uploaded; triangles moved since the snapshot get their current positions and a refit.
static int maybe_install_refined(fs_context* ctx) {
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

// ---- pipelined frames ---------------------------------------------------------------------------------------------
// What a held frame still owes once its connect pass has been enqueued: the fixed-point -> fp32 rounding, the sum over
// the ranks, and the reconstruct the caller asked for in the meantime.  The source may already have moved on to later
// frames (cur rotated): the per-frame fields are switched back for the duration.
static int finish_held_frame(fs_context* ctx, const fs_context::PipeFrame& q) {
    for (const fs_context::PipeFrame::Item& it : q.items) {
        Source* s = it.s;
        const bool moved_on = s->cur != it.cur;
        const int cur = s->cur;
        const bool cur_fixed = s->cur_fixed, reduced = s->reduced, handed_off = s->handed_off;
        s->cur = it.cur; s->cur_fixed = q.fixed; s->reduced = false; s->handed_off = false;
        int rc = FS_OK;
        if (q.fixed) launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), ctx->cfg.num_bands * ctx->num_bins, ctx->stream);
        if (ctx->comm) rc = reduce_energy(ctx, s);
        if (!rc && it.want_recon) rc = reconstruct_now(ctx, s, &it.recon);
        if (moved_on) { s->cur = cur; s->cur_fixed = cur_fixed; s->reduced = reduced; s->handed_off = handed_off; }
        if (rc) return rc;
    }
    return FS_OK;
}

static void held_connect_part(const fs_context::PipeFrame& q, FrameParts& f) {
    const fs_context::PipeFrame::Item& it = q.items[0];
    f.has_connect = true; f.kpc = q.kp; f.stc = q.st; f.energy = it.s->d_energy[it.cur];
    f.fixed = q.fixed ? it.s->d_fixed[it.cur] : nullptr; f.scratch_c = q.wl.queue_head; f.ppw = q.ppw;
    f.energy_tab = q.energy_tab; f.fixed_tab = q.fixed_tab;
}
static void held_walk_part(const fs_context::PipeFrame& q, FrameParts& f) {
    f.has_walk = true; f.kpw = q.kp; f.stw = q.st; f.wl = q.wl; f.perm = q.perm;
}

// Let every held frame finish on its own kernels, oldest first: something needs their results (or their buffers) now.
static int flush_pending(fs_context* ctx) {
    if (!ctx->held[0].has && !ctx->held[1].has) return FS_OK;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    for (int k = 0; k < 2; ++k) {
        if (!ctx->held[k].has) continue;
        const fs_context::PipeFrame q = ctx->held[k];
        ctx->held[k].has = false;
        if (!q.walked) launch_walk(ctx->scene, q.kp, q.st, q.wl, q.perm, ctx->stream);
        launch_connect(ctx->cfg.num_bands, ctx->scene, q.kp, q.st, q.items[0].s->d_energy[q.items[0].cur],
                       q.fixed ? q.items[0].s->d_fixed[q.items[0].cur] : nullptr, q.wl.queue_head, q.ppw, q.energy_tab, q.fixed_tab,
                       ctx->stream);
        FS_HIP(ctx, hipGetLastError());
        const int rc = finish_held_frame(ctx, q);
        if (rc) return rc;
    }
    return FS_OK;
}

static int trace_sources(fs_context* ctx, Source* const* srcs, int count, const fs_params* p) {
    Source* s = srcs[0];
    const bool batch = count > 1;
    { int ir = maybe_install_refined(ctx); if (ir) return ir; }                     // fs_scene_commit_progressive: the better tree is ready
    if (ctx->refit_pending) { int rr = fs_scene_refit(ctx); if (rr) return rr; }   // moved triangles: refit before tracing
    int rc = check_params(ctx, p);
    if (rc) return rc;
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    // Pipelined frames: this frame's connect pass is held back (to be launched with the next frame's walk) when the
    // frame has the default shape; any other frame first lets the held-back one finish on its own.
    const bool pipe_ok = ctx->pipelining > 0 && ctx->profiling < 2 && p->depth > 0 &&
                         !(p->flags & (FS_FLAG_MATERIAL_LOBES | FS_FLAG_MIS_BALANCE | FS_FLAG_ALL_CONNECTIONS | FS_FLAG_ACCUMULATE_ENERGY));
    if (!pipe_ok) FS_FLUSH(ctx);

    const int B = ctx->cfg.num_bands;
    const uint64_t P = p->num_rays / 2;
    uint32_t p0 = 0, pn = 0;
    (void)fs_shard_range(p->num_rays, ctx->cfg.rank, ctx->cfg.world_size, &p0, &pn);   // validated by check_params / create
    KParams kp{};
    kp.seed_lo = (uint32_t)p->seed;
    kp.seed_hi = (uint32_t)(p->seed >> 32);
    kp.pair_begin = p0;
    kp.pairs_per_source = pn;
    kp.num_local = kp.pairs_per_source * (uint32_t)count;
    kp.src_table = nullptr;
    // depth = 0: no cap, like the reference's while (true) (ARTS.cpp:294) — the roulette ends every walk; the records of
    // steps beyond FS_MAX_DEPTH go to the second tier.  Without roulette an uncapped walk would never end: FS_MAX_DEPTH.
    const bool unbounded = p->depth == 0 && p->russian_roulette && p->rr_prob < 1.0f;
    const int levels = p->depth > 0 ? p->depth : FS_MAX_DEPTH;
    kp.depth = unbounded ? FS_MAX_DEPTH + kOverLevels : levels;
    kp.mis_depth = unbounded ? kUnboundedDepth : levels;
    kp.russian_roulette = p->russian_roulette;
    kp.cosine = (p->flags & FS_FLAG_COSINE_SAMPLING) ? 1 : 0;
    kp.rr_prob = p->rr_prob;
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
    std::memcpy(kp.src, s->pos, sizeof(kp.src));
    std::memcpy(kp.lis, ctx->listener, sizeof(kp.lis));
    kp.count = ctx->profiling >= 3 ? 1 : 0;
    kp.num_bins = ctx->num_bins;
    kp.hist_window = std::min(ctx->num_bins, ctx->hist_window);

    kp.lobes = (p->flags & FS_FLAG_MATERIAL_LOBES) ? 1 : 0;
    const bool mis = (p->flags & FS_FLAG_MIS_BALANCE) != 0;
    const bool all_conn = mis || (p->flags & FS_FLAG_ALL_CONNECTIONS) != 0;
    kp.mis = mis ? 1 : 0;
    if (2 * (size_t)kp.num_local > ctx->cap_lanes || (size_t)levels * 2 * (size_t)kp.num_local > ctx->cap_seg)
        FS_FLUSH(ctx);   // the state arrays are about to be reallocated: held frames still read them
    rc = ensure_state(ctx, kp.num_local, levels, unbounded, all_conn, mis);
    if (rc) return rc;
    SubpathState st = ctx->st;
    const unsigned fidx = ctx->frame_index++;   // consecutive frames rotate through the state / schedule / scratch sets
    if (fidx & 1u) {
        st.end_pos += ctx->cap_lanes; st.end_misc += ctx->cap_lanes; st.slot_of += ctx->cap_lanes;
        st.seg_np += ctx->cap_seg; st.seg_mat += ctx->cap_seg;
    }
    unsigned* const scratch = ctx->walk.queue_head + (size_t)(fidx % kScratchSets) * kScratchAllocWords;
    uint32_t* const perm_buf = ctx->walk.perm ? ctx->walk.perm + (size_t)(fidx % kPermSets) * ctx->perm_words : nullptr;
    st.seg_pos = all_conn ? ctx->d_seg_pos : nullptr;
    st.seg_nrm = mis ? ctx->d_seg_pos + (size_t)levels * 2 * (size_t)kp.num_local : nullptr;
    st.main_levels = levels;
    st.over_levels = unbounded ? kOverLevels : 0;
    st.over_cap = unbounded ? ctx->over_cap : 0;
    st.over_np = ctx->d_over_np; st.over_mat = ctx->d_over_mat;
    st.over_pos = all_conn ? ctx->d_over_pos : nullptr;
    st.over_nrm = mis && ctx->d_over_pos ? ctx->d_over_pos + (size_t)kOverLevels * ctx->over_cap : nullptr;
    st.overflow = ctx->d_overflow;
    if (unbounded) ctx->overflow_armed = true;

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
    const bool fixed = (p->flags & FS_FLAG_DETERMINISTIC) != 0;
    const bool accumulate = (p->flags & FS_FLAG_ACCUMULATE_ENERGY) != 0;
    for (int i = 0; i < count; ++i) {
        Source* si = srcs[i];
        if (fixed && !si->d_fixed[0]) {
            for (int k = 0; k < kEnergyBufs; ++k) {
                FS_HIP(ctx, hipMalloc((void**)&si->d_fixed[k], sizeof(unsigned long long) * (size_t)B * (size_t)ctx->num_bins));
                FS_HIP(ctx, hipMemsetAsync(si->d_fixed[k], 0, sizeof(unsigned long long) * (size_t)B * (size_t)ctx->num_bins, ctx->stream));
            }
        }
        // this frame deposits into the other buffer; the tail may still be busy with the last one.  (FS_FLAG_ACCUMULATE_ENERGY
        // stays in the buffer of the previous frame — behind its reduce / reconstruct — and adds to what it holds.)
        if (!accumulate) si->cur = (si->cur + 1) % kEnergyBufs;
        si->cur_fixed = fixed;
        si->reduced = false; si->handed_off = false;
        FS_HIP(ctx, wait_energy_readers(ctx, si));
    }
    // FlushEnergyBuffer ARTS.cpp:157-161 is folded into the plan pass (one launch); plain memset otherwise.
    // Deterministic mode zeroes the fixed-point histogram instead (the fp32 buffer is rewritten from it).
    float* zero_ptr = accumulate ? nullptr : (fixed ? reinterpret_cast<float*>(s->d_fixed[s->cur]) : s->energy());
    const int zero_words = (fixed ? 2 : 1) * B * ctx->num_bins;
    float* const* energy_tab = nullptr;
    float* const* zero_tab = nullptr;   // batched frame: the table of buffers the plan pass zeroes
    unsigned long long* const* fixed_tab = nullptr;
    if (batch) {
        // per-frame tables in one pinned staging block: energy pointers [count] | fixed-point buffer pointers [count] |
        // source positions [count][3].  The block is rewritten only after the previous frame's copy has left it.
        const size_t bytes = 2 * (size_t)count * sizeof(void*) + (size_t)count * 3 * sizeof(float);
        if (bytes > ctx->batch_cap) {
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
            }
        }
        const int slot = (int)(ctx->batch_frame++ % fs_context::kBatchSlots);
        if (ctx->batch_pending[slot]) FS_HIP(ctx, hipEventSynchronize(ctx->ev_batch[slot]));   // its last copy has left the block
        char* hb = ctx->h_batch + (size_t)slot * ctx->batch_cap;
        char* db = ctx->d_batch + (size_t)slot * ctx->batch_cap;
        void** h_en = reinterpret_cast<void**>(hb);
        void** h_fx = h_en + count;
        float* h_pos = reinterpret_cast<float*>(h_fx + count);
        for (int i = 0; i < count; ++i) {
            h_en[i] = srcs[i]->energy();
            h_fx[i] = fixed ? (void*)srcs[i]->d_fixed[srcs[i]->cur] : nullptr;
            std::memcpy(h_pos + 3 * i, srcs[i]->pos, sizeof(float) * 3);
        }
        FS_HIP(ctx, hipMemcpyAsync(db, hb, bytes, hipMemcpyHostToDevice, ctx->stream));
        FS_HIP(ctx, hipEventRecord(ctx->ev_batch[slot], ctx->stream));
        ctx->batch_pending[slot] = true;
        energy_tab = reinterpret_cast<float* const*>(db);
        fixed_tab = fixed ? reinterpret_cast<unsigned long long* const*>(db + (size_t)count * sizeof(void*)) : nullptr;
        kp.src_table = reinterpret_cast<const float*>(db + 2 * (size_t)count * sizeof(void*));
        zero_ptr = nullptr;
        if (!accumulate) zero_tab = fixed ? reinterpret_cast<float* const*>(fixed_tab) : energy_tab;
    }
    WalkLaunch wplan = ctx->walk;
    wplan.queue_head = scratch;
    wplan.perm = perm_buf;
    if (unbounded) wplan.plan = 1;   // the second record tier relies on the schedule: the longest walks own the lowest slots
    bool sort = false;
    const bool plan_runs = plan_shape(kp, wplan, nullptr, &sort);   // the plan pass (and the flush with it) runs for this frame
    const uint32_t* perm = plan_runs && sort ? perm_buf : nullptr;
    if (!perm) st.slot_of = nullptr;   // no schedule: slot == subpath index
    const bool plan_held = pipe_ok && ctx->pipelining >= 2 && plan_runs;   // depth 2: the pass joins the fused launch below
    if (plan_runs && !plan_held)
        (void)launch_plan(kp, wplan, zero_ptr, (zero_ptr || zero_tab) ? zero_words : 0, zero_tab, count, ctx->stream);
    if (!plan_runs) {
        if (zero_ptr) FS_HIP(ctx, hipMemsetAsync(zero_ptr, 0, sizeof(float) * (size_t)zero_words, ctx->stream));
        for (int i = 0; i < count && zero_tab; ++i) {
            float* zp = fixed ? reinterpret_cast<float*>(srcs[i]->d_fixed[srcs[i]->cur]) : srcs[i]->energy();
            FS_HIP(ctx, hipMemsetAsync(zp, 0, sizeof(float) * (size_t)zero_words, ctx->stream));
        }
    }
    if (!kp.russian_roulette) ctx->host_segments += 2ull * kp.num_local * (unsigned long long)kp.depth;   // no plan pass to count them
    if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[0], ctx->stream));
    WalkLaunch wl = ctx->walk;
    wl.queue_head = scratch;
    wl.rays_per_wave = ctx->walk_rays_per_wave > 0 ? ctx->walk_rays_per_wave : auto_rays_per_wave(2ull * kp.num_local, kp.depth);
    const int ppw = ctx->connect_pairs_per_wave > 0 ? ctx->connect_pairs_per_wave : auto_pairs_per_wave(kp.num_local);
    if (pipe_ok) {   // (anything else has flushed the held frames above)
        fs_context::PipeFrame me;
        me.has = true; me.kp = kp; me.st = st; me.wl = wl; me.perm = perm; me.walked = false; me.fixed = fixed; me.ppw = ppw;
        me.items.resize((size_t)count);
        for (int i = 0; i < count; ++i) { me.items[(size_t)i].s = srcs[i]; me.items[(size_t)i].cur = srcs[i]->cur; }
        me.energy_tab = energy_tab; me.fixed_tab = fixed_tab;
        FrameParts f;
        f.wl = wl;
        const bool deep = ctx->pipelining >= 2;
        // the held frames (copies: the slots are rewritten below): one that only owes its connect pass, one that was only planned
        fs_context::PipeFrame to_connect, to_walk;
        if (ctx->held[0].has && ctx->held[0].walked) to_connect = ctx->held[0];
        if (ctx->held[1].has) to_walk = ctx->held[1];
        else if (ctx->held[0].has && !ctx->held[0].walked) to_walk = ctx->held[0];
        if (deep) {   // {plan of this frame, walk of the planned frame, connect of the walked one}
            if (plan_held) {
                f.has_plan = true; f.kpp = kp; f.scratch_p = scratch; f.perm_p = sort ? perm_buf : nullptr;
                f.zero_p = zero_ptr; f.zero_words_p = (zero_ptr || zero_tab) ? zero_words : 0; f.zero_tab_p = zero_tab; f.zero_count_p = count;
            }
            if (to_walk.has) held_walk_part(to_walk, f);
        } else {      // {walk of this frame, connect of the walked one}
            held_walk_part(me, f);
        }
        if (to_connect.has) held_connect_part(to_connect, f);
        if (timed_frame && !(f.has_walk && (f.has_connect || !deep))) {
            // a pipeline-fill launch (no walk, or depth 2 without its connect part yet) is not a sample of the frame kernel:
            // take the first event back out of the stream's timing (it was recorded above; both go back to the pool unused)
            for (int i = 0; i < 3; ++i) if (tf.e[i]) { ctx->free_events.push_back(tf.e[i]); tf.e[i] = nullptr; }
            timed_frame = false;
        }
        if (f.has_walk || f.has_connect || f.has_plan) {
            if (!launch_frame(B, ctx->scene, f, ctx->stream)) {   // no fused form: the same passes one after the other
                if (f.has_connect) launch_connect(B, ctx->scene, f.kpc, f.stc, f.energy, f.fixed, f.scratch_c, f.ppw, f.energy_tab, f.fixed_tab, ctx->stream);
                if (f.has_walk) launch_walk(ctx->scene, f.kpw, f.stw, f.wl, f.perm, ctx->stream);
                if (f.has_plan) (void)launch_plan(kp, wplan, zero_ptr, (zero_ptr || zero_tab) ? zero_words : 0, zero_tab, count, ctx->stream);
            }
        }
        if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[1], ctx->stream));
        FS_HIP(ctx, hipGetLastError());
        ctx->held[0].has = false; ctx->held[1].has = false;
        if (to_connect.has) { rc = finish_held_frame(ctx, to_connect); if (rc) return rc; }
        if (deep) {   // the planned frame has been walked now; this one has only been planned
            if (to_walk.has) { ctx->held[0] = to_walk; ctx->held[0].walked = true; ctx->held[1] = me; }
            else ctx->held[0] = me;
        } else {
            me.walked = true;
            ctx->held[0] = me;
        }
        if (timed_frame) { tf.has_trace = true; ctx->pending.push_back(tf); }
        ctx->stats.frames++;
        ctx->stats.pairs += kp.num_local;
        ctx->stats.rays += 2ull * kp.num_local;
        return FS_OK;
    }
    launch_walk(ctx->scene, kp, st, wl, perm, ctx->stream);
    if (timed_frame) FS_HIP(ctx, hipEventRecord(tf.e[1], ctx->stream));
    if (all_conn)
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
        for (int i = 0; i < count; ++i) { int rr = reduce_energy(ctx, srcs[i]); if (rr) return rr; }
    return FS_OK;
}

int fs_compute_energy_response_async(fs_context* ctx, fs_source h, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (!ctx->committed) return ctx->fail(FS_ERR_NOT_COMMITTED, "scene not committed");
    return trace_sources(ctx, &s, 1, p);
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

int fs_reconstruct_impulse_response_async(fs_context* ctx, fs_source h, const fs_params* p) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    fs_params def;
    if (!p) { fs_params_default(&def); p = &def; }
    if (p->struct_size != sizeof(fs_params)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_params.struct_size mismatch");
    // pipelined frames: the source's current frame still waits for its connect pass — the reconstruct goes with it
    for (int k = 1; k >= 0; --k) {   // the source's CURRENT frame is the newest held one
        fs_context::PipeFrame& q = ctx->held[k];
        if (!q.has) continue;
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
    return reconstruct_now(ctx, s, p);
}

}  // extern "C"

static int reconstruct_now(fs_context* ctx, Source* s, const fs_params* p) {
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

    poll_published(s);
    // never overwrite the front buffer: at most two publishes in flight
    if (s->enqueued >= 2) {
        int slot = (int)((s->enqueued - 1) % kIrRing);
        if (s->seq_of[slot] == s->enqueued - 1 && s->front.load(std::memory_order_relaxed) < s->enqueued - 1) {
            FS_HIP(ctx, hipEventSynchronize(s->ev[slot]));
            poll_published(s);
        }
    }
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
    }
    // The tail stream takes over: it waits for this frame's deposit (and runs behind any collective the caller
    // put there after fs_energy_handoff), reconstructs and publishes while the compute stream goes on to the
    // next frame.  Reconstructs and publishes of one source are ordered among themselves by the tail stream.
    FS_HIP(ctx, handoff_energy(ctx, s));
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
        s->rec_recorded[s->cur] = true;
        s->last_rec = s->cur;
    }
    uint64_t seq = s->enqueued + 1;
    int slot = (int)(seq % kIrRing);
    FS_HIP(ctx, hipMemcpyAsync(s->h_ir[slot], s->d_ir_mono, sizeof(float) * (size_t)ctx->num_samples,
                               hipMemcpyDeviceToHost, tail));
    FS_HIP(ctx, hipEventRecord(s->ev[slot], tail));
    s->seq_of[slot] = seq;
    s->enqueued = seq;
    if (timed) {
        FS_HIP(ctx, hipEventRecord(tf.e[4], tail));
        tf.has_recon = true;
        ctx->pending.push_back(tf);
    }
    return FS_OK;
}

extern "C" {

int fs_set_impulse_response(fs_context* ctx, fs_source h, const float* ir, int32_t n) {
    if (!ctx || !ir) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    Source* s = get_source(ctx, h);
    if (!s) return ctx->fail(FS_ERR_BAD_HANDLE, "bad source handle");
    if (n != ctx->num_samples) return ctx->fail(FS_ERR_SIZE_MISMATCH, "n != num_samples");
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    poll_published(s);
    if (s->enqueued >= 2) {   // never overwrite the front buffer: at most two publishes in flight
        int slot = (int)((s->enqueued - 1) % kIrRing);
        if (s->seq_of[slot] == s->enqueued - 1 && s->front.load(std::memory_order_relaxed) < s->enqueued - 1) {
            FS_HIP(ctx, hipEventSynchronize(s->ev[slot]));
            poll_published(s);
        }
    }
    hipStream_t tail = ctx->copy_stream;   // ordered with reconstructs and publishes of this source
    const size_t bytes = sizeof(float) * (size_t)n;
    {
        std::lock_guard<std::mutex> g(s->ir_mu);   // against fs_reverb_process on the audio thread
        if (s->rev_recorded) FS_HIP(ctx, hipStreamWaitEvent(tail, s->ev_rev, 0));
        FS_HIP(ctx, hipMemcpyAsync(s->d_ir_mono, ir, bytes, hipMemcpyHostToDevice, tail));
        for (int b = 0; b < ctx->cfg.num_bands; ++b)
            FS_HIP(ctx, hipMemcpyAsync(s->d_ir_bands + (size_t)b * (size_t)n, s->d_ir_mono, bytes, hipMemcpyDeviceToDevice, tail));
        const int cur = s->last_rec >= 0 ? s->last_rec : s->cur;
        FS_HIP(ctx, hipEventRecord(s->ev_rec[cur], tail));   // the reverb waits on this before reading d_ir_mono
        s->rec_recorded[cur] = true;
        s->last_rec = cur;
    }
    uint64_t seq = s->enqueued + 1;
    int slot = (int)(seq % kIrRing);
    FS_HIP(ctx, hipMemcpyAsync(s->h_ir[slot], s->d_ir_mono, bytes, hipMemcpyDeviceToHost, tail));
    FS_HIP(ctx, hipEventRecord(s->ev[slot], tail));
    s->seq_of[slot] = seq;
    s->enqueued = seq;
    FS_HIP(ctx, hipStreamSynchronize(tail));   // `ir` is the caller's memory
    poll_published(s);
    return FS_OK;
}

int fs_synchronize(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    FS_FLUSH(ctx);   // pipelined frames: a held-back connect pass goes first
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (Source* s : ctx->sources)
        if (s && s->alive) poll_published(s);
    resolve_timings(ctx);
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
    FS_HIP(ctx, wait_energy_readers(ctx, s));
    FS_HIP(ctx, hipMemcpyAsync(s->energy(), values, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    FS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FS_OK;
}

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

int fs_scene_set_objects(fs_context* ctx, const uint32_t* object_id, int32_t T) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (object_id && T != ctx->T) return ctx->fail(FS_ERR_SIZE_MISMATCH, "object ids: T != number of triangles");
    if (object_id) ctx->h_obj.assign(object_id, object_id + T);
    else ctx->h_obj.clear();
    ctx->committed = false;
    return FS_OK;
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
    if (!any_hit && N > 0 && (!t || !tri || !normal)) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "closest-hit outputs required");
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
    FS_TRY(hipMalloc((void**)&d_hit, n1));
    FS_TRY(hipMalloc((void**)&d_tri, n1));
    FS_TRY(hipMemcpyAsync(d_o, origins, n3, hipMemcpyHostToDevice, ctx->stream));
    FS_TRY(hipMemcpyAsync(d_d, dirs, n3, hipMemcpyHostToDevice, ctx->stream));
    FS_TRY(hipMemcpyAsync(d_tm, tmax, n1, hipMemcpyHostToDevice, ctx->stream));
    launch_trace_rays(ctx->scene, d_o, d_d, d_tm, N, any_hit, d_hit, d_t, d_tri, d_n, ctx->stream);
    FS_TRY(hipGetLastError());
    FS_TRY(hipMemcpyAsync(hit, d_hit, n1, hipMemcpyDeviceToHost, ctx->stream));
    if (!any_hit) {
        FS_TRY(hipMemcpyAsync(t, d_t, n1, hipMemcpyDeviceToHost, ctx->stream));
        FS_TRY(hipMemcpyAsync(tri, d_tri, n1, hipMemcpyDeviceToHost, ctx->stream));
        FS_TRY(hipMemcpyAsync(normal, d_n, n3, hipMemcpyDeviceToHost, ctx->stream));
    }
    FS_TRY(hipStreamSynchronize(ctx->stream));
#undef FS_TRY
    cleanup();
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
        if (s->last_rec >= 0) FS_HIP(ctx, hipStreamWaitEvent(rs, s->ev_rec[s->last_rec], 0));
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

// ---- measurement ----------------------------------------------------------------------------------------------
int fs_set_pipelining(fs_context* ctx, int32_t on) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (on < 0 || on > 2) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "fs_set_pipelining: 0 (off), 1 or 2 frames held back");
    if (on != ctx->pipelining) FS_FLUSH(ctx);
    ctx->pipelining = on;
    return FS_OK;
}

int fs_submit(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    return flush_pending(ctx);
}

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
        ctx->stats.segments = c[0] + ctx->host_segments;
        ctx->stats.connections_tested = c[1];
        ctx->stats.deposits = c[2];
        ctx->stats.walk_node_fetches = c[3];
        ctx->stats.walk_tri_fetches = c[4];
        ctx->stats.any_node_fetches = c[5];
        ctx->stats.any_tri_fetches = c[6];
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
