// fs_context.hpp — the context behind the C ABI (include/frequensee.h) and the helpers its translation units share.
// Internal: included by fs_capi_context.cpp (lifetime, helpers, stats), fs_capi_scene.cpp (geometry, commits, refit),
// fs_capi_frame.cpp (sources, the traced frame: describe / resources / commit / launch), fs_capi_pipeline.cpp (held frames, the
// drain), fs_capi_publish.cpp (reconstruct + publish: the IR ring, the host word, the fused launch's reconstruct parts),
// fs_capi_ir.cpp (reconstruct / tick / IR / energy entry points), fs_capi_comm.cpp (RCCL behind the ABI) and fs_capi_aux.cpp
// (legacy tracer, line trace, text interchange, reverb, material FD).
//
// Mirrors the roles of UAudioRayTracingSubsystem (context lifetime, geometry/source registries,
// per-source update: AudioRayTracingSubsystem.cpp:32-53, 128-195) and of UFrequenSeeAudioComponent's
// buffers (EnergyBuffer, ImpulseBuffer: FrequenSeeAudioComponent.h:69-91, 113, 133-143).  All compute
// is HIP on the context's streams; there is no CPU fallback.
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <thread>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl is opened at run time (fs_comm_*), never linked

#include "fs_internal.hpp"

using namespace fs;

namespace fsi {

constexpr int kIrRing = 8;  // published IR ring: up to 7 publishes of a source in flight (grouped frames of one source finish
                            // together); a returned pointer stays valid until the 7th-next publish (3 and "second-next" until round 3)

// Energy buffers per source, used in rotation: frame f deposits into one while the tail stream still reduces /
// reconstructs frame f - 1 from another; pipelined frames finish frame f - 2 only in the launch that plans frame f
// (and flushes ITS buffer), so that a fourth one lets frame f + 1 start without waiting for that tail.
// Frames in flight.  A pipelined frame lives through several launches — plan, its walk (one launch, or one launch per
// stage of a staged depth = 0 walk), connect — while the frames behind it are in their earlier steps; everything a
// frame owns rotates with the frame index.  kMaxSets bounds the rotation: energy buffers per source (zeroed by the
// plan pass of frame f, deposited into up to kMaxWalkParts + 1 launches later, then read by the tail), frame scratch
// sets (plan counts, cursors, work counters), per-frame tables of batched frames.  The big per-frame arrays (subpath
// state, segment records, schedules, continuation records) come in fs_context::state_sets sets, as many as the deepest
// pipeline in use needs (3 until a staged walk is pipelined).
constexpr int kMaxSets = 12;
// Energy buffers of a source.  A frame OWNS its buffer from the launch that plans it (and zeroes the buffer) to the launch
// that reconstructs it: stages + 3 launches for a staged depth = 0 frame, times the frames of the source per launch
// (fs_set_frames_per_launch).  Up to two frames per launch rotate through the first kEnergyBufsBase (11 launches x 2 at
// most; with 2 or 3 frames per launch the rotation then repeats every kBatchSlots launches or sooner, which is what lets a
// batched frame find its table already on the device — 48 buffers at two per launch: 928 -> 914 M rays/s); three and four
// per launch rotate through all kEnergyBufs (11 launches x 4), the buffers beyond the base allocated at their first use.
// frame_resources never hands out a buffer that a frame in flight still owns, whatever the rotation says (round 3 did:
// four staged frames per launch wrapped the 24-buffer rotation after six launches and zeroed buffers still being deposited into).
constexpr int kEnergyBufsBase = 2 * kMaxSets;
constexpr int kEnergyBufs = 4 * kMaxSets;
constexpr int kScratchSets = kMaxSets;
static_assert(kMaxSets >= kMaxWalkParts + 3, "a staged frame is in flight for stages + 2 launches and its buffer is read one more");
static_assert(kEnergyBufs >= 4 * (kMaxWalkParts + 3), "four frames of one source per launch, each owning its buffer for stages + 3 launches");

// Events that order work of two streams of the SAME device (the compute stream's hand-over to the tail stream, the sum over the
// ranks back to the compute stream, the reverb stream's reads): a DEVICE-scope release.  The default — a system-scope fence when
// the event is recorded — writes every dirty L2 line back and invalidates the caches: behind a frame kernel that has just written
// 100 MB of records that is tens of microseconds in which nothing runs, and the next launch starts on cold caches
// (tools/rccl_tax.sh: a one-rank communicator cost 7.5 % of the plain rate with default events).  Events the HOST inspects
// before it reads host memory (the publishes' ev[slot], tail_batch_ev) keep the system scope.
constexpr unsigned kDeviceEventFlags = hipEventDisableTiming | hipEventReleaseToDevice;

struct Source {
    bool alive = false;
    float pos[3] = {0, 0, 0};
    uint32_t object = FS_NO_OBJECT;    // the actor this source belongs to: its own walks ignore it (fs_source_set_object)
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
    bool tail_ordered = false;         // the tail stream already waits for everything the compute stream wrote into the current buffer
    hipEvent_t ev_red[kEnergyBufs] = {};   // tail stream: the library's all-reduce of buffer i is done
    bool red_recorded[kEnergyBufs] = {};
    // deterministic mode (FS_FLAG_DETERMINISTIC): u64 fixed-point histograms [B][bins], allocated on first use,
    // alternating like the energy buffers; cur_fixed = the current frame deposited into d_fixed[cur]
    unsigned long long* d_fixed[kEnergyBufs] = {};
    bool cur_fixed = false;
    float* d_ir_bands = nullptr;  // [B + 1][samples]: the bands, then ...
    float* d_ir_mono = nullptr;   // ... (= d_ir_bands + B * samples) the channel view (all channels identical, FSAC.cpp:331)
    // newest publish that reads or writes the device IR set from the TAIL stream (a copy command behind reconstruct_now, a batch
    // kernel): a reconstruct on the compute stream waits for it before it writes the set.  0: none / a compute-stream publish.
    uint64_t cur_pub_seq = 0;
    uint64_t recon_sync_mark = ~0ull;    // fs_context::syncs at this source's last stand-alone reconstruct (fs_reconstruct_impulse_response_async)
    // ... and the other way round: the newest COMPUTE-stream launch that writes the set (its id in fs_context::h_pub_word; 0: none) —
    // a reconstruct on the tail stream lets the compute stream hand over first while that launch may still be running
    uint64_t dev_ir_word = 0;
    float* h_ir[kIrRing] = {};  // pinned host copies of the channel view
    int mask_index = -1;        // this source's row of fs_context::d_slot_masks (its handle; -1: none — every block is always written)
    hipEvent_t ev[kIrRing] = {};
    // How ring slot `slot` is known to be published, in this order: pub_word[slot] != 0 — a launch on the COMPUTE stream whose
    // reconstruct workgroups wrote the slot themselves and whose id appears in fs_context::h_pub_word (publish_arrive, fs_device.hpp:
    // no event, nothing on the tail stream); pub_batch[slot] != 0 — a batched reconstruct on the TAIL stream, one event for all its
    // sources (fs_context::tail_batch_ev); else the slot's own event ev[slot] behind a copy command on the tail stream.
    // rec_batch[i] != 0 names the tail batch whose event stands for ev_rec[i]; rec_on_compute[i]: the reconstruct ran on the compute
    // stream — nothing that stream does later needs to wait for it (ev_rec[i] is recorded only for a source with a reverb).
    uint64_t rec_batch[kEnergyBufs] = {};
    bool rec_on_compute[kEnergyBufs] = {};
    std::atomic<uint64_t> pub_word[kIrRing] = {};
    std::atomic<uint64_t> pub_batch[kIrRing] = {};
    std::atomic<uint64_t> seq_of[kIrRing] = {};   // (atomics: fs_get_impulse_response_sequence may look from another thread)
    std::atomic<uint64_t> enqueued{0};            // publishes enqueued so far
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

}  // namespace fsi
using namespace fsi;

// fs_scene_commit_progressive: the host's SAH build of a snapshot of the registered triangles, on its own thread, while
// the frames already trace through the device-built tree.  The thread touches nothing but this object.
struct RefineJob {
    std::vector<float> xyz; std::vector<uint16_t> mat; std::vector<uint32_t> obj;
    int T = 0;
    fs::HostBVH bvh;
    std::mutex mu; std::condition_variable cv;
    bool done = false;
    std::atomic<bool> cancel{false};   // a newer registration made this build useless: the builder gives up at its next check
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
    std::string advice;                // what fs_context_create found worth telling the host (fs_context_advice); never an error
    std::mutex err_mu;                 // the audio thread may fail too

    // scene (host staging + device)
    std::vector<float> h_xyz;
    std::vector<uint16_t> h_mat;
    std::vector<uint32_t> h_obj;   // actor id per triangle (empty = one actor per triangle)
    std::vector<float> h_absorption, h_transmission, h_scattering;
    int32_t T = 0, M = 0;
    bool committed = false;
    NodeQ4* d_nodes = nullptr;
    CoopChild* d_coop = nullptr;      // per-child records of the 4-wide nodes (fs_internal.hpp), rebuilt behind every commit and refit ...
    size_t coop_cap = 0;              // nodes it has room for
    CoopChild* d_coop16 = nullptr;    // ... and folded into the 16-wide nodes the cooperative traversal walks
    size_t coop16_cap = 0;
    int32_t* d_coop_levels = nullptr; // [2][kMaxBuildLevels + 2]: level_begin | dense index of every even level (uploaded at commit)
    int coop16_nodes = 0, coop_levels = 0;
    CoopInfo coop_info;               // what DeviceScene.coop_info points at
    float stage_margin = 1.3f;        // KParams.stage_margin
    DeepStore deep;                   // HBM spill area of the bounded LDS traversal stacks (DeviceScene.deep)
    int stack_rows_cap = kStackRowsCap;   // FS_STACK_ROWS_CAP
    Tri64* d_tris = nullptr;          // authoring records (builders, refit, fs_scene_update_triangles)
    Tri48* d_tris48 = nullptr;        // what the kernels traverse (fs_internal.hpp: Tri48), derived from d_tris
    float4* d_tri_nrm = nullptr;      // unit normals, leaf order
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
    uint32_t listener_object = FS_NO_OBJECT;   // fs_listener_set_object
    std::vector<Source*> sources;

    // multi-GPU: RCCL communicator over the ranks that share the pairs of every frame
    ncclComm_t comm = nullptr;
    // cfg5 (independent sources, one per GPU): a communicator that never touches a frame — only fs_gather_energy uses it
    // fs_comm_enable_oneshot: the energy buffer's sum as one peer-write exchange instead of ncclAllReduce (fs_oneshot.hip)
    struct OneShot {
        bool on = false;
        OneShotView view{};
        void* own_mail = nullptr;         // this rank's mailbox (hipMalloc); the others are IPC mappings
        unsigned* d_err = nullptr;        // raised by a sum kernel that gave up waiting for a peer
        uint32_t seq = 0;                 // reduces issued so far
        bool broken = false;              // a sum gave up waiting for a peer: the sets are out of step, every later reduce is refused
    } oneshot;
    char* d_comm_stage = nullptr;         // 2 KB of device staging for the small collectives of fs_comm_enable_oneshot, allocated with the
                                          //   context: a rank that is out of memory later can still take part in them
    // One event per batched reconstruct launch on the tail stream, in a ring: the tail stream runs in order, so a batch is done as
    // soon as its own or ANY younger batch's event is (a batch whose ring entry has been recycled is covered by the oldest entry
    // still there).  Created with the context: the handles never change (the audio thread may query them).
    static constexpr int kTailBatches = 64;
    hipEvent_t tail_batch_ev[kTailBatches] = {};
    std::atomic<uint64_t> tail_batch_newest{0}, tail_batch_done{0};
    static constexpr int kReconTabSlots = 8, kReconTabItems = 256;
    ReconItem* h_recon_tab = nullptr;                  // pinned host: [kReconTabSlots][kReconTabItems], read by the batch kernel in place
    uint64_t recon_tab_batch[kReconTabSlots] = {};     // the tail-stream batch that last read slot k, or ...
    uint64_t recon_tab_word[kReconTabSlots] = {};      // ... the compute-stream launch (publish word id) that did
    unsigned recon_tab_next = 0;
    ncclComm_t peers = nullptr;
    int peers_size = 0;
    float* d_gather = nullptr; size_t gather_cap = 0;   // [peers][B][bins] fp32
    // Pipelined frames (fs_set_pipelining).  depth 1: the connect pass of frame f is held back and launched together with
    // the walk of frame f + 1 as ONE kernel; depth 2: the walk is held back as well — call f launches {plan of f, walk of
    // f - 1, connect of f - 2} as one kernel.  Anything that needs a held frame's result lets it finish alone (flush_pending).
    struct PipeFrame {
        KParams kp; SubpathState st;
        WalkLaunch wl;               // queue_head = the frame's scratch set, rays_per_wave of its first stage
        const uint32_t* perm = nullptr;   // its schedule (nullptr: none)
        std::vector<WalkStage> stages;    // the walk in one piece ({0, depth}) or the stages of a staged depth = 0 walk
        int next_stage = 0;          // stages [0, next_stage) have been launched; == stages.size(): only the connect pass is owed
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
        unsigned long long* const* fixed_tab = nullptr;   // rotate; a held frame is connected kMaxWalkParts + 1 calls later at most)
    };
    std::deque<PipeFrame> held;
    // fs_set_frames_per_launch(n > 1): plain pipelinable frames are collected until n are waiting, then traced as ONE
    // batched frame in which every item keeps its own seed (and its own energy buffer, also for the same source twice) —
    // a 262 144-ray frame leaves a tenth of the chip idle that a launch of two such frames fills.  Everything that
    // flushes held frames dispatches a partial group first.
    struct GroupEntry {
        Source* s = nullptr; fs_params p; bool want_recon = false; fs_params recon;
        float pos[3] = {}, lis[3] = {};   // source and listener position AT THE CALL (either may move before the group is sent off)
        uint32_t object = FS_NO_OBJECT;
    };
    std::vector<GroupEntry> group;
    int frames_per_launch = 1;
    // Reconstructs that ride in the NEXT fused launch (single GPU): the frames the last launch connected owe their IR; a
    // reconstruct part behind the plan part produces it (fs_frame.hip) instead of a kernel on the tail stream that has to
    // squeeze into a chip the frame kernel fills (2 % of the frame kernel's time for 0.5 % of its work).  Everything that
    // flushes runs what is owed on the tail stream as before.  A frame that is not the source's last in the launch writes
    // its IR into a temporary buffer (two frames of one source would write the same d_ir_* at once) and is published from there.
    // reduced: the library has summed the buffer over the ranks on the tail stream (ev_red[cur]); such an entry rides in the
    // launch AFTER next (age >= 1 when a launch is assembled): the all-reduce runs behind the launch that connected the
    // frame, i.e. while the next one executes — a launch that waited for it would leave the GPU idle meanwhile
    struct ReconOwed { Source* s = nullptr; int cur = 0; bool fixed = false; fs_params p; bool reduced = false; int age = 0; };
    std::vector<ReconOwed> recon_owed;
    bool fused_recon = true;                       // FS_FUSED_RECON=0: always the tail stream
    bool fused_recon_comm = true;                  // FS_FUSED_RECON_COMM=0: with a communicator, always the tail stream
    bool fused_drain = true;                       // FS_FUSED_DRAIN=0: a flush lets every held frame finish on kernels of its own (round 4's form)
    // Publishes of the compute stream (fused reconstruct parts, batches behind a tick or a flush): the launch writes the ring slots
    // and then its id into *h_pub_word (pinned, coherent) — see Source::pub_word.  pub_issued = id of the newest such launch.
    unsigned* d_pub_tickets = nullptr;             // device: the ticket cell of publish_arrive (re-armed by the launch that used it)
    // zero-block masks of the sources' host ring slots (fs_device.hpp: host_block_wanted): [kMaxMaskSources][kIrRing] words on the
    // device, bit b of word (source, slot) = block b of that slot may hold non-zero samples.  Kept by the reconstruct kernels; a
    // copy command into a slot (reconstruct_now, fs_set_impulse_response) sets the slot's word to all ones behind itself.
    uint32_t* d_slot_masks = nullptr;
    unsigned long long* h_pub_word = nullptr;
    uint64_t pub_issued = 0;
    int state_sets = 3;              // sets of the per-frame arrays (subpath state, records, schedules): frames in flight + 1
    bool state_cont = false;         // the sets include continuation records (staged walks)
    std::vector<int> stage_bounds;   // staged depth = 0 walks: the steps at which a walk changes launch (FS_WALK_STAGES)
    bool stage_bounds_default = true; // not set by the host: grouped frames (two or more per launch) take kGroupedStageBounds
    std::vector<int> sync_stage_bounds;   // the same for depth = 0 frames that are waited for (stages back to back; FS_SYNC_WALK_STAGES)
    int sync_first_rays_per_wave = 0;     // subpaths per wave of the first stage, 0 = by frame size (FS_SYNC_FIRST_RPW)
    int sync_late_rays_per_wave = 0;      // subpaths per wave of the later stages, 0 = by the number of survivors (FS_SYNC_LATE_RPW)
    std::vector<int> sync_stage_rpw;      // subpaths per wave of stage k of such a frame, overriding the two above where > 0 (FS_SYNC_STAGE_RPW: "32,64,0")
    // the long-walk lane of such a frame (WalkLane): walks of sync_lane_len steps or more take cooperative waves of their
    // own from step 0 on, steps [0, sync_lane_end) beside the first stage, the rest beside the second (FS_SYNC_LANE: "len,end"; len 0: off)
    // -1 (default): by the frame's size and the roulette — sync_lane_plan, fs_capi_frame.cpp; with it the ONE default bound moves too
    int sync_lane_len = -1, sync_lane_end = 1 << 30;
    bool sync_stage_bounds_default = true;
    uint64_t syncs = 0;                   // fs_synchronize calls (a stand-alone reconstruct asks whether its producer waits for every frame: Source::recon_sync_mark)
    bool sync_stage_from_default = true;
    int sync_stage_from = 15000;          // ... of at least this many subpaths (smaller frames: every walk has a wave of its own anyway; with the long-walk lane an 8-source tick — 16 000 — gains 5 %, 14 000 lose: profiles/r05_stage_from.jsonl; 16 384 until then)
    int stage_dense_from = 4096;     // stages with at least this many (provisioned) walks use dense waves (FS_STAGE_DENSE_FROM)
    std::shared_ptr<RefineJob> refine;   // fs_scene_commit_progressive: the background build whose tree replaces the device-built one
    std::vector<std::thread> refine_threads;   // every background build ever started (cancelled ones too): joined before the context goes
    bool moved_since_refine = false;     // fs_scene_update_triangles since the snapshot: re-apply the positions after the swap
    fs::HostBVH* prebuilt = nullptr;     // fs_scene_commit takes this tree instead of building one (install of a refined tree)
    int pipelining = 0;              // 0 off, 1 / 2 = frames held back
    unsigned frame_index = 0;        // consecutive traced frames rotate through the state / schedule / scratch sets
    size_t perm_words = 0;           // words of ONE schedule set (walk.perm holds state_sets)
    bool comm_owned = false;           // created by fs_comm_init (destroyed with the context) vs attached by the caller

    // subpath state (sized on demand)
    SubpathState st{};
    size_t cap_lanes = 0, cap_seg = 0;
    float4* d_seg_pos = nullptr;   // node positions per walk step, all-connections mode only (row f3)
    size_t cap_pos = 0;
    // second record tier of depth = 0 frames (walk steps beyond FS_MAX_DEPTH): [kOverLevels][over_cap] each, grown when
    // a frame raises the overflow word; d_overflow = that word
    float2* d_over_np = nullptr; uint32_t* d_over_mat = nullptr; float4* d_over_pos = nullptr;   // state_sets tiers each (positions: one)
    double* d_end_posd = nullptr; size_t cap_posd = 0;   // FS_FLAG_DOUBLE_POSITIONS: end points in double [lanes][3] (such frames are never held: one set)
    float4* d_cont = nullptr;        // continuation records of staged walks: [state_sets][2][cap_lanes]
    uint32_t over_cap = 0, over_cap_pos = 0;
    // FS_DEBUG_STALLS: where the producer waited (printed by fs_context_destroy): flushes of held frames, host waits for a publish
    // (count, microseconds), waits enqueued on the compute stream for another stream's event, owed reconstructs run on the tail stream
    bool flush_recon_on_compute = true;   // FS_FLUSH_RECON_ON_COMPUTE (fs_capi_publish.cpp: flush_reconstruct)
    bool debug_stalls = false;
    struct { uint64_t flushes = 0, flushed_frames = 0, sync_publish = 0, sync_publish_us = 0, waits_enqueued = 0, waits_skipped = 0, owed_on_tail = 0, launches = 0,
             tail_ops = 0, pub_word = 0, pub_event = 0, lane_launches = 0; } dbg;   // fs_get_pipeline_counters
    int over_cap_forced = 0;       // FS_OVER_CAP, read at fs_context_create
    unsigned* d_overflow = nullptr;   // the device's address of ...
    unsigned* h_overflow = nullptr;   // ... this pinned host word
    bool overflow_armed = false;   // an unbounded frame has been enqueued since the word was last read
    // batched frames (fs_compute_energy_response_batch_async): per-frame tables of pointers and source positions
    static constexpr int kBatchSlots = 2 * kMaxSets;   // frames the host may run ahead of the table copies (a held frame keeps its tables); >= the
                                                       //   launches after which the energy-buffer rotation of grouped frames repeats (24 / 2, 48 / 3, 48 / 4)
    char* d_batch = nullptr; char* h_batch = nullptr; size_t batch_cap = 0;   // kBatchSlots blocks of batch_cap bytes
    hipEvent_t ev_batch[kBatchSlots] = {};
    bool batch_pending[kBatchSlots] = {};
    size_t batch_bytes[kBatchSlots] = {};   // bytes of the table the slot's device copy holds (0: none) — same table again: no copy
    std::vector<char> batch_build;
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

#define FS_FLUSH(ctx)                       \
    do {                                    \
        int fr_ = fsi::flush_pending(ctx);  \
        if (fr_) return fr_;                \
    } while (0)

#define FS_HIP(ctx, call)                                        \
    do {                                                         \
        hipError_t e_ = (call);                                  \
        if (e_ != hipSuccess) return (ctx)->hip_fail(e_, #call); \
        if ((ctx)->deep.failed) {   /* a launcher could not grow the traversal stacks' deep store and skipped its launch */ \
            (ctx)->deep.failed = false;                          \
            return (ctx)->fail(FS_ERR_OUT_OF_MEMORY, "no device memory for the traversal stacks' deep store"); \
        }                                                        \
    } while (0)

#define FS_NCCL(ctx, call)                                               \
    do {                                                                 \
        ncclResult_t r_ = (call);                                        \
        if (r_ != ncclSuccess) return fsi::nccl_fail((ctx), r_, #call);  \
    } while (0)

constexpr float kMaxUnboundedRr = 0.95f;   // depth = 0 (uncapped walks): the strongest survival probability the record store covers

namespace fsi {

// ---- fs_capi_context.cpp ------------------------------------------------------------------------------------------
Source* get_source(fs_context* ctx, fs_source h);
hipError_t wait_energy_readers(fs_context* ctx, Source* s);            // before the compute stream writes the current energy buffer
hipError_t compute_waits_for(fs_context* ctx, hipEvent_t ev);   // hipStreamWaitEvent on the compute stream unless ev has completed
hipError_t wait_energy_readers(fs_context* ctx, Source* s, int buf);   // ... or the buffer the next frame will use
hipError_t handoff_energy(fs_context* ctx, Source* s);        // compute stream -> tail stream
void free_source(fs_context* ctx, Source* s);
void free_scene(fs_context* ctx);
void free_state(fs_context* ctx);
hipEvent_t take_event(fs_context* ctx);
void resolve_timings(fs_context* ctx);
void resolve_completed_timings(fs_context* ctx);
void poll_published(fs_context* ctx, Source* s);
bool tail_batch_done(fs_context* ctx, uint64_t id);                 // any thread
hipEvent_t tail_batch_event(fs_context* ctx, uint64_t id);          // the event that covers batch `id`
hipError_t stream_waits_for_rec(fs_context* ctx, hipStream_t st, Source* s, int buf);   // st waits for the reconstruct that last read energy buffer `buf` (and wrote d_ir_*)
hipError_t wait_event_polling(hipEvent_t ev);   // short producer waits: poll, do not sleep (fs_capi_context.cpp)
bool pub_word_done(const fs_context* ctx, uint64_t id);             // any thread: has compute-stream publish `id` reached the host word?
hipError_t wait_pub_word(fs_context* ctx, uint64_t id);             // producer: poll the host word (falls back to a stream wait)
PublishWord next_pub_word(fs_context* ctx);                         // the arguments of the next self-publishing launch (id = pub_issued + 1; commit with ++pub_issued)
bool slot_published(fs_context* ctx, Source* s, int slot);          // any thread: the publish in ring slot `slot` has completed
uint32_t* slot_mask_ptr(const fs_context* ctx, const Source* s, int slot);   // the slot's zero-block mask word on the device (nullptr: none)
hipError_t slot_mask_all_dirty(fs_context* ctx, const Source* s, int slot, hipStream_t st);   // behind a copy command into the slot
hipError_t sync_publish(fs_context* ctx, Source* s, int slot);      // block until the publish in ring slot `slot` has completed
int ensure_state(fs_context* ctx, uint32_t n_local, int levels, bool unbounded, double rr_prob, bool want_positions, bool want_normals,
                 int sets, bool staged);
int auto_rays_per_wave(unsigned long long lanes, int depth, unsigned long long waves = 0);
int auto_pairs_per_wave(unsigned long long pairs);
int check_params(fs_context* ctx, const fs_params* p);

// ---- fs_capi_scene.cpp --------------------------------------------------------------------------------------------
// fs_scene_commit_progressive: swap the finished background tree in; drop / wait for a background build
int maybe_install_refined(fs_context* ctx);
void cancel_refine(fs_context* ctx);
void wait_refine(const std::shared_ptr<RefineJob>& j);
void join_refine_threads(fs_context* ctx);

// ---- fs_capi_frame.cpp --------------------------------------------------------------------------------------------
int dispatch_group(fs_context* ctx);         // the frames collected by fs_set_frames_per_launch as ONE batched frame

// ---- fs_capi_pipeline.cpp -----------------------------------------------------------------------------------------
int flush_pending(fs_context* ctx);          // pipelined frames: drain the pipeline (fused launches, or every held frame on its own kernels)
int check_overflow(fs_context* ctx);         // depth = 0: did a record miss both tiers?  (stream just synchronised)
// what a held frame still owes once its connect pass has been enqueued (fixed-point rounding, the sum over the ranks, its reconstruct)
int finish_held_frame(fs_context* ctx, const fs_context::PipeFrame& q, bool may_defer_recon = false, bool tail_waits_already = false, bool draining = false);
void held_connect_part(const fs_context::PipeFrame& q, FrameParts& f);                         // a held frame's connect pass as a part of the fused launch
bool held_walk_part(const fs_context* ctx, const fs_context::PipeFrame& q, FrameParts& f);   // ... its next walk stage (false: no room for more walk parts)
WalkLaunch stage_launch(const fs_context* ctx, const fs_context::PipeFrame& q, int stage);

// ---- fs_capi_publish.cpp ------------------------------------------------------------------------------------------
// The reconstruct parts of a fused launch: owed_prepare turns the owed reconstructs into parts of the launch being assembled (and
// takes their sources' IR mutexes), owed_publish notes the publishes behind the launch (which announces them itself).
struct OwedLaunch {
    std::vector<fs_context::ReconOwed> owed;
    std::vector<uint64_t> seq;                             // publish number of each entry (ring slot = seq % kIrRing)
    std::vector<char> newest;                              // the source's newest frame of the launch: its IR becomes the device-resident set
    std::vector<std::unique_lock<std::mutex>> locks;
    PublishWord pub;
    unsigned tab_slot = 0;                                 // the slot of fs_context::h_recon_tab that holds the launch's items
};
int owed_prepare(fs_context* ctx, FrameParts& fp, OwedLaunch& ol);
int owed_publish(fs_context* ctx, OwedLaunch& ol, bool launched_fused);
int run_owed_reconstructs(fs_context* ctx);  // the reconstructs that were waiting for the next fused launch, on kernels of their own after all
int flush_reconstruct(fs_context* ctx, Source* s, const fs_params* p);   // a reconstruct no launch is fused with (one GPU: a compute-stream batch of one)
int reconstruct_now(fs_context* ctx, Source* s, const fs_params* p);
int reconstruct_batch(fs_context* ctx, Source* const* srcs, int count, const fs_params* p, bool on_compute);   // many sources: one launch, one event
int ir_ring_backpressure(fs_context* ctx, Source* s, int more);   // the IR ring's throttle before `more` publishes (may block; not under ir_mu)
int ir_ring_backpressure_for(fs_context* ctx, Source* s);
hipError_t tail_waits_for_compute_ir(fs_context* ctx, Source* s);   // before the tail stream writes the source's device IR set

// ---- fs_capi_comm.cpp ---------------------------------------------------------------------------------------------
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
RcclApi* rccl();   // nullptr: librccl is not loadable
int nccl_fail(fs_context* ctx, ncclResult_t r, const char* what);
int reduce_energy(fs_context* ctx, Source* s);   // sum the source's current energy buffer over the ranks (tail stream)
void oneshot_release(fs_context* ctx);
int oneshot_check(fs_context* ctx);             // FS_ERR_COMM if a one-shot sum gave up waiting (tail stream synchronised)

}  // namespace fsi
