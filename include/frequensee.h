/*
 * frequensee.h — C ABI of the MI355X-native FrequenSee acoustic BDPT path.
 *
 * Drop-in boundary for the per-frame bidirectional path trace + energy-buffer loop of the
 * FrequenSee Unreal plugin (henreedev/audio-pathtracer).  Every entry point names the reference
 * interface it replaces; paths are relative to Plugins/FrequenSee/Source/FrequenSee/ in the
 * reference tree:
 *   ARTS.h/.cpp = Public/AudioRayTracingSubsystem.h, Private/AudioRayTracingSubsystem.cpp
 *   FSAC.h/.cpp = Public/FrequenSeeAudioComponent.h, Private/FrequenSeeAudioComponent.cpp
 *   MAT.h, GEO.h = Public/AcousticMaterial.h, Public/AcousticGeometryComponent.h
 *
 * Conventions: extern "C", POD only, every call returns an int status (FS_OK == 0), no exception
 * crosses the boundary, output buffers are caller-allocated, handles are opaque.  Positions are
 * Unreal units (cm).  One context drives one HIP device.  Its tracing is ordered on one HIP stream (the
 * "compute" stream: its own, or the caller's via fs_config.stream).  On one GPU the reconstruct and the publish of
 * the impulse response ride on that stream too: the reconstruct workgroups write the pinned host ring slot themselves
 * and announce it in a pinned host word — no second queue, no copy command, no event in the steady state of a stream
 * of frames.  The context's second ("tail") stream carries what has to run beside the tracing: a multi-GPU reduce
 * (the library's or a caller's behind fs_energy_handoff) with the reconstruct of such a frame, and the few reconstructs
 * with per-kernel timing or a literal second flush.  A context is not re-entrant:
 * one producer thread (the game thread) calls compute/reconstruct; concurrently with it any number of threads may
 * read published impulse responses (fs_get_impulse_response), and ONE audio render thread may run the reverb callback
 * (fs_reverb_process) — it has a HIP stream of its own and is never queued behind a traced frame.
 * The tail stream and the reverb stream are created with the highest HIP stream priority (small work somebody waits for;
 * priority streams have hardware queues of their own); FS_TAIL_STREAM_PRIORITY=0 in the environment makes them ordinary.
 *
 * There is no CPU fallback: if no HIP device is usable every compute entry point fails with
 * FS_ERR_NO_DEVICE.
 *
 * TWO TIERS.  CORE = the drop-in boundary itself: what a UE shim inside UpdateSource / TickComponent
 * binds, and nothing a host has to know beyond the reference's own interface —
 *   CORE:     fs_abi_version fs_config_default fs_params_default fs_context_create fs_context_destroy fs_last_error
 *             fs_scene_set_triangles fs_scene_set_materials fs_scene_set_objects fs_scene_commit
 *             fs_source_create fs_source_destroy fs_source_set_position fs_source_set_object
 *             fs_listener_set_position fs_listener_set_object
 *             fs_compute_energy_response fs_reconstruct_impulse_response fs_update_sources
 *             fs_get_impulse_response fs_copy_impulse_response fs_get_impulse_response_sequence
 *             fs_get_energy_buffer fs_flush_energy_buffer fs_add_energy_at_delay fs_update_energy_buffer
 *             fs_num_bins fs_num_samples fs_get_occlusion_attenuation fs_update_sound fs_sound_params_default
 *             fs_get_stats fs_reset_stats
 * EXTENDED = everything else: the asynchronous / batched / pipelined forms of the two hot calls, multi-GPU plumbing,
 * run-time scene changes, the rows SURVEY.md 8(f) ranks next (text interchange, reverb, material filter), measurement
 * and tools.  A host can ignore all of it and still be correct; it is there for throughput and for the tests —
 *   EXTENDED: fs_context_advice fs_scene_commit_fast fs_scene_commit_progressive fs_scene_refine_pending
 *             fs_scene_refine_wait fs_scene_update_triangles fs_scene_refit
 *             fs_compute_energy_response_async fs_compute_energy_response_batch_async
 *             fs_reconstruct_impulse_response_async fs_reconstruct_impulse_response_batch_async fs_synchronize fs_submit
 *             fs_set_pipelining fs_set_walk_stages fs_set_frames_per_launch
 *             fs_energy_device_ptr fs_energy_handoff fs_shard_range fs_comm_unique_id fs_comm_init fs_comm_attach
 *             fs_comm_enable_oneshot fs_comm_detach fs_comm_info fs_peers_init fs_peers_detach fs_gather_energy fs_gather_energy_async
 *             fs_copy_band_impulse_response fs_set_impulse_response fs_trace_rays
 *             fs_save_array_to_file fs_load_float_array fs_save_impulse_response
 *             fs_reverb_init fs_reverb_process fs_reverb_release fs_apply_material_fd
 *             fs_set_profiling fs_set_profiling_interval fs_get_pipeline_counters fs_get_streams
 * (tests/test_capi_cpu.py checks that every exported symbol is in exactly one of the two lists.)
 * Environment variables (FS_*) are tuning and diagnostic knobs only; all of them are read ONCE — at fs_context_create, at a
 * scene commit (builder knobs) or at the first launch of a kernel family — never per frame.
 */
#ifndef FREQUENSEE_H
#define FREQUENSEE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(FS_BUILDING_LIBRARY) && defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: only this header is exported */
#endif

#define FS_ABI_VERSION 5
#define FS_MAX_BANDS 8
#define FS_NO_MATERIAL 0xFFFFu /* actor without UAcousticGeometryComponent / Material (ARTS.cpp:383) */
#define FS_MAX_DEPTH 64        /* largest explicit depth cap.  depth == 0 means NO cap, like the reference's while (true)
                                * (ARTS.cpp:294): a walk ends when the roulette ends it (without roulette, or with
                                * rr_prob >= 1, depth == 0 means FS_MAX_DEPTH) */

/* status codes (replace the reference's check()/UE_LOG error behaviour) */
enum {
    FS_OK = 0,
    FS_ERR_INVALID_ARGUMENT = 1,
    FS_ERR_NO_DEVICE = 2,     /* no usable HIP device / HIP runtime error at init */
    FS_ERR_HIP = 3,           /* a HIP call failed; fs_last_error() has the text */
    FS_ERR_NOT_COMMITTED = 4, /* scene not committed */
    FS_ERR_BAD_HANDLE = 5,
    FS_ERR_SIZE_MISMATCH = 6, /* UpdateEnergyBuffer's check(Num()==NumBins), FSAC.h:83 */
    FS_ERR_OUT_OF_MEMORY = 7,
    FS_ERR_COMM = 8,          /* multi-GPU: librccl not loadable, an RCCL call failed, or a sharded frame was not reduced */
    FS_ERR_OVERFLOW = 9       /* depth = 0 only: more walks than provisioned outlived FS_MAX_DEPTH steps; the library has grown
                               * its record store — trace the frame again (fs_compute_energy_response does so by itself,
                               * the _async entry points report it at the next fs_synchronize) */
};

/* compat flags: reproduce a reference quirk literally (default 0 = evident intent) */
#define FS_FLAG_FIXED_NORM_1000 1u          /* ARTS.cpp:164 normaliser 1/USED_RAY_COUNT whatever NumRays is */
#define FS_FLAG_FLUSH_BEFORE_RECONSTRUCT 2u /* build-owned: what ARTS.cpp:191 would do if FlushEnergyBuffer zeroed the buffer as its name
                                             * and comment say (FSAC.h:76-79) — the IR becomes all zero.  At HEAD it does NOT:
                                             * TArray::SetNumZeroed only zero-fills elements it ADDS, so after the first call
                                             * both flushes are no-ops; the literal behaviour is FS_FLAG_ACCUMULATE_ENERGY */
#define FS_FLAG_ACCUMULATE_ENERGY 128u      /* HEAD literally: EnergyBuffer.SetNumZeroed(NumBins) (FSAC.h:78) keeps the old values,
                                             * so the deposits of every UpdateSource add to those of all earlier ones (the
                                             * "energy accumulation" the reference's README lists as a known bug).  The frame
                                             * deposits into the buffer the previous frame of this source used, without clearing it
                                             * (single-GPU contexts only; with FS_FLAG_DETERMINISTIC the fixed-point histogram
                                             * accumulates — keep one mode for the whole accumulation) */
#define FS_FLAG_COSINE_SAMPLING 4u          /* cosine-weighted bounce instead of VRandCone(n, 90 deg) */
#define FS_FLAG_ALL_CONNECTIONS 16u         /* row f3, the reference's unfinished draft (Is_NaiveConnections, ARTS.cpp:518-546): connect every
                                             * forward prefix F0..Fi with every backward prefix B0..Bj of a pair (visibility test and
                                             * EvaluatePath as for the end-to-end connection) and combine the (i, j) that give the same
                                             * path length with uniform weights 1/N(i+j); ~(k+1)(m+1) contributions per pair instead of 1 */
#define FS_FLAG_MIS_BALANCE 32u             /* row f3: balance-heuristic weights for the all-connections mode (implies it) — the intent of
                                             * the draft's MISEnergy / getExpectedWeight, ARTS.cpp:548-597 ("the weight considers other
                                             * strategies that could have produced the same path"): weight of the (i, j) that produced a
                                             * path = its sampling density / the sum over all (i', j') of the same path length, with the
                                             * densities the walk uses (1/4pi at the end points, CosTheta/PI at surfaces, ARTS.cpp:306-318)
                                             * in area measure; uniform weight when a segment is degenerate */
#define FS_FLAG_MATERIAL_LOBES 64u          /* row f4: a walk vertex reached by a hit picks ONE of three lobes from the material's
                                             * Absorption / Transmission / Scattering arrays (MAT.h:22-30), split as ApplyMaterialFD does
                                             * per bin (MaterialAcousticProcessor.cpp:51-72: Refl = 1 - alpha, tau clamped to Refl + tau
                                             * <= 1, specular Refl (1 - sigma), diffuse Refl sigma, transmitted tau): diffuse = the
                                             * reference's cone sample, specular = mirror direction, transmitted = straight on from the
                                             * far side; EvaluatePath then uses that lobe's gain (diffuse / pi at a connection vertex)
                                             * instead of Absorption / pi.  The reference's walk is diffuse only ("FIXME assuming
                                             * diffuse", ARTS.cpp:304).  Not combinable with FS_FLAG_MIS_BALANCE. */
#define FS_FLAG_DOUBLE_POSITIONS 256u       /* node positions in double like the reference's FVector (ARTS.h:61): hit point, surface
                                             * offset, subpath end points, the connection ray's direction and length and every
                                             * segment length are computed in double and narrowed where the reference narrows them
                                             * (FVector::Dist(...) / 1000.f assigned to a float, ARTS.cpp:372-373); the line trace itself
                                             * starts from the float-rounded node as before.  Bit-for-bit the oracle's double-position
                                             * build (oracle/Makefile target dpos).  Costs a few % of the walk; such frames are never
                                             * held by fs_set_pipelining.  Not combinable with the lobe / all-connections modes. */
#define FS_FLAG_DETERMINISTIC 8u            /* deposits are summed as 64-bit integers of 2^-40 energy quanta: the
                                             * histogram no longer depends on the order of the atomics, so it is bit-identical
                                             * from run to run and for every split of the pairs over GPUs (sum-reduce the u64
                                             * buffer fs_energy_handoff returns; it is rounded to fp32 once, after the reduce) */

typedef struct fs_context fs_context;
typedef int32_t fs_source; /* handle of one registered UFrequenSeeAudioComponent */

/* Subsystem/component sizing constants (FSAC.h:133-139). Zero fields take the reference value. */
typedef struct fs_config {
    uint32_t struct_size;      /* = sizeof(fs_config) */
    int32_t device;            /* HIP device ordinal */
    int32_t num_bands;         /* B, 1..FS_MAX_BANDS; 1 = the reference (band slot 0 == Absorption[2]) */
    int32_t sample_rate;       /* 48000  FSAC.h:133 */
    int32_t num_channels;      /* 2      FSAC.h:135 */
    float simulated_duration;  /* 1.0 s  FSAC.h:136 */
    float bin_duration;        /* 0.001 s FSAC.h:137 */
    int32_t rank;              /* multi-GPU: this process traces pairs [rank*P/W, (rank+1)*P/W) */
    int32_t world_size;        /* W >= 1 */
    void* stream;              /* optional hipStream_t owned by the caller (e.g. the harness's); NULL = own stream */
} fs_config;

/* Per-update parameters; defaults (fs_params_default) are the constants compiled into the reference. */
typedef struct fs_params {
    uint32_t struct_size;      /* = sizeof(fs_params) */
    uint32_t flags;            /* FS_FLAG_* */
    uint64_t seed;             /* counter-based RNG key (replaces the global rand() behind FMath::FRand) */
    uint32_t num_rays;         /* R = source + listener subpaths per frame over all ranks; pairs P = R/2.
                                  reference: NumRays = USED_RAY_COUNT = 1000 pairs = 2000 (ARTS.h:176) */
    int32_t depth;             /* max segments per subpath, 1..FS_MAX_DEPTH; 0 = unbounded like ARTS.cpp:294 */
    int32_t russian_roulette;  /* 1 = ARTS.cpp:300-301 */
    float rr_prob;             /* 0.9       ARTS.cpp:282 */
    float max_trace_dist;      /* 1e6 cm    ARTS.cpp:284 */
    float surface_offset;      /* 0.1 cm    ARTS.cpp:345 */
    float connect_pullback;    /* 0.1 cm    ARTS.cpp:253 */
    float dist_divisor;        /* 1000      ARTS.cpp:373 */
    float min_seg;             /* 1.0       ARTS.cpp:375 */
    float prob_exponent;       /* 0.1       ARTS.cpp:398 */
    float energy_clamp;        /* 1.0       ARTS.cpp:410 */
    float energy_gain;         /* 10        ARTS.cpp:413 */
    float sound_speed;         /* 343       ARTS.cpp:362 */
    float air_absorption[FS_MAX_BANDS]; /* 0.05 per band, ARTS.cpp:395 */
    int32_t samples_per_bin;   /* 0 = reference's ceil(0.001f*48000) = 49 (FSAC.cpp:324) */
    /* SURVEY A.6-h, HEAD literally: the traces of GeneratePath and ConnectSubpaths query ECC_Pawn as well (ARTS.cpp:243-246,
     * 331-334).  A walk ignores the actor it starts from (AddIgnoredActor, :322-327) but can hit the OTHER end point's
     * collision — and then goes on from there with no material; ConnectSubpaths ignores nothing (:252-254): a connection
     * that starts or ends inside a collision sphere (every connection to B_0 or from F_0) is blocked.  Build-owned engine
     * semantics: an end point's collision is a sphere around its position (ADefaultPawn: 34 cm), a ray that starts
     * inside leaves through the far side.  0 (default) = the end point is a point, nothing collides with it. */
    float listener_radius;     /* cm, 0 = off */
    float source_radius;       /* cm, 0 = off */
} fs_params;

typedef struct fs_stats {
    uint64_t frames;             /* compute_energy_response calls */
    uint64_t rays;               /* subpaths traced by this rank */
    uint64_t pairs;              /* pairs traced by this rank */
    double walk_kernel_ms_sum;    /* HIP-event time on the context's stream, profiling enabled: walk_kernel */
    double walk_kernel_ms_last;
    double connect_kernel_ms_sum; /* connect_kernel */
    double reconstruct_ms_sum;    /* reconstruct_kernel + IR publish copy */
    uint64_t timed_frames;        /* frames contributing to the walk sum */
    uint64_t timed_connects;      /* frames contributing to the connect sum (profiling level 2) */
    uint64_t timed_reconstructs;  /* reconstructs contributing to reconstruct_ms_sum */
    uint32_t bvh_nodes;
    uint32_t triangles;
    uint32_t bvh_stack_need;     /* worst-case traversal stack entries of the committed tree */
    uint32_t bvh_depth;          /* depth of the binary tree before the 4-wide collapse */
    uint64_t scene_bytes;        /* device bytes of BVH + triangles + materials */
    /* work counters kept on the device since the last fs_reset_stats */
    uint64_t segments;           /* walk segments = closest-hit queries the walks TOOK: every walker counts its applied hits and misses and
                                  * leaves the count with its end state; the connect pass, which reads both of a pair's, sums them (a sum
                                  * inside the walk kernel cost the fused launch 20 spilled registers and 3 % of the headline) */
    uint64_t connections_tested; /* any-hit queries: one per pair, or one per (i, j) in all-connections mode */
    uint64_t deposits;           /* unobstructed connections = paths evaluated and deposited */
    /* profiling level 3 only (counting instantiations of the kernels, not for timed frames): records the traversal
     * fetched — 64-B nodes of the 4-wide tree and 48-B triangle records — by the closest-hit queries of the walk and
     * by the any-hit queries of the connections: the kernel's OWN algorithmic bytes (SURVEY.md 8d prices the oracle's
     * BVH2 instead) */
    uint64_t walk_node_fetches, walk_tri_fetches, any_node_fetches, any_tri_fetches;
    uint64_t node_request_insts, node_request_lanes, node_request_distinct;   /* profiling level 3, the dense walk: node-record request
                                  * instructions of its waves, the lanes that took part in them, the distinct 64-B records among those
                                  * lanes — how coherent the requests are (lanes / insts of 64; distinct / lanes: 1 = no two lanes share) */
    uint64_t planned_segments;   /* walk segments as the plan pass predicts them from the RNG stream alone (the roulette does not
                                  * depend on geometry); `segments`, `connections_tested` and `deposits` are what the walk and the
                                  * connect kernels counted as they worked: both must agree (the tests assert it) */
} fs_stats;

/* ---- lifecycle: UAudioRayTracingSubsystem::Initialize/Deinitialize (ARTS.cpp:32-42) ------------- */
void fs_config_default(fs_config* cfg);
void fs_params_default(fs_params* p);
int fs_abi_version(void);
int fs_context_create(const fs_config* cfg, fs_context** out);
int fs_context_destroy(fs_context* ctx);
const char* fs_last_error(const fs_context* ctx); /* replaces UE_LOG warnings; "" if none */
/* Advice of fs_context_create to the host, "" if none — never an error.  Today: GPU_MAX_HW_QUEUES.  The context overlaps
 * the tail of a frame (all-reduce, reconstruct, publish) with the next frame's tracing on two HIP streams; the HIP
 * runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read once when the runtime initialises.  With
 * other streams in the process the two can share a queue and serialise: export GPU_MAX_HW_QUEUES=16 before the
 * process touches HIP.  The library does not set it for the host. */
const char* fs_context_advice(const fs_context* ctx);

/* ---- scene: RegisterGeometry/UnregisterGeometry (ARTS.h:99-100) + UAcousticMaterial (MAT.h:22-33) -- */
/* xyz: [T][3][3] vertices, mat_id: [T] index into the material table or FS_NO_MATERIAL. Caller keeps ownership. */
int fs_scene_set_triangles(fs_context* ctx, const float* xyz, const uint16_t* mat_id, int32_t T);
/* absorption/transmission/scattering: [M][B] (FAcousticBand arrays, MAT.h:22-30); transmission and
 * scattering may be NULL (the BDPT path reads Absorption only, ARTS.cpp:385). */
int fs_scene_set_materials(fs_context* ctx, const float* absorption, const float* transmission,
                           const float* scattering, int32_t M, int32_t B);
int fs_scene_commit(fs_context* ctx); /* builds the flattened BVH (binned SAH, on the host: 18 ms per 100 000 triangles) and uploads it */
/* The same commit with the tree built ON THE DEVICE (Morton codes, radix sort, Karras' binary radix tree, 4-wide
 * collapse, then the refit pass): well under a millisecond for 100 000 triangles, for actors that register or unregister
 * at run time (RegisterGeometry / UnregisterGeometry, ARTS.h:99-100) in a frame that cannot wait.  Results are identical
 * (closest hits do not depend on the tree); a Morton tree costs more node visits per ray than the SAH tree, so call
 * fs_scene_commit again when there is time.  Falls back to fs_scene_commit by itself for empty scenes, sharded contexts
 * with a communicator (rank 0's build is broadcast) and degenerate inputs. */
int fs_scene_commit_fast(fs_context* ctx);
/* Both: the device-built tree NOW (frames trace through it right away) and the host's SAH tree as soon as a background
 * thread has built it from a snapshot of the registered triangles (18 ms per 100 000 triangles) — the next call that
 * traces anything after that swaps it in (held frames finish first, the stream drains, 8 MB of records are uploaded:
 * ~1 ms once).  Results never change, only the tracing speed (the SAH tree traces 1.6x faster).  Triangles moved by
 * fs_scene_update_triangles in the meantime keep their current positions (re-applied + refit after the swap); a new
 * fs_scene_set_triangles / commit abandons the background build.  fs_scene_refine_pending: is a build still outstanding;
 * fs_scene_refine_wait: block until it has finished and swap now.  fs_context_destroy waits for an outstanding build. */
int fs_scene_commit_progressive(fs_context* ctx);
int fs_scene_refine_pending(fs_context* ctx, int32_t* pending);
int fs_scene_refine_wait(fs_context* ctx);
/* Moving geometry without a rebuild (row f4).  The reference's line traces run against the live physics scene and
 * include ECC_WorldDynamic objects (ARTS.cpp:333-336, FSAC.cpp:229-232): a prop that moved is seen by the next
 * frame.  fs_scene_update_triangles overwrites `count` committed triangles starting at input index `first` with new
 * vertex positions (layout of fs_scene_set_triangles; materials and actor ids are kept); fs_scene_refit recomputes
 * the boxes of the acceleration structure bottom-up on the device (same topology).  A pending refit is also run
 * automatically by the next trace.  Results equal those of a fresh fs_scene_commit of the moved geometry. */
int fs_scene_update_triangles(fs_context* ctx, int32_t first, int32_t count, const float* xyz /* [count][3][3] */);
int fs_scene_refit(fs_context* ctx);

/* ---- sources and listener: RegisterSource/UnRegisterSource (ARTS.h:103-104, ARTS.cpp:45-53),
 *      GetActorLocation of the source owner / the player pawn (ARTS.cpp:287) ------------------------- */
int fs_source_create(fs_context* ctx, fs_source* out);
int fs_source_destroy(fs_context* ctx, fs_source src);
int fs_source_set_position(fs_context* ctx, fs_source src, const float xyz[3]);
int fs_listener_set_position(fs_context* ctx, const float xyz[3]);
/* The actor a walk starts from is ignored by that walk's traces (FCollisionQueryParams::AddIgnoredActor, ARTS.cpp:322-327): a
 * source whose own mesh is registered geometry does not trap its walks inside it.  object_id = the actor's id among the
 * ids of fs_scene_set_objects; FS_NO_OBJECT (the default) = the end point belongs to no registered actor.  The walks from
 * the source skip the source's actor, the walks from the listener the listener's; ConnectSubpaths ignores nothing
 * (:252-254).  Takes effect with the next traced frame; such frames are not held by fs_set_pipelining. */
#define FS_NO_OBJECT 0xFFFFFFFFu
int fs_source_set_object(fs_context* ctx, fs_source src, uint32_t object_id);
int fs_listener_set_object(fs_context* ctx, uint32_t object_id);

/* ---- the hot path ------------------------------------------------------------------------------- */
/* ComputeEnergyResponse() == UpdateSource up to the deposit (ARTS.cpp:128-173): GenerateFullPaths
 * (:201-233) -> GeneratePath x2 (:279-355) -> ConnectSubpaths (:235-277) -> EvaluatePath (:360-420)
 * -> FlushEnergyBuffer + AddEnergyAtDelay(delay, gain/P) (:157-173, FSAC.h:76-91).
 * Traces this rank's share of params->num_rays, leaves the band-major energy [B][num_bins] resident
 * on the device and, if energy_out != NULL, copies it to the host (synchronous). */
int fs_compute_energy_response(fs_context* ctx, fs_source src, const fs_params* params, float* energy_out);
/* Same, enqueue only (no host sync). */
int fs_compute_energy_response_async(fs_context* ctx, fs_source src, const fs_params* params);
/* Several sources in ONE traced frame — UpdateSource over ActiveSources (ForceUpdateSources, ARTS.cpp:60-68, :128-195).
 * Every listed source gets exactly the result of its own fs_compute_energy_response_async(ctx, src, params) call (same
 * pairs, same random streams), but the device runs one plan / walk / connect sequence over all of them: many small
 * frames become one large one (8 sources x 131 072 rays: 1.9x the rays/s of eight separate frames).  Afterwards each
 * source is reconstructed as usual.  The all-connections modes fall back to one frame per source.  A source may appear
 * only once in the list. */
int fs_compute_energy_response_batch_async(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* params);
/* Device pointer of the energy buffer [B][num_bins] fp32 the source's CURRENT frame deposits into.  A source
 * owns four such buffers and every fs_compute_energy_response* moves on to the next one, so that the tail of
 * frame f (reduce, reconstruct, publish) overlaps the tracing of frame f+1: query the pointer per frame. */
int fs_energy_device_ptr(fs_context* ctx, fs_source src, void** dptr, size_t* bytes);
/* Multi-GPU hook (one sum all-reduce of [B][1000] fp32 between ARTS.cpp:173 and :192).  Hands
 * the current frame's energy buffer over to the context's tail stream: every deposit enqueued so far completes
 * before anything enqueued on *tail_stream after this call.  The caller issues its collective there
 * (ncclAllReduce(dptr, dptr, B*1000, ncclFloat, ncclSum, comm, (hipStream_t)*tail_stream)) and then calls
 * fs_reconstruct_impulse_response_async, which runs behind it on the same stream — all of it concurrent with
 * the next frame's tracing on the compute stream.  Any of the three out-pointers may be NULL.
 * If the frame was computed with FS_FLAG_DETERMINISTIC, *dptr is the [B][num_bins] uint64 fixed-point histogram
 * and *bytes = 8 * B * num_bins: reduce it with an integer sum (ncclUint64 / ncclSum); the reconstruct converts it. */
int fs_energy_handoff(fs_context* ctx, fs_source src, void** dptr, size_t* bytes, void** tail_stream);

/* ---- multi-GPU: the collective behind the boundary --------------------------------------------------
 * The reference's loop over the pairs (GenerateFullPaths, ARTS.cpp:215-230) carries no state from one pair to the next,
 * so rank r of W traces pairs [P r / W, P (r+1) / W) of every frame (fs_config.rank / world_size, one process per GPU)
 * and the ranks sum their [B][bins] histograms: ONE all-reduce per source and frame (fp32, or uint64 in deterministic
 * mode), issued by the library on the context's tail stream at the end of fs_compute_energy_response*, so that it and
 * the reconstruct behind it overlap the next frame's tracing.  With a communicator attached fs_scene_commit also lets
 * rank 0 alone build the acceleration structure and broadcasts it (nodes, triangle records, refit tables).
 * RCCL is opened at run time ($FS_RCCL_LIB if set, else a librccl the process has already loaded, else the system's).
 *   rank 0:     fs_comm_unique_id(id, FS_COMM_ID_BYTES)  -> ship the 128 bytes to the other ranks by any means
 *   every rank: fs_comm_init(ctx, id, FS_COMM_ID_BYTES)  (collective: ncclCommInitRank(world_size, id, rank))
 * or hand over a communicator the host already owns (fs_comm_attach; not destroyed with the context).
 * A world_size > 1 context without a communicator refuses to reconstruct (FS_ERR_COMM) unless the caller reduced the
 * frame itself behind fs_energy_handoff. */
#define FS_COMM_ID_BYTES 128
int fs_comm_unique_id(void* id_out, size_t bytes);
int fs_comm_init(fs_context* ctx, const void* unique_id, size_t bytes);
int fs_comm_attach(fs_context* ctx, void* nccl_comm /* ncclComm_t */);
/* Optional, collective over the ranks, after fs_comm_init / fs_comm_attach: sum the 32 KB energy buffer in ONE exchange
 * step instead of ncclAllReduce (a ring of that size is latency-bound on xGMI).  Every rank owns a mailbox with one slot per
 * rank in its HBM, mapped into the other ranks' processes through HIP IPC (the 64-byte handles travel over the
 * communicator); a reduce = every rank writes its histogram into its slot of every mailbox and raises a sequence flag,
 * then sums the slots of its own mailbox in rank order (bit-identical on all ranks, fp32 or the deterministic mode's
 * u64).  Same stream (the tail stream), same place in the frame as the all-reduce.  If any rank cannot map a peer, all
 * ranks keep ncclAllReduce and the call returns FS_ERR_COMM.  A peer that stops sending makes the next fs_synchronize
 * return FS_ERR_COMM after a bounded wait. */
int fs_comm_enable_oneshot(fs_context* ctx);
int fs_comm_detach(fs_context* ctx);   /* destroys a communicator made by fs_comm_init; fs_context_destroy calls it */
/* What the attached communicator says about itself (ncclCommCount / ncclCommUserRank asked of RCCL now, not the configured
 * values) and how the energy buffer is summed: *collective = 0 none (no communicator), 1 ncclAllReduce on the tail stream,
 * 2 the one-shot peer-write exchange (fs_comm_enable_oneshot).  Without a communicator: *ranks = 0, *rank = -1.
 * A measurement that reports these proves by itself how many ranks took part in its collective. */
int fs_comm_info(fs_context* ctx, int32_t* ranks, int32_t* rank, int32_t* collective);
/* cfg5 — independent sources, one per GPU ("optional ncclAllGather of 8 x 32 KB so any rank can serve any
 * source's IR").  Nothing of a frame is sharded or reduced there (fs_config.world_size stays 1); a PEER communicator
 * of the processes that each own a source serves one collective only:
 *   every rank: fs_peers_init(ctx, id, FS_COMM_ID_BYTES, rank, world_size)        (id from fs_comm_unique_id on rank 0)
 *   every rank: fs_gather_energy(ctx, my_source, out, world_size * B * bins)      (collective, in rank order)
 * out[r] is the [B][bins] histogram of the source rank r passed (its current frame, behind the deposit — and behind the
 * all-reduce on a sharded context).  To serve a peer's IR: fs_update_energy_buffer(mirror_source, out + r * B * bins, …)
 * and fs_reconstruct_impulse_response(mirror_source) — the same energy gives the same samples on every rank.
 * fs_gather_energy_async leaves the gathered histograms on the device, in tail-stream order (valid until the next gather). */
int fs_peers_init(fs_context* ctx, const void* unique_id, size_t bytes, int32_t rank, int32_t world_size);
int fs_peers_detach(fs_context* ctx);   /* fs_context_destroy calls it */
int fs_gather_energy(fs_context* ctx, fs_source src, float* out /* host [world_size][B][bins] */, int32_t n);
int fs_gather_energy_async(fs_context* ctx, fs_source src, void** dptr, size_t* bytes);
/* The partition rule itself, host-only (no device needed): pairs [*pair_begin, *pair_begin + *pair_count) of a frame of
 * num_rays subpaths belong to `rank` of `world_size`. */
int fs_shard_range(uint32_t num_rays, int32_t rank, int32_t world_size, uint32_t* pair_begin, uint32_t* pair_count);

/* Pipelined frames (off by default).  A frame is three passes in a row — plan, walk, connect — and each leaves wave
 * slots idle that the others could use: the walk's longest waves end in a thin tail, the connect pass is one thin
 * round, the plan pass is short.  fs_set_pipelining(ctx, depth):
 *   depth 1  fs_compute_energy_response_async HOLDS BACK the connect pass of its frame; the next call launches it
 *            together with its own walk as ONE kernel;
 *   depth 2  the walk is held back as well: call f launches {plan of frame f, walk of frame f-1, connect of frame f-2}
 *            as one kernel (two kernel boundaries per frame disappear too).
 * The frames in one launch share nothing but the scene (rotating sets of subpath state, schedule, frame scratch and
 * energy buffers).  A held frame's fs_reconstruct_impulse_response_async is recorded and runs right behind its connect
 * pass.  Everything that observes, synchronises or changes what a held frame needs (fs_synchronize, the blocking
 * variants, energy / stats / scene / communicator calls, fs_submit) lets the held frames finish on their own kernels
 * first, so results never depend on the setting; only WHEN work reaches the GPU does: a producer that streams frames
 * (many sources, offline rendering, bench.py) gains 12-14 % (36 % on 16 384-ray frames), a producer that issues one
 * frame per game tick should end the tick with fs_submit (or leave pipelining off) or the frame's IR is published one
 * or two ticks later (three on a single GPU: there the reconstruct of a held frame is itself a part of the launch after
 * the one that connects it, instead of a kernel on the tail stream; four with the library's collective: the launch after
 * next, behind the all-reduce — FS_FUSED_RECON=0 / FS_FUSED_RECON_COMM=0 keep the reconstructs on the tail stream).  Batched frames are held like any other (they gain little: a frame of several chip-fulls has no
 * thin tail to fill).  Frames with lobes, all-connections modes, FS_FLAG_ACCUMULATE_ENERGY and profiling level >= 2
 * are never held.
 * depth = 0 frames (the reference's uncapped walks, ARTS.cpp:294) are held at depth 2 as STAGED WALKS: the longest walk
 * of a frame is a chain of log(subpaths) / log(1 / rr) dependent bounces (118 at 262 144 subpaths) while 97 % of the
 * walks end within 32, so such a frame alone leaves the chip idle for most of its duration.  Launch s + 1 of the frame
 * walks only steps [bound[s-1], bound[s]) of the walks still alive (a 32-byte continuation record per walk carries them
 * from launch to launch), next to the other stages of the frames around it: every launch holds one frame's worth of
 * work and no dependent chain longer than a stage; the frame's IR is published stages + 2 calls after its own (9 with
 * the default bounds 8, 18, 30, 46, 64, 96 — 16, 36, 64, 96 for launches of two to four frames, fs_set_frames_per_launch —;
 * fs_set_walk_stages changes them, count 0 = do not hold such frames). */
int fs_set_pipelining(fs_context* ctx, int32_t depth);   /* 0 = off, 1, 2 */
int fs_set_walk_stages(fs_context* ctx, const int32_t* bounds /* ascending, 1..511 */, int32_t count /* 0..7 */);
/* Frames per launch (with pipelining on; default 1): a 262 144-ray frame leaves a tenth of an MI355X idle that a launch of
 * two such frames fills (866 -> 965 M rays/s).  With n > 1 fs_compute_energy_response_async lets a plain pipelinable frame
 * WAIT until n of its kind have come — the same fs_params but for the low 32 bits of the seed, any sources, the same
 * source several times — and traces them as ONE batched frame in which every item keeps its own seed, energy buffer and
 * recorded fs_reconstruct_impulse_response_async: results are exactly those of n single frames.  Everything that observes
 * or synchronises (and a frame of another kind, and fs_submit) sends a partial group off first.  The price is latency:
 * the IR of a frame is published up to n - 1 calls later than with n = 1.  A waiting frame is traced with the source and
 * listener positions of ITS call (a moved listener sends the group off: a batched frame has one listener). */
int fs_set_frames_per_launch(fs_context* ctx, int32_t n /* 1..4 */);
int fs_submit(fs_context* ctx);   /* hand everything requested so far to the GPU; does not wait */

/* ReconstructImpulseResponse (FSAC.cpp:320-380, called at ARTS.cpp:192): energy -> per-band IR
 * [B][num_samples] and the num_channels-channel view (both channels identical, FSAC.cpp:331) built
 * from the band-mean energy; publishes the channel view to the host front buffer. */
int fs_reconstruct_impulse_response(fs_context* ctx, fs_source src, const fs_params* params);
int fs_reconstruct_impulse_response_async(fs_context* ctx, fs_source src, const fs_params* params);
/* The tick's reconstructs in one go (UpdateSources loops over ActiveSources, ARTS.cpp:100-126, each UpdateSource ending in
 * ReconstructImpulseResponse :192): exactly the result of fs_reconstruct_impulse_response_async on every listed source, as
 * ONE launch that also writes the published channel views, and one completion event — per source the single call costs a
 * stream wait, a kernel, a copy and three event records (32 sources: 2.8 ms per tick against 0.9 ms). */
int fs_reconstruct_impulse_response_batch_async(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* params);
int fs_synchronize(fs_context* ctx);
/* UpdateSources (ARTS.cpp:100-126) as the game thread runs it: one UpdateSource (:128-195) for every listed source — trace,
 * deposit, reconstruct — and every IR is in its published host buffer when the call returns.  = the batched compute call +
 * the batched reconstruct + fs_synchronize, with the reconstructs riding on the compute stream (nothing else to overlap
 * with when the caller waits); a depth = 0 frame whose records overflowed is traced again like fs_compute_energy_response
 * — the impulse responses (and sequence numbers) such a failed attempt published meanwhile are PROVISIONAL: the retry publishes
 * the complete ones behind them before the call returns (a concurrent reader may see one for a few hundred microseconds). */
int fs_update_sources(fs_context* ctx, const fs_source* sources, int32_t count, const fs_params* params);

/* GetImpulseResponse() (FSAC.h:113): pointer to the PUBLISHED [num_samples] channel buffer — a slot of the source's ring of 8
 * pinned host buffers — valid for the next 7 publishes of the source; lock-free and without a runtime call, for the audio
 * thread (RVB.cpp:136).  The buffer is written by the launch that reconstructs the frame (ReconstructImpulseResponse leaves the IR
 * in the component's own buffer, FSAC.cpp:377-378) and becomes the front when fs_get_impulse_response_sequence (or any producer
 * call) has noticed the launch's announcement. */
int fs_get_impulse_response(fs_context* ctx, fs_source src, int32_t channel, const float** data, int32_t* n);
int fs_copy_impulse_response(fs_context* ctx, fs_source src, int32_t channel, float* out, int32_t n);
/* Number of IRs of this source published so far (0: the zero-initialised buffer of FSAC.cpp:24-28 is in front): the k-th
 * reconstruct / fs_set_impulse_response of a source is publish k, and fs_get_impulse_response returns publish
 * `*completed` or a newer one.  Any thread, no lock; it also notices publishes that completed since the producer's last
 * call into the library (a load of the context's publish word; an event query only for the few publishes that went through the
 * tail stream), which fs_get_impulse_response alone does not.  A consumer that reads it before and after
 * copying the buffer knows that the copy is whole (the pointer stays valid for 7 publishes), and the reverb callback can
 * keep the IR's spectrum while the number stands still instead of transforming the IR every callback (RVB.cpp:188). */
int fs_get_impulse_response_sequence(fs_context* ctx, fs_source src, uint64_t* completed);
int fs_copy_band_impulse_response(fs_context* ctx, fs_source src, int32_t band, float* out, int32_t n);
/* GetImpulseResponse() returns a MUTABLE reference in the reference (FSAC.h:113): consumers may install an IR of their
 * own (the authors' convolver checks used synthetic and downloaded IRs: GenerateDummyImpulseResponse FSAC.cpp:408-452,
 * a delta at samples 0 and N-1; LoadFloatArray :454-490).  Replaces the source's published IR (all channels and
 * bands) with ir[num_samples]; the reverb callback and fs_get_impulse_response see it until the next reconstruct. */
int fs_set_impulse_response(fs_context* ctx, fs_source src, const float* ir, int32_t n);

/* ---- energy-buffer helpers of the component (FSAC.h:72-91), so a UE shim or a test can drive the
 *      same sequence as ARTS.cpp:157-192 ------------------------------------------------------------ */
int fs_get_energy_buffer(fs_context* ctx, fs_source src, float* out, int32_t n);          /* EnergyBuffer */
int fs_flush_energy_buffer(fs_context* ctx, fs_source src);                                /* FSAC.h:76-79 */
int fs_add_energy_at_delay(fs_context* ctx, fs_source src, int32_t band, float delay_seconds,
                           float energy);                                                  /* FSAC.h:87-91 */
int fs_update_energy_buffer(fs_context* ctx, fs_source src, const float* values, int32_t n); /* FSAC.h:81-85 */
int fs_num_bins(const fs_context* ctx);    /* FSAC.h:137 */
int fs_num_samples(const fs_context* ctx); /* FSAC.h:138 */

/* ---- legacy per-frame forward tracer (row a9): UpdateSound (FSAC.cpp:283-306) = RaycastsPerTick specular
 *      chains CastAudioRay (:132-207), each bounce firing a listener-directed CastDirectAudioRay (:209-280),
 *      then OcclusionAttenuation (:295-299), the only live output at HEAD (OCC.cpp:43 reads it).
 *      Engine semantics owned by the build: an actor = an object id per triangle; the player pawn = a sphere. */
typedef struct fs_sound_params {
    uint32_t struct_size;        /* = sizeof(fs_sound_params) */
    int32_t raycasts_per_tick;   /* 1500  FSAC.h:39 */
    uint64_t seed;
    int32_t raycast_bounces;     /* 10    FSAC.h:42 */
    float raycast_distance;      /* 5000  FSAC.h:45 */
    float simulated_duration;    /* 1.0   FSAC.h:136 */
    float listener_radius;       /* pawn collision sphere radius, cm */
} fs_sound_params;

typedef struct fs_sound_result {
    float total_energy;          /* TotalEnergy / RaycastsPerTick, FSAC.cpp:294 (computed then dropped at HEAD) */
    float occlusion_attenuation; /* FSAC.cpp:299 */
    float direct_energy_sum;     /* sum of the per-bounce CastDirectAudioRay results (what Accumulate would get) */
    uint32_t rays_reaching_listener;
    uint32_t direct_hits;
    uint64_t traces;             /* line traces issued */
} fs_sound_result;

void fs_sound_params_default(fs_sound_params* p);
/* actor id per triangle [T] (AActor the collision belongs to); NULL = every triangle its own actor.
 * Takes effect at the next fs_scene_commit. */
int fs_scene_set_objects(fs_context* ctx, const uint32_t* object_id, int32_t T);
int fs_update_sound(fs_context* ctx, fs_source src, const fs_sound_params* p, fs_sound_result* out);
/* GetOcclusionAttenuation() FSAC.h:112: value of the last fs_update_sound (1.0 before the first) */
int fs_get_occlusion_attenuation(fs_context* ctx, fs_source src, float* out);

/* ---- engine line trace the BVH kernel replaces (UWorld::LineTraceSingleByObjectType; call sites
 *      ARTS.cpp:252-254 any-hit, :340-342 closest-hit). Batch query, host arrays. ------------------- */
/* origins/dirs: [N][3] (dirs unit), tmax: [N]; out: hit[N] (0/1), t[N], tri[N] (input triangle index or -1),
 * normal[N][3] (unit, facing the ray origin side). any_hit == 1: only hit[] is written; any_hit == 2 .. 8: the closest
 * hit again, found by the cooperative traversal the small frames' walks use (2, 3, 4: 1, 2, 4 rays per wave — a group of
 * 64, 32, 16 lanes searches each ray — with every node record fetched from memory; 5, 6, 7: the same with the top of the
 * tree resident in LDS; 8: four rays per wave, as much of the tree resident as fits) — same answers, for tests and tools. */
int fs_trace_rays(fs_context* ctx, const float* origins, const float* dirs, const float* tmax, int32_t N,
                  int32_t any_hit, int32_t* hit, float* t, int32_t* tri, float* normal);

/* ---- row f1: the reference's only interchange format (one float per line) --------------------------------
 *      SaveArrayToFile FSAC.cpp:492-505 (FString::SanitizeFloat per value, joined by '\n'),
 *      LoadFloatArray FSAC.cpp:454-490 (split on '\n' culling empty lines, FCString::Atof per line).
 *      Host-side utilities; they need no context and no device. */
int fs_save_array_to_file(const float* data, int32_t n, const char* path);
/* reads at most cap values into out (out may be NULL to count); *n_out = number of lines parsed */
int fs_load_float_array(const char* path, float* out, int32_t cap, int32_t* n_out);
/* "saved_ir.txt": SaveArrayToFile(ImpulseBuffer[channel]) FSAC.cpp:302 */
int fs_save_impulse_response(fs_context* ctx, fs_source src, int32_t channel, const char* path);

/* ---- row f2: the reverb plugin's per-callback convolution (audio render thread) --------------------------
 *      FFrequenSeeAudioReverbPlugin::Initialize/OnInitSource (RVB.cpp:74-109), ProcessSourceAudio (:118-170),
 *      ConvolveFFT (:172-213), FCircularAudioBuffer (CircularBuffer.cpp).  RVB.cpp =
 *      Private/FrequenSeeAudioReverbPlugin.cpp.  The source's most recent impulse response is used on the device. */
#define FS_REVERB_LITERAL_TAIL 1u /* RVB.cpp:147-148 literally: the interleaved buffer's first `frame` floats feed both channels */
int fs_reverb_init(fs_context* ctx, fs_source src, int32_t frame_size /* BufferLength, 1024 */);
/* in/out: interleaved stereo [frame_size * 2]; apply_reverb == 0 is the bApplyReverb bypass (memcpy, RVB.cpp:128-132).
 * Audio-thread safe: runs on the context's reverb stream, reads the source's newest device-resident IR behind the
 * reconstruct that wrote it (events, exchanged under a per-source mutex held only while work is enqueued) and waits for
 * its own stream only.  fs_reverb_init / fs_reverb_release of a source must not run concurrently with its callback. */
int fs_reverb_process(fs_context* ctx, fs_source src, const float* in, float* out, int32_t apply_reverb, uint32_t flags);
int fs_reverb_release(fs_context* ctx, fs_source src); /* OnReleaseSource: ClearBuffers */

/* ---- row f4: frequency-dependent material response of one audio block ------------------------------------
 *      UMaterialAcousticProcessor::ApplyMaterialFD (Private/MaterialAcousticProcessor.cpp:8-107, MAP.cpp):
 *      N = next power of two >= L (:15-16); forward real FFT of the zero-padded block (:29-47); per bin
 *      Refl = 1 - absorption, transmission clamped so Refl + tau <= 1, specular = Refl*(1 - scattering),
 *      diffuse = Refl*scattering, transmitted = tau (:51-72); three inverse FFTs scaled by 1/N (:75-92).
 *      The three response curves (FMaterialAcousticFD, MaterialAcousticProcessor.h:24-37) must each hold
 *      num_responses == N/2 + 1 values, otherwise FS_ERR_SIZE_MISMATCH (the reference logs the error and
 *      returns empty outputs, :20-26).  in and the three outputs are host arrays of L floats. */
int fs_apply_material_fd(fs_context* ctx, const float* in, int32_t L, const float* absorption, const float* transmission,
                         const float* scattering, int32_t num_responses, float* specular, float* diffuse,
                         float* transmitted);

/* ---- measurement --------------------------------------------------------------------------------- */
/* HIP events on the context's stream: 0 = off, 1 = around the dominant (walk) kernel only, 2 = every kernel,
 * 3 = level 2 + the kernels count the node / triangle records they fetch (slower: not for timed frames) */
int fs_set_profiling(fs_context* ctx, int32_t level);
/* level 1 only: put the event pair around every n-th frame (default 1 = every frame).  An event pair costs a frame a few
 * microseconds of queue bubbles; sampling keeps a live measurement inside a timed region without paying that per frame. */
int fs_set_profiling_interval(fs_context* ctx, int32_t frames);
int fs_get_stats(fs_context* ctx, fs_stats* out);
int fs_reset_stats(fs_context* ctx);
/* What the producer's side of a stream of frames did since the context was created (host counters, no device access, any time):
 * where a timed region can lose time that is not kernel time.  In the steady state of a single-GPU stream of pipelined frames
 * tail_stream_ops, stream_waits_enqueued and publishes_by_event stay constant: every launch goes onto the compute stream and
 * publishes its impulse responses by itself (the reference's contract: the IR is in the component's buffer when
 * ReconstructImpulseResponse returns, FSAC.cpp:377-378 — here: when the launch's id appears in a pinned host word). */
typedef struct fs_pipeline_counters {
    uint32_t struct_size;            /* = sizeof(fs_pipeline_counters), set by the caller */
    uint32_t reserved;
    uint64_t fused_launches;         /* launches that carry parts of several pipelined frames */
    uint64_t flushes, flushed_frames;/* held frames that had to finish on kernels of their own (fs_submit, fs_synchronize, an observer) */
    uint64_t host_waits, host_wait_us;   /* the producer waited for a publish: the IR ring's back-pressure, its only throttle */
    uint64_t stream_waits_enqueued;  /* waits for another stream's event put on the compute stream ... */
    uint64_t stream_waits_skipped;   /* ... and those not needed because the event had completed */
    uint64_t tail_stream_ops;        /* commands enqueued on the tail stream: hand-overs, reconstruct kernels, copies, event records, collectives */
    uint64_t owed_on_tail;           /* reconstructs that missed their fused launch and ran on kernels of their own */
    uint64_t publishes_by_word;      /* impulse responses published by the launch itself (compute stream, pinned host word) */
    uint64_t publishes_by_event;     /* ... through an event on the tail stream (a copy command or a batch kernel there) */
    uint64_t lane_launches;          /* first-stage launches of waited-for uncapped frames that carried a long-walk lane (cooperative waves for the longest walks) */
} fs_pipeline_counters;
int fs_get_pipeline_counters(fs_context* ctx, fs_pipeline_counters* out);
/* The context's HIP streams as hipStream_t values: the compute stream (fs_config.stream if the caller gave one, else the
 * context's own) and the tail stream (what fs_energy_handoff returns too).  For measurement — events recorded on the stream
 * the launches really go to — and for hosts that order work of their own behind a frame.  Either pointer may be NULL. */
int fs_get_streams(fs_context* ctx, void** compute_stream, void** tail_stream);

#if defined(FS_BUILDING_LIBRARY) && defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FREQUENSEE_H */
