// fs_internal.hpp — internal types shared by the host library and the HIP kernels (gfx950).
// Not part of the public boundary (that is include/frequensee.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/frequensee.h"

#if defined(FS_TRAV_STATS) && !defined(FS_EXPERIMENTS)
#define FS_EXPERIMENTS   // the traversal statistics are recorded by the one-subpath-per-lane walk kernel
#endif

namespace fs {

// ---- device data layout (HBM) --------------------------------------------------------------------
// 4-wide BVH node, 64 B = half a 128-B cache line = four 16-byte loads:
//   q0 = (origin.x origin.y origin.z, step.x)  step: the power-of-two grid step of each axis, as a float (the
//                                             traversal multiplies it by the ray's reciprocal direction right away)
//   q1 = (lox4 loy4 loz4 hix4)  q2 = (hiy4 hiz4 step.y step.z)   byte c of each word = child c's plane on that
//                                             grid: box = origin + q * step, rounded outwards
//   q3 = child[4] as int bits; child >= 0: inner node index; child < 0: leaf,
//        ~child = first_tri * 4 + (count - 1), count in 1..4 (the builder makes 1..2); empty slot: lo = 255 > hi = 0 (never hit).
struct alignas(16) NodeQ4 {
    float ox, oy, oz;
    float sx;
    uint32_t lox, loy, loz, hix;
    uint32_t hiy, hiz;
    float sy, sz;
    int32_t child[4];
};
static_assert(sizeof(NodeQ4) == 64, "NodeQ4 must be 64 B");

// Triangle record, 64 B (the intersection test reads the first 48 B, the hit shading the last 16 B),
// stored in leaf order:
//   a = (v0.x v0.y v0.z e1.x)  b = (e1.y e1.z e2.x e2.y)  c = (e2.z, material, input index, object id)
//   d = (unit geometric normal of cross(e1, e2), -)
struct alignas(16) Tri64 {
    float4 a, b, c, d;
};
static_assert(sizeof(Tri64) == 64, "Tri64 must be 64 B");

// What the kernels read: the first 48 B of every Tri64 at a 48-B stride (the intersection test never sees the normal:
// without it the records the traversal keeps pulling through L1/L2 take a quarter less room) and the unit normals in
// an array of their own (read once per walk step, at the hit).  Tri64 stays the authoring format of the builders, the
// refit and fs_scene_update_triangles; launch_pack_triangles / update_tris_kernel derive the kernels' view from it.
struct alignas(16) Tri48 {
    float4 a, b, c;
};
static_assert(sizeof(Tri48) == 48, "Tri48 must be 48 B");

// What the cooperative traversal of small frames reads (fs_device.hpp: trav_coop): one 16-byte record per CHILD — a lane
// tests one child box per step and fetches exactly its own record, with one ds_read_b128 (the first DeviceScene.lds_nodes
// nodes of the breadth-first array are staged in LDS by every workgroup) or one global_load_dwordx4 — derived from the
// NodeQ4 array after every commit and refit (fs_refit.hip: coop_nodes_kernel):
//   lo.x lo.y | lo.z hi.x | hi.y hi.z   the child's box as fp16, rounded outwards (conservative; an empty slot is the
//                                       inverted box +inf / -inf), ref = NodeQ4.child[c]
// Record 4 * node + child (an intermediate array), then folded TWO LEVELS AT A TIME into 16-wide nodes — record
// 16 * dense node + slot, 256 B per node, inner references = dense indices of the even levels (fs_refit.hip:
// coop16_kernel): a query takes half as many steps.  DeviceScene.coop points at the 16-wide array.
struct alignas(16) CoopChild {
    uint32_t lo_xy, loz_hix, hi_yz;
    int32_t ref;
};
static_assert(sizeof(CoopChild) == 16, "CoopChild must be 16 B");

// What a cooperative kernel is given beside the scene: one of the two node arrays of that traversal.
//   wshift 4: the 16-wide nodes (two levels of the 4-wide tree folded together, dense order of the even levels): half the
//             steps per query — what waves of one or two rays walk (a group of 64 / 32 lanes takes 4 / 2 nodes per step);
//   wshift 2: the per-child records of the 4-wide nodes as they are — what waves of FOUR rays walk: a group of 16 lanes
//             takes four of those nodes per step but only one 16-wide one (old_mine, 128 sources: 1.76 against 1.99 ms per tick).
// lds_nodes: nodes [0, lds_nodes) are resident in the workgroup's LDS (set by the launcher: what fits).
// stack_need: worst-case pending entries of a one-node-at-a-time descent (what trav_coop keeps free before it widens).
struct CoopView {
    const CoopChild* rec;     // [nodes << wshift]
    int32_t lds_nodes, nodes;
    int16_t wshift, stack_need;
};
struct CoopInfo { CoopView wide16{}, wide4{}; };
constexpr float kCoopMaxCoordinate = 16384.0f;   // largest |coordinate| (UE units) of a scene whose small frames use the cooperative traversal (fp16 ulp 8 there)

// Per-lane traversal stack in LDS: DeviceScene.stack_rows rows of kBlock ints, sized at run time from the committed
// tree (its worst-case need + kStackSlack), passed as dynamic shared memory.  A node visit that pushes writes its
// three candidate entries at sp .. sp + hits - 2 — never above the new top — so the worst-case need itself would do;
// one spare row is kept.  LDS size matters beyond occupancy: the walk workgroups of frame f+1 share the CUs with the
// reconstruct workgroups of frame f (tail stream, 4 KB of LDS each), and 3 x walk + 1 x reconstruct must fit the CU's
// 160 KB or the third walk workgroup waits (measured: 52 224 B of dynamic LDS 0.356 ms, 51 200 B 0.315 ms).
constexpr int kStackDepth = 64;   // largest worst-case stack need the builder accepts before it rebuilds shallower
constexpr int kStackSlack = 1;
constexpr int kBlock = 256;       // 4 waves of 64 lanes
// LDS-privatised part of the [bands][bins] energy histogram in the connect kernels.  With the reference's distance
// scale (cm / 1000, ARTS.cpp:373) bin = path length in metres / 3.43: 256 bins cover 878 m, and a 262 144-ray frame
// at cfg3 touches bins 0..60.  All 1000 bins cost 32 KB per workgroup at 8 bands — the 8 KB window keeps three
// connect workgroups on a CU instead of two.  FS_HIST_WINDOW overrides it (tests exercise the far path with 16).
constexpr int kHistWindow = 256;
constexpr int kUnboundedDepth = 1 << 24;   // the weights' depth cap when the walk has none: every strategy exists
constexpr int kOverLevels = 448;            // second-tier walk steps of depth = 0 frames (64 + 448 = 512 steps)

struct DeviceScene {
    const NodeQ4* nodes;
    const Tri48* tris;        // traversal records, leaf order
    const float4* tri_nrm;    // unit geometric normals, leaf order
    const float* absorption;  // [M][B]
    const float* lobe_gain;   // [M][3][B] diffuse / specular / transmitted gains (FS_FLAG_MATERIAL_LOBES)
    const float* lobe_prob;   // [M][3] probability of each lobe
    int32_t num_nodes;        // 0 = empty scene
    int32_t num_tris;
    int32_t num_materials;
    int32_t stack_rows;       // LDS rows of the traversal stack area per lane (see kStackDepth); with a deep store the last
                              //   one holds the lanes' deep counts
    int32_t stack_limit;      // rows a lane's pending entries may use in LDS: stack_rows without a deep store (the tree's
                              //   worst case + 1, nothing can overflow), else kStackRowsCap (more goes to `deep`)
    // Deep store (only for trees whose worst case exceeds kStackRowsCap rows): [deep_rows][deep_lanes] ints in HBM, column
    // blockIdx.x * kBlock + threadIdx.x.  A lane whose LDS rows are full moves its oldest entries there in chunks and
    // takes them back when its LDS rows run low (fs_device.hpp: trav_maintain).  The worst case assumes
    // that a ray hits every child box at every level of the deepest path; rays that need more than kStackRowsCap rows
    // are rare enough that the trips to HBM do not show, and the bounded LDS stack lets four workgroups share a CU.
    uint32_t stack_attn;      // stack_limit - 4 with a deep store, else 0x7FFFFFFF: (unsigned)(sp + sb) >= stack_attn sends a
                              //   lane to trav_maintain at the top of a step (fs_device.hpp)
    int32_t stack_worst;      // rows of a stack that cannot overflow (worst case + 1) if a workgroup may have that much LDS, else
                              //   0: what the wide flavour of the frame kernel runs with (fs_frame.hip)
    const struct CoopInfo* coop_info;   // host bookkeeping: the cooperative traversal's node arrays (CoopView below); unused on the device
    int32_t* deep;
    uint32_t deep_lanes;
    struct DeepStore* deep_owner;   // host bookkeeping (grows the store when a launch has more lanes); unused on the device
};

// per-update constants handed to the kernels by value
struct KParams {
    uint32_t seed_lo, seed_hi;
    uint32_t pair_begin;   // first global pair index of this rank
    uint32_t num_local;    // pairs traced by this rank (all sources of a batched frame together)
    uint32_t pairs_per_source;   // == num_local for one source; a batched frame lays its sources' pairs end to end:
                                 // pair li belongs to source li / pairs_per_source, RNG pair index li % pairs_per_source
    const float* src_table;      // [sources][4] source positions (+ the source's actor id as bits) of a batched frame (device), else null (kp.src)
    uint32_t src_object, lis_object;   // the actors the walks ignore (fs_source_set_object / fs_listener_set_object), FS_NO_OBJECT: none
    uint32_t item_seed[4];       // grouped frames (fs_set_frames_per_launch): the low seed word of each item of the batched
    int32_t item_seeds;          //   frame (item_seeds of them; 0 = every item uses seed_lo, the batch of one call)
    int32_t depth;         // max segments per subpath: 1..FS_MAX_DEPTH, or main_levels + over_levels for depth = 0 (a bound the
                           //   roulette practically never reaches: 0.9^512 ~ 4e-24)
    int32_t mis_depth;     // depth cap D of the all-connections weights (kUnboundedDepth for depth = 0)
    int16_t russian_roulette;
    int16_t cosine;
    int16_t ignore_on;     // some walk of the frame ignores an actor: the EXT instantiations run (16 bits each: the fused launch's 4 KB of arguments are full)
    int32_t lobes;         // 1 = FS_FLAG_MATERIAL_LOBES: the walk picks a specular / diffuse / transmitted lobe per vertex
    int16_t mis;           // all-connections mode: 1 = balance-heuristic weights, 0 = uniform
    int16_t plan_coop;     // the plan pass evaluates 64 bounces of a subpath at once, a wave per 8 subpaths (plan_coop_body): frames of <= 32 768 subpaths, and uncapped walks
                           //   that are waited for (the host decides: frame_describe)
    float rr_prob, max_trace_dist, surface_offset, connect_pullback;
    float stage_margin;    // staged walks: lanes a later stage is given = stage_margin x the expected survivors + 1024 (1.3; doubled
                           //   by the overflow retry if a frame ever had more)
    float dist_divisor, min_seg, prob_exponent, energy_clamp, energy_gain, sound_speed;
    float norm;            // 1/P or 1/1000 (ARTS.cpp:164)
    float air[FS_MAX_BANDS];
    float src[3], lis[3];
    int32_t dpos;          // 1 = FS_FLAG_DOUBLE_POSITIONS: node positions, segment and connection lengths in double
    float listener_radius, source_radius;   // collision spheres of the end points, 0 = points
    int32_t count;         // 1 = COUNT instantiations: the kernels also count the records they fetch (fs_set_profiling level 3)
    int32_t num_bins;
    int32_t num_bands;     // bands of the context (the connect kernels are instantiated for 1, 4 and 8; any other count reads this)
    int32_t hist_window;   // the connect kernels privatise bins [0, hist_window) of every band in LDS; deposits beyond go
                           // straight to the energy buffer with global atomics (kHistWindow, or all bins if fewer)
};

#ifndef FS_PLAN_COOP_MAX_UNCAPPED
#define FS_PLAN_COOP_MAX_UNCAPPED (1u << 17)
#endif
constexpr uint32_t kPlanCoopMax = 32768, kPlanCoopMaxUncapped = FS_PLAN_COOP_MAX_UNCAPPED;   // KParams.plan_coop (fs_device.hpp: plan_coop_body)

// legacy forward tracer (UpdateSound) constants and device-side accumulators
struct SoundKParams {
    uint32_t seed_lo, seed_hi;
    int32_t raycasts_per_tick, raycast_bounces;
    float raycast_distance, simulated_duration, listener_radius;
    float src[3], lis[3];
};
struct SoundAccum {
    unsigned long long traces;
    unsigned reaching, direct_hits;
    float direct_energy_sum, occlusion;
};

// What the walk kernels leave for connect_kernel.  A subpath is known by its index g (side-major: [0,n) source side,
// [n,2n) listener side) and by its LAUNCH SLOT: the lane that walks it under the length-sorted schedule.  Everything
// the walk writes is indexed by slot — a wave's lanes write consecutive words — and slot_of[g] leads back to it
// (null = no schedule, slot == g).  total = 2 * num_local.
struct SubpathState {
    float4* end_pos;    // [total] by slot: xyz = last node position, w = last node probability
    uint2* end_misc;    // [total] by slot: x = last node material, y = segments taken
    float2* seg_np;     // [main_levels][total] per walk step, by slot: x = scaled segment length, y = probability of the
                        //   node EvaluatePath pairs with the segment (departure node on the source side, arrival node
                        //   on the listener side)
    uint32_t* seg_mat;  // [main_levels][total] material of that node
    float4* seg_pos;    // [main_levels][total] xyz = position of the node a walk step arrives at; only written (and only
                        //   non-null) in all-connections mode, which connects interior nodes too (row f3)
    float4* seg_nrm;    // [main_levels][total] xyz = normal of that node; only with balance-heuristic weights
    uint32_t* slot_of;  // [total] launch slot of subpath g, written by the walk; null: slot == g
    // Walk steps beyond main_levels exist only for depth = 0 (no cap, ARTS.cpp:294): 0.9^64 of the walks take more than
    // 64 steps, and the schedule puts the longest walks into the lowest slots, so their later records live in a small
    // second tier [over_levels][over_cap] indexed by (step - main_levels, slot).  A record that fits neither tier
    // raises *overflow; the host then grows the tier and traces the frame again (fs_capi.cpp).
    float2* over_np; uint32_t* over_mat; float4* over_pos; float4* over_nrm;
    unsigned* overflow;
    uint32_t over_cap;
    int32_t main_levels, over_levels;
    // Staged walks (pipelined depth = 0 frames): a walk that reaches the last step of its stage leaves a continuation
    // record by slot — cont_a = (position, probability of the current node), cont_b = (normal, material | kCont* flags) —
    // and the next stage (one launch later) resumes from it.  Null for walks that run in one piece.
    float4* cont_a; float4* cont_b;
    // FS_FLAG_DOUBLE_POSITIONS: the last node's position in double [total][3] by slot (end_pos keeps the float rounding)
    double* end_posd;
};
constexpr uint32_t kContHasNormal = 1u << 16, kContArrived = 1u << 17, kContAlive = 1u << 18;   // cont_b.w above the 16-bit material

// steps [begin, end) of a walk; slots_cap = the slots this launch has lanes for (a stage that starts at step begin > 0
// only covers the walks longer than that — the first slots of the length-sorted schedule)
struct WalkStage {
    int32_t begin = 0, end = 1 << 30;
    uint32_t slots_cap = 0xFFFFFFFFu;
};
// The long-walk lane of a waited-for staged frame (fs_capi_frame.cpp): the longest walks — those of
// len steps or more, at most cap of them: the FIRST slots of the length-sorted schedule — walk on cooperative waves of
// their own from step 0 on, with their own stage bounds, beside the launch's other walks.  len = 0: no such lane.
struct WalkLane {
    int32_t len = 0;
    uint32_t cap = 0;
    int32_t begin = 0, end = 1 << 30;   // the lane's stage in this launch
    int32_t mode = 0;                   // kLaneBoth / kLaneOnly / kLaneSkip: which of the two classes of slots this part walks
};
constexpr int32_t kLaneBoth = 0, kLaneOnly = 1, kLaneSkip = 2;
constexpr int32_t kLaneSplit = 3;   // (to launch_walk: ONE launch whose first workgroups are the lane's cooperative waves — kLaneOnly — and whose others walk the rest — kLaneSkip)

// host side of DeviceScene.deep: owned by the context, grown by the launchers (attach_deep) when a grid has more lanes
// than the store has columns.  A replaced buffer stays allocated until the scene is freed: launches already in the
// stream still point at it.
struct DeepStore {
    int32_t* buf = nullptr;
    size_t lanes = 0;
    int rows = 0;                       // 0: the committed tree cannot overflow its LDS rows, no store
    std::vector<int32_t*> retired;
    bool failed = false;                // an allocation failed: the launch that needed it was skipped (sticky; fs_synchronize reports it)
};
constexpr int kStackRowsCap = 21;       // LDS rows for pending entries when the tree's worst case needs more (+ 1 row of deep counts)
constexpr int kDeepChunk = 8;           // entries moved per trip to / from the deep store
constexpr int kMaxReconParts = 256;     // reconstruct parts of a fused launch = items of one table slot (fs_context::kReconTabItems)

// ---- host BVH builder ------------------------------------------------------------------------------
struct HostBVH {
    std::vector<NodeQ4> nodes;
    std::vector<Tri64> tris;   // leaf order
    int max_depth = 0;         // of the binary tree before the 4-wide collapse
    int stack_need = 0;        // worst-case pending traversal-stack entries (<= kStackDepth by construction)
    // refit support (fs_refit.hip): input triangle -> leaf-order position; node range [level_begin[l],
    // level_begin[l+1]) of every tree level (breadth-first layout); box padding used by the build
    std::vector<uint32_t> leaf_pos;
    std::vector<int32_t> level_begin;
    float pad = 0.01f;
};
// xyz [T][3][3], mat [T]; binned SAH BVH2 collapsed to a quantised 4-wide tree, <= 2 triangles per leaf.
// cancel (optional): polled by the builder; once set it stops splitting and leaves `out` empty (background builds of
// fs_scene_commit_progressive that a newer registration made useless)
void build_bvh(const float* xyz, const uint16_t* mat, const uint32_t* object_id, int32_t T, HostBVH& out,
               const std::atomic<bool>* cancel = nullptr);

// ---- kernel launchers (fs_kernels.hip) -----------------------------------------------------------------
struct WalkLaunch {
    int variant;         // 2 = wave work sharing (the only one in a default build); 0 = one subpath per lane (FS_EXPERIMENTS)
    int num_cus;         // compute units of the device
    unsigned* queue_head;  // frame scratch: [0] unused, then plan counts + cursors; zero at launch
    int plan;            // 1 = sort subpaths by their (RNG-determined) length before walking
    uint32_t* perm;      // [depth + 1][total] subpath indices bucketed by planned length
    int rays_per_wave = 64;   // < 64: sparse waves for small frames (variant 2): a wave owns this many subpaths, the other lanes help
    int coop = 1;             // 1: waves of 1, 2 or 4 subpaths search every ray with ALL the lanes of its group (walk_kernel_coop) instead of
                              //    lane-private descents that hand subtrees to idle lanes; 0: the sparse kernel for every rays_per_wave < 64
};
// subpaths per wave the cooperative walk exists for (groups of 64, 32, 16 lanes)
inline bool coop_rays_per_wave(int r) { return r == 1 || r == 2 || r == 4; }
constexpr int kScratchWords = 1 + 2 * (FS_MAX_DEPTH + 1);
// frame scratch allocation: kScratchWords rearmed every frame, then (8-byte aligned) kNumCounters u64 work
// counters that accumulate until fs_reset_stats: walk segments, connections tested, deposits
constexpr int kCounterWord = (kScratchWords + 1) & ~1;
constexpr int kNumCounters = 12;  // walk segments (observed), connections tested, deposits | level 3: walk node / triangle records, any-hit node / triangle records |
                                  // planned segments | level 3: node-record request instructions of the walk, their active lanes, the distinct 64-B records among those | spare
constexpr int kScratchAllocWords = kCounterWord + 2 * kNumCounters;
// plan pass (length-bucketed schedule + FlushEnergyBuffer); returns the bucket array to walk through, or
// nullptr when no plan applies (the caller then clears the energy buffer itself)
// energy / energy_words: buffer the pass zeroes (FlushEnergyBuffer); energy_tab / energy_count: a device table of such
// buffers instead (batched frame).  Returns nullptr — and zeroes nothing — when there is no roulette to plan for.
const uint32_t* launch_plan(const KParams& kp, const WalkLaunch& wl, float* energy, int energy_words,
                            float* const* energy_tab, int energy_count, hipStream_t s);
// Pipelined frames: ONE launch with several parts — walks (or walk stages) of frames planned before, the connect pass
// of an older one (walked before), the plan pass of the newest.  false = no fused form for this shape (lobes, counting
// instantiations, experiment walk variants, nothing to do): the caller launches the kernels one after the other.
constexpr int kMaxWalkParts = 8;   // walk parts of one fused launch = stages of a staged walk in flight
constexpr int kSlotMaskOffsetWords = 16;  // the mask table starts this many 32-bit words behind the ticket cell (one device allocation)
constexpr int kMaxMaskSources = 1024;     // sources (by handle) whose ring slots have zero-block masks; later ones always write every block
struct PublishWord { unsigned* tickets = nullptr; unsigned long long* host_word = nullptr; unsigned long long id = 0; };
struct WalkPart {   // a frame's walks from step stage.begin up to step stage.end
    KParams kp; SubpathState st; WalkLaunch wl;   // wl.queue_head = the frame's scratch set, wl.rays_per_wave
    const uint32_t* perm = nullptr;               // its schedule (nullptr: none)
    WalkStage stage;
};
struct FrameParts {
    int num_walk = 0;           // walk parts, in grid order
    WalkPart walk[kMaxWalkParts];
    bool has_connect = false;   // kpc, stc, energy / fixed, scratch_c (re-armed by the pass), ppw
    KParams kpc; SubpathState stc; float* energy = nullptr; unsigned long long* fixed = nullptr; unsigned* scratch_c = nullptr; int ppw = 64;
    float* const* energy_tab = nullptr; unsigned long long* const* fixed_tab = nullptr;   // batched frame: per-source buffers
    bool has_plan = false;      // kpp, wl_p (plan switch), scratch_p, perm_p (the schedule to write, nullptr: counts only), zero_p / zero_words_p (the flush)
    KParams kpp; WalkLaunch wl_p; unsigned* scratch_p = nullptr; uint32_t* perm_p = nullptr; float* zero_p = nullptr; int zero_words_p = 0;
    float* const* zero_tab_p = nullptr; int zero_count_p = 0;                              // batched frame: the buffers to flush
    // reconstruct parts: ReconstructImpulseResponse of the frames the PREVIOUS launch connected (single GPU; a sharded
    // frame is reduced on the tail stream first and reconstructed there)
    // a table of ReconItem in pinned host memory (fs_context::h_recon_tab: the parts read their item across the bus, as the batch
    // kernel does — the launch's 4 KB of arguments hold no per-item data, so a launch carries as many reconstructs as are owed:
    // cfg5's eight sources per frame no longer fall back to kernels of their own on the tail stream).  ir_bands / ir_mono =
    // nullptr for a frame whose IR is superseded within the launch; host = the pinned ring slot (the publish).
    int num_recon = 0;
    const struct ReconItem* recon_tab = nullptr;
    int recon_B = 0, recon_nb = 0, recon_samples = 0;
    // the publish of those host slots (fs_device.hpp: publish_arrive): the ticket cell, the pinned host word, this launch's id (tickets == nullptr: by an event)
    PublishWord pub;
};
bool launch_frame(int B, const DeviceScene& sc, const FrameParts& f, hipStream_t s);
// does this frame have a plan pass (roulette on, not empty)?  blocks / sort: its grid and whether it writes the schedule
bool plan_shape(const KParams& kp, const WalkLaunch& wl, uint32_t* blocks, bool* sort);
void launch_walk(const DeviceScene& sc, const KParams& kp, const SubpathState& st, const WalkLaunch& wl,
                 const uint32_t* perm, hipStream_t s, const WalkStage& stage = WalkStage(), const WalkLane& lane = WalkLane());
// can a staged frame whose first stage launches like `first` and whose later ones like `late` have a long-walk lane
// (WalkLane)?  (sparse first stage, cooperative later stages, a cooperative view of the tree for one walk per wave)
bool walk_lane_possible(const DeviceScene& sc, const KParams& kp, const WalkLaunch& first, const WalkLaunch& late, const uint32_t* perm);
// lanes a walk stage needs: all subpaths for a stage that starts at step 0, else the expected number of walks longer than
// stage.begin under the roulette (x1.3 + 1024: the count is binomial, the margin is hundreds of standard deviations)
uint32_t walk_stage_slots(const KParams& kp, int begin);
// fixed != nullptr: deterministic mode, deposits go to the [B][bins] u64 fixed-point histogram instead
// energy_tab / fixed_tab: per-source buffers of a batched frame (device arrays of kp.num_local / kp.pairs_per_source
// pointers), null for one source (`energy` / `fixed` are used)
void launch_connect(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                    unsigned long long* fixed, unsigned* queue_head, int pairs_per_wave, float* const* energy_tab,
                    unsigned long long* const* fixed_tab, hipStream_t s);
// row f3: every forward prefix x every backward prefix of each pair (one wave per pair), uniform MIS weights
void launch_connect_all(int B, const DeviceScene& sc, const KParams& kp, const SubpathState& st, float* energy,
                        unsigned long long* fixed, unsigned* queue_head, hipStream_t s);
void launch_fixed_to_energy(const unsigned long long* fixed, float* energy, int words, hipStream_t s);
void launch_reconstruct(const float* energy, int B, int num_bins, int sample_rate, int num_samples, int spb,
                        float* ir_bands, float* ir_mono, hipStream_t s);
// ReconstructImpulseResponse of MANY sources as one launch (fs_reconstruct_impulse_response_batch_async): a table of items in
// pinned host memory (read by the kernel as it stands); host != nullptr: the channel view is also written straight into
// that pinned host buffer (the publish: 16-byte stores, a block's 4 096 samples staged in LDS) — no copy command per source.
// mask (optional): the device word that says which 4 096-sample blocks of the host slot may hold non-zero samples (bit b = block b):
// a block whose samples are all exactly zero is not written across the bus again when the slot's block is known to be zero.
struct ReconItem { const float* energy; float* ir_bands; float* ir_mono; float* host; uint32_t* mask; int32_t spb; int32_t pad; };
// pub.tickets != nullptr: the launch announces its own completion in the context's pinned host word (publish_arrive)
void launch_reconstruct_batch(const ReconItem* table, int count, int B, int num_bins, int num_samples, hipStream_t s, const PublishWord& pub = PublishWord());
void launch_trace_rays(const DeviceScene& sc, const float* o, const float* d, const float* tmax, int N, int any_hit,
                       int32_t* hit, float* t, int32_t* tri, float* normal, hipStream_t s);
// rays_per_wave < 64: sparse waves whose other lanes help with every closest-hit query; 64 = one ray per lane
void launch_update_sound(const DeviceScene& sc, const SoundKParams& sp, SoundAccum* acc, int rays_per_wave, hipStream_t s);
constexpr int kReverbRing = 65536;   // per-channel history ring (floats), matches kRevRing in the kernels
void launch_reverb(const float* ir, int ir_size, float* ring, unsigned head, const float* in, float* cur, float* out,
                   int frame, int literal_tail, hipStream_t s);
void launch_add_energy(float* energy_row, int num_bins, float delay_s, float e, hipStream_t s);
// dynamic LDS the traversal kernels of a frame need for a tree with `stack_rows` stack rows: the larger of the walk
// kernel (stack + work-sharing area) and the connect kernels (stack + [bands][bins] histogram + work-sharing area)
size_t traversal_lds_bytes(int stack_rows, int bands, int num_bins);
int default_hist_window(int bands);   // LDS histogram bins of the connect part when FS_HIST_WINDOW is not set
// fs_oneshot.hip: the sum of a [bands][bins] buffer over the ranks as one peer-write exchange (fs_comm_enable_oneshot).
// A rank's mailbox: kOneShotHeaderBytes of flags ([2 sets][kOneShotMaxRanks] u32 sequence numbers), then
// [2 sets][world] slots of slot_bytes each.  mail[r] = rank r's mailbox as mapped into this process.
constexpr int kOneShotMaxRanks = 16;
constexpr size_t kOneShotHeaderBytes = 256;
struct OneShotView {
    void* mail[kOneShotMaxRanks];
    int32_t world, rank;
    size_t slot_bytes;
};
// buffer: fp32 [words] (or u64 [words], deterministic mode), summed in place over the ranks; set = seq & 1
void launch_oneshot_reduce(const OneShotView& v, void* buffer, int words, bool u64, int set, uint32_t seq, unsigned* err,
                           hipStream_t s);
// row f4 (fs_refit.hip): moving geometry without a rebuild.  xyz = `count` new triangles [count][3][3] on the
// device, written to the leaf-order records through leaf_pos; then one refit launch per tree level, deepest
// first (node_box = scratch [num_nodes][2] float4 holding each node's fp32 bounds).
void launch_update_triangles(Tri64* tris, Tri48* packed, float4* nrm, const uint32_t* leaf_pos, int first, int count,
                             const float* xyz, hipStream_t s);
void launch_pack_triangles(const Tri64* tris, int count, Tri48* packed, float4* nrm, hipStream_t s);
void launch_refit(NodeQ4* nodes, const Tri64* tris, float4* node_box, const int32_t* level_begin, int levels, float pad,
                  hipStream_t s);
void launch_coop_nodes(const NodeQ4* nodes, int n, CoopChild* out, hipStream_t s);   // per-child records of the current 4-wide nodes
// ... folded two levels at a time into 16-wide nodes in the dense order of the even levels (fs_refit.hip)
void launch_coop16(const CoopChild* in, const int32_t* level_begin_dev, const int32_t* dense_dev, int levels, int n16, CoopChild* out, hipStream_t s);
// fs_build.hip: the acceleration structure built on the device (Morton codes, radix sort, Karras' binary radix tree,
// breadth-first collapse to 4-wide nodes); launch_refit then derives the quantised boxes.  DeviceBuildInfo is what the
// host reads back: levels < 0 = the tree is deeper than kMaxBuildLevels or ran out of node space (use the host build).
constexpr int kMaxBuildLevels = 96;
struct DeviceBuildInfo {
    int32_t num_nodes, levels, stack_need, reserved;
    int32_t level_begin[kMaxBuildLevels + 1];
};
size_t device_build_scratch_bytes(int T);
bool launch_device_build(const float* xyz, const uint16_t* mat, const uint32_t* object_id, int T, const float lo[3],
                         const float hi[3], NodeQ4* nodes, Tri64* tris, uint32_t* leaf_pos, void* scratch, size_t scratch_bytes,
                         DeviceBuildInfo* info_dev, hipStream_t s);
// row f4 (fs_fft.hip): ApplyMaterialFD.  x [N] and y [3][N] complex work buffers, W [N/2] twiddles,
// resp [3][N/2+1] = absorption | transmission | scattering, out [3][L] = specular | diffuse | transmitted
constexpr int kFftChunkLog = 11;     // FFT stages with spans below 2^11 points run inside LDS
void launch_apply_material_fd(const float* in, int L, int n, float2* x, float2* y, const float2* W, const float* resp,
                              float* out, hipStream_t s);

}  // namespace fs
