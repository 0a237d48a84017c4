// fs_capi_comm.cpp — multi-GPU behind the C ABI.
#include "fs_context.hpp"

// ---- multi-GPU: RCCL over xGMI behind the C ABI ---------------------------------------------------
// The pairs of a frame are sharded over world_size ranks (one process per GPU); what the ranks exchange is
//   * one sum all-reduce of the [B][bins] energy histogram per source and frame (fp32, or the u64 fixed-point
//     histogram of FS_FLAG_DETERMINISTIC), on the tail stream, concurrent with the next frame's tracing;
//   * one broadcast of the acceleration structure at fs_scene_commit (rank 0 builds it, the others receive nodes,
//     triangle records and the refit tables).
// librccl is opened at run time — an already loaded copy first (a host process that uses torch.distributed has its
// own), then FS_RCCL_LIB, then the system's — so single-GPU users need no RCCL at all.
namespace fsi {

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

// Sum the source's current energy buffer over the ranks, on the tail stream, behind everything the compute stream
// has enqueued so far.  No-op without a communicator or when the frame has been summed already.
int reduce_energy(fs_context* ctx, Source* s) {
    if (!ctx->comm || s->reduced) return FS_OK;
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
    if (!s->tail_ordered) FS_HIP(ctx, handoff_energy(ctx, s));   // (a later item of the launch the tail already waits behind: no second event)
    const size_t words = (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins;
    if (ctx->oneshot.broken)
        return ctx->fail(FS_ERR_COMM, "one-shot reduce: a peer did not arrive within the time limit of an earlier frame — the mailboxes are no "
                                      "longer in step; detach the communicator (fs_comm_detach) and attach it again");
    if (ctx->oneshot.on) {   // one exchange step through the peers' mailboxes (fs_comm_enable_oneshot)
        const uint32_t seq = ++ctx->oneshot.seq;
        launch_oneshot_reduce(ctx->oneshot.view, s->cur_fixed ? (void*)s->d_fixed[s->cur] : (void*)s->energy(), (int)words, s->cur_fixed,
                              (int)(seq & 1u), seq, ctx->oneshot.d_err, ctx->copy_stream);
        FS_HIP(ctx, hipGetLastError());
        if (s->cur_fixed) launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), (int)words, ctx->copy_stream);
    } else if (s->cur_fixed) {   // deterministic mode: integer sum of the fixed-point histogram, rounded to fp32 once, behind it
        FS_NCCL(ctx, a->AllReduce(s->d_fixed[s->cur], s->d_fixed[s->cur], words, ncclUint64, ncclSum, ctx->comm, ctx->copy_stream));
        launch_fixed_to_energy(s->d_fixed[s->cur], s->energy(), (int)words, ctx->copy_stream);
    } else
        FS_NCCL(ctx, a->AllReduce(s->energy(), s->energy(), words, ncclFloat32, ncclSum, ctx->comm, ctx->copy_stream));
    FS_HIP(ctx, hipEventRecord(s->ev_red[s->cur], ctx->copy_stream));
    ctx->dbg.tail_ops += 2;   // the collective and its event
    s->red_recorded[s->cur] = true;
    s->reduced = true;
    return FS_OK;
}

// unmap the peers' mailboxes and free this rank's (fs_comm_detach; a no-op when the one-shot reduce was never enabled)
void oneshot_release(fs_context* ctx) {
    fs_context::OneShot& o = ctx->oneshot;
    if (!o.own_mail && !o.d_err) return;
    for (int r = 0; r < o.view.world; ++r)
        if (r != o.view.rank && o.view.mail[r]) (void)hipIpcCloseMemHandle(o.view.mail[r]);
    if (o.own_mail) (void)hipFree(o.own_mail);
    if (o.d_err) (void)hipFree(o.d_err);
    o = fs_context::OneShot{};
}

// did a one-shot sum give up waiting for a peer?  (called where the tail stream has been synchronised)
int oneshot_check(fs_context* ctx) {
    if (!ctx->oneshot.on || !ctx->oneshot.d_err) return FS_OK;
    unsigned e = 0;
    FS_HIP(ctx, hipMemcpy(&e, ctx->oneshot.d_err, sizeof(e), hipMemcpyDeviceToHost));
    if (!e) return FS_OK;
    FS_HIP(ctx, hipMemset(ctx->oneshot.d_err, 0, sizeof(e)));
    ctx->oneshot.broken = true;   // the mailbox sets are no longer in step with the peers': every later reduce is refused (reduce_energy)
    return ctx->fail(FS_ERR_COMM, "one-shot reduce: a rank's contribution did not arrive within the time limit (FS_ONESHOT_TIMEOUT_MS, 10 s): "
                                  "the frame's sum is incomplete and the communicator must be detached and attached again");
}

}  // namespace fsi

extern "C" {

int fs_comm_enable_oneshot(fs_context* ctx) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    if (!ctx->device_ok) return ctx->fail(FS_ERR_NO_DEVICE, "no HIP device available (no CPU fallback)");
    if (!ctx->comm) return ctx->fail(FS_ERR_COMM, "fs_comm_enable_oneshot needs a communicator (fs_comm_init / fs_comm_attach) first");
    if (ctx->oneshot.on) return FS_OK;
    const int W = ctx->cfg.world_size, me = ctx->cfg.rank;
    if (W > kOneShotMaxRanks) return ctx->fail(FS_ERR_INVALID_ARGUMENT, "one-shot reduce: at most 16 ranks");
    RcclApi* a = rccl();
    if (!a) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
    FS_FLUSH(ctx);
    FS_HIP(ctx, hipSetDevice(ctx->cfg.device));
    FS_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    fs_context::OneShot& o = ctx->oneshot;
    // Every step below is collective: a rank that fails locally still takes part in the exchanges and the ranks agree on
    // the outcome at the end (MIN over an "ok" word) — either all of them use the mailboxes from now on or none does.
    int ok = 1;
    std::string why;                                   // this rank's first local failure, for the error text
    auto hip_step = [&](hipError_t err, const char* what) {
        if (err != hipSuccess && ok) { ok = 0; why = std::string(what) + ": " + hipGetErrorString(err); }
        return err == hipSuccess;
    };
    const size_t slot = (sizeof(unsigned long long) * (size_t)ctx->cfg.num_bands * (size_t)ctx->num_bins + 255) & ~(size_t)255;   // room for the u64 histogram
    const size_t bytes = kOneShotHeaderBytes + 2 * (size_t)W * slot;
    hipIpcMemHandle_t mine{};
    // The mailbox is written by other devices while this one polls it: fine-grained memory (no stale L2 lines on either
    // side) where the runtime can share such an allocation; plain device memory otherwise (the accesses are system-scope
    // atomics either way).
    // ... and it MUST be fine-grained: peer writes over xGMI into coarse-grained memory are not guaranteed visible to a kernel
    // that polls it on the owner's device (its L2 may hold stale lines), system-scope atomics or not.  A rank that cannot
    // get (or export) such an allocation votes against the one-shot path and every rank keeps ncclAllReduce.
    // (FS_ONESHOT_COARSE_OK=1: plain device memory is accepted — ranks that share ONE device, i.e. tests, see one L2.)
    bool fine = hipExtMallocWithFlags(&o.own_mail, bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (fine && hipIpcGetMemHandle(&mine, o.own_mail) != hipSuccess) { (void)hipFree(o.own_mail); o.own_mail = nullptr; fine = false; }
    (void)hipGetLastError();
    static const bool coarse_ok = std::getenv("FS_ONESHOT_COARSE_OK") && std::atoi(std::getenv("FS_ONESHOT_COARSE_OK")) != 0;
    if (!fine && !coarse_ok) { ok = 0; why = "no fine-grained device memory that HIP IPC can share for the mailbox"; }
    if (ok && (fine || hip_step(hipMalloc(&o.own_mail, bytes), "hipMalloc(mailbox)")) && hip_step(hipMemset(o.own_mail, 0, bytes), "hipMemset(mailbox)") &&
        hip_step(hipMalloc((void**)&o.d_err, sizeof(unsigned)), "hipMalloc(err)") && hip_step(hipMemset(o.d_err, 0, sizeof(unsigned)), "hipMemset(err)"))
        (void)hip_step(hipIpcGetMemHandle(&mine, o.own_mail), "hipIpcGetMemHandle");
    (void)hipGetLastError();
    // all-gather of the 64-byte IPC handles over the communicator, then map every peer's mailbox.  The staging buffer was
    // allocated with the context (fs_context::d_comm_stage): a rank that is out of memory NOW still takes part in both
    // collectives below — returning before them would leave the other ranks blocked in the all-gather for ever.
    char* d_h = ctx->d_comm_stage;
    std::vector<hipIpcMemHandle_t> all((size_t)W);
    hipError_t e = hipMemcpyAsync(d_h + sizeof(mine) * (size_t)W, &mine, sizeof(mine), hipMemcpyHostToDevice, ctx->copy_stream);
    ncclResult_t nr = a->AllGather(d_h + sizeof(mine) * (size_t)W, d_h, sizeof(mine), ncclUint8, ctx->comm, ctx->copy_stream);
    if (e == hipSuccess) e = hipMemcpyAsync(all.data(), d_h, sizeof(mine) * (size_t)W, hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
    if (nr != ncclSuccess && ok) { ok = 0; why = "ncclAllGather(IPC handles) failed"; }
    (void)hip_step(e, "handle exchange");
    o.view = OneShotView{};
    o.view.world = W; o.view.rank = me; o.view.slot_bytes = slot;
    for (int r = 0; r < W && ok; ++r) {
        if (r == me) { o.view.mail[r] = o.own_mail; continue; }
        if (!hip_step(hipIpcOpenMemHandle(&o.view.mail[r], all[(size_t)r], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) o.view.mail[r] = nullptr;
    }
    (void)hipGetLastError();
    // agreement
    int* d_ok = reinterpret_cast<int*>(d_h);
    e = hipMemcpyAsync(d_ok, &ok, sizeof(int), hipMemcpyHostToDevice, ctx->copy_stream);
    nr = a->AllReduce(d_ok, d_ok, 1, ncclInt32, ncclMin, ctx->comm, ctx->copy_stream);
    int all_ok = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&all_ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, ctx->copy_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->copy_stream);
    if (nr != ncclSuccess || e != hipSuccess) all_ok = 0;
    if (!all_ok) {
        oneshot_release(ctx);
        return ctx->fail(FS_ERR_COMM, "one-shot reduce: the ranks could not map each other's mailboxes (HIP IPC); ncclAllReduce stays in use" +
                         (why.empty() ? std::string(" (another rank failed)") : " (" + why + ")"));
    }
    o.seq = 0;
    o.on = true;
    o.broken = false;
    return FS_OK;
}

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

int fs_comm_info(fs_context* ctx, int32_t* ranks, int32_t* rank, int32_t* collective) {
    if (!ctx) return FS_ERR_INVALID_ARGUMENT;
    int n = 0, r = -1;
    if (ctx->comm) {
        RcclApi* a = rccl();
        if (!a) return ctx->fail(FS_ERR_COMM, "communicator attached but librccl is not loadable");
        FS_NCCL(ctx, a->CommCount(ctx->comm, &n));
        FS_NCCL(ctx, a->CommUserRank(ctx->comm, &r));
    }
    if (ranks) *ranks = n;
    if (rank) *rank = r;
    if (collective) *collective = !ctx->comm ? 0 : (ctx->oneshot.on && !ctx->oneshot.broken ? 2 : 1);
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
    oneshot_release(ctx);
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
    ctx->dbg.tail_ops += 2;   // the collective and its event
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
