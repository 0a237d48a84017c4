// fake_rccl.cpp — TEST DOUBLE, not part of the product: the handful of RCCL entry points libfrequensee.so opens at run
// time (csrc/fs_capi.cpp: RcclApi), implemented over POSIX shared memory so that SEVERAL RANKS CAN SHARE ONE GPU.
// Real RCCL refuses two ranks on one device, and the GPU box of the test tier has one device: with FS_RCCL_LIB
// pointing here, tests/test_two_ranks_one_gpu.py runs the library's whole multi-rank protocol — unique-id rendezvous,
// scene broadcast at fs_scene_commit, per-frame all-reduce on the tail stream, the partial-reconstruct refusal — as two
// processes, and compares every rank's IR with the single-rank IR.  (Real RCCL is exercised with one rank by
// tests/test_gpu_parity.py::test_library_collective_one_rank and with N ranks by bench.py on a multi-GPU node.)
//
// Semantics: every collective first drains the stream it was given (so it is stream-ordered), stages through host
// shared memory, and returns when the result is in the receive buffer.  Sums are taken in rank order on every rank.
// A rank that waits longer than FAKE_RCCL_TIMEOUT_S (default 60) for its peers returns ncclSystemError instead of hanging.
//   build: g++ -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.cpp -o <dir>/libfake_rccl.so
//          -L/opt/rocm/lib -lamdhip64 -lrt
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr size_t kSlotBytes = 48u << 20;   // per-rank staging area: a test scene's tree and a frame's histograms fit
constexpr int kMaxRanks = 8;

struct Control {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};

struct FakeComm {
    int rank = 0, nranks = 1;
    char name[64] = {0};
    Control* ctl = nullptr;
    unsigned char* slots = nullptr;
    size_t map_bytes = 0;
    double timeout_s = 60.0;
    unsigned char* slot(int r) const { return slots + (size_t)r * kSlotBytes; }
};

bool barrier(FakeComm* c) {
    const int gen = c->ctl->generation.load(std::memory_order_acquire);
    if (c->ctl->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->nranks) {
        c->ctl->arrived.store(0, std::memory_order_relaxed);
        c->ctl->generation.fetch_add(1, std::memory_order_acq_rel);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->generation.load(std::memory_order_acquire) == gen) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) return false;
    }
    return true;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

template <typename T>
void sum_into(T* acc, const T* x, size_t n) { for (size_t i = 0; i < n; ++i) acc[i] += x[i]; }

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    static std::atomic<unsigned> counter{0};
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/fs_fake_rccl_%d_%u", (int)getpid(), counter.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm* c = new FakeComm;
    c->rank = rank; c->nranks = nranks;
    if (const char* t = std::getenv("FAKE_RCCL_TIMEOUT_S")) c->timeout_s = std::atof(t);
    std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = 4096 + (size_t)nranks * kSlotBytes;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { if (fd >= 0) close(fd); delete c; return ncclSystemError; }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->ctl = static_cast<Control*>(p);             // a fresh shm object is zero-filled: counters start at 0
    c->slots = static_cast<unsigned char*>(p) + 4096;
    c->ctl->attached.fetch_add(1);
    if (!barrier(c)) { munmap(p, c->map_bytes); delete c; return ncclSystemError; }   // like the real one: returns when all ranks are in
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    if (!c) return ncclInvalidArgument;
    if (c->ctl->attached.fetch_sub(1) == 1) shm_unlink(c->name);   // the last rank out removes the object
    munmap(c->ctl, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { *count = reinterpret_cast<const FakeComm*>(comm)->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { *rank = reinterpret_cast<const FakeComm*>(comm)->rank; return ncclSuccess; }

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    const size_t bytes = count * type_bytes(dt);
    const bool words32 = dt == ncclInt32 || dt == ncclUint32;     // the library's agreement / overflow words: sum, min or max
    if (bytes > kSlotBytes || !((op == ncclSum && (dt == ncclFloat32 || dt == ncclUint64)) ||
                                (words32 && (op == ncclSum || op == ncclMin || op == ncclMax))))
        return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    std::vector<unsigned char> acc(bytes, 0);
    if (words32) {
        for (size_t i = 0; i < count; ++i) {
            long long v = 0;
            for (int r = 0; r < c->nranks; ++r) {
                const long long x = dt == ncclInt32 ? (long long)reinterpret_cast<const int32_t*>(c->slot(r))[i]
                                                    : (long long)reinterpret_cast<const uint32_t*>(c->slot(r))[i];
                v = r == 0 ? x : (op == ncclSum ? v + x : (op == ncclMin ? std::min(v, x) : std::max(v, x)));
            }
            reinterpret_cast<uint32_t*>(acc.data())[i] = (uint32_t)v;
        }
    }
    for (int r = 0; r < c->nranks && !words32; ++r) {
        if (dt == ncclFloat32) sum_into(reinterpret_cast<float*>(acc.data()), reinterpret_cast<const float*>(c->slot(r)), count);
        else sum_into(reinterpret_cast<uint64_t*>(acc.data()), reinterpret_cast<const uint64_t*>(c->slot(r)), count);
    }
    if (!barrier(c)) return ncclSystemError;        // every rank has read every slot: they may be overwritten now
    if (hipMemcpy(recv, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    const size_t bytes = count * type_bytes(dt);
    if (bytes > kSlotBytes || root < 0 || root >= c->nranks) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == root && bytes && hipMemcpy(c->slot(root), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    if (c->rank != root) {
        if (bytes && hipMemcpy(recv, c->slot(root), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    } else if (send != recv && bytes) {
        if (hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = reinterpret_cast<FakeComm*>(comm);
    const size_t bytes = count * type_bytes(dt);
    if (bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpy(c->slot(c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->nranks && bytes; ++r)
        if (hipMemcpy(static_cast<unsigned char*>(recv) + (size_t)r * bytes, c->slot(r), bytes, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "fake rccl: HIP call failed";
        case ncclSystemError: return "fake rccl: peer rank did not arrive (timeout) or shared memory unavailable";
        case ncclInvalidArgument: return "fake rccl: invalid argument";
        default: return "fake rccl: error";
    }
}

}  // extern "C"
