// fs_oneshot.hip — the energy buffer's sum over the ranks as ONE exchange step (the [bands][bins]
// histogram is 32 KB; on 8 GPUs a ring all-reduce of that size is 14 latency-bound hops over xGMI).  Every rank owns a
// mailbox in its HBM with one slot per rank; a reduce is
//   push   rank r writes its histogram into slot r of EVERY rank's mailbox (peer memory mapped through HIP IPC) and
//          then raises flag r there to the reduce's sequence number — one kernel, one workgroup per destination;
//   sum    every rank waits until all flags of its own mailbox show the sequence number and adds the slots up in rank
//          order (the same order everywhere: all ranks hold bit-identical sums) — one workgroup.
// Both kernels run on the context's tail stream, like the RCCL all-reduce they replace.  Two mailbox sets alternate:
// a rank pushes reduce k + 2 only after its own sum k + 1 has finished, which needed every peer's push k + 1, which every
// peer enqueued behind its own sum k — so nobody still reads set k % 2 when it is overwritten.
// Payload and flags cross devices while kernels run on both sides: they are written and read with system-scope atomics
// (no caching on either side), the flag store is a release behind a system-scope fence.  The wait is bounded so that no
// kernel spins for ever: a peer that has not shown up after `spins` polls (~1 us each; 10 s by default,
// FS_ONESHOT_TIMEOUT_MS — a legitimate straggler, a first-launch code-object load, a descheduled host thread must never
// trip it: ncclAllReduce would simply wait) sets the error word and lets the kernel end.  After that the two-set argument
// above no longer holds (a late push could land in a set that is being reused), so the host treats it as fatal for the
// communicator: every later reduce fails with FS_ERR_COMM until the host detaches and attaches again (oneshot_check).
// Flags are compared wrap-safe: a flag AT OR BEYOND the awaited sequence number ends the wait.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "fs_internal.hpp"

namespace fs {
namespace {

constexpr int kOneShotBlock = 1024;

__device__ __forceinline__ uint32_t* flag_of(char* mail, int set, int r) {
    return reinterpret_cast<uint32_t*>(mail) + set * kOneShotMaxRanks + r;
}
__device__ __forceinline__ char* slot_of_mail(char* mail, const OneShotView& v, int set, int r) {
    return mail + kOneShotHeaderBytes + ((size_t)set * (size_t)v.world + (size_t)r) * v.slot_bytes;
}

__global__ __launch_bounds__(kOneShotBlock) void oneshot_push_kernel(OneShotView v, const uint32_t* __restrict__ src,
                                                                     int words32, int set, uint32_t seq) {
    char* mail = static_cast<char*>(v.mail[blockIdx.x]);           // one workgroup per destination rank
    uint32_t* dst = reinterpret_cast<uint32_t*>(slot_of_mail(mail, v, set, v.rank));
    for (int i = (int)threadIdx.x; i < words32; i += kOneShotBlock)
        __hip_atomic_store(dst + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag_of(mail, set, v.rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <typename T>
__global__ __launch_bounds__(kOneShotBlock) void oneshot_sum_kernel(OneShotView v, T* __restrict__ dst, int words, int set,
                                                                    uint32_t seq, unsigned* __restrict__ err, unsigned max_spins) {
    char* mail = static_cast<char*>(v.mail[v.rank]);
    if ((int)threadIdx.x < v.world) {
        uint32_t* f = flag_of(mail, set, (int)threadIdx.x);
        unsigned spins = 0;
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            if (++spins > max_spins) { *err = 1u; break; }          // the host reports FS_ERR_COMM; the kernel ends either way
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    for (int i = (int)threadIdx.x; i < words; i += kOneShotBlock) {
        T sum = T(0);
        for (int r = 0; r < v.world; ++r) {
            T* slot = reinterpret_cast<T*>(slot_of_mail(mail, v, set, r));
            if (sizeof(T) == 4) {
                const uint32_t bits = __hip_atomic_load(reinterpret_cast<uint32_t*>(slot) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                float x;
                __builtin_memcpy(&x, &bits, 4);
                sum += (T)x;
            } else {
                sum += (T)__hip_atomic_load(reinterpret_cast<unsigned long long*>(slot) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        dst[i] = sum;
    }
}

}  // namespace

void launch_oneshot_reduce(const OneShotView& v, void* buffer, int words, bool u64, int set, uint32_t seq, unsigned* err,
                           hipStream_t s) {
    static const unsigned max_spins = [] {   // ~1 us per poll
        const char* e = std::getenv("FS_ONESHOT_TIMEOUT_MS");
        const long ms = e ? std::atol(e) : 10000;
        return (unsigned)std::max(1000L, std::min(ms, 3600000L)) * 1000u;
    }();
    const int words32 = u64 ? 2 * words : words;
    hipLaunchKernelGGL(oneshot_push_kernel, dim3((unsigned)v.world), dim3(kOneShotBlock), 0, s, v,
                       static_cast<const uint32_t*>(buffer), words32, set, seq);
    if (u64)
        hipLaunchKernelGGL(oneshot_sum_kernel<unsigned long long>, dim3(1), dim3(kOneShotBlock), 0, s, v,
                           static_cast<unsigned long long*>(buffer), words, set, seq, err, max_spins);
    else
        hipLaunchKernelGGL(oneshot_sum_kernel<float>, dim3(1), dim3(kOneShotBlock), 0, s, v, static_cast<float*>(buffer), words,
                           set, seq, err, max_spins);
}

}  // namespace fs
