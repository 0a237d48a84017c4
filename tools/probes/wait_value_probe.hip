// wait_value_probe.hip — does this ROCm stack let a stream wait for a VALUE a running kernel of another stream writes
// (hipStreamWaitValue64 on signal memory), and what does that cost compared with an event between two launches?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wvp tools/probes/wait_value_probe.hip && /tmp/wvp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_then_store(unsigned long long* word, unsigned long long v, long long cycles, float* sink) {
    const long long t0 = wall_clock64();
    float a = 0.f;
    while (wall_clock64() - t0 < cycles) a += 1.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sink[0] = a;
        __threadfence_system();
        __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void stamp(long long* out) { if (threadIdx.x == 0) out[0] = wall_clock64(); }
__global__ void busy(long long cycles, float* sink) {
    const long long t0 = wall_clock64();
    float a = 0.f;
    while (wall_clock64() - t0 < cycles) a += 1.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[1] = a;
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    std::printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    float* sink; CK(hipMalloc(&sink, 64));
    long long* d_stamp; CK(hipMalloc(&d_stamp, 64));
    int wc = 0; CK(hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0));   // kHz
    std::printf("wall clock %d kHz\n", wc);
    const long long cyc_500us = (long long)wc * 500 / 1000;
    // (1) signal memory written by a kernel on stream a, waited for by stream b
    for (int kind = 0; kind < 2; ++kind) {
        unsigned long long* word = nullptr;
        hipError_t e = kind == 0 ? hipExtMallocWithFlags((void**)&word, 64, hipMallocSignalMemory) : hipHostMalloc((void**)&word, 64, hipHostMallocCoherent);
        if (e != hipSuccess) { std::printf("kind %d: allocation failed: %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        CK(hipMemset(word, 0, 8));
        CK(hipDeviceSynchronize());
        hipLaunchKernelGGL(spin_then_store, dim3(1), dim3(64), 0, a, word, 7ull, cyc_500us, sink);
        e = hipStreamWaitValue64(b, word, 7ull, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull);
        if (e != hipSuccess) { std::printf("kind %d: hipStreamWaitValue64 -> %s\n", kind, hipGetErrorString(e)); (void)hipGetLastError(); CK(hipDeviceSynchronize()); continue; }
        hipLaunchKernelGGL(stamp, dim3(1), dim3(64), 0, b, d_stamp);
        const auto t0 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(b));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CK(hipStreamSynchronize(a));
        std::printf("kind %d (%s): stream b released after %.3f ms (the store comes 0.5 ms into the kernel)\n", kind, kind == 0 ? "signal memory" : "coherent pinned host", ms);
    }
    // (2) what an event record between two back-to-back kernels of one stream costs: 200 x 100 us kernels, with / without
    const long long cyc_100us = (long long)wc * 100 / 1000;
    hipEvent_t ev[3];
    CK(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming | hipEventReleaseToDevice));
    CK(hipEventCreateWithFlags(&ev[2], hipEventDisableTiming | hipEventDisableSystemFence));
    for (int mode = 0; mode < 5; ++mode) {
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 200; ++i) {
            hipLaunchKernelGGL(busy, dim3(256), dim3(256), 0, a, cyc_100us, sink);
            if (mode >= 1 && mode <= 3) CK(hipEventRecord(ev[mode - 1], a));
            if (mode == 4) { CK(hipEventRecord(ev[1], a)); CK(hipStreamWaitEvent(b, ev[1], 0)); }
        }
        CK(hipStreamSynchronize(a));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        static const char* names[] = {"no event", "default event after every kernel", "ReleaseToDevice event", "DisableSystemFence event", "ReleaseToDevice event + wait on another stream"};
        std::printf("200 x 100 us kernels, %s: %.3f ms (%.1f us per kernel beyond 100)\n", names[mode], ms, (ms * 1000.0 / 200.0) - 100.0);
    }
    return 0;
}
