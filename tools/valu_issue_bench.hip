// valu_issue_bench.hip — what one MI355X SIMD issues per cycle, measured (the denominator of bench.py's VALU-issue
// roofline).  Every wave runs a long stream of independent v_fma_f32 / v_cvt_f32_ubyte / v_cndmask / v_min3 style
// instructions; the grid puts W waves on every SIMD (W = 1, 2, 3, 4, 8) and the in-kernel clock (s_memtime) over the
// loop gives wave-instructions per SIMD-cycle.  Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/valu_issue_bench tools/valu_issue_bench.hip && /tmp/valu_issue_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kIters = 4096;
constexpr int kUnroll = 32;   // independent accumulators per lane

template <int KIND>
__global__ __launch_bounds__(256) void issue_kernel(float* out, unsigned long long* cycles, float a, float b) {
    float acc[kUnroll];
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) acc[i] = (float)(threadIdx.x + i);
    const float2 a2 = make_float2(a, a), b2 = make_float2(b, b);
    const unsigned long long mask = __ballot((threadIdx.x & 1) != 0);
    unsigned long long tmpm = 0;
    (void)mask; (void)tmpm;
    if (KIND == 4 || KIND == 13 || KIND == 14 || KIND == 16) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {   // inline asm: the compiler would pack or fold plain C
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 1) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 2) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(acc[i]));
            if (KIND == 3) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            if (KIND == 5 && (i & 1) == 0)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*reinterpret_cast<float2*>(&acc[i])) : "v"(a2), "v"(b2));
            if (KIND == 6) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "s"(mask));
            if (KIND == 7) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(acc[i]), "v"(a) : "vcc");
            if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "+v"(acc[i]) : "v"(a));
            if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(a));
            if (KIND == 10) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(acc[i]) : "v"(a));
            if (KIND == 11) asm volatile("v_max3_f32 %0, %0, %1, 0" : "+v"(acc[i]) : "v"(a));
            if (KIND == 13) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            if (KIND == 14) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(acc[i]) : "v"(a), "v"(b));
            if (KIND == 15) {   // realistic: one compare into vcc feeds four selects
                if ((i & 3) == 0) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(acc[i]), "v"(a) : "vcc");
                asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a));
            }
            if (KIND == 16) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(acc[(i + 7) & 31]));
            if (KIND == 12) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(acc[i]), "=&s"(tmpm) : "v"(a));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cycles[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t1 - t0;
        cycles[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = r1 - r0;   // 100 MHz
    }
}

template <int KIND>
void run(const char* name, int half_insts_per_step) {   // instructions per accumulator per step, in halves
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 2 * 4 * cus * 8);
    for (int wps : {1, 2, 3, 4}) {   // waves per SIMD: one 256-thread block = 1 wave on each of the CU's 4 SIMDs
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);   // warm
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
        (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        double mean = 0, real = 0;
        for (size_t k = 0; k < h.size(); k += 2) { mean += (double)h[k]; real += (double)h[k + 1]; }
        mean /= (double)(h.size() / 2); real /= (double)(h.size() / 2);
        const double clock_mhz = real > 0 ? mean / real * 100.0 : 0.0;   // shader clock inside the loop
        const double inst_per_wave = (double)kIters * kUnroll * half_insts_per_step * 0.5;
        // s_memtime ticks at a fixed 100 MHz on this part; convert with the wall time instead: cycles are reported both ways
        const double wave_inst_per_simd = inst_per_wave * wps;
        const double simd_inst_per_us = wave_inst_per_simd / (ms * 1e3);
        printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"shader_clock_mhz\": %.0f, "
               "\"cycles_per_inst_one_wave\": %.3f, \"cycles_per_inst_simd\": %.3f, \"wave_insts_per_simd_per_us\": %.1f}\n",
               name, wps, ms, clock_mhz, mean / inst_per_wave, mean / inst_per_wave / wps, simd_inst_per_us);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32", 2);
    run<1>("v_min3_f32", 2);
    run<2>("v_cvt_f32_ubyte1", 2);
    run<3>("v_and_or_b32", 2);
    run<4>("v_cndmask_b32", 2);
    run<5>("v_pk_fma_f32", 1);   // one packed instruction per two accumulators
    run<6>("v_cndmask_b32_e64 (sgpr mask)", 2);
    run<7>("v_cmp_lt_f32_e32", 2);
    run<8>("v_mov_b32", 2);
    run<9>("v_add_f32", 2);
    run<10>("v_lshl_add_u32", 2);
    run<11>("v_max3_f32", 2);
    run<12>("v_cmp_e64 + v_cndmask_e64", 4);
    run<13>("v_cndmask_b32_e64 (vcc)", 2);
    run<14>("v_cndmask_b32_e32 (vcc, dst != src)", 2);
    run<15>("v_cmp_e32 vcc + 4 v_cndmask_e32", 2);   // counted as the 32 selects only (8 compares ride along)
    run<16>("v_cndmask_b32_e32 (vcc, two vgpr sources)", 2);
    return 0;
}
