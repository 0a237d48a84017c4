// vmem_issue_bench.hip — what one MI355X CU's vector-memory pipe (TA -> L1 tag lookup -> TD return) sustains for the
// access shape of the BVH traversal: every lane fetches ITS OWN 64-byte record (divergent addresses) from a table that
// lives in L2 (1.6 MB = cfg3's node array, 8 MB = nodes + triangles) with 16-byte global_load_dwordx4 instructions.
// It is the memory-side twin of tools/valu_issue_bench.hip and gives bench.py's `vmem` roofline its denominator.
//
// Shapes (one "step" = what a lane needs per traversal iteration):
//   rec64x4   4 x dwordx4 on the lane's own 64-B record        (the node fetch as shipped: 4 L1 lookups per lane)
//   rec48x3   3 x dwordx4 on the lane's own 48-B record        (the triangle fetch / a 48-B node)
//   rec32x2   2 x dwordx4                                      (a 32-B record)
//   rec16x1   1 x dwordx4                                      (one lookup per lane)
//   quad64    4 x dwordx4, the 4 lanes of a quad fetch the 4 quarters of ONE record per instruction (16 distinct
//             lines per wave instruction instead of 64; the data would then be redistributed by DPP)
//   bcast     4 x dwordx4, all lanes the same record            (what round 1's dummy fetches were)
//   half      rec64x4 with only the even lanes enabled          (exec-masked lanes: do they cost lookups?)
// Each step ends with s_waitcnt vmcnt(0) like the traversal step does (the next address depends on the data there; here
// it comes from an LCG so that the latency chain is the wait, not the address).  Waves per SIMD = 1..4 by grid size.
// Output: JSON lines with cycles per wave-step per CU, lookups (lane-instructions on distinct 16-B pieces) per
// CU-cycle, bytes per CU-cycle and the in-kernel shader clock.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/vmem_issue_bench tools/vmem_issue_bench.hip && /tmp/vmem_issue_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kIters = 2048;

enum Shape { REC64 = 0, REC48 = 1, REC32 = 2, REC16 = 3, QUAD64 = 4, BCAST = 5, HALF = 6 };

template <int SHAPE>
__global__ __launch_bounds__(256) void vmem_kernel(const char* __restrict__ table, uint32_t rec_mask, float* out,
                                                   unsigned long long* cycles) {
    const unsigned lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        x = x * 1664525u + 1013904223u;
        uint32_t rec = (x >> 8) & rec_mask;
        if (SHAPE == BCAST) rec = __builtin_amdgcn_readfirstlane(rec);
        v4f q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
        if (SHAPE == QUAD64) {
            // instruction j fetches, for every quad, quarter (lane & 3) of the record of the quad's lane j
            const uint32_t r0q = __shfl(rec, (lane & ~3u) + 0u), r1q = __shfl(rec, (lane & ~3u) + 1u),
                           r2q = __shfl(rec, (lane & ~3u) + 2u), r3q = __shfl(rec, (lane & ~3u) + 3u);
            const char* p0 = table + (size_t)r0q * 64 + (lane & 3u) * 16;
            const char* p1 = table + (size_t)r1q * 64 + (lane & 3u) * 16;
            const char* p2 = table + (size_t)r2q * 64 + (lane & 3u) * 16;
            const char* p3 = table + (size_t)r3q * 64 + (lane & 3u) * 16;
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %5, off\n\t"
                         "global_load_dwordx4 %2, %6, off\n\tglobal_load_dwordx4 %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        } else {
            const char* p = table + (size_t)rec * 64;
            const bool on = SHAPE != HALF || (lane & 1u) == 0u;
            if (on) {
                if (SHAPE == REC64 || SHAPE == BCAST || SHAPE == HALF)
                    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:16\n\t"
                                 "global_load_dwordx4 %2, %4, off offset:32\n\tglobal_load_dwordx4 %3, %4, off offset:48"
                                 : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(p) : "memory");
                if (SHAPE == REC48)
                    asm volatile("global_load_dwordx4 %0, %3, off\n\tglobal_load_dwordx4 %1, %3, off offset:16\n\t"
                                 "global_load_dwordx4 %2, %3, off offset:32"
                                 : "=&v"(q0), "=&v"(q1), "=&v"(q2) : "v"(p) : "memory");
                if (SHAPE == REC32)
                    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                                 : "=&v"(q0), "=&v"(q1) : "v"(p) : "memory");
                if (SHAPE == REC16)
                    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(q0) : "v"(p) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        }
        acc += q0.x + q1.y + q2.z + q3.w;
        x ^= __float_as_uint(q0.x) & 1u;   // a (harmless) data dependence, like the traversal's next address
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (lane == 0) {
        cycles[2 * (blockIdx.x * 4 + threadIdx.x / 64)] = t1 - t0;
        cycles[2 * (blockIdx.x * 4 + threadIdx.x / 64) + 1] = r1 - r0;   // 100 MHz
    }
}

template <int SHAPE>
void run(const char* name, const char* table, size_t table_bytes, int loads_per_step, double lookups_per_step,
         double bytes_per_step) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float* out;
    unsigned long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * cus * 8);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 2 * 4 * cus * 8);
    const uint32_t rec_mask = (uint32_t)(table_bytes / 64) - 1u;
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const int blocks = cus * wps;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(blocks), dim3(256), 0, 0, table, rec_mask, out, cyc);   // warm: L2 + code
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(vmem_kernel<SHAPE>, dim3(blocks), dim3(256), 0, 0, table, rec_mask, out, cyc);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h((size_t)blocks * 4 * 2);
        (void)hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        double mean = 0, real = 0;
        for (size_t k = 0; k < h.size(); k += 2) { mean += (double)h[k]; real += (double)h[k + 1]; }
        mean /= (double)(h.size() / 2); real /= (double)(h.size() / 2);
        const double clock_mhz = real > 0 ? mean / real * 100.0 : 0.0;
        // per CU: 4 * wps waves run kIters steps each in `mean` shader cycles
        const double wave_steps_per_cu = 4.0 * wps * kIters;
        const double cyc_per_step_cu = mean / wave_steps_per_cu;
        printf("{\"shape\": \"%s\", \"table_mb\": %.1f, \"waves_per_simd\": %d, \"ms\": %.4f, \"shader_clock_mhz\": %.0f, "
               "\"cycles_per_wave_step_one_wave\": %.1f, \"cu_cycles_per_wave_step\": %.2f, \"cu_cycles_per_load_inst\": %.2f, "
               "\"lookups_per_cu_cycle\": %.3f, \"bytes_per_cu_cycle\": %.1f, \"chip_tb_s\": %.2f}\n",
               name, table_bytes / 1048576.0, wps, ms, clock_mhz, mean / kIters, cyc_per_step_cu,
               cyc_per_step_cu / loads_per_step, lookups_per_step / cyc_per_step_cu, bytes_per_step / cyc_per_step_cu,
               bytes_per_step * wave_steps_per_cu * cus / (ms * 1e-3) / 1e12);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(out); (void)hipFree(cyc);
}

int main(int argc, char** argv) {
    // table sizes in KB (powers of two): default 2 MB ~ the node array (fits every XCD's L2 with room), 8 MB ~ nodes +
    // triangles; 16 KB is the L1-hit regime (tag lookup and data return rates without misses)
    std::vector<size_t> kbs;
    for (int i = 1; i < argc; ++i) kbs.push_back((size_t)atoi(argv[i]));
    if (kbs.empty()) kbs = {2048, 8192};
    for (size_t kb : kbs) {
        const size_t bytes = kb << 10;
        char* table;
        (void)hipMalloc(&table, bytes);
        std::vector<float> h(bytes / 4);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i & 1023) * 1e-3f;
        (void)hipMemcpy(table, h.data(), bytes, hipMemcpyHostToDevice);
        // lookups = lane-instructions (one 16-B piece each); quad64: a quad's 4 lanes share one 64-B piece of a line
        run<REC64>("rec64x4", table, bytes, 4, 256.0, 4096.0);
        run<REC48>("rec48x3", table, bytes, 3, 192.0, 3072.0);
        run<REC32>("rec32x2", table, bytes, 2, 128.0, 2048.0);
        run<REC16>("rec16x1", table, bytes, 1, 64.0, 1024.0);
        run<QUAD64>("quad64", table, bytes, 4, 256.0, 4096.0);
        run<BCAST>("bcast", table, bytes, 4, 256.0, 4096.0);
        run<HALF>("half", table, bytes, 4, 128.0, 2048.0);
        (void)hipFree(table);
    }
    return 0;
}
