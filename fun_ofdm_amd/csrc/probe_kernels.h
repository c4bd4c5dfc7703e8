// probe_kernels.h -- what does one wave64 instruction cost a SIMD on THIS part, now?  (foa_rx_probe_issue)
//
// bench.py's roofline prices the forward pass against the rate at which the machine can issue the instructions the pass is
// made of.  That rate is measured live, in the run that reports it (counters and clocks of another box do
// not belong in the line): 1024 SIMDs x W single-wave workgroups issue one instruction class from eight independent
// chains for a fixed WINDOW of shader clocks (s_memtime ticks = shader cycles, MI355X_MICROARCH.md) and report how far
// they got; the wall time of the launch (HIP events) gives the clock the part sustained meanwhile.
// Same method as tools/probe_issue.hip, reduced to the two classes the roofline needs.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace foa {

template <int KIND>      // 0: v_pk_add_u16 ... clamp (packed: two of the counted ops per lane), 1: v_add_u32 (plain VOP2)
__global__ __launch_bounds__(256) void k_probe_issue(unsigned long long *out, unsigned seed, int window_k)
{
    unsigned r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761u + seed + i;
    const unsigned long long t0 = __builtin_readcyclecounter(), window = (unsigned long long)window_k * 1000ull;
    unsigned long long t1 = t0;
    unsigned done = 0;
    for (; t1 - t0 < window; t1 = __builtin_readcyclecounter(), done++) {
        for (int it = 0; it < 8; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(seed));
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
            }
        }
    }
    unsigned a = seed;
#pragma unroll
    for (int i = 0; i < 8; i++) a += r[i];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[3 * w] = t1 - t0; out[3 * w + 1] = done; out[3 * w + 2] = a;
    }
}

}  // namespace foa
