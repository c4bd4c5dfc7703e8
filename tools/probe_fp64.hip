// Issue cost of fp64 VALU instructions on gfx950 (8 waves per SIMD, 8 independent chains per wave), like probe_issue.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 2048
template <int V>
__global__ __launch_bounds__(64) void k(double *out, double seed)
{
    double r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 1e-3 + seed + i;
    const double c = seed * 0.5, one = seed / seed, nz = -0.0 * seed;
    for (int it = 0; it < REP; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (V == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(c));
            if (V == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(one));
            if (V == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(one), "v"(c));
            if (V == 3) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(one), "v"(nz));
            if (V == 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(*(float *)&r[i]) : "v"((float)c));
            if (V == 5) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(r[i]) : "v"(*(float *)&r[(i + 1) & 7]));
        }
    }
    double a = 0;
    for (int i = 0; i < 8; i++) a += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <int V> void run(const char *name, double *d, double per_iter_instr)
{
    const int nblk = 256 * 4 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; it++) { hipEventRecord(e0); k<V><<<nblk, 64>>>(d, 1.25); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    double instr_per_simd = 8.0 * REP * 8 * per_iter_instr;
    printf("%-34s %8.3f ms   %6.2f clk per wave-instr per SIMD @2.4GHz\n", name, ms, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    double *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 8);
    run<0>("v_add_f64", d, 1); run<1>("v_mul_f64", d, 1); run<2>("v_fma_f64", d, 1); run<3>("v_fma_f64 (+ -0.0)", d, 1);
    run<4>("v_add_f32", d, 1); run<5>("v_cvt_f64_f32", d, 1);
    return 0;
}
