// Issue-throughput probe: how many clocks of SIMD time does one wave64 instruction of each kind cost on
// gfx950 when 8 waves per SIMD keep the pipes full?  (8192 waves, each loops over 8 independent chains.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
#define REP 2048
template <int V>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned seed)
{
    unsigned r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761u + seed + i;
    unsigned long long sacc = 0;
    for (int it = 0; it < REP; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (V == 0) r[i] = r[i] + 0x9E3779B9u;                                                      // v_add_u32
            if (V == 1) r[i] = __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2_t, r[i]), __builtin_bit_cast(ushort2_t, 0x00030005u)));   // v_pk_add_u16 clamp
            if (V == 2) r[i] = __builtin_amdgcn_perm(r[i], r[(i + 1) & 7], 0x07020500u);               // v_perm_b32
            if (V == 3) r[i] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)r[i], 0xA0, 0xF, 0xF, true);   // v_mov_dpp
            if (V == 4) { auto s = __builtin_amdgcn_permlane32_swap(r[i], r[(i + 1) & 7], false, false); r[i] = s[0]; r[(i + 1) & 7] = s[1]; }
            if (V == 5) sacc += __ballot((r[i] & 0xFFFFu) <= (r[(i + 1) & 7] & 0xFFFFu)), r[i] += 1;   // v_cmp_sdwa (+v_add) (+s_add)
            if (V == 6) asm volatile("v_writelane_b32 %0, %1, 7" : "+v"(r[i]) : "s"(seed));
            if (V == 7) r[i] ^= r[(i + 1) & 7];                                                         // v_xor
            if (V == 8) asm volatile("s_nop 0");
            if (V == 9) asm volatile("s_add_u32 %0, %0, 3" : "+s"(seed));
        }
    }
    unsigned a = (unsigned)sacc + seed;
    for (int i = 0; i < 8; i++) a += r[i];
    out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <int V> void run(const char *name, unsigned *d, double per_iter_instr)
{
    const int nblk = 256 * 4 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 3; it++) { hipEventRecord(e0); k<V><<<nblk, 64>>>(d, 7); hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); }
    double instr_per_simd = 8.0 * REP * 8 * per_iter_instr;      // 8 waves per SIMD
    printf("%-34s %8.3f ms   %6.2f ns per wave-instr per SIMD (= %5.2f clk @2.4GHz)\n", name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main()
{
    unsigned *d; hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    run<0>("v_add_u32", d, 1); run<7>("v_xor_b32", d, 1); run<1>("v_pk_add_u16 clamp", d, 1); run<2>("v_perm_b32", d, 1);
    run<3>("v_mov_b32_dpp", d, 1); run<4>("v_permlane32_swap (+movs?)", d, 1); run<5>("v_cmp_sdwa + v_add + s_add64", d, 3);
    run<6>("v_writelane_b32", d, 1); run<8>("s_nop 0", d, 1); run<9>("s_add_u32", d, 1);
    return 0;
}
