// Single-wave instruction-sequence latencies on gfx950 (cycles per iteration, s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_add_sat(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b))); }
__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b))); }
#define N 4096
template <int V>
__global__ void k(unsigned *out, long long *cyc, unsigned seed)
{
    unsigned M = threadIdx.x * 2654435761u + seed, acc = 0, acc2 = 0;
    unsigned inc = (threadIdx.x & 3) + 1;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; i++) {
        if (V == 0) { M = pk_add_sat(M, inc); }                                   // dependent packed add
        if (V == 1) { M = pk_min(pk_add_sat(M, inc), pk_add_sat(M ^ 1, inc)); }   // add,add,min
        if (V == 2) { M = (unsigned)__builtin_amdgcn_update_dpp(0, (int)M, 0xA0, 0xF, 0xF, true) + 1; }   // dpp + add
        if (V == 3) { auto r = __builtin_amdgcn_permlane32_swap(M, M, false, false); M = r[0] + r[1]; }
        if (V == 4) { unsigned long long d = __ballot((M & 0xFFFF) <= (inc << 8)); asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(acc) : "s"((unsigned)d)); asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(acc2) : "s"((unsigned)(d >> 32))); M = pk_add_sat(M, inc); }
        if (V == 5) { M = pk_add_sat(M, inc); unsigned c0 = __builtin_amdgcn_readfirstlane(M); if (c0 == 0x12345678u) M ^= acc++; }   // never-taken branch on VALU-derived scalar
        if (V == 6) { M = pk_add_sat(M, inc); unsigned c0 = __builtin_amdgcn_readfirstlane(M); if (c0 != 0x12345678u) M ^= 1; }      // always-"taken" body
        if (V == 7) { unsigned long long d = __ballot((M & 0xFFFF) <= (inc << 8)); acc += (unsigned)d; M = pk_add_sat(M, inc); }      // ballot consumed by SALU->VALU add
        if (V == 8) { auto r = __builtin_amdgcn_permlane16_swap(M, M, false, false); M = r[0] + r[1]; }
        if (V == 9) { unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x118, 0xF, 0xC, false); unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x108, 0xF, 0x3, false); M = lo + hi; }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = M + acc + acc2;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char *name, unsigned *d, long long *c)
{
    long long h = 0;
    for (int it = 0; it < 2; it++) { k<V><<<1, 64>>>(d, c, 7); hipDeviceSynchronize(); }
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-46s %8.2f memtime-ticks/iter\n", name, (double)h / N);
}
int main()
{
    unsigned *d; long long *c;
    hipMalloc(&d, 256); hipMalloc(&c, 8);
    run<0>("dependent v_pk_add_u16 clamp", d, c);
    run<1>("add,add,min chain", d, c);
    run<2>("dpp quad_perm mov + add", d, c);
    run<3>("permlane32_swap + add", d, c);
    run<8>("permlane16_swap + add", d, c);
    run<9>("2x dpp row_shr/shl:8 masked + add", d, c);
    run<4>("cmp->2x writelane (+add)", d, c);
    run<7>("cmp->SGPR->VALU add (+add)", d, c);
    run<5>("add+readfirstlane+cmp+branch(not taken)", d, c);
    run<6>("add+readfirstlane+cmp+branch(body)", d, c);
    return 0;
}
