// Latency of the forward pass's renormalisation path on gfx950: one wave alone on the device (s_memtime ticks per iteration; a
// tick is about one shader clock here -- the same loop timed over a grid gives 54 clocks at 2.4 GHz for 50.5 ticks) and 1 or 5
// waves on every SIMD (ms for the whole grid, clocks of SIMD time per iteration).  Variants of "find the smallest
// 16-bit half over the 64 lanes and subtract it from every lane":
//   0  the loop alone: packed add + v_readfirstlane + scalar test + branch never taken (the step's frame)
//   1  six v_min_u16_dpp + v_readlane + s_sub + v_sub, in line (what viterbi_v3.h's cold path runs)
//   2  the same behind a branch that is always taken, laid out cold (as in the kernel)
//   3  LDS: ds_min_u32 of all lanes on one word, ds_read of it, v_sub with the vector operand, cell reset by lane 0
//   4  four v_min_u16_dpp inside the rows, 4 v_readlane, 3 s_min_u32
//   5  v_permlane32_swap + v_permlane16_swap levels, then four DPP levels, v_readfirstlane
//   7, 8  no branch: the six DPP levels under EXEC = all or none (never / always due), v_readlane and v_sub unconditional
//   9, 10  no branch: reduction, v_readlane and v_sub in VSKIP mode (s_setvskip) when nothing is due (never / always due)
//   6  variant 1 with the wave at raised priority while it reduces (s_setprio)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_add_sat(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b))); }

__device__ __forceinline__ unsigned min_dpp6(unsigned v)
{
    unsigned r;
    asm("s_nop 1\n\t"
        "v_min_u16_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return __builtin_amdgcn_readlane(r, 63) & 0xFFFFu;
}
__device__ __forceinline__ unsigned min_rows4(unsigned v)
{
    unsigned r;
    asm("s_nop 1\n\t"
        "v_min_u16_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return r;
}

#define N 4096
template <int V>
__global__ __launch_bounds__(256) void k(unsigned *out, long long *cyc, unsigned seed)
{
    __shared__ unsigned cell[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned M = 0xFF00FF00u + ((threadIdx.x * 2654435761u + seed) & 0x003F003Fu);
    const unsigned inc = (threadIdx.x & 3) + 1;
    if (lane == 0) cell[wave] = 0xFFFFFFFFu;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < N; i++) {
        M = pk_add_sat(M, inc);
        const unsigned s0 = __builtin_amdgcn_readfirstlane(M);
        if (V == 0) { if (__builtin_expect(s0 == 0x12345678u, 0)) M ^= 1; }
        if (V == 1) { const unsigned mn = min_dpp6(M); M -= mn - 0xFF00u; }
        if (V == 2) { if (__builtin_expect(s0 != 0x12345678u, 0)) { const unsigned mn = min_dpp6(M); M -= mn - 0xFF00u; } }
        if (V == 6) { if (__builtin_expect(s0 != 0x12345678u, 0)) { __builtin_amdgcn_s_setprio(3); const unsigned mn = min_dpp6(M); M -= mn - 0xFF00u; __builtin_amdgcn_s_setprio(0); } }
        if constexpr (V == 7 || V == 8) {        // no branch at all: the six DPP levels under EXEC = (due ? all : none), the rest unconditional
            const bool due = V == 8 ? s0 != 0x12345678u : s0 == 0x12345678u;
            unsigned r = M;
            const unsigned ds = __builtin_amdgcn_readfirstlane((unsigned)due);
            asm volatile("s_cmp_lg_u32 %1, 0\n\ts_cselect_b64 exec, -1, 0\n\t"
                "v_min_u16_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                "s_mov_b64 exec, -1" : "+v"(r) : "s"(ds) : "scc");
            const unsigned mn = __builtin_amdgcn_readlane(r, 63) & 0xFFFFu;
            M -= due ? mn - 0xFF00u : 0u;
        }
        if constexpr (V == 9 || V == 10) {      // no branch: VSKIP mode (s_setvskip) over the reduction when nothing is due
            const bool due = V == 10 ? s0 != 0x12345678u : s0 == 0x12345678u;
            const unsigned skip = __builtin_amdgcn_readfirstlane((unsigned)!due);
            unsigned r = M, adj;
            asm volatile("s_setvskip %3, 0\n\t"
                "v_min_u16_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
                "v_min_u16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0\n\t"
                "v_readlane_b32 %1, %0, 63\n\t"
                "s_and_b32 %1, %1, 0xffff\n\ts_sub_u32 %1, %1, 0xff00\n\ts_nop 0\n\t"
                "v_subrev_u32 %2, %1, %2\n\t"
                "s_setvskip 0, 0" : "+v"(r), "=&s"(adj), "+v"(M) : "s"(skip) : "scc");
        }
        if (V == 3) {
            if (__builtin_expect(s0 != 0x12345678u, 0)) {
                __hip_atomic_fetch_min(&cell[wave], M & 0xFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                const unsigned mn = *(volatile unsigned *)&cell[wave];
                if (lane == 0) *(volatile unsigned *)&cell[wave] = 0xFFFFFFFFu;
                M -= mn - 0xFF00u;
            }
        }
        if (V == 4) {
            if (__builtin_expect(s0 != 0x12345678u, 0)) {
                const unsigned r = min_rows4(M);
                const unsigned a = __builtin_amdgcn_readlane(r, 0) & 0xFFFFu, b = __builtin_amdgcn_readlane(r, 16) & 0xFFFFu,
                               c = __builtin_amdgcn_readlane(r, 32) & 0xFFFFu, d = __builtin_amdgcn_readlane(r, 48) & 0xFFFFu;
                const unsigned mn = min(min(a, b), min(c, d));
                M -= mn - 0xFF00u;
            }
        }
        if (V == 5) {
            if (__builtin_expect(s0 != 0x12345678u, 0)) {
                unsigned v = M & 0xFFFFu;
                auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
                v = min(r[0], r[1]);
                auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
                v = min(q[0], q[1]);
                const unsigned mn = __builtin_amdgcn_readfirstlane(min_rows4(v)) & 0xFFFFu;
                M -= mn - 0xFF00u;
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = M;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char *name, unsigned *d, long long *c)
{
    long long h = 0;
    for (int it = 0; it < 2; it++) { k<V><<<1, 64>>>(d, c, 7); hipDeviceSynchronize(); }
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    // W waves per SIMD: 256 CUs x W blocks of 4 waves
    float ms[2] = { 0, 0 };
    const int Ws[2] = { 1, 5 };
    for (int wi = 0; wi < 2; wi++) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<V><<<256 * Ws[wi], 256>>>(d, c, 7); hipDeviceSynchronize();
        hipEventRecord(e0); k<V><<<256 * Ws[wi], 256>>>(d, c, 7); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[wi], e0, e1);
    }
    printf("%-62s %7.2f ticks/iter alone; grid at 1 wave/SIMD %.3f ms (%.0f clk per iteration at 2.4 GHz), at 5 waves/SIMD %.3f ms (%.0f clk per iteration and SIMD)\n",
           name, (double)h / N, ms[0], ms[0] * 1e-3 * 2.4e9 / N, ms[1], ms[1] * 1e-3 * 2.4e9 / N / 5);
}
int main()
{
    unsigned *d; long long *c;
    hipMalloc(&d, 256 * 5 * 256 * 4); hipMalloc(&c, 8);
    run<0>("0 step frame: add + readfirstlane + test + branch not taken", d, c);
    run<1>("1 + 6 x v_min_u16_dpp, readlane, s_sub, v_sub in line", d, c);
    run<2>("2 + the same behind a taken branch, cold", d, c);
    run<6>("6 + the same, cold, s_setprio 3 around it", d, c);
    run<7>("7 six DPP levels under EXEC = 0 (never due), readlane + v_sub always, no branch", d, c);
    run<8>("8 the same, always due", d, c);
    run<9>("9 reduction + readlane + v_sub under VSKIP (never due), no branch", d, c);
    run<10>("10 the same, always due", d, c);
    run<3>("3 + ds_min_u32 on one LDS word, ds_read, v_sub, cold", d, c);
    run<4>("4 + 4 DPP levels, 4 readlane, 3 s_min, cold", d, c);
    run<5>("5 + permlane32/16 swap levels, 4 DPP levels, readfirstlane, cold", d, c);
    return 0;
}
