// Issue-cost probe for gfx950: how many SHADER CLOCKS of one SIMD does one wave64 instruction of each class cost, with
// W = 1, 2, 4, 8 waves resident per SIMD (1024 SIMDs x W single-wave workgroups, every wave running 8 independent
// dependency chains)?  Answers the question DESIGN.md 4 hangs on: is the VALU issue roof of the forward pass
// 1024 SIMDs x f / 2 (MI355X_MICROARCH.md: SIMD-32, v_fma_f32 issues over 2 cycles) or / 4, for the instruction
// classes the kernel is made of (v_pk_add_u16 clamp, v_pk_min_u16, v_pk_sub_u16, v_mov_b32_dpp, v_permlane32_swap,
// v_permlane16_swap, v_bfi_b32, v_lshrrev_b32, v_readfirstlane_b32, ds_read_b64)?
// Clocks come from s_memtime inside the kernel (tick = shader cycle, MI355X_MICROARCH.md), so the figure does not
// depend on an assumed frequency; the effective frequency itself is reported as window ticks / wall time (HIP events).
//   hipcc --offload-arch=gfx950 -O3 -o probe_issue.bin tools/probe_issue.hip && ./probe_issue.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP 1500        // window = REP x 1000 shader clocks (~0.65 ms)

enum { V_ADD, V_FMA, V_PK_ADD_CLAMP, V_PK_MIN, V_PK_SUB, V_DPP_QUAD, V_DPP_ROW, V_PERMLANE32, V_PERMLANE16, V_BFI, V_LSHR, V_READFIRST, DS_READ_B64, V_FMA_F64, MIX_FWD, MIX_FWD_CHAIN,
       V_MIN_U16, V_MIN_U32, V_CNDMASK, V_AND, V_SUB_U32, V_MOV, V_PERM, V_ADD_U16, V_MIN_U16_DPP, DS_SWIZZLE, DS_SWIZZLE_MIN, V_READLANE, V_AND_OR, V_CNDMASK_E64, V_ADD_U32_DPP, DS_BPERMUTE };

template <int V>
__global__ __launch_bounds__(256) void k(unsigned long long *out, unsigned seed, int rep)
{
    __shared__ unsigned long long lds[256 * 8];
    unsigned r[8];
    for (int i = 0; i < 8; i++) r[i] = threadIdx.x * 2654435761u + seed + i;
    for (int i = 0; i < 8; i++) lds[threadIdx.x * 8 + i] = r[i];
    unsigned sacc = 0;
    double d[8];
    for (int i = 0; i < 8; i++) d[i] = (double)r[i];
    __syncthreads();
    // Fixed WINDOW instead of fixed work: under the SIMD's oldest-first arbitration waves with equal work finish at very
    // different times and the tail runs at reduced occupancy; here every wave issues until `rep` x 1000 ticks have passed and
    // reports how far it got, so the occupancy is W from the first tick to the last.
    const unsigned long long t0 = __builtin_readcyclecounter(), window = (unsigned long long)rep * 1000ull;
    unsigned long long t1 = t0;
    unsigned done = 0;
    for (; t1 - t0 < window; t1 = __builtin_readcyclecounter(), done++) {
      for (int it = 0; it < 8; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (V == V_ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
            if (V == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(r[i]) : "v"(seed));
            if (V == V_PK_ADD_CLAMP) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(r[i]) : "v"(seed));
            if (V == V_PK_MIN) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            if (V == V_PK_SUB) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
            if (V == V_DPP_QUAD) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
            if (V == V_DPP_ROW) asm volatile("v_mov_b32_dpp %0, %0 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(r[i]));
            if (V == V_PERMLANE32) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
            if (V == V_PERMLANE16) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 7]));
            if (V == V_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[i]) : "s"(0x00010001u << i), "v"(r[(i + 1) & 7]));
            if (V == V_LSHR) asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i]));
            if (V == V_READFIRST) { unsigned s; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(r[i])); sacc += s; }
            if (V == DS_READ_B64) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((threadIdx.x * 8 + i) * 8)); r[i] ^= (unsigned)v; }
            if (V == V_FMA_F64) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[i]));
            if (V == V_MIN_U16) asm volatile("v_min_u16 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            if (V == V_MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            if (V == V_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 1) & 7]) : "vcc");
            if (V == V_CNDMASK_E64) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(0x00ff00ff00ff00ffull));
            if (V == V_AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            if (V == V_SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
            if (V == V_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(r[(i + 1) & 7]));
            if (V == V_PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(0x07030501u));
            if (V == V_ADD_U16) asm volatile("v_add_u16 %0, %0, %1" : "+v"(r[i]) : "v"(seed));
            if (V == V_MIN_U16_DPP) asm volatile("v_min_u16_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
            if (V == DS_SWIZZLE) { asm volatile("ds_swizzle_b32 %0, %0 offset:0x101f\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i])); }
            if (V == DS_SWIZZLE_MIN) { unsigned t; asm volatile("ds_swizzle_b32 %1, %0 offset:0x101f\n\ts_waitcnt lgkmcnt(0)\n\tv_min_u16 %0, %0, %1" : "+v"(r[i]), "=&v"(t)); }
            if (V == V_READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(r[i])); sacc += s; }
            if (V == V_AND_OR) asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(0x01000100u));
            if (V == V_ADD_U32_DPP) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(seed));
            if (V == DS_BPERMUTE) { asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(r[i]) : "v"((threadIdx.x ^ 32) * 4)); }
            if (V == MIX_FWD) {
                // one forward-pass step of viterbi_v3.h as an instruction multiset: 2 DPP moves, 2 clamped adds, sub, bfi, min, readfirstlane
                unsigned lo, hi, x, y, t, s;
                asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
                             : "=&v"(lo), "=&v"(hi) : "v"(r[i]));
                asm volatile("v_pk_add_u16 %0, %2, %4 clamp\n\tv_pk_add_u16 %1, %3, %4 clamp" : "=&v"(x), "=&v"(y) : "v"(lo), "v"(hi), "v"(seed));
                asm volatile("v_pk_sub_u16 %0, %1, %2" : "=&v"(t) : "v"(x), "v"(y));
                asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[(i + 1) & 7]) : "s"(0x01000100u), "v"(t));
                asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r[i]) : "v"(x), "v"(y));
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(r[i]));
                sacc += s;
            }
            if (V == MIX_FWD_CHAIN) {
                // the same eight instructions as ONE dependent chain per wave (r[0] -> r[0]), the renormalisation test's scalar
                // compare-and-branch included: what a lone frame pair's step looks like to the SIMD
                unsigned lo, hi, x, y, t, s;
                asm volatile("v_mov_b32_dpp %0, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
                             : "=&v"(lo), "=&v"(hi) : "v"(r[0]));
                asm volatile("v_pk_add_u16 %0, %2, %4 clamp\n\tv_pk_add_u16 %1, %3, %4 clamp" : "=&v"(x), "=&v"(y) : "v"(lo), "v"(hi), "v"(seed));
                asm volatile("v_pk_sub_u16 %0, %1, %2" : "=&v"(t) : "v"(x), "v"(y));
                asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(r[1]) : "s"(0x01000100u), "v"(t));
                asm volatile("v_pk_min_u16 %0, %1, %2" : "=v"(r[0]) : "v"(x), "v"(y));
                asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s) : "v"(r[0]));
                if (__builtin_expect(((s & 0x00FF00FFu) + 0x002D002Du) & 0x01000100u, 0)) r[0] -= 0x00010001u;
            }
        }
      }
    }
    unsigned a = sacc + seed;
    for (int i = 0; i < 8; i++) a += r[i] + (unsigned)d[i];
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * 4 + (threadIdx.x >> 6); out[3 * w] = t1 - t0; out[3 * w + 1] = done; out[3 * w + 2] = a; }
}

template <int V> void run(const char *name, unsigned long long *d, double per_iter_instr, int rep = REP)
{
    for (int W : { 1, 2, 4, 5, 8 }) {
        const int nw = 256 * 4 * W;      // waves
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float ms = 0;
        std::vector<unsigned long long> h(3 * nw);
        for (int it = 0; it < 2; it++) {
            hipEventRecord(e0);
            k<V><<<nw / 4, 256>>>(d, 7, rep);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipMemcpy(h.data(), d, 3 * nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double ticks = 0, instr = 0, lo = 1e30, hi = 0;
        for (int i = 0; i < nw; i++) {
            ticks += (double)h[3 * i];
            const double n = (double)h[3 * i + 1] * 8 * 8 * per_iter_instr;       // trips x 8 reps x 8 chains x instructions
            instr += n; lo = std::min(lo, n); hi = std::max(hi, n);
        }
        // SIMD clocks per wave-instruction = window / (instructions the SIMD's W waves issued in it)
        const double per_simd = instr / (256.0 * 4), window = ticks / nw;
        printf("%-30s W=%d  %7.3f ms  %5.2f clk per wave-instr per SIMD   (a wave issues one per %5.1f clk; slowest / fastest wave %4.2f)  %4.2f GHz\n", name, W, ms,
               window / per_simd, window / (instr / nw), lo / hi, window / (ms * 1e6));
    }
}

int main()
{
    unsigned long long *d;
    hipMalloc(&d, 3 * 256 * 4 * 8 * sizeof(unsigned long long));
    for (int i = 0; i < 200; i++) k<V_FMA><<<2048, 256>>>(d, 7, 100);      // bring the clocks up before the first row
    hipDeviceSynchronize();
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s  CUs %d  clockRate %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<V_ADD>("v_add_u32", d, 1);
    run<V_FMA>("v_fma_f32", d, 1);
    run<V_PK_ADD_CLAMP>("v_pk_add_u16 clamp", d, 1);
    run<V_PK_MIN>("v_pk_min_u16", d, 1);
    run<V_PK_SUB>("v_pk_sub_u16", d, 1);
    run<V_DPP_QUAD>("v_mov_b32_dpp quad_perm", d, 1);
    run<V_DPP_ROW>("v_mov_b32_dpp row_shr:8", d, 1);
    run<V_PERMLANE32>("v_permlane32_swap_b32", d, 1);
    run<V_PERMLANE16>("v_permlane16_swap_b32", d, 1);
    run<V_BFI>("v_bfi_b32", d, 1);
    run<V_LSHR>("v_lshrrev_b32", d, 1);
    run<V_READFIRST>("v_readfirstlane_b32 (+s_add)", d, 1);
    run<DS_READ_B64>("ds_read_b64 (+v_xor)", d, 1);
    run<V_FMA_F64>("v_fma_f64", d, 1);
    run<V_MIN_U16>("v_min_u16", d, 1);
    run<V_MIN_U32>("v_min_u32", d, 1);
    run<V_CNDMASK>("v_cndmask_b32 (vcc)", d, 1);
    run<V_CNDMASK_E64>("v_cndmask_b32 (sgpr pair)", d, 1);
    run<V_AND>("v_and_b32", d, 1);
    run<V_SUB_U32>("v_sub_u32", d, 1);
    run<V_MOV>("v_mov_b32", d, 1);
    run<V_PERM>("v_perm_b32", d, 1);
    run<V_ADD_U16>("v_add_u16", d, 1);
    run<V_AND_OR>("v_and_or_b32", d, 1);
    run<V_MIN_U16_DPP>("v_min_u16_dpp", d, 1);
    run<V_ADD_U32_DPP>("v_add_u32_dpp", d, 1);
    run<V_READLANE>("v_readlane_b32 (+s_add)", d, 1);
    run<DS_SWIZZLE>("ds_swizzle_b32 (+wait)", d, 1);
    run<DS_SWIZZLE_MIN>("ds_swizzle_b32 + wait + v_min_u16", d, 1);
    run<DS_BPERMUTE>("ds_bpermute_b32 (+wait)", d, 1);
    run<MIX_FWD>("forward-step mix (8 VALU)", d, 8);
    run<MIX_FWD_CHAIN>("forward mix, one chain/wave", d, 8);
    return 0;
}
