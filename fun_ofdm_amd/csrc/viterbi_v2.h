// viterbi_v2.h -- K=7 Viterbi forward pass with TWO frames per wavefront, packed 16-bit metrics.
//
// Same recursion as viterbi_v1.h (viterbi.cpp:208-457, exact uint8 saturation / state-0 renormalisation /
// tie-break), restructured for gfx950 throughput:
//   * lane s = trellis state s of BOTH frames: a VGPR holds the two frames' path metrics as two u16
//     halves, biased by 0xFF00 so that `v_pk_add_u16 ... clamp` saturates exactly where the reference's
//     `_mm_adds_epu8` does (0xFFFF <-> 255); min and compares are bias-invariant;
//   * one `ds_bpermute_b32` pair per step moves both frames' metrics (lane s needs states s>>1, (s>>1)+32);
//   * branch metrics arrive precomputed from the data-symbol kernel as one dword per step
//     (m00,m01,m10,m11); a lane picks its butterfly's class with one `v_perm_b32`;
//   * the 64 decision bits of a step are the lane mask of a `v_cmp` -- already in the reference's
//     decision_t bit order (viterbi.h:36-41) -- and leave the wave through the scalar data cache
//     (`s_store_dwordx2`), so decision traffic costs no vector-memory or VALU issue slots;
//   * renormalisation (on average every ~9 steps per frame) reduces with DPP row operations.
// Chain-back, descrambling and CRC run in a second kernel, one wave per frame (finish_frame_wave).
#pragma once

#include "viterbi_v1.h"

namespace foa {

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t kBias = 0xFF00u;                      // stored metric = metric + kBias (per 16-bit half)
constexpr uint32_t kBias2 = kBias | (kBias << 16);
constexpr uint32_t kRenormThr = kBias + 210u;            // viterbi.cpp:314: renormalise when state 0 > 210

__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b));
    return __builtin_bit_cast(uint32_t, r);
}

template <int J>
__device__ __forceinline__ void sstore_u64(uint64_t *base, uint64_t v)
{
    asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(v), "s"(base), "n"(J * 8) : "memory");
}

struct Fwd2Lane {
    uint32_t sel, flip;
    int src_lo, src_hi;
};

// One trellis step for both frames.  SA/SB: store frame A's / B's decision word (at da[J] / db[J]).
template <int J, bool SA, bool SB>
__device__ __forceinline__ uint32_t fwd2_step(uint32_t M, const uint2 *bml, const Fwd2Lane &c, uint64_t *da, uint64_t *db)
{
    const uint2 w = bml[J];                                              // same address in every lane: LDS broadcast
    const uint32_t m = __builtin_amdgcn_perm(w.y, w.x, c.sel);
    const uint32_t ma = m ^ c.flip, mb = ma ^ 0x003F003Fu;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(c.src_lo, (int)M);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(c.src_hi, (int)M);
    const uint32_t x = pk_add_sat(lo, ma), y = pk_add_sat(hi, mb);
    if (SA) sstore_u64<J>(da, __ballot((y & 0xFFFFu) <= (x & 0xFFFFu)));    // upper predecessor wins ties
    if (SB) sstore_u64<J>(db, __ballot((y >> 16) <= (x >> 16)));
    uint32_t Mn = pk_min(x, y);
    // viterbi.cpp:314-332 per frame: renormalise when the new metric of state 0 (lane 0) exceeds 210
    const ushort2_t over = __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2_t, Mn), __builtin_bit_cast(ushort2_t, kRenormThr | (kRenormThr << 16)));
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, over));
    if (c0) {
        uint32_t v = Mn;
        v = pk_min(v, dpp_mov<FOA_DPP_XOR1>(v));
        v = pk_min(v, dpp_mov<FOA_DPP_XOR2>(v));
        v = pk_min(v, dpp_mov<FOA_DPP_HALF_MIRROR>(v));
        v = pk_min(v, dpp_mov<FOA_DPP_MIRROR>(v));
        uint32_t r = pk_min(pk_min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                            pk_min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
        r -= kBias2;                                                        // per-half minimum, unbiased
        const uint32_t amt = ((c0 & 0xFFFFu) ? (r & 0xFFFFu) : 0u) | ((c0 >> 16) ? (r & 0xFFFF0000u) : 0u);
        Mn -= amt;                                                          // no borrow: every half >= its minimum
    }
    return Mn;
}

template <bool SA, bool SB>
__device__ __forceinline__ uint32_t fwd2_run(uint32_t M, int t0, int nn, const uint2 *bml, const Fwd2Lane &c, uint64_t *dA, uint64_t *dB)
{
    int j = 0;
    for (; j + 8 <= nn; j += 8) {
        uint64_t *da = dA + t0 + j, *db = dB + t0 + j;
        const uint2 *b = bml + j;
        M = fwd2_step<0, SA, SB>(M, b, c, da, db);
        M = fwd2_step<1, SA, SB>(M, b, c, da, db);
        M = fwd2_step<2, SA, SB>(M, b, c, da, db);
        M = fwd2_step<3, SA, SB>(M, b, c, da, db);
        M = fwd2_step<4, SA, SB>(M, b, c, da, db);
        M = fwd2_step<5, SA, SB>(M, b, c, da, db);
        M = fwd2_step<6, SA, SB>(M, b, c, da, db);
        M = fwd2_step<7, SA, SB>(M, b, c, da, db);
    }
    for (; j < nn; j++) M = fwd2_step<0, SA, SB>(M, bml + j, c, dA + t0 + j, dB + t0 + j);
    return M;
}

__global__ __launch_bounds__(64) void k_viterbi_fwd2(const FrameInfo *__restrict__ info, int n_frames, const uint32_t *__restrict__ bm,
                                                     uint64_t *__restrict__ dec)
{
    __shared__ uint2 bml[64];
    const int lane = threadIdx.x;
    const int fA = 2 * blockIdx.x, fB = fA + 1;
    if (fA >= n_frames) return;
    const FrameInfo ia = info[fA];
    FrameInfo ib = ia;
    if (fB < n_frames) ib = info[fB];
    const int TA = ia.nsym > 0 ? ia.nsteps : 0;
    const int TB = (fB < n_frames && ib.nsym > 0) ? ib.nsteps : 0;
    const int T = max(TA, TB), Tboth = min(TA, TB);
    if (T == 0) return;
    const uint32_t *bmA = bm + ia.dec_off, *bmB = bm + ib.dec_off;
    uint64_t *dA = dec + ia.dec_off, *dB = dec + ib.dec_off;

    // lane constants: butterfly i = lane>>1, its Branchtab class (viterbi.cpp:86-91); odd states swap m / 63-m
    const int i = lane >> 1;
    const uint32_t cls = ((__popc((2 * i) & 121) & 1) << 1) | (__popc((2 * i) & 91) & 1);
    Fwd2Lane c;
    c.sel = 0x0C000C00u | ((4u + cls) << 16) | cls;                      // {0, B.byte[cls], 0, A.byte[cls]}
    c.flip = (lane & 1) ? 0x003F003Fu : 0u;
    c.src_lo = i * 4; c.src_hi = (i + 32) * 4;
    uint32_t M = lane == 0 ? kBias2 : kBias2 + 0x003F003Fu;               // viterbi.cpp:71-78

    for (int t0 = 0; t0 < T; t0 += 64) {
        const int nn = min(64, T - t0);
        __builtin_amdgcn_wave_barrier();
        bml[lane] = make_uint2(t0 + lane < TA ? bmA[t0 + lane] : 0u, t0 + lane < TB ? bmB[t0 + lane] : 0u);
        wave_lds_sync();
        // steps where both frames are alive, then the tail of the longer one (all wave-uniform)
        const int nb = max(0, min(nn, Tboth - t0));
        M = fwd2_run<true, true>(M, t0, nb, bml, c, dA, dB);
        if (nb < nn) {
            if (TA > TB) M = fwd2_run<true, false>(M, t0 + nb, nn - nb, bml + nb, c, dA, dB);
            else M = fwd2_run<false, true>(M, t0 + nb, nn - nb, bml + nb, c, dA, dB);
        }
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}

// Chain-back (viterbi.cpp:108-146), descrambler and CRC-32 (ppdu.cpp:256-293), one LANE per frame: the
// recursion is serial per frame and a handful of integer ops per step, so 64 frames share a wave and
// nothing runs on the (CU-shared) scalar unit.
__global__ __launch_bounds__(64) void k_viterbi_finish2(const FrameInfo *__restrict__ info, int n_frames, const uint64_t *__restrict__ dec,
                                                        uint32_t *__restrict__ decoded, uint8_t *__restrict__ psdu, size_t slot_bytes,
                                                        foa_frame_result *__restrict__ results)
{
    __shared__ uint32_t crc_tab[256];
    const int lane = threadIdx.x, f = blockIdx.x * 64 + lane;
    for (int i = lane; i < 256; i += 64) crc_tab[i] = g_tab.crc_table[i];
    __syncthreads();
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0; fi.soft_off = 0; fi.dec_off = 0;
    if (f < n_frames) fi = info[f];
    const bool live = f < n_frames && fi.nsym > 0;
    const int T = live ? fi.nsteps : 0, data_bits = T - 6;
    const uint64_t *dp = dec + fi.dec_off;
    uint32_t *out = decoded + fi.dec_off;                  // T/8 bytes needed; the region holds >= T dwords
    int maxbits = data_bits;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxbits = max(maxbits, __shfl_xor(maxbits, o));
    // chain-back: bit n uses the decision word of step n+6; endstate 0
    uint32_t e = 0, word = 0;
    for (int n = maxbits - 1; n >= 0; n--) {
        if (n < data_bits) {
            const uint64_t w = dp[n + 6];
            const uint32_t k = (uint32_t)(w >> (e >> 2)) & 1u;
            e = (e >> 1) | (k << 7);
            if ((n & 7) == 0) {
                word = (word << 8) | e;
                if ((n & 31) == 0) out[n >> 5] = word;
            }
        }
    }
    // descramble (one LFSR bit per byte, ppdu.cpp:256-264) + CRC over service+payload (ppdu.cpp:267-271)
    const int len = fi.length, ncrc = live ? 2 + len : 0, nwords = live ? (ncrc + 4 + 3) / 4 : 0;
    int maxw = nwords;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxw = max(maxw, __shfl_xor(maxw, o));
    uint32_t crc = 0xFFFFFFFFu, given = 0;
    for (int q = 0; q < maxw; q++) {
        uint32_t scr = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) scr |= (uint32_t)g_tab.scramble[(4 * q + b) % 127] << (8 * b);
        if (q < nwords) {
            const uint32_t d = out[q] ^ scr;
            out[q] = d;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int x = 4 * q + b;
                const uint32_t byte = (d >> (8 * b)) & 0xFFu;
                if (x < ncrc) crc = crc_tab[(crc ^ byte) & 0xFFu] ^ (crc >> 8);
                else if (x < ncrc + 4) given |= byte << (8 * (x - ncrc));
            }
        }
    }
    const bool ok = live && (crc ^ 0xFFFFFFFFu) == given;
    // payload = descrambled bytes [2, 2+len) (ppdu.cpp:283-285), only for frames whose CRC matched
    if (ok) {
        uint8_t *slot = psdu + (size_t)f * slot_bytes;
        const int ncopy = min((size_t)len, slot_bytes);
        int y = 0;
        if ((((uintptr_t)slot) & 3) == 0)
            for (; y + 4 <= ncopy; y += 4) {
                const int q = (y + 2) >> 2;                                  // bytes y+2 .. y+5 straddle words q, q+1
                ((uint32_t *)slot)[y >> 2] = (out[q] >> 16) | (out[q + 1] << 16);
            }
        for (; y < ncopy; y++) slot[y] = (uint8_t)(out[(y + 2) >> 2] >> (8 * ((y + 2) & 3)));
    }
    if (f < n_frames) write_result(&results[f], fi, live ? (ok ? FOA_ST_OK : FOA_ST_CRC_FAIL) : fi.status);
}

inline void launch_viterbi_v2(hipStream_t st, const FrameInfo *info, int nf, const uint32_t *bm, uint64_t *dec, uint32_t *decoded,
                              uint8_t *psdu, size_t slot_bytes, foa_frame_result *results)
{
    hipLaunchKernelGGL(k_viterbi_fwd2, dim3((nf + 1) / 2), dim3(64), 0, st, info, nf, bm, dec);
    hipLaunchKernelGGL(k_viterbi_finish2, dim3((nf + 63) / 64), dim3(64), 0, st, info, nf, dec, decoded, psdu, slot_bytes, results);
}

}  // namespace foa
