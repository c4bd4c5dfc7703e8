// viterbi_v2.h -- K=7 Viterbi forward pass, TWO frames per wavefront, in-place trellis, packed u16 metrics.
//
// Same recursion as viterbi_v1.h (viterbi.cpp:208-457: uint8 saturating metrics, renormalise when state 0
// exceeds 210, upper predecessor wins ties), laid out for gfx950:
//
//   * In-place butterflies.  Physical slot p (= lane) holds state label rotl6^t(p) at trellis time t, so the
//     butterfly of step t pairs slots that differ in ONE lane bit q = 5 - (t mod 6): old states (i, i+32) sit in
//     the pair's low/high slot and the new states (2i, 2i+1) are written back to the same two slots.  Label 0
//     is slot 0 at all times.  The exchange is a lane-xor by 32/16/8/4/2/1, done with v_permlane32_swap,
//     v_permlane16_swap and DPP row/quad moves -- no LDS round trip on the step's critical path.  After the
//     exchange every lane holds (low-slot metric, high-slot metric) in the same two registers, so one formula
//     serves both lanes of a pair; only the branch-metric increments differ per lane (constants per phase).
//   * A VGPR carries the two frames' metrics as u16 halves biased by 0xFF00: `v_pk_add_u16 ... clamp`
//     saturates at 0xFFFF exactly where `_mm_adds_epu8` saturates at 255; min/compare are bias-invariant.
//   * The front end delivers the two soft bytes of every step; when a chunk is staged, lane = step turns its pair into
//     the dword (m00,m01,m10,m11) and in the step a lane selects its butterfly's Branchtab class with one v_perm_b32.
//   * The 64 decision bits of a step are the lane mask of one v_cmp per frame; the mask is parked in lane J of a
//     VGPR pair (plain v_mov under a one-lane EXEC) and a chunk of 60 steps leaves as one coalesced store.  Bit p of the word of step t says
//     "slot p's survivor came from the pair's HIGH slot", which is all the chain-back needs:
//         p <- (p & ~(1<<q)) | (bit << q).
//   * Renormalisation (about every 9th step per frame) reduces with DPP inside rows and four v_readlane.
//   * The scalar unit is shared by a CU's four SIMDs, so per-step scalar work is kept to the renorm test
//     and loop control; chain-back / descramble / CRC run in a second kernel with one LANE per frame.
#pragma once

#include "viterbi_v1.h"

namespace foa {

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t kBias = 0xFF00u;                      // stored metric = metric + kBias (per 16-bit half)
constexpr uint32_t kBias2 = kBias | (kBias << 16);
constexpr uint32_t kRenormThr = kBias + 210u;            // viterbi.cpp:314: renormalise when state 0 > 210
constexpr int kChunk = 60;                               // trellis steps staged per LDS refill (multiple of 6)

__device__ __forceinline__ uint32_t pk_add_sat(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_elementwise_add_sat(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_elementwise_sub_sat(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b));
    return __builtin_bit_cast(uint32_t, r);
}

// 6-bit rotate left
__host__ __device__ constexpr int rotl6(int v, int r)
{
    r %= 6;
    return ((v << r) | (v >> (6 - r))) & 63;
}

#if FOA_XCHECK      // cross-check build only: decision words of the v2 forward pass
// Decision words are collected in VGPRs, lane J holding the word of the chunk's step J, and leave as one
// coalesced 8-byte-per-lane store per chunk.  (Scalar stores would cost no VALU slot at all, but gfx950's
// scalar data cache retires only about one s_store per 10 clocks per CU -- measured, tools/probe_sstore.hip.)
// The word goes into lane J by narrowing EXEC to that lane around plain v_mov_b32 (about 2 clocks each; a
// v_writelane_b32 costs about 4.2).  The kernel runs with all 64 lanes active, so EXEC is restored to -1.
// The two SALU instructions in front also provide the wait states a VALU needs before it may read an SGPR that
// a VALU compare has just written.
struct DecAcc { uint32_t lo, hi; };

template <int J>
__device__ __forceinline__ void dec_put2(DecAcc &a, DecAcc &b, uint64_t va, uint64_t vb)
{
    asm volatile("s_mov_b64 exec, 0\n\ts_bitset1_b64 exec, %8\n\t"
                 "v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(a.lo), "+v"(a.hi), "+v"(b.lo), "+v"(b.hi)
                 : "s"((uint32_t)va), "s"((uint32_t)(va >> 32)), "s"((uint32_t)vb), "s"((uint32_t)(vb >> 32)), "n"(J));
}
template <int J>
__device__ __forceinline__ void dec_put(DecAcc &a, uint64_t v)
{
    asm volatile("s_mov_b64 exec, 0\n\ts_bitset1_b64 exec, %4\n\t"
                 "v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(a.lo), "+v"(a.hi)
                 : "s"((uint32_t)v), "s"((uint32_t)(v >> 32)), "n"(J));
}
__device__ __forceinline__ void dec_put_dyn(DecAcc &a, uint64_t v, int j, int lane)
{
    a.lo = lane == j ? (uint32_t)v : a.lo;
    a.hi = lane == j ? (uint32_t)(v >> 32) : a.hi;
}

// viterbi.cpp:242-247 for the four Branchtab classes of one step: byte 2*b0 + b1 = (((s0 ^ b0*255) + (s1 ^ b1*255) + 1) >> 1) >> 2,
// from the step's soft pair (s0 | s1 << 8)
__device__ __forceinline__ uint32_t bm_word_of_pair(uint32_t pair)
{
    const uint32_t s0 = pair & 255u, s1 = (pair >> 8) & 255u, n0 = s0 ^ 255u, n1 = s1 ^ 255u;
    return ((s0 + s1 + 1u) >> 3) | (((s0 + n1 + 1u) >> 3) << 8) | (((n0 + s1 + 1u) >> 3) << 16) | (((n0 + n1 + 1u) >> 3) << 24);
}

#endif  // FOA_XCHECK

// After this every lane has lo = metric of its pair's low slot, hi = metric of the high slot (pair = lanes that
// differ in bit Q).
template <int Q>
__device__ __forceinline__ void pair_exchange(uint32_t M, uint32_t &lo, uint32_t &hi)
{
    if constexpr (Q == 5) {
        auto r = __builtin_amdgcn_permlane32_swap(M, M, false, false);
        lo = r[0]; hi = r[1];
    } else if constexpr (Q == 4) {
        auto r = __builtin_amdgcn_permlane16_swap(M, M, false, false);
        lo = r[0]; hi = r[1];
    } else if constexpr (Q == 3) {
        lo = (uint32_t)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x118, 0xF, 0xC, false);   // row_shr:8 into lanes 8-15
        hi = (uint32_t)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x108, 0xF, 0x3, false);   // row_shl:8 into lanes 0-7
    } else if constexpr (Q == 2) {
        lo = (uint32_t)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x114, 0xF, 0xA, false);   // row_shr:4 into banks 1,3
        hi = (uint32_t)__builtin_amdgcn_update_dpp((int)M, (int)M, 0x104, 0xF, 0x5, false);   // row_shl:4 into banks 0,2
    } else if constexpr (Q == 1) {
        lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M, 0x44, 0xF, 0xF, true);           // quad_perm [0,1,0,1]
        hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M, 0xEE, 0xF, 0xF, true);           // quad_perm [2,3,2,3]
    } else {
        lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M, 0xA0, 0xF, 0xF, true);           // quad_perm [0,0,2,2]
        hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)M, 0xF5, 0xF, 0xF, true);           // quad_perm [1,1,3,3]
    }
}

#if FOA_XCHECK      // cross-check build only: the v2 forward pass
// per-lane constants of the six phases
struct Fwd2Lane {
    uint32_t sel[6];      // v_perm selector {0, B.byte[cls], 0, A.byte[cls]} of this lane's butterfly class
    uint32_t flip[6];     // 0x003F003F when this lane is the pair's high slot (it adds 63-m on the low branch)
};

__device__ __forceinline__ Fwd2Lane fwd2_lane_init(int lane)
{
    Fwd2Lane c;
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const int q = 5 - ph;
        const int i = rotl6(lane & ~(1 << q), ph);                       // old label of the pair's low slot (< 32)
        const uint32_t cls = ((__popc((2 * i) & 121) & 1) << 1) | (__popc((2 * i) & 91) & 1);   // viterbi.cpp:86-91
        c.sel[ph] = 0x0C000C00u | ((4u + cls) << 16) | cls;
        c.flip[ph] = ((lane >> q) & 1) ? 0x003F003Fu : 0u;
    }
    return c;
}

// One trellis step (phase PH = t mod 6) for both frames.  SA/SB: record frame A's / B's decision word as the
// chunk's step J (J < 0: run-time index jdyn).
template <int PH, int J, bool SA, bool SB>
__device__ __forceinline__ uint32_t fwd2_step(uint32_t M, const uint2 w, const Fwd2Lane &c, DecAcc &accA, DecAcc &accB, int jdyn, int lane)
{
    const uint32_t m = __builtin_amdgcn_perm(w.y, w.x, c.sel[PH]);
    const uint32_t inc_lo = m ^ c.flip[PH], inc_hi = inc_lo ^ 0x003F003Fu;
    uint32_t lo, hi;
    pair_exchange<5 - PH>(M, lo, hi);
    const uint32_t x = pk_add_sat(lo, inc_lo), y = pk_add_sat(hi, inc_hi);
    // high (old MSB = 1) predecessor wins ties; the lane mask of the compare is the step's decision word
    if constexpr (SA && SB && J >= 0) {
        dec_put2<(J >= 0 ? J : 0)>(accA, accB, __ballot((y & 0xFFFFu) <= (x & 0xFFFFu)), __ballot((y >> 16) <= (x >> 16)));
    } else {
        if (SA) {
            const uint64_t d = __ballot((y & 0xFFFFu) <= (x & 0xFFFFu));
            if constexpr (J >= 0) dec_put<(J >= 0 ? J : 0)>(accA, d); else dec_put_dyn(accA, d, jdyn, lane);
        }
        if (SB) {
            const uint64_t d = __ballot((y >> 16) <= (x >> 16));
            if constexpr (J >= 0) dec_put<(J >= 0 ? J : 0)>(accB, d); else dec_put_dyn(accB, d, jdyn, lane);
        }
    }
    uint32_t Mn = pk_min(x, y);
    // viterbi.cpp:314-332 per frame: renormalise when the new metric of state 0 (slot 0 = lane 0) exceeds 210
    const uint32_t c0 = __builtin_amdgcn_readfirstlane(pk_sub_sat(Mn, kRenormThr | (kRenormThr << 16)));
    if (c0) {
        // usually only one of the two frames is due: reduce just that half, as plain u32 (v_min_u32 takes the DPP
        // operand directly, one instruction per round)
        if (c0 & 0xFFFFu) Mn -= wave_min_u32(Mn & 0xFFFFu) - kBias;
        if (c0 >> 16) Mn -= (wave_min_u32(Mn >> 16) - kBias) << 16;
    }
    return Mn;
}

// six steps (one of each phase) whose chunk-relative indices are J0 .. J0+5
template <int J0, bool SA, bool SB>
__device__ __forceinline__ uint32_t fwd2_group(uint32_t M, const uint2 *bml, const Fwd2Lane &c, DecAcc &accA, DecAcc &accB, int lane)
{
    // the six LDS broadcast reads (same address in every lane) are issued together, ahead of the dependent chain
    const uint2 w0 = bml[J0], w1 = bml[J0 + 1], w2 = bml[J0 + 2], w3 = bml[J0 + 3], w4 = bml[J0 + 4], w5 = bml[J0 + 5];
    M = fwd2_step<0, J0 + 0, SA, SB>(M, w0, c, accA, accB, 0, lane);
    M = fwd2_step<1, J0 + 1, SA, SB>(M, w1, c, accA, accB, 0, lane);
    M = fwd2_step<2, J0 + 2, SA, SB>(M, w2, c, accA, accB, 0, lane);
    M = fwd2_step<3, J0 + 3, SA, SB>(M, w3, c, accA, accB, 0, lane);
    M = fwd2_step<4, J0 + 4, SA, SB>(M, w4, c, accA, accB, 0, lane);
    M = fwd2_step<5, J0 + 5, SA, SB>(M, w5, c, accA, accB, 0, lane);
    return M;
}

// a full chunk of kChunk = 60 steps, completely unrolled so that every decision word is parked in a constant lane
template <bool SA, bool SB>
__device__ __forceinline__ uint32_t fwd2_chunk(uint32_t M, const uint2 *bml, const Fwd2Lane &c, DecAcc &accA, DecAcc &accB, int lane)
{
    M = fwd2_group<0, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<6, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<12, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<18, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<24, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<30, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<36, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<42, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<48, SA, SB>(M, bml, c, accA, accB, lane);
    M = fwd2_group<54, SA, SB>(M, bml, c, accA, accB, lane);
    return M;
}

// a single step with run-time index, phase and store flags (chunks in which a frame ends)
__device__ __forceinline__ uint32_t fwd2_step_dyn(uint32_t M, int j, bool sa, bool sb, const uint2 *bml, const Fwd2Lane &c, DecAcc &accA,
                                                  DecAcc &accB, int lane)
{
    const uint2 w = bml[j];
#define FOA_DYN(PH)                                                                          \
    case PH:                                                                                 \
        if (sa && sb) return fwd2_step<PH, -1, true, true>(M, w, c, accA, accB, j, lane);    \
        if (sa) return fwd2_step<PH, -1, true, false>(M, w, c, accA, accB, j, lane);         \
        return fwd2_step<PH, -1, false, true>(M, w, c, accA, accB, j, lane);
    switch (j % 6) {
        FOA_DYN(0) FOA_DYN(1) FOA_DYN(2) FOA_DYN(3) FOA_DYN(4)
    default:
        FOA_DYN(5)
    }
#undef FOA_DYN
}

#endif  // FOA_XCHECK

#ifndef FOA_FWD_WAVES
#define FOA_FWD_WAVES 4
#endif
constexpr int kFwdWaves = FOA_FWD_WAVES;     // waves (frame pairs) per workgroup: whole workgroups spread evenly over a CU's four SIMDs

#if FOA_XCHECK      // cross-check build only
__global__ __launch_bounds__(64 * kFwdWaves) void k_viterbi_fwd2(const FrameInfo *__restrict__ info, int n_frames,
                                                                 const uint16_t *__restrict__ sp, uint64_t *__restrict__ dec)
{
    __shared__ uint2 bml_all[kFwdWaves][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 *bml = bml_all[wave];
    const int fA = 2 * (blockIdx.x * kFwdWaves + wave), fB = fA + 1;
    if (fA >= n_frames) return;
    const FrameInfo ia = info[fA];
    FrameInfo ib = ia;
    if (fB < n_frames) ib = info[fB];
    const int TA = ia.nsym > 0 ? ia.nsteps : 0;
    const int TB = (fB < n_frames && ib.nsym > 0) ? ib.nsteps : 0;
    const int T = max(TA, TB), Tboth = min(TA, TB);
    if (T == 0) return;
    const uint16_t *spA = sp + ia.dec_off, *spB = sp + ib.dec_off;
    uint64_t *dA = dec + ia.dec_off, *dB = dec + ib.dec_off;
    const Fwd2Lane c = fwd2_lane_init(lane);
    uint32_t M = lane == 0 ? kBias2 : kBias2 + 0x003F003Fu;               // viterbi.cpp:71-78 (label 0 = slot 0)

    for (int t0 = 0; t0 < T; t0 += kChunk) {                              // t0 mod 6 == 0: phase = chunk-relative index mod 6
        const int nn = min(kChunk, T - t0);
        __builtin_amdgcn_wave_barrier();
        bml[lane] = make_uint2(t0 + lane < TA ? bm_word_of_pair(spA[t0 + lane]) : 0u, t0 + lane < TB ? bm_word_of_pair(spB[t0 + lane]) : 0u);
        wave_lds_sync();
        DecAcc accA = { 0u, 0u }, accB = { 0u, 0u };
        if (t0 + kChunk <= Tboth) M = fwd2_chunk<true, true>(M, bml, c, accA, accB, lane);
        else if (nn == kChunk && t0 >= TB) M = fwd2_chunk<true, false>(M, bml, c, accA, accB, lane);
        else if (nn == kChunk && t0 >= TA) M = fwd2_chunk<false, true>(M, bml, c, accA, accB, lane);
        else
            for (int j = 0; j < nn; j++) M = fwd2_step_dyn(M, j, t0 + j < TA, t0 + j < TB, bml, c, accA, accB, lane);
        if (t0 + lane < TA && lane < nn) dA[t0 + lane] = ((uint64_t)accA.hi << 32) | accA.lo;
        if (t0 + lane < TB && lane < nn) dB[t0 + lane] = ((uint64_t)accB.hi << 32) | accB.lo;
    }
    // the chain-back kernel reads whole 48-step chunks: the words after a frame's last step, up to the end of its
    // region (dec_words(T)), must read as "no decision" (zero keeps the walk in slot 0)
    for (int i = TA + lane; i < dec_words(TA); i += 64) dA[i] = 0;
    if (TB > 0)
        for (int i = TB + lane; i < dec_words(TB); i += 64) dB[i] = 0;
}

#endif  // FOA_XCHECK

// descramble + CRC-32 + payload copy of one frame per lane; shared by k_viterbi_finish2 and k_tb_finish
struct FinishTables { uint32_t crc[1024]; uint32_t scr[128]; };

__device__ __forceinline__ void finish_tables_init(FinishTables &t, int tid, int nthreads)
{
    for (int i = tid; i < 256; i += nthreads) {
        const uint32_t t0 = g_tab.crc_table[i];
        const uint32_t t1 = (t0 >> 8) ^ g_tab.crc_table[t0 & 0xFFu];
        const uint32_t t2 = (t1 >> 8) ^ g_tab.crc_table[t1 & 0xFFu];
        const uint32_t t3 = (t2 >> 8) ^ g_tab.crc_table[t2 & 0xFFu];
        t.crc[i] = t0; t.crc[256 + i] = t1; t.crc[512 + i] = t2; t.crc[768 + i] = t3;
    }
    for (int i = tid; i < 127; i += nthreads) {
        uint32_t m = 0;
        for (int b = 0; b < 4; b++) m |= (uint32_t)g_tab.scramble[(4 * i + b) % 127] << (8 * b);
        t.scr[i] = m;
    }
}

// LDS of the wave-cooperative finish (one per 64-frame wave)
struct FinishWave {
    uint32_t tile[64][17];         // 16 words of each of the wave's 64 frames (row stride 17: one bank per lane)
    int64_t base[64];              // each frame's offset (in words) into the decoded buffer
    int nwords[64], ncopy[64];     // words to descramble; payload bytes to copy (0 unless the CRC matched)
};

// descramble (one LFSR bit per byte, ppdu.cpp:256-264) + CRC over service+payload (ppdu.cpp:267-271) + payload copy
// (ppdu.cpp:283-285), one frame per lane.  Whole words go through four table look-ups that do not depend on each other
// (slicing-by-4); the scrambler's 127-byte period makes a 127-word table of descrambling masks.
// A lane that walks its own frame's words touches a different cache line than every other lane with each load and
// store, which is what this step used to spend its time on.  So memory is moved by the wave as a whole: four lanes
// per frame fetch 64 contiguous bytes of each of 16 frames per instruction into an LDS tile, each lane then works on
// its own row, and the descrambled words (and later the payload) leave the same way.  Wave-uniform control flow.
__device__ __forceinline__ void finish_crc_psdu(const FinishTables &t, FinishWave &fw, const FrameInfo &fi, bool live, int f, int n_frames,
                                                uint32_t *__restrict__ decoded, uint8_t *__restrict__ psdu, size_t slot_bytes,
                                                foa_frame_result *__restrict__ results)
{
    const int lane = threadIdx.x & 63, sub = lane >> 2, pc = lane & 3;
    const int len = fi.length, ncrc = live ? 2 + len : 0, nwords = live ? (ncrc + 4 + 3) / 4 : 0;
    int maxw = nwords;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxw = max(maxw, __shfl_xor(maxw, o));
    fw.base[lane] = live ? fi.dec_off : 0;
    fw.nwords[lane] = nwords;
    wave_lds_sync();
    uint32_t crc = 0xFFFFFFFFu, given = 0;
    const int full = ncrc >> 2;                                            // words that lie entirely inside the CRC range
    int qs = 0;                                                            // q mod 127
    // in: frame 16 r + sub, words q0 + 4 pc .. + 3 (the regions are 256-byte aligned and padded past their last word);
    // the next 16-word chunk is fetched into registers while the current one is worked on
    uint4 pre[4];
    auto fetch = [&](int q0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            pre[r] = make_uint4(0u, 0u, 0u, 0u);
            if (q0 + 4 * pc < fw.nwords[fr]) pre[r] = *(const uint4 *)(decoded + fw.base[fr] + q0 + 4 * pc);
        }
    };
    fetch(0);
    for (int q0 = 0; q0 < maxw; q0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            fw.tile[fr][4 * pc] = pre[r].x; fw.tile[fr][4 * pc + 1] = pre[r].y; fw.tile[fr][4 * pc + 2] = pre[r].z; fw.tile[fr][4 * pc + 3] = pre[r].w;
        }
        if (q0 + 16 < maxw) fetch(q0 + 16);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int q = q0 + i;
            const uint32_t scr = t.scr[qs];
            qs = qs == 126 ? 0 : qs + 1;
            if (q < nwords) {
                const uint32_t d = fw.tile[lane][i] ^ scr;
                fw.tile[lane][i] = d;
                if (q < full) {
                    const uint32_t c = crc ^ d;
                    crc = t.crc[768 + (c & 0xFFu)] ^ t.crc[512 + ((c >> 8) & 0xFFu)] ^ t.crc[256 + ((c >> 16) & 0xFFu)] ^ t.crc[c >> 24];
                } else {
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int x = 4 * q + b;
                        const uint32_t byte = (d >> (8 * b)) & 0xFFu;
                        if (x < ncrc) crc = t.crc[(crc ^ byte) & 0xFFu] ^ (crc >> 8);
                        else if (x < ncrc + 4) given |= byte << (8 * (x - ncrc));
                    }
                }
            }
        }
        wave_lds_sync();
        // out: the descrambled words, same pieces (words of a piece beyond the frame's last one are pad bits, descrambled or not)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub;
            if (q0 + 4 * pc < fw.nwords[fr])
                *(uint4 *)(decoded + fw.base[fr] + q0 + 4 * pc) =
                    make_uint4(fw.tile[fr][4 * pc], fw.tile[fr][4 * pc + 1], fw.tile[fr][4 * pc + 2], fw.tile[fr][4 * pc + 3]);
        }
        wave_lds_sync();
    }
    const bool ok = live && (crc ^ 0xFFFFFFFFu) == given;
    const bool fits = (size_t)len <= slot_bytes;                           // a payload longer than the caller's slot is reported, never cut short
    if (f < n_frames) write_result(&results[f], fi, live ? (ok ? (fits ? FOA_ST_OK : FOA_ST_NO_SPACE) : FOA_ST_CRC_FAIL) : fi.status);

    // payload = descrambled bytes [2, 2+len) (ppdu.cpp:283-285), only for frames whose CRC matched: 16 bytes per lane,
    // payload bytes 64 c + 16 pc .. + 15 of frame 16 r + sub = bytes 2 .. 17 of the five words from 16 c + 4 pc on
    const int ncopy = ok && fits ? len : 0;
    fw.ncopy[lane] = ncopy;
    int maxc = ncopy;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxc = max(maxc, __shfl_xor(maxc, o));
    __threadfence();                                                       // the words were stored by other lanes of this wave
    wave_lds_sync();
    const int f0 = f - lane;                                               // the wave's first frame
    for (int c = 0; 64 * c < maxc; c++) {
        const int y = 64 * c + 16 * pc;
        uint4 a[4];
        uint32_t a4[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {                                      // all loads of the trip first
            const int fr = 16 * r + sub;
            // unconditional: the words exist for every frame of the wave (a frame that is not copied starts at offset 0
            // and the buffer is far longer than one PSDU), and four loads in flight beat four round trips
            const uint32_t *src = decoded + fw.base[fr] + 16 * c + 4 * pc;
            a[r] = *(const uint4 *)src;
            a4[r] = src[4];
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int fr = 16 * r + sub, nc = fw.ncopy[fr];
            if (y >= nc) continue;
            const uint4 v = make_uint4((a[r].x >> 16) | (a[r].y << 16), (a[r].y >> 16) | (a[r].z << 16), (a[r].z >> 16) | (a[r].w << 16),
                                       (a[r].w >> 16) | (a4[r] << 16));
            uint8_t *dst = psdu + (size_t)(f0 + fr) * slot_bytes + y;
            if (y + 16 <= nc && (((uintptr_t)dst) & 15) == 0) {
                *(uint4 *)dst = v;
            } else {
                const uint32_t w4[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (y + i < nc) dst[i] = (uint8_t)(w4[i >> 2] >> (8 * (i & 3)));
            }
        }
    }
}

#if FOA_XCHECK      // cross-check build only: serial chain-back, one lane per frame
constexpr int kTbChunk = 48;      // chain-back steps per LDS-DMA chunk (multiple of 6 and 8; two chunks = 48 loads in flight)

// Chain-back (viterbi.cpp:108-146) in slot space, descrambler and CRC-32 (ppdu.cpp:256-293), one LANE per frame:
// the recursion is serial per frame and a handful of integer ops per step, so 64 frames share a wave and
// nothing runs on the CU-shared scalar unit.
__global__ __launch_bounds__(64) void k_viterbi_finish2(const FrameInfo *__restrict__ info, int n_frames, const uint64_t *__restrict__ dec,
                                                        uint32_t *__restrict__ decoded, uint8_t *__restrict__ psdu, size_t slot_bytes,
                                                        foa_frame_result *__restrict__ results)
{
    __shared__ FinishTables tabs;
    __shared__ FinishWave fwave;
    __shared__ ulonglong2 tbuf[2][kTbChunk / 2][64];     // two chunks of decision words, [piece][lane] x 16 B
    const int lane = threadIdx.x, f = blockIdx.x * 64 + lane;
    finish_tables_init(tabs, lane, 64);
    __syncthreads();
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0; fi.soft_off = 0; fi.dec_off = 0;
    fi.seg_off = 0;
    if (f < n_frames) fi = info[f];
    const bool live = f < n_frames && fi.nsym > 0;
    const int T = live ? fi.nsteps : 0, data_bits = T - 6;
    const uint64_t *dp = dec + fi.dec_off;
    uint32_t *out = decoded + fi.dec_off;                  // T/8 bytes needed; the region holds >= T dwords
    // Data bit n is the decision bit read at step n+6 (viterbi.cpp:131-142); the walk starts in state 0 = slot 0
    // at time T and follows  p <- (p & ~(1<<q)) | (bit << q),  q = 5 - t mod 6.
    // The walk is a few integer ops per step; what limits this kernel is getting 8 B per step and frame out of HBM
    // with only one wave per 64 frames.  The words are therefore streamed with LDS-DMA (global_load_lds_dwordx4: each
    // lane fetches 16 B = two steps of its own frame straight into LDS, no VGPRs), 48 steps per chunk, two chunks
    // (48 loads, 768 B per frame) in flight while the previous chunk is walked.  Addresses do not depend on the walk.
    // Lanes whose frame is shorter than the wave's longest join at their own top chunk; the words after a frame's
    // last step are zero (written by the forward kernel) and leave the walk parked in slot 0.
    // The slot index is kept as 5 low bits (p) plus the top bit as a predicate (hi5): bit 5 changes only on every
    // sixth step, so the right half of the 64-bit word is picked off the critical path and the walk itself is a
    // 32-bit bit-field extract followed by a shift-or.  Decoded bits are shifted into a 32-bit register, newest
    // (lowest n) last, and turned into four MSB-first bytes by one bit-reverse + byte swap per 32 steps.
    const int top_lane = live ? (T - 1) / kTbChunk * kTbChunk : -kTbChunk;
    int top = top_lane;
#pragma unroll
    for (int o = 32; o; o >>= 1) top = max(top, __shfl_xor(top, o));
    uint32_t p = 0, acc = 0;
    bool hi5 = false;
    // chunk tc -> LDS buffer `which`: piece i = steps tc+2i, tc+2i+1 of every lane's frame
    auto fetch = [&](int tc, int which) {
        // lanes that have not started (or dead lanes) fetch their own first words instead: always inside their region
        const uint64_t *src = dp + (tc <= top_lane ? tc : 0);
        // Issued from inline asm on purpose: hipcc tracks LDS-DMA issued through the builtin and puts `s_waitcnt vmcnt(0)`
        // in front of every later LDS read, which would serialise fetch and walk; the counted wait below is ours.
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&tbuf[which][0][0];
#pragma unroll
        for (int i = 0; i < kTbChunk / 2; i++) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + 2 * i), "s"(lds0 + (uint32_t)i * 1024u)
                         : "memory");
        }
    };
    auto step = [&](uint64_t w, int j, int tc) {                         // j: compile-time position inside the chunk
        const int q = 5 - j % 6;                                           // tc is a multiple of 6
        const uint32_t half = hi5 ? (uint32_t)(w >> 32) : (uint32_t)w;
        const uint32_t k = __builtin_amdgcn_ubfe(half, p, 1);              // offset taken modulo 32
        if (q == 5) hi5 = k != 0;
        else p = (p & ~(1u << q)) | (k << q);
        acc = (acc << 1) | k;
        if (((j - 6) & 7) == 0) {                                          // n = tc + j - 6 is a multiple of 8 (tc is)
            // acc bit i = data bit n+i; the reference packs MSB first: byte n/8+b holds bits n+8b .. n+8b+7, high to low
            const int n = tc + j - 6;
            if ((n & 31) == 0 && n < data_bits) out[n >> 5] = __builtin_bswap32(__builtin_bitreverse32(acc));
        }
    };
    auto walk = [&](int which, int tc, int jlo) {
        if (tc <= top_lane) {
#pragma unroll
            for (int j = kTbChunk - 1; j >= 0; j--)
                if (j >= jlo) {
                    const ulonglong2 v = tbuf[which][j >> 1][lane];
                    step((j & 1) ? v.y : v.x, j, tc);
                }
        }
    };
    if (top >= 0) {
        int tc = top, which = 0;
        fetch(tc, 0);
        for (;;) {
            if (tc >= kTbChunk) {
                fetch(tc - kTbChunk, which ^ 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kTbChunk / 2) : "memory");   // the older chunk has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_wave_barrier();
            walk(which, tc, tc == 0 ? 6 : 0);                              // steps 0..5 carry no data bit
            __builtin_amdgcn_wave_barrier();
            if (tc == 0) break;
            tc -= kTbChunk;
            which ^= 1;
        }
    }
    if (psdu == nullptr) return;                                           // foa_conv_decode: the decoded bits are the result
    finish_crc_psdu(tabs, fwave, fi, live, f, n_frames, decoded, psdu, slot_bytes, results);
}

inline void launch_viterbi_v2(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec, uint32_t *decoded,
                              uint8_t *psdu, size_t slot_bytes, foa_frame_result *results, hipEvent_t between)
{
    hipLaunchKernelGGL(k_viterbi_fwd2, dim3(((nf + 1) / 2 + kFwdWaves - 1) / kFwdWaves), dim3(64 * kFwdWaves), 0, st, info, nf, sp, dec);
    if (between) (void)hipEventRecord(between, st);
    hipLaunchKernelGGL(k_viterbi_finish2, dim3((nf + 63) / 64), dim3(64), 0, st, info, nf, dec, decoded, psdu, slot_bytes, results);
}

#endif  // FOA_XCHECK

}  // namespace foa
