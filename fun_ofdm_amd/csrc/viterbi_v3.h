// viterbi_v3.h -- K=7 Viterbi: forward pass with lane-resident decision bits, segment-parallel chain-back.
//
// Same recursion and the same in-place trellis as viterbi_v2.h (two frames per wave, packed u16 metrics, slot
// p = lane holds label rotl6^t(p)).  What changes is where the decision bits go, because on gfx950 the forward
// pass is bound by VALU issue (a wave64 instruction occupies its SIMD for four clocks) and v2 spends six of its
// eighteen VALU instructions per step on turning the compare into a lane mask and parking it in a VGPR lane:
//
//   * The decision of slot p stays in lane p.  `x - y` as a packed 16-bit subtract has the decision of both
//     frames in the sign bits of its halves, and since |x - y| <= 255 bits 8..15 of a half all equal that sign: one
//     v_bfi_b32 files them at bit 8 + (step mod 8) of a packed accumulator (2 VALU for both frames).  Two
//     accumulators make a 16-step block: one v_perm_b32, then one 16-bit store per lane and frame.
//     Decision memory is therefore TRANSPOSED relative to v2: u16 [block of 16 steps][slot], same 8 bytes per
//     step and frame.  Stored bit = 1 means "survivor came from the pair's LOW slot" (the complement of v2's bit),
//     and slot p's word sits at index 63 - p, so that a chain-back that keeps the complemented slot index
//     pbar = 63 - p simply copies the bit it reads:   pbar <- (pbar & ~(1<<q)) | (bit << q),  q = 5 - n mod 6.
//     Steps 0..5 of the trellis carry no data bit (viterbi.cpp:131-142) and are not recorded at all: "data step"
//     n = t - 6 is the index used from here on; block b holds data steps 16b .. 16b+15.
//   * Branch-metric increments cost no VALU in the step: when a chunk's soft pairs are staged in LDS, each step
//     gets all eight (Branchtab class, side of the butterfly) variants of the packed increment pair, and a lane
//     reads the 8 bytes of its variant (v2: one v_perm + two v_xor per step; half the LDS bytes of reading both
//     frames' dwords and their complements).
//   * The renormalisation test reads state 0 with one v_readfirstlane and decides on the scalar unit.
//
// Chain-back.  With the decisions transposed, the address of the bit a path needs depends on the path, so the
// serial walk of one frame cannot be fed ahead of time.  Instead the walk is cut into SEGMENTS of S data steps,
// one LANE per segment, all segments of all frames at once (k_tb_walk).  A lane does not know the state at the
// top of its segment, so it starts L steps higher in an arbitrary state and discards those bits; survivor paths
// merge quickly, so after L steps it is on the true path -- almost always.  "Almost" is not bit-exact, so each
// lane records the state it assumed at its top boundary (e) and the state it reached at its bottom boundary (s);
// k_tb_finish checks e(k) == s(k+1) down each frame (the top segment starts from the true terminal state, so by
// induction every segment that passes is the true path) and re-walks a segment from the proven state when the
// check fails.  That keeps the result identical to the serial chain-back whatever L is; L only trades overlap
// work against the frequency of re-walks (tests run with L = 0, where nearly every segment is re-walked).
#pragma once

#include "viterbi_v2.h"

namespace foa {

#ifndef FOA_ABL
#define FOA_ABL 0        // timing experiments only (tools/ablate.sh): 1 no decision ops, 2 no renormalisation test, 4 no LDS reads of the
                         // increments, 8 no lane exchange, 16 no decision stores, 32 test without the renormalisation itself,
                         // 64..256 made-up events, 512 four extra taken branches per event, 1024 one extra reduction per event
#endif
#ifndef FOA_EXP
#define FOA_EXP 0        // timing experiments only (tools/exp_chainback.sh; results are wrong by design): 1 the chain-back walk is not launched, 2 / 4 one / two
                         // extra VALU instructions in every forward step -- the bounds on what a chain-back fused into the forward pass could gain and must cost
#endif
constexpr int kChunk3 = 48;                  // data steps per forward chunk: 3 decision blocks, 8 phase groups
constexpr int kTbBlockBytes = 8 * 1024;      // LDS of one 16-step decision block of the wave's 64 lanes: [lane / 8][lane % 8] x 128 B
constexpr int kTbRing = 3;                   // blocks resident per wave: one being walked, two streaming in
constexpr int kTbMaxSeg = 3072;              // largest segment length (LDS of the re-walk path: 8 B per step)

__device__ __forceinline__ uint32_t pk_sub_wrap(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_bit_cast(ushort2_t, a) - __builtin_bit_cast(ushort2_t, b);
    return __builtin_bit_cast(uint32_t, r);
}

#ifndef FOA_ADD_FIRST
#define FOA_ADD_FIRST 1
#endif
#ifndef FOA_RN_PRIO
#define FOA_RN_PRIO 1      // priority of a wave while it renormalises (0: unchanged; profiles/r03_ab_renorm_prio.txt)
#endif
#ifndef FOA_WALK_PRIO
#define FOA_WALK_PRIO 0    // priority of the chain-back walk's waves (A/B only; profiles/r03_ab_renorm_prio.txt)
#endif
#ifndef FOA_FWD_PRIO
#define FOA_FWD_PRIO 0     // priority of the forward pass's waves (A/B only: 1 .. 3, 4 + p = p and 3 over the last third of a frame; profiles/r03_ab_fwd_prio.txt)
#endif
#if FOA_RN_PRIO && (FOA_FWD_PRIO & 4)
#error "FOA_FWD_PRIO 4 + p (priority 3 over the last third of a frame) and FOA_RN_PRIO do not combine: a renormalisation would end at priority p"
#endif
#ifndef FOA_MIN16
#define FOA_MIN16 1
#endif
// Smallest metric of one frame of the packed register, wave-uniform.  High half: the unsigned 32-bit minimum of the packed
// words has the smallest high half in its high half, so the register goes into the reduction as it is and the shift happens
// on the scalar side.  Low half: v_min_u16 compares low halves only and takes the DPP operand like v_min_u32 does, so no
// mask is needed either.  (s_nop: a DPP operand written by the instruction before needs two wait states; the compiler
// does not look into an asm block.)
__device__ __forceinline__ uint32_t wave_min_hi16_word(uint32_t v)      // the smallest high half, still in the high half of the result
{
    uint32_t r;
    asm("s_nop 1\n\t"
        "v_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return __builtin_amdgcn_readlane(r, 63);
}
__device__ __forceinline__ uint32_t wave_min_lo16(uint32_t v)
{
#if FOA_MIN16
    uint32_t r;
    asm("s_nop 1\n\t"
        "v_min_u16_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_min_u16_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "=&v"(r) : "v"(v));
    return __builtin_amdgcn_readlane(r, 63) & 0xFFFFu;
#else
    return wave_min_u32(v & 0xFFFFu);
#endif
}

struct Fwd3Lane {
    uint32_t ofs[6];      // byte offset of this lane's variant inside a step's 64-byte staging entry: 16 cls + 8 side
};

__device__ __forceinline__ Fwd3Lane fwd3_lane_init(int lane)
{
    Fwd3Lane c;
#pragma unroll
    for (int ph = 0; ph < 6; ph++) {
        const int q = 5 - ph;
        const int i = rotl6(lane & ~(1 << q), ph);                       // old label of the pair's low slot (< 32)
        const uint32_t cls = ((__popc((2 * i) & 121) & 1) << 1) | (__popc((2 * i) & 91) & 1);   // viterbi.cpp:86-91
        c.ofs[ph] = 16u * cls + 8u * ((lane >> q) & 1);                   // the pair's high slot adds 63-m on the low branch
    }
    return c;
}

// Renormalisation of viterbi.cpp:314-332 per frame, given s0 = the packed metrics of state 0 (slot 0 = lane 0): subtract the
// frame's smallest metric when state 0 exceeds 210.  Stored halves are 0xFF00 + metric: adding 45 to both halves at once
// carries out of a half iff its metric exceeds 210 (bit 15 / bit 31 of the sum then reads 0).  A carry out of the low half
// can push a high half of exactly 210 over, so the cold path looks at both halves again; the common path only needs
// "nothing is due" (two scalar instructions in front of the branch).
__device__ __forceinline__ bool fwd3_due(uint32_t s0)
{
    return (~(s0 + 0x002D002Du) & 0x80008000u) != 0u;
}
// Both frames' state 0 at or below 147: the NEXT step cannot bring either above 210 (new[0] <= old[0] + m, m <= 63: viterbi.cpp:242-247), so that step
// needs no test.  Adding 108 carries out of a half iff its metric exceeds 147 (as fwd3_due; a carry out of the low half can only make the answer "no").
#ifndef FOA_TEST_SKIP
#define FOA_TEST_SKIP 0     // A/B only.  1: a step whose predecessor showed both state-0 metrics <= 147 (a third of config 2's steps) runs without the renormalisation
                            // test -- no v_readfirstlane; the bookkeeping is scalar.  Bit-exact, and SLOWER: the pass alone 1.00 -> 1.15 ms, a lone wave +16 %, the
                            // pipelined step +8.5 % (profiles/r04_ab_test_skip.txt): the branch around the test is taken on a third of the steps, and a taken branch
                            // costs a wave more than the vector instruction it saves (round 1 found the same on another kernel)
#endif
__device__ __forceinline__ uint32_t fwd3_safe(uint32_t s0)
{
    return (~(s0 + 0x006C006Cu) & 0x80008000u) == 0u ? 1u : 0u;
}
__device__ __forceinline__ uint32_t fwd3_renorm(uint32_t Mn, uint32_t &s0)
{
#if FOA_RN_PRIO
    __builtin_amdgcn_s_setprio(FOA_RN_PRIO);      // the wave's recursion stands still until this is through
#endif
    if constexpr (FOA_ABL & 512) asm volatile("s_branch 0\n\ts_branch 0\n\ts_branch 0\n\ts_branch 0");      // what do four more taken branches cost an event?
    if constexpr (FOA_ABL & 1024) { uint32_t z = wave_min_lo16(Mn ^ s0); asm volatile("" :: "s"(z)); }   // ... and one more reduction?
    // which halves: the low one iff bit 15 of s0 + 0x002D002D reads 0 (nothing carries into it), the high one by comparing it
    // alone; the amount is wave-uniform, so the bias comes off on the scalar side and the vector side is one v_sub per half
    // (the scalar subtractions are asm so that they stay scalar: the compiler reassociates Mn - (mn - bias) into two vector ops)
    if (!((s0 + 0x002D002Du) & 0x8000u)) {
        uint32_t adj;
        asm("s_sub_u32 %0, %1, %2" : "=s"(adj) : "s"(wave_min_lo16(Mn)), "s"(kBias) : "scc");
        Mn -= adj;
        s0 -= adj;                                    // (state 0 as the next step will find it: fwd3_safe)
    }
    if (s0 >= ((kRenormThr + 1u) << 16)) {
        uint32_t adj;
        asm("s_and_b32 %0, %1, 0xffff0000\n\ts_sub_u32 %0, %0, %2" : "=s"(adj) : "s"(wave_min_hi16_word(Mn)), "s"(kBias << 16) : "scc");
        Mn -= adj;
        s0 -= adj;
    }
#if FOA_RN_PRIO
    __builtin_amdgcn_s_setprio(FOA_FWD_PRIO & 3);      // back to the wave's own priority (0 in the product; the A/B builds with FOA_FWD_PRIO keep theirs)
#endif
    return Mn;
}

// Exchange + add-compare-select of one trellis step (phase PH) for both frames, decision bits filed.  J >= 0: data step J
// of the chunk (compile time); J == -1: no decision is recorded (trellis steps 0..5); J == -2: data step jdyn (run time).
// Decision bits: x and y lie in [0xFF00, 0xFFFF], so the 16-bit difference x - y lies in [-255, 255] and its bits
// 8..15 ALL equal its sign.  An accumulator therefore takes eight steps in bits 8..15 of each half with no shift at
// all -- v_pk_sub_u16 + v_bfi_b32 under the mask 0x01000100 << (step mod 8) --, and the two accumulators of a 16-step
// block are merged by one v_perm_b32 when the block is stored (2 + 1/16 instead of 3 VALU instructions per step).
#ifndef FOA_ACC2
#define FOA_ACC2 1       // 1: two decision accumulators, a 16-step block's word is formed (and its store queued) as soon as the block is complete;
                         // 0: six accumulators, a chunk's three words formed at its end (the round-2 arrangement, kept for A/B)
#endif
constexpr int fwd3_acc_index(int j) { return FOA_ACC2 ? (j >> 3) & 1 : j >> 3; }

#ifndef FOA_FILE_LATE
#define FOA_FILE_LATE 1   // 1: the two decision instructions are issued BEHIND the v_readfirstlane of the renormalisation test, in the shadow of its
                          // way to the scalar unit, instead of in front of it (they do not depend on it; a wave issues in order)
#endif
template <int PH, int J>
__device__ __forceinline__ uint32_t fwd3_acs(uint32_t M, const uint2 w, uint32_t (&acc)[6], int jdyn, uint32_t &xo, uint32_t &yo)
{
    const uint32_t inc_lo = w.x, inc_hi = w.y;
    uint32_t x, y;
#if FOA_ADD_FIRST
    if constexpr (PH < 2) {
        // The two phases that exchange with v_permlane32/16_swap add FIRST: every lane adds both of its increments to its OWN
        // metric -- the pair's low lane forms (x of the low lane, x of the high lane), the high lane (y of the low lane, y of the
        // high lane): with the staging entry's layout those are the very (first, second) words each lane reads anyway -- and the
        // swap then hands each lane its x and y.  Saturation is per sum, so the order does not matter; what it saves is the
        // register copy the swap needs when it runs on the metrics themselves (an atomic exchange of two registers, unlike the
        // two masked DPP moves of the other phases, which would overwrite each other's source).
        const uint32_t p = pk_add_sat(M, inc_lo), q = pk_add_sat(M, inc_hi);
        if constexpr (PH == 0) { auto r = __builtin_amdgcn_permlane32_swap(p, q, false, false); x = r[0]; y = r[1]; }
        else { auto r = __builtin_amdgcn_permlane16_swap(p, q, false, false); x = r[0]; y = r[1]; }
    } else
#endif
    {
        uint32_t lo, hi;
        if constexpr (FOA_ABL & 8) { lo = M; hi = M ^ 0x00010001u; }
        else pair_exchange<5 - PH>(M, lo, hi);
        x = pk_add_sat(lo, inc_lo); y = pk_add_sat(hi, inc_hi);
    }
    xo = x; yo = y;
    // upper predecessor wins ties (viterbi.cpp): survivor = low slot iff x < y iff the 16-bit difference is negative
    if constexpr (FOA_FILE_LATE && J >= 0) {
        // (filed by fwd3_step, behind the test's v_readfirstlane)
    } else if constexpr (J >= 0 && !(FOA_ABL & 1)) {
        // Written as one volatile block: left to itself the compiler sinks these instructions of all 48 steps
        // to the end of the chunk and keeps every step's x and y alive until then (148 VGPRs, 3 waves per SIMD).
        // (Filing them one step later instead, into the gaps between that step's exchange, adds and min: no change.)
        constexpr uint32_t m = 0x01000100u << (J & 7);
        uint32_t tmp;
        asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m));
    } else if constexpr (J == -2) {
        const int a = fwd3_acc_index(jdyn);
        const uint32_t m = 0x01000100u << (jdyn & 7);
        const uint32_t tmp = pk_sub_wrap(x, y);
#pragma unroll
        for (int b = 0; b < (FOA_ACC2 ? 2 : 6); b++) acc[b] = a == b ? ((acc[b] & ~m) | (tmp & m)) : acc[b];
    }
    return pk_min(x, y);
}

// One trellis step: exchange + ACS, then the renormalisation test on its result (v_readfirstlane of state 0, the scalar
// test, the branch).  (Measured and dropped: doing the next step on the metrics as they are while the scalar unit tests this
// step's state 0, and repeating that step from the renormalised metrics in the 2 cases in 9 where the test fires.  It takes the
// readfirstlane -> scalar -> branch chain off a lone wave's critical path, yet a lone wave's 8 424-step frame went from 0.639 to
// 0.616 ms only -- what a lone wave waits for is the chain of dependent VALU instructions itself, about 14 clocks each -- and
// at five waves per SIMD the repeated steps cost 11 %.)
template <int PH, int J>
__device__ __forceinline__ uint32_t fwd3_step(uint32_t M, const uint2 w, uint32_t (&acc)[6], int jdyn, uint32_t &skip)
{
    uint32_t x, y;
    uint32_t Mn = fwd3_acs<PH, J>(M, w, acc, jdyn, x, y);
    if constexpr (FOA_FILE_LATE && J >= 0 && !(FOA_ABL & 1) && !(FOA_ABL & 2)) {
        constexpr uint32_t m = 0x01000100u << (J & 7);
        uint32_t tmp;
        if constexpr (FOA_TEST_SKIP) {
            // the decisions first (ONE block for both ways through the step: two copies of it make the compiler shuffle the accumulator between them), then
            // the test -- unless the step before left both state-0 metrics at or below 147: then nothing can be due now and no v_readfirstlane is issued
            asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m));
            if (skip) { skip = 0u; return Mn; }                 // (wave-uniform)
            uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn);
            if (__builtin_expect(fwd3_due(s0), 0)) Mn = fwd3_renorm(Mn, s0);
            skip = fwd3_safe(s0);
            return Mn;
        }
        uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn);
        // (s0 is named as an input only to keep the block behind the v_readfirstlane)
        asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m), "s"(s0));
        if constexpr (FOA_EXP & 2) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(acc[5]) : "s"(0u), "v"(Mn));
        if constexpr (FOA_EXP & 4) asm volatile("v_bfi_b32 %0, %1, %2, %0\n\tv_bfi_b32 %0, %1, %2, %0" : "+v"(acc[5]) : "s"(0u), "v"(Mn));
        if (__builtin_expect(fwd3_due(s0), 0)) Mn = fwd3_renorm(Mn, s0);
        return Mn;
    }
    skip = 0u;                                        // (steps without a recorded decision, the run-time steps of a last partial chunk, ablation builds: always tested)
    if constexpr (FOA_FILE_LATE && J >= 0 && !(FOA_ABL & 1)) {           // (ablation build without the test: file in line)
        constexpr uint32_t m = 0x01000100u << (J & 7);
        uint32_t tmp;
        asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m));
    }
    if constexpr (FOA_EXP & 2) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(acc[5]) : "s"(0u), "v"(Mn));          // (mask 0: acc[5] unchanged)
    if constexpr (FOA_EXP & 4) asm volatile("v_bfi_b32 %0, %1, %2, %0\n\tv_bfi_b32 %0, %1, %2, %0" : "+v"(acc[5]) : "s"(0u), "v"(Mn));
    if constexpr (FOA_ABL & 2) return Mn;
    uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn);
    if constexpr (FOA_ABL & 32) { if (__builtin_expect(s0 == 0x12345678u, 0)) Mn = fwd3_renorm(Mn, s0); return Mn; }
    if constexpr (FOA_ABL & 64) {               // events at a made-up 1 step in 8; 128: the cold path without the reduction; 256: low half only
        if (__builtin_expect(((s0 >> 1) & 7u) == 0u, 0)) {
            if constexpr (FOA_ABL & 128) { uint32_t z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); Mn -= z; }
            else if constexpr (FOA_ABL & 256) { uint32_t z = 0xFF00FFD3u; Mn = fwd3_renorm(Mn, z); }
            else { uint32_t z = (s0 & 16u) ? 0xFF00FFD3u : 0xFFD3FF00u; Mn = fwd3_renorm(Mn, z); }
        }
        return Mn;
    }
    if (__builtin_expect(fwd3_due(s0), 0)) Mn = fwd3_renorm(Mn, s0);      // cold: keeps the common path free of taken branches
    return Mn;
}

// Staging entry of a step: for each Branchtab class c, 16 bytes {lo, hi, hi, lo} with lo = (A.m[c], B.m[c]) as u16
// halves and hi = lo ^ 0x003F003F (63 - m); the pair's low slot reads (lo, hi) at +0, its high slot (hi, lo) at +8.
__device__ __forceinline__ uint2 fwd3_inc(const uint4 *bml, int e, uint32_t ofs)
{
    if constexpr (FOA_ABL & 4) return make_uint2(ofs + e, ofs ^ e);
    return *(const uint2 *)((const uint8_t *)bml + 64 * e + ofs);
}

// six steps (one of each phase) on staging entries E0 .. E0+5
// (Measured and dropped: issuing the LDS reads of group G + 1 before the steps of group G, so that only a chunk's first group
// waits out an LDS round trip: 10 more VGPRs, no change in time at five waves per SIMD, 1.5 % for a lone wave.)
struct Fwd3NoFlush { __device__ __forceinline__ void operator()(int) const {} };

// flush(blk): called right behind the step that completes the 16-step block blk of the chunk (FOA_ACC2)
template <int E0, int J0, typename Flush = Fwd3NoFlush>
__device__ __forceinline__ uint32_t fwd3_group(uint32_t M, const uint4 *bml, const Fwd3Lane &c, uint32_t (&acc)[6], uint32_t &skip, const Flush &flush = Flush())
{
    const uint2 w0 = fwd3_inc(bml, E0 + 0, c.ofs[0]), w1 = fwd3_inc(bml, E0 + 1, c.ofs[1]), w2 = fwd3_inc(bml, E0 + 2, c.ofs[2]),
                w3 = fwd3_inc(bml, E0 + 3, c.ofs[3]), w4 = fwd3_inc(bml, E0 + 4, c.ofs[4]), w5 = fwd3_inc(bml, E0 + 5, c.ofs[5]);
    M = fwd3_step<0, (J0 < 0 ? J0 : J0 + 0)>(M, w0, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 0) & 15) == 15) flush((J0 + 0) >> 4);
    M = fwd3_step<1, (J0 < 0 ? J0 : J0 + 1)>(M, w1, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 1) & 15) == 15) flush((J0 + 1) >> 4);
    M = fwd3_step<2, (J0 < 0 ? J0 : J0 + 2)>(M, w2, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 2) & 15) == 15) flush((J0 + 2) >> 4);
    M = fwd3_step<3, (J0 < 0 ? J0 : J0 + 3)>(M, w3, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 3) & 15) == 15) flush((J0 + 3) >> 4);
    M = fwd3_step<4, (J0 < 0 ? J0 : J0 + 4)>(M, w4, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 4) & 15) == 15) flush((J0 + 4) >> 4);
    M = fwd3_step<5, (J0 < 0 ? J0 : J0 + 5)>(M, w5, acc, 0, skip);
    if constexpr (FOA_ACC2 && J0 >= 0 && ((J0 + 5) & 15) == 15) flush((J0 + 5) >> 4);
    __builtin_amdgcn_sched_barrier(0);          // keep the next groups' LDS reads from being hoisted (registers)
    return M;
}

__device__ __forceinline__ uint32_t fwd3_step_dyn(uint32_t M, int j, const uint4 *bml, const Fwd3Lane &c, uint32_t (&acc)[6], uint32_t &skip)
{
    switch (j % 6) {
    case 0: return fwd3_step<0, -2>(M, fwd3_inc(bml, j, c.ofs[0]), acc, j, skip);
    case 1: return fwd3_step<1, -2>(M, fwd3_inc(bml, j, c.ofs[1]), acc, j, skip);
    case 2: return fwd3_step<2, -2>(M, fwd3_inc(bml, j, c.ofs[2]), acc, j, skip);
    case 3: return fwd3_step<3, -2>(M, fwd3_inc(bml, j, c.ofs[3]), acc, j, skip);
    case 4: return fwd3_step<4, -2>(M, fwd3_inc(bml, j, c.ofs[4]), acc, j, skip);
    default: return fwd3_step<5, -2>(M, fwd3_inc(bml, j, c.ofs[5]), acc, j, skip);
    }
}

// five resident waves per SIMD (10 000 frames = 4.9 waves per SIMD on 256 CUs): cap the register budget accordingly
__global__ __launch_bounds__(64 * kFwdWaves) void k_viterbi_fwd3(const FrameInfo *__restrict__ info, int n_frames,
                                                                 const uint16_t *__restrict__ sp, uint64_t *__restrict__ dec)
{
    __shared__ uint4 bml_all[kFwdWaves][4 * kChunk3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 *bml = bml_all[wave];
#if FOA_FWD_PRIO
    __builtin_amdgcn_s_setprio(FOA_FWD_PRIO & 3);      // (A/B: the forward pass's waves ahead of its guests' at the issue arbiter; 4 + p: p, and 3 over the last third of the frame)
#endif
    const int fA = 2 * (blockIdx.x * kFwdWaves + wave), fB = fA + 1;
    if (fA >= n_frames) return;
    const FrameInfo ia = info[fA];
    FrameInfo ib = ia;
    if (fB < n_frames) ib = info[fB];
    // (wave-uniform by construction; saying so lets the compiler keep the loop control and the test counter on the scalar unit)
    const int TA = __builtin_amdgcn_readfirstlane(ia.nsym > 0 ? ia.nsteps : 0);
    const int TB = __builtin_amdgcn_readfirstlane((fB < n_frames && ib.nsym > 0) ? ib.nsteps : 0);
    if (max(TA, TB) == 0) return;
    const int NA = max(TA - 6, 0), NB = max(TB - 6, 0), N = max(NA, NB);             // data steps
    const int NAtop = (NA + kChunk3 - 1) / kChunk3 * kChunk3, NBtop = (NB + kChunk3 - 1) / kChunk3 * kChunk3;
    auto uniform64 = [](int64_t v) {
        return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v));
    };
    const int64_t offA = uniform64(ia.dec_off), offB = uniform64(ib.dec_off);     // scalar bases: addresses become base + lane
    const uint16_t *spA = sp + offA, *spB = sp + offB;
    uint16_t *dA = (uint16_t *)(dec + offA), *dB = (uint16_t *)(dec + offB);
    const Fwd3Lane c = fwd3_lane_init(lane);
    uint32_t M = lane == 0 ? kBias2 : kBias2 + 0x003F003Fu;               // viterbi.cpp:71-78 (label 0 = slot 0)
    uint32_t acc[6] = { 0u, 0u, 0u, 0u, 0u, 0u };

    // The soft pairs (s0 | s1 << 8, as the front end left them: 2 bytes per step, half of what metric words would move) of
    // trellis steps t0 .. t0+47 (lane = step) are fetched one chunk ahead into registers, so a chunk never waits for HBM;
    // put() turns them into the branch metrics of viterbi.cpp:242-247 for both frames at once --
    //   (((s0 ^ b0*255) + (s1 ^ b1*255) + 1) >> 1) >> 2  =  (s0+s1+1)>>3, (s0-s1+256)>>3, (s1-s0+256)>>3, (511-s0-s1)>>3
    // for (b0,b1) = 00, 01, 10, 11, frame A in the low half, frame B in the high half -- and into staging entries 0 .. cnt-1.
    // (Measured: these ~17 extra VALU and 4 extra LDS instructions per 48 steps cost the kernel 4.7 %; hand-placed
    // ds_write2_b32 pairs instead of the compiler's stores, or v_perm byte gathers, made it slower still.)
    uint32_t pa = 0, pb = 0;
#ifndef FOA_TRIM
#define FOA_TRIM 1       // 1: the chunk's loads and stores are buffer instructions (scalar descriptor + per-lane offset + scalar chunk offset): no vector
                         // instruction forms an address or tests a bound (round 4); 0: flat addresses, the compiler's arithmetic
#endif
#if FOA_TRIM
    // Soft pairs come in and decision words go out through BUFFER instructions: base address and extent in a scalar resource descriptor, the
    // lane's byte offset in one VGPR that never changes, the chunk's position in a scalar offset -- no vector instruction forms an
    // address (the flat-address form spent nine per chunk on it, and three registers), and a lane whose step lies beyond its frame's last
    // one needs no test: a load outside the descriptor's extent returns zero (the soft pair of "no more steps").  The compiler's own
    // builtins, so that its wait-count bookkeeping sees the loads (a first version issued global_load ... saddr from inline asm: put()
    // read the registers before the data had landed, every chain-back segment was re-walked, 8 ms per step).
    constexpr int kRsrcFlags = 0x00020000;          // gfx9 raw buffer, dword 3: DATA_FORMAT 32 (as composable_kernel's CK_BUFFER_RESOURCE_3RD_DWORD for gfx90a / gfx94x / gfx950)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)spA, 0, 2 * TA, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)spB, 0, 2 * TB, kRsrcFlags);
    const int lane2 = 2 * lane;
    auto get = [&](int t0) {
        // soft pairs of steps t0 .. t0+47 of both frames (lanes 48..63 fetch the next sixteen, unused); beyond a frame's end: 0
        pa = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsA, lane2, 2 * t0, 0);
        pb = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsB, lane2, 2 * t0, 0);
    };
#else
    auto get = [&](int t0) {
        pa = (lane < kChunk3 && t0 + lane < TA) ? (uint32_t)spA[t0 + lane] : 0u;
        pb = (lane < kChunk3 && t0 + lane < TB) ? (uint32_t)spB[t0 + lane] : 0u;
    };
#endif
    const uint32_t bml_addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)bml + 64u * (uint32_t)lane;
    auto put = [&](int cnt) {
        __builtin_amdgcn_wave_barrier();
        if (lane < cnt) {
            const uint32_t s0 = (pa & 255u) | ((pb & 255u) << 16), s1 = (pa >> 8) | ((pb >> 8) << 16);     // packed halves (A, B), each <= 255
            const uint32_t P = s0 + s1, D = s0 - s1 + 0x01000100u;                                              // <= 510, 1 .. 511 per half: no carries
            // (one 16-byte store per class, spelled out: left to itself the compiler breaks these into 4- and 8-byte stores at
            // a 64-byte lane stride, which the LDS serves sixteen lanes to a bank; one class at a time, fenced, so that the four
            // vectors share their registers: this block, not the steps, was the kernel's register peak)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (uint32_t cls = 0; cls < 4; cls++) {
                const uint32_t t = cls == 0 ? P + 0x00010001u : cls == 1 ? D : cls == 2 ? 0x02000200u - D : 0x01FF01FFu - P;
                const uint32_t lo = (t >> 3) & 0x003F003Fu, hi = lo ^ 0x003F003Fu;
                const u32x4 v = { lo, hi, hi, lo };
                asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(bml_addr), "v"(v), "n"(16 * cls) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the stores above are not the compiler's to wait for)
        wave_lds_sync();
    };
    get(0);
    put(6);
    get(6);
    uint32_t skip = 0u;
    M = fwd3_group<0, -1>(M, bml, c, acc, skip);
    // The decision words of a chunk leave one chunk LATE (FOA_LATE_ST = 1), right behind the loads that fetch the soft pairs
    // of the chunk after: vector loads and stores share one counter on gfx950 and complete out of order with respect to each
    // other, so the wait for those loads is a wait for everything -- and with the stores queued at the end of a chunk it was
    // a wait for six stores issued a moment ago, at every chunk.  Queued a whole chunk earlier they are long done by then.
    // Bits of data steps at or beyond a frame's last one, up to the end of the frame's last chunk, are stored
    // as 1: they hold a chain-back that starts above the frame's end in pbar = 63 (state 0) until it gets there.
#ifndef FOA_LATE_ST
#define FOA_LATE_ST 1
#endif
#if FOA_ACC2
    // A 16-step block's word -- bytes 1 and 3 of its two accumulators: (A steps 0-7, A steps 8-15, B steps 0-7, B steps 8-15) -- is formed
    // the moment the block is complete.  The first two blocks of a chunk are stored at once (thirty-two and sixteen steps before the
    // next wait for loads: old enough), the third one chunk LATE, behind the next chunk's loads, as all three were in round 2.  Three
    // live registers instead of nine; the accumulators need no initialisation either (eight steps overwrite all the bits the word takes,
    // and bits at or beyond a frame's last step are forced to 1 by the store).
#if FOA_TRIM
    const int slot2 = 2 * (63 - lane);                                    // byte offset of this lane's word inside a 16-step block (slot p's word sits at index 63 - p)
    const __amdgpu_buffer_rsrc_t rdA = __builtin_amdgcn_make_buffer_rsrc((void *)dA, 0, 8 * NAtop, kRsrcFlags);      // 8 bytes of decisions per step
    const __amdgpu_buffer_rsrc_t rdB = __builtin_amdgcn_make_buffer_rsrc((void *)dB, 0, 8 * NBtop, kRsrcFlags);
#endif
    auto store_block = [&](int b0, uint32_t w) {
        if constexpr (FOA_ABL & 16) { if (w == 0x12345678u) dA[lane] = 1; return; }
        if (__builtin_expect(b0 < NAtop, 1)) {             // (said so that the stores stay in line: as unlikely blocks each cost two taken branches)
            const int v = NA - b0;
            const uint32_t word = w | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v);
#if FOA_TRIM
            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)word, rdA, slot2, 128 * (b0 >> 4), 0);
#else
            dA[(size_t)(b0 >> 4) * 64 + 63 - lane] = (uint16_t)word;
#endif
        }
        if (__builtin_expect(b0 < NBtop, 1)) {
            const int v = NB - b0;
#if FOA_TRIM
            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)((w >> 16) | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v)), rdB, slot2, 128 * (b0 >> 4), 0);
#else
            dB[(size_t)(b0 >> 4) * 64 + 63 - lane] = (uint16_t)((w >> 16) | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v));
#endif
        }
    };
    uint32_t late = 0u;
    int n_chunk = 0;                                                      // (the flush hook reads the chunk's first step through this)
    auto flush = [&](int blk) {
        const uint32_t w = __builtin_amdgcn_perm(acc[1], acc[0], 0x07030501u);
        if (blk < 2) store_block(n_chunk + 16 * blk, w);
        else late = w;
    };
    for (int n0 = 0; n0 < N; n0 += kChunk3) {                             // n0 mod 6 == 0: phase = chunk-relative index mod 6
        const int nn = min(kChunk3, N - n0);
#if FOA_FWD_PRIO & 4
        if (3 * n0 >= 2 * N && 3 * (n0 - kChunk3) < 2 * N) __builtin_amdgcn_s_setprio(3);
#endif
        put(kChunk3);
        get(n0 + kChunk3 + 6);
        if (n0 > 0) store_block(n0 - 16, late);
        n_chunk = n0;
        if constexpr (FOA_TEST_SKIP) skip = __builtin_amdgcn_readfirstlane(skip);      // (it IS uniform; said so that it stays on the scalar side across the loop)
        if (nn == kChunk3) {
            M = fwd3_group<0, 0>(M, bml, c, acc, skip, flush);   M = fwd3_group<6, 6>(M, bml, c, acc, skip, flush);   M = fwd3_group<12, 12>(M, bml, c, acc, skip, flush);
            M = fwd3_group<18, 18>(M, bml, c, acc, skip, flush); M = fwd3_group<24, 24>(M, bml, c, acc, skip, flush); M = fwd3_group<30, 30>(M, bml, c, acc, skip, flush);
            M = fwd3_group<36, 36>(M, bml, c, acc, skip, flush); M = fwd3_group<42, 42>(M, bml, c, acc, skip, flush);
        } else {
            // the last, partial chunk: blocks the steps do not reach are still stored (all ones: the store's mask)
            for (int j = 0; j < nn; j++) {
                M = fwd3_step_dyn(M, j, bml, c, acc, skip);
                if ((j & 15) == 15) flush(j >> 4);
            }
            for (int blk = nn >> 4; blk < 3; blk++) flush(blk);
        }
    }
    if (N > 0) store_block((N - 1) / kChunk3 * kChunk3 + 32, late);
#else
    uint32_t word[3] = { 0u, 0u, 0u };                                    // a chunk's three 16-step blocks: (A, B) per lane
    auto store = [&](int n0) {
#pragma unroll
        for (int blk = 0; blk < 3; blk++) {
            const int b0 = n0 + 16 * blk;
            if constexpr (FOA_ABL & 16) { if (word[blk] == 0x12345678u) dA[lane] = 1; continue; }
            if (__builtin_expect(b0 < NAtop, 1)) {         // (said so that the stores stay in line: as unlikely blocks each cost two taken branches)
                const int v = NA - b0;
                dA[(size_t)(b0 >> 4) * 64 + 63 - lane] = (uint16_t)(word[blk] | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v));
            }
            if (__builtin_expect(b0 < NBtop, 1)) {
                const int v = NB - b0;
                dB[(size_t)(b0 >> 4) * 64 + 63 - lane] = (uint16_t)((word[blk] >> 16) | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v));
            }
        }
    };
    for (int n0 = 0; n0 < N; n0 += kChunk3) {                             // n0 mod 6 == 0: phase = chunk-relative index mod 6
        const int nn = min(kChunk3, N - n0);
        put(kChunk3);
        get(n0 + kChunk3 + 6);
        if (FOA_LATE_ST && n0 > 0) store(n0 - kChunk3);
        acc[0] = acc[1] = acc[2] = acc[3] = acc[4] = acc[5] = 0xFFFFFFFFu;
        if (nn == kChunk3) {
            M = fwd3_group<0, 0>(M, bml, c, acc, skip);   M = fwd3_group<6, 6>(M, bml, c, acc, skip);   M = fwd3_group<12, 12>(M, bml, c, acc, skip);
            M = fwd3_group<18, 18>(M, bml, c, acc, skip); M = fwd3_group<24, 24>(M, bml, c, acc, skip); M = fwd3_group<30, 30>(M, bml, c, acc, skip);
            M = fwd3_group<36, 36>(M, bml, c, acc, skip); M = fwd3_group<42, 42>(M, bml, c, acc, skip);
        } else {
            for (int j = 0; j < nn; j++) M = fwd3_step_dyn(M, j, bml, c, acc, skip);
        }
        // bytes 1 and 3 of a block's two accumulators: (A steps 0-7, A steps 8-15, B steps 0-7, B steps 8-15)
#pragma unroll
        for (int blk = 0; blk < 3; blk++) word[blk] = __builtin_amdgcn_perm(acc[2 * blk + 1], acc[2 * blk], 0x07030501u);
        if (!FOA_LATE_ST) store(n0);
    }
    if (FOA_LATE_ST && N > 0) store((N - 1) / kChunk3 * kChunk3);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// chain-back, one lane per segment
// ---------------------------------------------------------------------------------------------------------
// LDS holds a ring of three 16-step decision blocks of every lane: [ring slot][lane / 8][lane % 8][slot] u16, whole
// 128-byte lines.  Eight consecutive lanes of an LDS-DMA instruction fetch the eight 16-byte pieces of ONE line (one
// block of one segment), so every instruction moves eight full lines.  The lane's LDS byte offset carries the
// complemented slot index in bits 1-6; the ring slot is an immediate offset.  A step is then: read 16 bits, shift the
// step's bit to its place, v_bfi it in.  24 KB per wave let six waves share a CU, which is what the kernel's speed
// hangs on: a lane's walk is one dependent chain of about 200 clocks per step, so time = steps per lane x rounds.
__device__ __forceinline__ uint32_t tb_lane_base(int lane) { return (uint32_t)(lane >> 3) * 1024u + (uint32_t)(lane & 7) * 128u; }
__device__ __forceinline__ uint32_t tb_gather(uint32_t a) { return (a >> 1) & 63u; }

// 16 steps of block KK (0..5) of a 96-step unit, i.e. data steps 96u + 16 KK + 15 .. 96u + 16 KK, decoded bits into w[]
// (data bit 96u+i at bit 31-(i mod 32) of w[i/32]: the reference's MSB-first byte order once the word is byte-swapped).
template <int KK>
__device__ __forceinline__ void tb_walk_block(const uint8_t *tb, uint32_t &a, uint32_t (&w)[3])
{
#pragma unroll
    for (int jj = 15; jj >= 0; jj--) {
        const int ju = 16 * KK + jj, q = 5 - ju % 6, pos = q + 1;            // unit starts are multiples of 96: phase = ju mod 6
        const uint32_t v = *(const uint16_t *)(tb + a + (KK % kTbRing) * kTbBlockBytes);
        const uint32_t tmp = pos >= jj ? v << (pos - jj) : v >> (jj - pos);
        asm("v_bfi_b32 %0, %1, %2, %0" : "+v"(a) : "s"(1u << pos), "v"(tmp));
        if (ju % 6 == 0) {
            // the six bits just written are the six data bits of this group, complemented; p bit i = data bit 5-i
            const int o = ju, wi = o >> 5, r = o & 31;
            const uint32_t p = tb_gather(a) ^ 63u;
            if (r <= 26) {
                w[wi] |= p << (26 - r);
            } else {
                w[wi] |= p >> (r - 26);
                w[wi + 1] |= p << (58 - r);
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_tb_walk(const FrameInfo *__restrict__ info, const int32_t *__restrict__ seg2frame,
                                                const int64_t *__restrict__ totals, const uint64_t *__restrict__ dec,
                                                uint32_t *__restrict__ decoded, uint16_t *__restrict__ tb_state, int S, int L)
{
    __shared__ __attribute__((aligned(16))) uint8_t tb[kTbRing * kTbBlockBytes];
#if FOA_WALK_PRIO
    __builtin_amdgcn_s_setprio(FOA_WALK_PRIO);         // (A/B only)
#endif
    const int lane = threadIdx.x, g = blockIdx.x * 64 + lane;
    const int n_seg = (int)totals[4];
    if (blockIdx.x * 64 >= n_seg) return;
    const int f = g < n_seg ? seg2frame[g] : -1;
    const bool live = f >= 0;
    FrameInfo fi;
    fi.nsteps = 0; fi.dec_off = 0; fi.seg_off = 0;
    if (live) fi = info[f];
    const int k = g - fi.seg_off, N = fi.nsteps - 6;
    const int n_lo = k * S, n_own = min(n_lo + S, N), n_hi = min(n_lo + S + L, N);
    const int cnt = live ? (n_hi - n_lo + kChunk3 - 1) / kChunk3 : 0;      // 48-step chunks this lane walks, numbered from its bottom
    const int own = live ? (n_own - n_lo + kChunk3 - 1) / kChunk3 : 0;     // of which the lowest `own` are its own
    const uint8_t *src = live ? (const uint8_t *)(dec + fi.dec_off) + (size_t)(n_lo / kChunk3) * 384 : (const uint8_t *)dec;
    uint32_t *out = decoded + fi.dec_off + n_lo / 32;
    int cmax = cnt;
#pragma unroll
    for (int o = 32; o; o >>= 1) cmax = max(cmax, __shfl_xor(cmax, o));
    uint32_t a = tb_lane_base(lane) + (63u << 1);                          // state 0; true at the frame's end, a guess elsewhere
    uint32_t e = 63u, w[3] = { 0u, 0u, 0u };

    // Fetch roles: in the instruction for lane group gi, this lane moves piece lane % 8 of a block of segment lane
    // 8 gi + lane / 8.  A segment lane that has no chunk i gets its chunk 0 again (always inside its region).
    const uint8_t *fsrc[8];
    int fcnt[8];
#pragma unroll
    for (int gi = 0; gi < 8; gi++) {
        const int from = 8 * gi + (lane >> 3);
        fsrc[gi] = (const uint8_t *)__shfl((unsigned long long)(uintptr_t)src, from) + (lane & 7) * 16;
        fcnt[gi] = __shfl(cnt, from);
    }
    // block bk (0..2) of chunk i of every segment lane -> ring slot bk
    auto fetch = [&](int i, int bk) {
        const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)&tb[bk * kTbBlockBytes];
#pragma unroll
        for (int gi = 0; gi < 8; gi++) {
            const uint8_t *p = fsrc[gi] + (size_t)(i < fcnt[gi] ? i : 0) * 384 + 128 * bk;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(p), "s"(lds0 + (uint32_t)gi * 1024u)
                         : "memory");
        }
    };
    // Block KK of unit u is block KK % 3 of chunk 2u + KK / 3.  While it is walked the two blocks below it are in flight
    // or landed, and the block two below is requested as soon as the ring slot above is free (the block walked before).
#define FOA_TB_BLOCK(KK)                                                                                   \
    {                                                                                                      \
        constexpr int below = (KK) - 2;           /* in-unit index of the block to request, may be negative */ \
        if (below >= 0) fetch(2 * u + below / 3, below % 3);                                               \
        else if (u > 0) fetch(2 * (u - 1) + (below + 6) / 3, (below + 6) % 3);                             \
        if (below >= 0 || u > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                         \
        else if ((KK) == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        __builtin_amdgcn_wave_barrier();                                                                   \
        if ((KK) % 3 == 2 && 2 * u + (KK) / 3 == own - 1) e = tb_gather(a);                                \
        if (2 * u + (KK) / 3 < cnt) tb_walk_block<(KK)>(tb, a, w);                                         \
        __builtin_amdgcn_wave_barrier();                                                                   \
    }
    const int U = (cmax - 1) / 2;                                           // top unit
    fetch(2 * U + 1, 2);
    fetch(2 * U + 1, 1);
    for (int u = U; u >= 0; u--) {
        FOA_TB_BLOCK(5) FOA_TB_BLOCK(4) FOA_TB_BLOCK(3) FOA_TB_BLOCK(2) FOA_TB_BLOCK(1) FOA_TB_BLOCK(0)
        // own is even except in a frame's last segment, whose chunks above `own` lie beyond the frame's end (zeros)
        if (2 * u < own) {
            out[3 * u] = __builtin_bswap32(w[0]); out[3 * u + 1] = __builtin_bswap32(w[1]); out[3 * u + 2] = __builtin_bswap32(w[2]);
        }
        w[0] = w[1] = w[2] = 0u;
    }
#undef FOA_TB_BLOCK
    if (live) tb_state[g] = (uint16_t)(e | (tb_gather(a) << 8));
}

// Serial walk of data steps n_hi-1 .. n_lo (n_lo a multiple of 32) of one frame from state pbar, every lane of the
// wave with the same arguments (decisions staged through lds by the whole wave; lane 0 writes the decoded words).
// Returns the state at n_lo.  Only used when a segment's assumed start state turned out wrong.
__device__ __noinline__ uint32_t tb_rewalk(const uint16_t *__restrict__ d16, int n_lo, int n_hi, uint32_t pbar, uint32_t *__restrict__ out,
                                           uint16_t *lds, int lane)
{
    const int b_lo = n_lo >> 4, nblk = ((n_hi + 15) >> 4) - b_lo;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < nblk * 64; i += 64) lds[i] = d16[(size_t)b_lo * 64 + i];
    wave_lds_sync();
    uint32_t word = 0;
    for (int n = n_hi - 1; n >= n_lo; n--) {
        const uint32_t s = (lds[((n >> 4) - b_lo) * 64 + pbar] >> (n & 15)) & 1u;
        const int q = 5 - n % 6;
        pbar = (pbar & ~(1u << q)) | (s << q);
        word |= (s ^ 1u) << (8 * ((n & 31) >> 3) + 7 - (n & 7));          // data bit n, MSB-first bytes in a little-endian word
        if ((n & 31) == 0) {
            if (lane == 0) out[n >> 5] = word;
            word = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return pbar;
}

// One lane per frame: stitch the segments (re-walking the rare one whose start state was wrong), then descramble,
// CRC and payload copy.
__global__ __launch_bounds__(64) void k_tb_finish(const FrameInfo *__restrict__ info, int n_frames, const uint64_t *__restrict__ dec,
                                                  uint32_t *__restrict__ decoded, const uint16_t *__restrict__ tb_state, int S,
                                                  uint8_t *__restrict__ psdu, size_t slot_bytes, foa_frame_result *__restrict__ results)
{
    __shared__ FinishTables tabs;
    __shared__ FinishWave fwave;
    __shared__ uint16_t rw[kTbMaxSeg / 16 * 64];
    const int lane = threadIdx.x, f = blockIdx.x * 64 + lane;
    finish_tables_init(tabs, lane, 64);
    __syncthreads();
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0; fi.soft_off = 0; fi.dec_off = 0;
    fi.seg_off = 0;
    if (f < n_frames) fi = info[f];
    const bool live = f < n_frames && fi.nsym > 0;
    const int N = live ? fi.nsteps - 6 : 0, nseg = live ? tb_segments(fi.nsteps, S) : 0;

    int maxseg = nseg;
#pragma unroll
    for (int o = 32; o; o >>= 1) maxseg = max(maxseg, __shfl_xor(maxseg, o));
    uint32_t s_next = nseg > 0 ? (uint32_t)(tb_state[fi.seg_off + nseg - 1] >> 8) : 0u;    // state at the bottom of the top segment
    for (int k = maxseg - 2; k >= 0; k--) {
        const bool has = k < nseg - 1;
        const uint32_t st = has ? tb_state[fi.seg_off + k] : 0u;
        uint32_t s_k = st >> 8;
        uint64_t redo = __ballot(has && (st & 0xFFu) != s_next);           // assumed start state != proven one
        while (redo) {
            const int l = __ffsll((unsigned long long)redo) - 1;
            redo &= redo - 1;
            const int64_t off = __shfl(fi.dec_off, l);
            const int n_l = __shfl(N, l);
            const uint32_t sn = __shfl(s_next, l);
            const uint32_t r = tb_rewalk((const uint16_t *)(dec + off), k * S, min(k * S + S, n_l), sn, decoded + off, rw, lane);
            if (lane == l) s_k = r;
        }
        if (has) s_next = s_k;
    }
    __threadfence();                                                       // re-walked words were written by lane 0
    if (psdu == nullptr) return;                                           // foa_conv_decode: the decoded bits are the result
    finish_crc_psdu(tabs, fwave, fi, live, f, n_frames, decoded, psdu, slot_bytes, results);
}

inline void launch_fwd3(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec)
{
    hipLaunchKernelGGL(k_viterbi_fwd3, dim3(((nf + 1) / 2 + kFwdWaves - 1) / kFwdWaves), dim3(64 * kFwdWaves), 0, st, info, nf, sp, dec);
}

// walk on st, stitch + descramble + CRC on st_fin (the same stream, or another one that then waits for walk_done)
inline void launch_finish3(hipStream_t st, hipStream_t st_fin, const FrameInfo *info, int nf, const uint64_t *dec, uint32_t *decoded, const int32_t *seg2frame,
                           const int64_t *totals, uint16_t *tb_state, size_t max_segs, int S, int L, uint8_t *psdu, size_t slot_bytes,
                           foa_frame_result *results, hipEvent_t walk_done = nullptr)
{
    if constexpr (!(FOA_EXP & 1))
    hipLaunchKernelGGL(k_tb_walk, dim3((unsigned)((max_segs + 63) / 64)), dim3(64), 0, st, info, seg2frame, totals, dec, decoded, tb_state, S, L);
    if (walk_done) (void)hipEventRecord(walk_done, st);
    if (st_fin != st) (void)hipStreamWaitEvent(st_fin, walk_done, 0);
    hipLaunchKernelGGL(k_tb_finish, dim3((nf + 63) / 64), dim3(64), 0, st_fin, info, nf, dec, decoded, tb_state, S, psdu, slot_bytes, results);
}

inline void launch_viterbi_v3(hipStream_t st, const FrameInfo *info, int nf, const uint16_t *sp, uint64_t *dec, uint32_t *decoded,
                              const int32_t *seg2frame, const int64_t *totals, uint16_t *tb_state, size_t max_segs, int S, int L,
                              uint8_t *psdu, size_t slot_bytes, foa_frame_result *results, hipEvent_t between)
{
    launch_fwd3(st, info, nf, sp, dec);
    if (between) (void)hipEventRecord(between, st);
    launch_finish3(st, st, info, nf, dec, decoded, seg2frame, totals, tb_state, max_segs, S, L, psdu, slot_bytes, results);
}

}  // namespace foa
