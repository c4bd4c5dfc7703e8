// viterbi_fwd.h -- K=7 Viterbi forward pass (viterbi.cpp:208-457), TWO frames per wavefront, a state per lane, packed u16 metrics,
// in-place trellis, decision bits resident in the lanes.
//
// The reference's uint8 saturating path metrics, its "renormalise when state 0 exceeds 210" rule (viterbi.cpp:314,438) and its
// tie-break (upper predecessor wins when equal, :260-267) are part of its observable behaviour, so the recursion is evaluated
// exactly and sequentially per frame; parallelism comes from frames.  Laid out for gfx950, where the pass is bound by VALU issue
// (a wave64 instruction occupies its SIMD for four clocks):
//
//   * In-place butterflies.  Physical slot p (= lane) holds state label rotl6^t(p) at trellis time t, so the butterfly of step t
//     pairs slots that differ in ONE lane bit q = 5 - (t mod 6): old states (i, i+32) sit in the pair's low / high slot and the
//     new states (2i, 2i+1) are written back to the same two slots.  Label 0 is slot 0 at all times.  The exchange is a lane-xor
//     by 32 / 16 / 8 / 4 / 2 / 1: v_permlane32_swap, v_permlane16_swap and DPP row / quad moves -- no LDS round trip on the step's
//     critical path.
//   * A VGPR carries the two frames' metrics as u16 halves biased by 0xFF00: `v_pk_add_u16 ... clamp` saturates at 0xFFFF
//     exactly where `_mm_adds_epu8` saturates at 255; min / compare are bias-invariant.
//   * The decision of slot p stays in lane p.  `x - y` as a packed 16-bit subtract has the decision of both frames in the sign
//     bits of its halves, and since |x - y| <= 255 bits 8..15 of a half all equal that sign: one v_bfi_b32 files them at bit
//     8 + (step mod 8) of a packed accumulator (2 VALU for both frames).  Two accumulators make a 16-step block: one v_perm_b32,
//     then one 16-bit store per lane and frame.  Decision memory is therefore TRANSPOSED: u16 [block of 16 steps][slot], 8 bytes
//     per step and frame.  Stored bit = 1 means "survivor came from the pair's LOW slot", and slot p's word sits at index
//     63 - p, so that a chain-back that keeps the complemented slot index pbar = 63 - p simply copies the bit it reads:
//     pbar <- (pbar & ~(1<<q)) | (bit << q),  q = 5 - n mod 6.  Steps 0..5 of the trellis carry no data bit (viterbi.cpp:131-142)
//     and are not recorded at all: "data step" n = t - 6 is the index used from here on; block b holds data steps 16b .. 16b+15.
//   * Branch-metric increments cost no VALU in the step: when a chunk's soft pairs are staged in LDS, each step gets all eight
//     (Branchtab class, side of the butterfly) variants of the packed increment pair, and a lane reads the 8 bytes of its variant.
//   * The renormalisation test reads state 0 with one v_readfirstlane and decides on the scalar unit.
//   * The same code with ONE frame per wave (template argument kPair = 1, the high halves idle and left alone) serves calls so small
//     that every wave has a SIMD nearly to itself: a wave's own step latency and its renormalisation events are what such a call waits for.
// What was measured on the way here and dropped is in HISTORY.md (patches and numbers under profiles/).
#pragma once

#include "device_math.h"

namespace foa {

typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t kBias = 0xFF00u;                      // stored metric = metric + kBias (per 16-bit half)
constexpr uint32_t kBias2 = kBias | (kBias << 16);
constexpr uint32_t kRenormThr = kBias + 210u;            // viterbi.cpp:314: renormalise when state 0 > 210
constexpr int kFwdWaves = 4;                             // waves (frame pairs) per workgroup: whole workgroups spread evenly over a CU's four SIMDs
constexpr int kChunk3 = 48;                              // data steps per forward chunk: 3 decision blocks, 8 phase groups

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
__device__ __forceinline__ uint32_t pk_sub_wrap(uint32_t a, uint32_t b)
{
    ushort2_t r = __builtin_bit_cast(ushort2_t, a) - __builtin_bit_cast(ushort2_t, b);
    return __builtin_bit_cast(uint32_t, r);
}

// 6-bit rotate left
__host__ __device__ constexpr int rotl6(int v, int r)
{
    r %= 6;
    return ((v << r) | (v >> (6 - r))) & 63;
}

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
}

struct Fwd3Lane {
    uint32_t ofs[6];      // byte offset of this lane's variant inside a step's 64-byte staging entry: 16 cls + 8 side
    uint32_t thr1;        // (one frame per wave) the renormalisation threshold as a per-lane operand: state 0's lane holds it, no other lane can exceed its 0xFFFF
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
    c.thr1 = lane == 0 ? kRenormThr : 0xFFFFu;
    return c;
}

// Renormalisation of viterbi.cpp:314-332 per frame, given s0 = the packed metrics of state 0 (slot 0 = lane 0): subtract the
// frame's smallest metric when state 0 exceeds 210.  Stored halves are 0xFF00 + metric: adding 45 to both halves at once
// carries out of a half iff its metric exceeds 210 (bit 15 / bit 31 of the sum then reads 0).  A carry out of the low half
// can push a high half of exactly 210 over, so the cold path looks at both halves again; the common path only needs
// "nothing is due" (two scalar instructions in front of the branch).
template <int kPair>
__device__ __forceinline__ bool fwd3_due(uint32_t s0)
{
    return (~(s0 + 0x002D002Du) & (kPair == 2 ? 0x80008000u : 0x00008000u)) != 0u;      // (one frame per wave: the high half is nobody's)
}
template <int kPair>
__device__ __forceinline__ uint32_t fwd3_renorm(uint32_t Mn, uint32_t &s0)
{
    if constexpr (kPair == 2) __builtin_amdgcn_s_setprio(1);   // the wave's recursion stands still until this is through (profiles/r03_ab_renorm_prio.txt; a lone wave has nobody to overtake: +1.2 % there)
    // which halves: the low one iff bit 15 of s0 + 0x002D002D reads 0 (nothing carries into it), the high one by comparing it
    // alone; the amount is wave-uniform, so the bias comes off on the scalar side and the vector side is one v_sub per half
    // (the scalar subtractions are asm so that they stay scalar: the compiler reassociates Mn - (mn - bias) into two vector ops)
    if (kPair == 1 || !((s0 + 0x002D002Du) & 0x8000u)) {               // (one frame per wave: the low half is the one that was due)
        uint32_t adj;
        asm("s_sub_u32 %0, %1, %2" : "=s"(adj) : "s"(wave_min_lo16(Mn)), "s"(kBias) : "scc");
        Mn -= adj;
        s0 -= adj;
    }
    if (kPair == 2 && s0 >= ((kRenormThr + 1u) << 16)) {
        uint32_t adj;
        asm("s_and_b32 %0, %1, 0xffff0000\n\ts_sub_u32 %0, %0, %2" : "=s"(adj) : "s"(wave_min_hi16_word(Mn)), "s"(kBias << 16) : "scc");
        Mn -= adj;
        s0 -= adj;
    }
    if constexpr (kPair == 2) __builtin_amdgcn_s_setprio(0);   // back to the wave's own priority
    return Mn;
}

// Exchange + add-compare-select of one trellis step (phase PH) for both frames.  x and y (the two candidates of every lane) go
// back to the caller, which files the decision bits: x and y lie in [0xFF00, 0xFFFF], so the 16-bit difference x - y lies in
// [-255, 255] and its bits 8..15 ALL equal its sign.  An accumulator therefore takes eight steps in bits 8..15 of each half with
// no shift at all -- v_pk_sub_u16 + v_bfi_b32 under the mask 0x01000100 << (step mod 8) --, and the two accumulators of a
// 16-step block are merged by one v_perm_b32 when the block is stored (2 + 1/16 instead of 3 VALU instructions per step).
template <int PH>
__device__ __forceinline__ uint32_t fwd3_acs(uint32_t M, const uint2 w, uint32_t &xo, uint32_t &yo)
{
    const uint32_t inc_lo = w.x, inc_hi = w.y;
    uint32_t x, y;
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
    } else {
        uint32_t lo, hi;
        pair_exchange<5 - PH>(M, lo, hi);
        x = pk_add_sat(lo, inc_lo); y = pk_add_sat(hi, inc_hi);
    }
    xo = x; yo = y;
    // upper predecessor wins ties (viterbi.cpp): survivor = low slot iff x < y iff the 16-bit difference is negative
    return pk_min(x, y);
}

constexpr int fwd3_acc_index(int j) { return (j >> 3) & 1; }

// One trellis step: exchange + ACS, then the renormalisation test on its result (v_readfirstlane of state 0, the scalar test,
// the branch).  J >= 0: data step J of the chunk (compile time), its decision filed BEHIND the test's v_readfirstlane, in the
// shadow of its way to the scalar unit (the two instructions do not depend on it; a wave issues in order); J == -1: no decision
// is recorded (trellis steps 0..5); J == -2: data step jdyn (run time: a frame's last, partial chunk).
template <int kPair, int PH, int J>
__device__ __forceinline__ uint32_t fwd3_step(uint32_t M, const uint2 w, uint32_t (&acc)[2], int jdyn, uint32_t thr1 = 0xFFFFu)
{
    uint32_t x, y;
    uint32_t Mn = fwd3_acs<PH>(M, w, x, y);
    if constexpr (J >= 0 && kPair == 1) {
        // One frame per wave: "state 0 exceeds 210" as a vector compare against a per-lane threshold (v_cmp_gt_u32_sdwa on the low half ->
        // VCC) and s_cbranch_vccnz -- two instructions where v_readfirstlane + subtract + bit test + branch are four, all of them on a lone
        // wave's chain: forward pass of 16 .. 256 frames of 1 024 bytes at 54 Mbps 0.430 -> 0.389 ms.  (With two frames per wave it would be
        // two compares: one more VALU instruction per step where those are what the machine runs out of.)
        constexpr uint32_t m = 0x01000100u << (J & 7);
        uint32_t tmp;
        const uint64_t due = __builtin_amdgcn_ballot_w64((uint16_t)Mn > (uint16_t)thr1);
        asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m), "s"((uint32_t)due));
        if (__builtin_expect(due != 0, 0)) { uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn); Mn = fwd3_renorm<1>(Mn, s0); }
        return Mn;
    }
    if constexpr (J >= 0) {
        constexpr uint32_t m = 0x01000100u << (J & 7);
        uint32_t tmp;
        uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn);
        // Written as one volatile block: left to itself the compiler sinks these instructions of all 48 steps to the end of the
        // chunk and keeps every step's x and y alive until then (148 VGPRs, 3 waves per SIMD).  (s0 is named as an input only to
        // keep the block behind the v_readfirstlane.)
        asm volatile("v_pk_sub_u16 %1, %2, %3\n\tv_bfi_b32 %0, %4, %1, %0" : "+v"(acc[fwd3_acc_index(J)]), "=&v"(tmp) : "v"(x), "v"(y), "s"(m), "s"(s0));
        if (__builtin_expect(fwd3_due<kPair>(s0), 0)) Mn = fwd3_renorm<kPair>(Mn, s0);
        return Mn;
    }
    if constexpr (J == -2) {
        const int a = fwd3_acc_index(jdyn);
        const uint32_t m = 0x01000100u << (jdyn & 7);
        const uint32_t tmp = pk_sub_wrap(x, y);
#pragma unroll
        for (int b = 0; b < 2; b++) acc[b] = a == b ? ((acc[b] & ~m) | (tmp & m)) : acc[b];
    }
    uint32_t s0 = __builtin_amdgcn_readfirstlane(Mn);
    if (__builtin_expect(fwd3_due<kPair>(s0), 0)) Mn = fwd3_renorm<kPair>(Mn, s0);      // cold: keeps the common path free of taken branches
    return Mn;
}

// Staging entry of a step: for each Branchtab class c, 16 bytes {lo, hi, hi, lo} with lo = (A.m[c], B.m[c]) as u16
// halves and hi = lo ^ 0x003F003F (63 - m); the pair's low slot reads (lo, hi) at +0, its high slot (hi, lo) at +8.
__device__ __forceinline__ uint2 fwd3_inc(const uint4 *bml, int e, uint32_t ofs)
{
    return *(const uint2 *)((const uint8_t *)bml + 64 * e + ofs);
}

struct Fwd3NoFlush { __device__ __forceinline__ void operator()(int) const {} };

// six steps (one of each phase) on staging entries E0 .. E0+5; flush(blk): called right behind the step that completes the
// 16-step block blk of the chunk
template <int kPair, int E0, int J0, typename Flush = Fwd3NoFlush>
__device__ __forceinline__ uint32_t fwd3_group(uint32_t M, const uint4 *bml, const Fwd3Lane &c, uint32_t (&acc)[2], const Flush &flush = Flush())
{
    const uint2 w0 = fwd3_inc(bml, E0 + 0, c.ofs[0]), w1 = fwd3_inc(bml, E0 + 1, c.ofs[1]), w2 = fwd3_inc(bml, E0 + 2, c.ofs[2]),
                w3 = fwd3_inc(bml, E0 + 3, c.ofs[3]), w4 = fwd3_inc(bml, E0 + 4, c.ofs[4]), w5 = fwd3_inc(bml, E0 + 5, c.ofs[5]);
    M = fwd3_step<kPair, 0, (J0 < 0 ? J0 : J0 + 0)>(M, w0, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 0) & 15) == 15) flush((J0 + 0) >> 4);
    M = fwd3_step<kPair, 1, (J0 < 0 ? J0 : J0 + 1)>(M, w1, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 1) & 15) == 15) flush((J0 + 1) >> 4);
    M = fwd3_step<kPair, 2, (J0 < 0 ? J0 : J0 + 2)>(M, w2, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 2) & 15) == 15) flush((J0 + 2) >> 4);
    M = fwd3_step<kPair, 3, (J0 < 0 ? J0 : J0 + 3)>(M, w3, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 3) & 15) == 15) flush((J0 + 3) >> 4);
    M = fwd3_step<kPair, 4, (J0 < 0 ? J0 : J0 + 4)>(M, w4, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 4) & 15) == 15) flush((J0 + 4) >> 4);
    M = fwd3_step<kPair, 5, (J0 < 0 ? J0 : J0 + 5)>(M, w5, acc, 0);
    if constexpr (J0 >= 0 && ((J0 + 5) & 15) == 15) flush((J0 + 5) >> 4);
    __builtin_amdgcn_sched_barrier(0);          // keep the next groups' LDS reads from being hoisted (registers)
    return M;
}

// The same six steps for a wave that has its SIMD to itself (one frame per wave): the increments of the group AFTER this one (entries
// ENEXT .., -1: none) are read before the group's own steps, which use the ones the group before fetched (w) -- a lone wave has nobody to
// hide an LDS round trip behind, and with the reads at the head of their own group it stood still for one every six steps (forward pass
// of 16 .. 256 frames of 1 024 bytes at 54 Mbps: 0.445 -> 0.430 ms; with five waves to a SIMD the same change bought nothing, HISTORY.md).
// (Tried on top of it and dropped, round 6: the six steps run WITHOUT the renormalisation test, state 0's largest metric of the group
// tested once behind them and the group run again step by step when it fires -- a renormalisation is due every ~9 steps at 54 Mbps, so
// half the groups run twice: 0.52 ms.)
template <int E0, int J0, int ENEXT, typename Flush>
__device__ __forceinline__ uint32_t fwd3_group_ahead(uint32_t M, const uint4 *bml, const Fwd3Lane &c, uint32_t (&acc)[2], uint2 (&w)[6], const Flush &flush)
{
    uint2 n[6] = {};
    if constexpr (ENEXT >= 0) {
#pragma unroll
        for (int i = 0; i < 6; i++) n[i] = fwd3_inc(bml, ENEXT + i, c.ofs[i]);
    }
    M = fwd3_step<1, 0, J0 + 0>(M, w[0], acc, 0, c.thr1);
    if constexpr (((J0 + 0) & 15) == 15) flush((J0 + 0) >> 4);
    M = fwd3_step<1, 1, J0 + 1>(M, w[1], acc, 0, c.thr1);
    if constexpr (((J0 + 1) & 15) == 15) flush((J0 + 1) >> 4);
    M = fwd3_step<1, 2, J0 + 2>(M, w[2], acc, 0, c.thr1);
    if constexpr (((J0 + 2) & 15) == 15) flush((J0 + 2) >> 4);
    M = fwd3_step<1, 3, J0 + 3>(M, w[3], acc, 0, c.thr1);
    if constexpr (((J0 + 3) & 15) == 15) flush((J0 + 3) >> 4);
    M = fwd3_step<1, 4, J0 + 4>(M, w[4], acc, 0, c.thr1);
    if constexpr (((J0 + 4) & 15) == 15) flush((J0 + 4) >> 4);
    M = fwd3_step<1, 5, J0 + 5>(M, w[5], acc, 0, c.thr1);
    if constexpr (((J0 + 5) & 15) == 15) flush((J0 + 5) >> 4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; i++) w[i] = n[i];
    return M;
}

template <int kPair>
__device__ __forceinline__ uint32_t fwd3_step_dyn(uint32_t M, int j, const uint4 *bml, const Fwd3Lane &c, uint32_t (&acc)[2])
{
    switch (j % 6) {
    case 0: return fwd3_step<kPair, 0, -2>(M, fwd3_inc(bml, j, c.ofs[0]), acc, j);
    case 1: return fwd3_step<kPair, 1, -2>(M, fwd3_inc(bml, j, c.ofs[1]), acc, j);
    case 2: return fwd3_step<kPair, 2, -2>(M, fwd3_inc(bml, j, c.ofs[2]), acc, j);
    case 3: return fwd3_step<kPair, 3, -2>(M, fwd3_inc(bml, j, c.ofs[3]), acc, j);
    case 4: return fwd3_step<kPair, 4, -2>(M, fwd3_inc(bml, j, c.ofs[4]), acc, j);
    default: return fwd3_step<kPair, 5, -2>(M, fwd3_inc(bml, j, c.ofs[5]), acc, j);
    }
}

// five resident waves per SIMD (10 000 frames = 4.9 waves per SIMD on 256 CUs): cap the register budget accordingly.
// kPair = 2: two frames per wave (the packed halves).  kPair = 1: one frame per wave, the high halves idle -- for calls so small that the
// SIMDs have a wave or two each: a wave's own step is then what takes the time, a renormalisation event costs a lone wave ~430 clocks,
// and a wave with one frame has half as many of them (launch_fwd3 chooses).
template <int kPair>
__global__ __launch_bounds__(64 * kFwdWaves) void k_viterbi_fwd3(const FrameInfo *__restrict__ info, int n_frames,
                                                                 const uint16_t *__restrict__ sp, uint64_t *__restrict__ dec)
{
    __shared__ uint4 bml_all[kFwdWaves][4 * kChunk3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 *bml = bml_all[wave];
    const int fA = kPair * (blockIdx.x * kFwdWaves + wave), fB = kPair == 2 ? fA + 1 : n_frames;
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
    uint32_t acc[2] = { 0u, 0u };

    // The soft pairs (s0 | s1 << 8, as the front end left them: 2 bytes per step, half of what metric words would move) of
    // trellis steps t0 .. t0+47 (lane = step) are fetched one chunk ahead into registers, so a chunk never waits for HBM;
    // put() turns them into the branch metrics of viterbi.cpp:242-247 for both frames at once --
    //   (((s0 ^ b0*255) + (s1 ^ b1*255) + 1) >> 1) >> 2  =  (s0+s1+1)>>3, (s0-s1+256)>>3, (s1-s0+256)>>3, (511-s0-s1)>>3
    // for (b0,b1) = 00, 01, 10, 11, frame A in the low half, frame B in the high half -- and into staging entries 0 .. cnt-1.
    uint32_t pa = 0, pb = 0;
    // Soft pairs come in and decision words go out through BUFFER instructions: base address and extent in a scalar resource descriptor, the
    // lane's byte offset in one VGPR that never changes, the chunk's position in a scalar offset -- no vector instruction forms an
    // address, and a lane whose step lies beyond its frame's last one needs no test: a load outside the descriptor's extent returns
    // zero (the soft pair of "no more steps").  The compiler's own builtins, so that its wait-count bookkeeping sees the loads.
    constexpr int kRsrcFlags = 0x00020000;          // gfx9 raw buffer, dword 3: DATA_FORMAT 32 (as composable_kernel's CK_BUFFER_RESOURCE_3RD_DWORD for gfx90a / gfx94x / gfx950)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)spA, 0, 2 * TA, kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)spB, 0, 2 * TB, kRsrcFlags);
    const int lane2 = 2 * lane;
    auto get = [&](int t0) {
        // soft pairs of steps t0 .. t0+47 of both frames (lanes 48..63 fetch the next sixteen, unused); beyond a frame's end: 0
        pa = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsA, lane2, 2 * t0, 0);
        pb = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsB, lane2, 2 * t0, 0);
    };
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
    M = fwd3_group<kPair, 0, -1>(M, bml, c, acc);
    // A 16-step block's word -- bytes 1 and 3 of its two accumulators: (A steps 0-7, A steps 8-15, B steps 0-7, B steps 8-15) -- is formed
    // the moment the block is complete.  The first two blocks of a chunk are stored at once (thirty-two and sixteen steps before the
    // next wait for loads: old enough), the third one chunk LATE, right behind the loads that fetch the soft pairs of the chunk after:
    // vector loads and stores share one counter on gfx950 and complete out of order with respect to each other, so the wait for those
    // loads is a wait for everything, and a store queued a whole chunk earlier is long done by then.  The accumulators need no
    // initialisation (eight steps overwrite all the bits the word takes).  Bits of data steps at or beyond a frame's last one, up to
    // the end of the frame's last chunk, are stored as 1: they hold a chain-back that starts above the frame's end in pbar = 63
    // (state 0) until it gets there.
    const int slot2 = 2 * (63 - lane);                                    // byte offset of this lane's word inside a 16-step block (slot p's word sits at index 63 - p)
    const __amdgpu_buffer_rsrc_t rdA = __builtin_amdgcn_make_buffer_rsrc((void *)dA, 0, 8 * NAtop, kRsrcFlags);      // 8 bytes of decisions per step
    const __amdgpu_buffer_rsrc_t rdB = __builtin_amdgcn_make_buffer_rsrc((void *)dB, 0, 8 * NBtop, kRsrcFlags);
    auto store_block = [&](int b0, uint32_t w) {
        if (__builtin_expect(b0 < NAtop, 1)) {             // (said so that the stores stay in line: as unlikely blocks each cost two taken branches)
            const int v = NA - b0;
            const uint32_t word = w | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v);
            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)word, rdA, slot2, 128 * (b0 >> 4), 0);
        }
        if (kPair == 2 && __builtin_expect(b0 < NBtop, 1)) {
            const int v = NB - b0;
            __builtin_amdgcn_raw_buffer_store_b16((uint16_t)((w >> 16) | (v >= 16 ? 0u : v <= 0 ? 0xFFFFu : 0xFFFFu << v)), rdB, slot2, 128 * (b0 >> 4), 0);
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
        put(kChunk3);
        get(n0 + kChunk3 + 6);
        if (n0 > 0) store_block(n0 - 16, late);
        n_chunk = n0;
        if (kPair == 1 && nn == kChunk3) {
            uint2 w[6];
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = fwd3_inc(bml, i, c.ofs[i]);
            M = fwd3_group_ahead<0, 0, 6>(M, bml, c, acc, w, flush);    M = fwd3_group_ahead<6, 6, 12>(M, bml, c, acc, w, flush);
            M = fwd3_group_ahead<12, 12, 18>(M, bml, c, acc, w, flush); M = fwd3_group_ahead<18, 18, 24>(M, bml, c, acc, w, flush);
            M = fwd3_group_ahead<24, 24, 30>(M, bml, c, acc, w, flush); M = fwd3_group_ahead<30, 30, 36>(M, bml, c, acc, w, flush);
            M = fwd3_group_ahead<36, 36, 42>(M, bml, c, acc, w, flush); M = fwd3_group_ahead<42, 42, -1>(M, bml, c, acc, w, flush);
        } else if (nn == kChunk3) {
            M = fwd3_group<kPair, 0, 0>(M, bml, c, acc, flush);   M = fwd3_group<kPair, 6, 6>(M, bml, c, acc, flush);   M = fwd3_group<kPair, 12, 12>(M, bml, c, acc, flush);
            M = fwd3_group<kPair, 18, 18>(M, bml, c, acc, flush); M = fwd3_group<kPair, 24, 24>(M, bml, c, acc, flush); M = fwd3_group<kPair, 30, 30>(M, bml, c, acc, flush);
            M = fwd3_group<kPair, 36, 36>(M, bml, c, acc, flush); M = fwd3_group<kPair, 42, 42>(M, bml, c, acc, flush);
        } else {
            // the last, partial chunk: blocks the steps do not reach are still stored (all ones: the store's mask)
            for (int j = 0; j < nn; j++) {
                M = fwd3_step_dyn<kPair>(M, j, bml, c, acc);
                if ((j & 15) == 15) flush(j >> 4);
            }
            for (int blk = nn >> 4; blk < 3; blk++) flush(blk);
        }
    }
    if (N > 0) store_block((N - 1) / kChunk3 * kChunk3 + 32, late);
}

}  // namespace foa
