// signal_decode.h -- ppdu::decode_header's Viterbi (ppdu.cpp:178-209) for ONE SIGNAL symbol per wave: 24 trellis steps with a
// lane per state, the plain form of viterbi.cpp:208-457.  (A frame's data symbols go through viterbi_fwd.h / viterbi_tb.h; a
// SIGNAL symbol is 24 steps that its frame's whole schedule waits for, so it is decoded where it is equalised: k_header and the
// stage entry point foa_decode_header_f64.)
#pragma once

#include "device_math.h"

namespace foa {

// Lane s owns NEW state s; it needs old metrics of states s>>1 and (s>>1)+32.
struct AcsLane {
    uint32_t b0, b1;     // Branchtab entries (0/255) of butterfly lane>>1 (viterbi.cpp:86-91)
    uint32_t flip;       // 63 for odd states (they take 63-m on the lower branch), else 0
    int src_lo, src_hi;  // byte addresses for ds_bpermute
};

__device__ __forceinline__ AcsLane acs_lane_init(int lane)
{
    AcsLane a;
    int i = lane >> 1;
    a.b0 = (__popc((2 * i) & 121) & 1) ? 255u : 0u;
    a.b1 = (__popc((2 * i) & 91) & 1) ? 255u : 0u;
    a.flip = (lane & 1) ? 63u : 0u;
    a.src_lo = i * 4;
    a.src_hi = (i + 32) * 4;
    return a;
}

// DPP controls (cdna4 ISA 'DPP_CTRL'): quad_perm xor-1 / xor-2, row_half_mirror, row_mirror
#define FOA_DPP_XOR1 0xB1
#define FOA_DPP_XOR2 0x4E
#define FOA_DPP_HALF_MIRROR 0x141
#define FOA_DPP_MIRROR 0x140

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// minimum over the 64 lanes, returned wave-uniform: four DPP rounds inside each row of 16 (xor 1, xor 2,
// half-mirror, mirror), then row_bcast:15 / row_bcast:31 carry the row results to lane 63
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_min_step(uint32_t v)
{
    // lanes outside ROW_MASK (and lanes whose source is out of range) see UINT_MAX, the identity of min, so that
    // the compiler can fold the move into v_min_u32_dpp
    uint32_t t = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)v, CTRL, ROW_MASK, 0xF, false);
    return umin32(v, t);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = dpp_min_step<FOA_DPP_XOR1, 0xF>(v);
    v = dpp_min_step<FOA_DPP_XOR2, 0xF>(v);
    v = dpp_min_step<FOA_DPP_HALF_MIRROR, 0xF>(v);
    v = dpp_min_step<FOA_DPP_MIRROR, 0xF>(v);
    v = dpp_min_step<0x142, 0xA>(v);          // row_bcast:15 -> rows 1 and 3
    v = dpp_min_step<0x143, 0xC>(v);          // row_bcast:31 -> rows 2 and 3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ uint32_t acs_step(uint32_t M, uint32_t s0, uint32_t s1, const AcsLane &a, uint64_t &dec)
{
    uint32_t m = ((s0 ^ a.b0) + (s1 ^ a.b1) + 1u) >> 3;      // avg_epu8 then >>2 (0..63)
    uint32_t ma = m ^ a.flip, mb = ma ^ 63u;
    uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(a.src_lo, (int)M);
    uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(a.src_hi, (int)M);
    uint32_t x = lo + ma, y = hi + mb;
    x = x > 255u ? 255u : x;                                  // _mm_adds_epu8
    y = y > 255u ? 255u : y;
    bool d = y <= x;                                          // cmpeq(min, upper)
    dec = __ballot(d);
    uint32_t Mn = d ? y : x;
    uint32_t m0 = __builtin_amdgcn_readfirstlane(Mn);         // lane 0 = state 0
    if (m0 > 210u) Mn -= wave_min_u32(Mn);                    // viterbi.cpp:314-332
    return Mn;
}

// ppdu.cpp:178-209: the 48 deinterleaved BPSK soft bytes of a SIGNAL symbol -> conv_decode(18 bits, i.e. 24
// trellis steps) -> parity / rate checks.  dem, decs: LDS (48 bytes, 24 words).  All lanes return the same values;
// rate < 0 on failure.
__device__ __forceinline__ void decode_signal_bits(const uint8_t *dem, uint64_t *decs, int lane, int &rate, int &length, int &nsym)
{
    const AcsLane acs = acs_lane_init(lane);
    uint32_t M = lane == 0 ? 0u : 63u;
    for (int t = 0; t < 24; t++) {
        uint64_t dec;
        M = acs_step(M, dem[2 * t], dem[2 * t + 1], acs, dec);
        if (lane == 0) decs[t] = dec;
    }
    __syncthreads();
    // viterbi.cpp:131-142 chain-back from state 0, 18 bits -> 3 bytes MSB first
    uint32_t e = 0, hb[3] = { 0, 0, 0 };
    for (int n = 17; n >= 0; n--) {
        uint32_t k = (uint32_t)((decs[n + 6] >> (e >> 2)) & 1ull);
        e = (e >> 1) | (k << 7);
        hb[n >> 3] = e;
    }
    const uint32_t field = (hb[0] << 16) | (hb[1] << 8) | hb[2];
    rate = -1; length = 0; nsym = 0;
    if ((__popc(field) & 1) == 0) {                                       // ppdu.cpp:187-191
        const int rf = (field >> 19) & 0xF;
        for (int r = 0; r < kNumRates; r++) if (g_tab.rates[r].rate_field == rf) rate = r;   // ppdu.cpp:198-203
    }
    if (rate >= 0) {
        length = (field >> 6) & 0xFFF;
        const int dbps = g_tab.rates[rate].dbps;
        nsym = (16 + 8 * (length + 4) + 6 + dbps - 1) / dbps;             // ppdu.cpp:206-209 (exact in integers)
    }
}

}  // namespace foa
