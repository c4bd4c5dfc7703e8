// viterbi_v1.h -- K=7 rate-1/2 soft-decision Viterbi, one wavefront per frame, lane = trellis state.
//
// Replaces viterbi::conv_decode (viterbi.cpp:31-37: viterbi_init :71-78, FULL_SPIRAL :208-457,
// viterbi_chainback :108-146) followed by the descrambler and CRC-32 of ppdu::decode_data
// (ppdu.cpp:256-293).
//
// The reference's uint8 saturating path metrics, its "renormalise when state 0 exceeds 210" rule and
// its tie-break (upper predecessor wins when equal) are part of its observable behaviour, so the
// recursion is evaluated exactly and sequentially per frame; parallelism comes from frames.
// Lane s holds the metric of NEW state s.  `v_cmp` writes the 64 decision bits of a step straight
// into an SGPR pair whose bit layout is exactly the reference's decision_t (viterbi.h:36-41).
// Decision words go to HBM (8 B per step, written as one coalesced 512-B store per 64 steps) and are
// read back in 512-B pieces by the scalar chain-back.
#pragma once

#include "frontend_kernels.h"

namespace foa {

constexpr int kMaxDecodedBytes = 4128;   // num_data_bytes <= 4104 for length <= 4095 at any rate

__device__ __forceinline__ void write_result(foa_frame_result *res, const FrameInfo &fi, int status)
{
    foa_frame_result r;
    r.status = status; r.rate = fi.rate; r.length = fi.length;
    r.num_symbols = fi.nsym > 0 ? fi.nsym : (fi.nsteps < 0 ? -fi.nsteps : 0);
    *res = r;
}

#if FOA_XCHECK      // cross-check build only (foa_common.h): the lane-per-state kernels

// Chain-back + descramble + CRC + payload copy, shared by v1 and v2.  `dec_at(n6)` must return the
// reference-layout decision word of trellis step n6.  Runs on one wave; `decoded` is LDS.
template <typename DecAt>
__device__ __forceinline__ int finish_frame_wave(const FrameInfo &fi, DecAt dec_chunk, uint8_t *decoded, const uint32_t *crc_tab,
                                                 uint8_t *psdu_slot, size_t slot_bytes, int lane)
{
    const int T = fi.nsteps, data_bits = T - 6, nbytes = T / 8;
    // viterbi.cpp:131-142: endstate 0; bit n uses decision word n+6
    uint32_t e = 0;
    for (int base = ((T - 1) / 64) * 64; base >= 0; base -= 64) {
        uint64_t w = dec_chunk(base, lane);               // lane j: word of step base+j (0 beyond T)
        uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
        int jtop = min(63, T - 1 - base);
        for (int j = jtop; j >= 0; j--) {
            int n = base + j - 6;
            if (n < 0) break;
            uint32_t lo = __builtin_amdgcn_readlane(wlo, j), hi = __builtin_amdgcn_readlane(whi, j);
            uint32_t sel = e >> 2;
            uint32_t word = sel & 32 ? hi : lo;
            uint32_t k = (word >> (sel & 31)) & 1u;
            e = (e >> 1) | (k << 7);
            if ((n & 7) == 0 || n == data_bits - 1) { if (lane == 0) decoded[n >> 3] = (uint8_t)e; }
        }
    }
    __syncthreads();
    // ppdu.cpp:256-264: descramble every byte x < num_data_bytes (only bit 0 changes)
    for (int x = lane; x < nbytes; x += 64) decoded[x] ^= g_tab.scramble[x % 127];
    __syncthreads();
    // ppdu.cpp:267-279: CRC-32 over service(2) + payload, compared with the next 4 bytes (LE)
    const int len = fi.length;
    int ok = 0;
    if (lane == 0) {
        uint32_t c = 0xFFFFFFFFu;
        for (int i = 0; i < 2 + len; i++) c = crc_tab[(c ^ decoded[i]) & 0xFFu] ^ (c >> 8);
        c ^= 0xFFFFFFFFu;
        uint32_t given = (uint32_t)decoded[2 + len] | ((uint32_t)decoded[3 + len] << 8) | ((uint32_t)decoded[4 + len] << 16) |
                         ((uint32_t)decoded[5 + len] << 24);
        ok = given == c;
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    if (ok && (size_t)len > slot_bytes) return 2;          // a payload longer than the caller's slot is reported, never cut short
    if (ok)                                                // ppdu.cpp:283-285
        for (int x = lane; x < len; x += 64) psdu_slot[x] = decoded[2 + x];
    return ok;
}


// Forward pass over T trellis steps (T even); sp = soft-byte pairs, dp = decision words (padded to a
// multiple of 64).  viterbi.cpp:71-78 (init), :208-457 (ACS).
__device__ __forceinline__ void viterbi_forward_wave(const uint16_t *__restrict__ sp, int T, uint64_t *__restrict__ dp, int lane)
{
    const AcsLane acs = acs_lane_init(lane);
    uint32_t M = lane == 0 ? 0u : 63u;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int nn = min(64, T - t0);
        uint32_t pair = lane < nn ? sp[t0 + lane] : 0;     // both soft bytes of step t0+lane
        uint32_t dlo = 0, dhi = 0;
        if (nn == 64) {
#pragma unroll
            for (int j = 0; j < 64; j++) {
                uint32_t s01 = __builtin_amdgcn_readlane(pair, j);
                uint64_t d;
                M = acs_step(M, s01 & 0xFFu, s01 >> 8, acs, d);
                dlo = lane == j ? (uint32_t)d : dlo;
                dhi = lane == j ? (uint32_t)(d >> 32) : dhi;
            }
        } else {
            for (int j = 0; j < nn; j++) {
                uint32_t s01 = __builtin_amdgcn_readlane(pair, j);
                uint64_t d;
                M = acs_step(M, s01 & 0xFFu, s01 >> 8, acs, d);
                dlo = lane == j ? (uint32_t)d : dlo;
                dhi = lane == j ? (uint32_t)(d >> 32) : dhi;
            }
        }
        dp[t0 + lane] = ((uint64_t)dhi << 32) | dlo;
    }
}

__global__ __launch_bounds__(64) void k_viterbi_v1(const FrameInfo *__restrict__ info, int n_frames, const uint16_t *__restrict__ sp,
                                                   uint64_t *__restrict__ dec, uint8_t *__restrict__ psdu, size_t slot_bytes,
                                                   foa_frame_result *__restrict__ results)
{
    __shared__ uint8_t decoded[kMaxDecodedBytes];
    __shared__ uint32_t crc_tab[256];
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_frames) return;
    const FrameInfo fi = info[f];
    if (fi.nsym <= 0) {
        if (lane == 0) write_result(&results[f], fi, fi.status);
        return;
    }
    for (int i = lane; i < 256; i += 64) crc_tab[i] = g_tab.crc_table[i];
    const int T = fi.nsteps;
    uint64_t *dp = dec + fi.dec_off;
    viterbi_forward_wave(sp + fi.dec_off, T, dp, lane);
    __syncthreads();
    int ok = finish_frame_wave(
        fi, [&](int base, int l) -> uint64_t { return base + l < T ? dp[base + l] : 0ull; }, decoded, crc_tab,
        psdu + (size_t)f * slot_bytes, slot_bytes, lane);
    if (lane == 0) write_result(&results[f], fi, ok == 2 ? FOA_ST_NO_SPACE : ok ? FOA_ST_OK : FOA_ST_CRC_FAIL);
}


// Stage-level viterbi::conv_decode (viterbi.cpp:31-37) for n_blocks independent blocks of equal size:
// symbols[b][2*(data_bits+6)] -> data[b][(data_bits+7)/8].  One wave per block.
__global__ __launch_bounds__(64) void k_conv_decode(const uint8_t *__restrict__ symbols, uint8_t *__restrict__ data, int data_bits,
                                                    int n_blocks, uint64_t *__restrict__ dec, int dec_stride)
{
    __shared__ uint8_t decoded[kMaxDecodedBytes];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks) return;
    const int nsteps = data_bits + 6, T = 2 * (nsteps / 2);          // viterbi.cpp:209 drops an odd last step
    uint64_t *dp = dec + (size_t)b * dec_stride;
    for (int i = lane; i < dec_stride; i += 64) dp[i] = 0;           // viterbi.cpp:193-194 memset
    __syncthreads();
    viterbi_forward_wave((const uint16_t *)(symbols + (size_t)b * 2 * nsteps), T, dp, lane);
    __syncthreads();
    const int nbytes = (data_bits + 7) / 8;
    uint32_t e = 0;
    for (int base = ((nsteps - 1) / 64) * 64; base >= 0; base -= 64) {
        uint64_t w = dp[base + lane];
        uint32_t wlo = (uint32_t)w, whi = (uint32_t)(w >> 32);
        for (int j = min(63, nsteps - 1 - base); j >= 0; j--) {
            int n = base + j - 6;
            if (n < 0) break;
            uint32_t lo = __builtin_amdgcn_readlane(wlo, j), hi = __builtin_amdgcn_readlane(whi, j);
            uint32_t sel = e >> 2;
            uint32_t k = ((sel & 32 ? hi : lo) >> (sel & 31)) & 1u;
            e = (e >> 1) | (k << 7);
            if (lane == 0) decoded[n >> 3] = (uint8_t)e;
        }
    }
    __syncthreads();
    for (int x = lane; x < nbytes; x += 64) data[(size_t)b * nbytes + x] = decoded[x];
}

#endif  // FOA_XCHECK

}  // namespace foa
