// frontend_lps.h -- data symbols, ONE LANE PER OFDM SYMBOL.
//
// Same arithmetic as k_data_symbols in frontend_kernels.h (fft_symbols.cpp:33-79 + fft.cpp:50-59,
// channel_est.cpp:77-81, phase_tracker.cpp:83-99, modulator.cpp:108-164 / qam.h:110-125, interleaver.cpp:28-38,
// puncturer.cpp:78-123), same operation order (so the results are bit-identical to that kernel), but each lane
// owns a whole symbol: the 64 samples, the radix-4 butterflies and the 48 equalised carriers live in its own
// registers.  Nothing is exchanged between lanes and nothing is computed 64-fold redundantly (LDS only stages the coalesced
// loads and stores and holds the demapping table);
// per symbol this issues roughly 1/10 of the instructions of the wave-per-symbol kernel.  All indices
// (butterfly wiring, subcarrier order, interleaver and puncturing positions) are compile-time constants of the
// unrolled code, selected per rate by a wave-uniform switch (lanes of other rates, if any, wait their turn).
#pragma once

#include "frontend_kernels.h"

namespace foa {

// three radix-4 DIF stages in registers; X[k] ends at index rev4(k) (see lane_subcarrier / subcarrier_lane)
__device__ __forceinline__ void fft64_regs(cpx (&x)[64])
{
#pragma unroll
    for (int st = 0; st < 3; st++) {
        const int span = 16 >> (2 * st);
#pragma unroll
        for (int blk = 0; blk < 64; blk += 4 * span) {
#pragma unroll
            for (int n = 0; n < span; n++) {
                const int i0 = blk + n, i1 = i0 + span, i2 = i1 + span, i3 = i2 + span;
                const cpx a = x[i0], b = x[i1], c = x[i2], d = x[i3];
                // y_m = (a + (-1)^m c) + ((-j)^m b + (j)^m d), the association fft64_lane uses
                const cpx t0 = cadd(a, c), t1 = cadd(a, cneg(c));
                const cpx u = cadd(b, d);
                const cpx v1 = cadd(cpx{ b.y, -b.x }, cpx{ -d.y, d.x });     // (-j) b + (j) d
                const cpx v3 = cadd(cpx{ -b.y, b.x }, cpx{ d.y, -d.x });     // (j) b + (-j) d
                cpx y0 = cadd(t0, u), y1 = cadd(t1, v1), y2 = cadd(t0, cadd(cneg(b), cneg(d))), y3 = cadd(t1, v3);
                if (st < 2 && n > 0) {
                    const int e = n * (16 / span);
                    y1 = cmul(y1, cpx{ g_tab.tw_re[e], g_tab.tw_im[e] });
                    y2 = cmul(y2, cpx{ g_tab.tw_re[2 * e], g_tab.tw_im[2 * e] });
                    y3 = cmul(y3, cpx{ g_tab.tw_re[3 * e], g_tab.tw_im[3 * e] });
                }
                x[i0] = y0; x[i1] = y1; x[i2] = y2; x[i3] = y3;
            }
        }
    }
}

// data subcarrier index (0..47) -> subcarrier (phase_tracker.cpp:46-50)
__host__ __device__ constexpr int data_subcarrier(int di)
{
    int s = 6 + di;
    if (s >= 11) s++;
    if (s >= 25) s++;
    if (s >= 32) s++;
    if (s >= 39) s++;
    if (s >= 53) s++;
    return s;
}

template <int NB>
__device__ __forceinline__ void qam_decode_n(double sym, double scale_d, uint32_t (&bits)[3])
{
#pragma clang fp contract(off)
    uint32_t pt = (uint32_t)trunc_to_int(sym * scale_d);
    int flip = 1, amp = 128;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        int v = (int)((uint32_t)flip * pt + 128u);
        bits[i] = (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        int bit = ((int)pt < 0) ? -1 : 1;
        pt -= (uint32_t)(bit * amp);
        flip = -bit;
        amp >>= 1;
    }
}

// qam.h:110-125 through the table: the three soft bytes of one axis in one dword
__device__ __forceinline__ uint32_t qam_lookup(const uint32_t *qam, double sym, double scale_d)
{
#pragma clang fp contract(off)
    const int pt = trunc_to_int(sym * scale_d);
    return qam[min(max(pt, -320), 320) + 320];
}

// LDS of one wave: staging rows for the coalesced input loads and output stores, and the look-up tables
struct LpsShared {
    uint32_t qam[641];
    int64_t in_base[64];                 // per lane: first sample index of its symbol
    int64_t out_base[64];                // per lane: index of its symbol's first soft pair
    float4 in[64][9];                    // 16 samples (8 x 16 B) per lane and round, one 16-B pad
    uint2 out[64][10];                   // up to 36 soft pairs (9 x 8 B) per lane and interleaver block, one pad
};

// Soft bytes of one symbol in depunctured order, two per trellis step.
// zx: the derotated carriers in FFT register order (data carrier di at index subcarrier_lane(data_subcarrier(di))).
// Works through the symbol in 48-byte interleaver blocks (BPSC of them).  COOP: all 64 lanes run this body together,
// so each block's pairs go through LDS and leave as 8-byte pieces, consecutive lanes writing consecutive pieces
// (a lane-private run of 24-36 pairs per block would otherwise cost 64 separate 2-byte transactions per store).
template <int BPSC, int PUNCT, bool COOP>
__device__ __forceinline__ void emit_symbol_lps(const cpx (&zx)[64], double scale_d, uint16_t *__restrict__ sp, int64_t my_out, LpsShared *sh, int lane)
{
    constexpr int NB = BPSC == 1 ? 1 : BPSC / 2;
    constexpr int CPB = 48 / BPSC;                                  // carriers per 48-byte block
    constexpr int STEPS = PUNCT == 0 ? 24 : (PUNCT == 1 ? 32 : 36); // trellis steps per block
    constexpr int PIECES = STEPS / 4;
#pragma unroll
    for (int q = 0; q < BPSC; q++) {
        uint32_t d[48];                                             // the block in deinterleaved order
#pragma unroll
        for (int cc = 0; cc < CPB; cc++) {
            const int di = q * CPB + cc;
            const cpx zc = zx[subcarrier_lane(data_subcarrier(di))];
            const uint32_t li = qam_lookup(sh->qam, zc.x, scale_d), lq = BPSC > 1 ? qam_lookup(sh->qam, zc.y, scale_d) : 0u;
#pragma unroll
            for (int b = 0; b < BPSC; b++) {
                const int w = cc * BPSC + b;                        // byte index inside the block, demodulated order
                d[16 * (w % 3) + w / 3] = b < NB ? (li >> (8 * b)) & 255u : (lq >> (8 * (b - NB))) & 255u;   // interleaver.cpp:33-36
            }
        }
        // puncturer.cpp:94-102,112-118: step -> (first, second) soft byte, 127 where punctured
        auto soft_pair = [&](int t) -> uint32_t {
            uint32_t s0, s1;
            if (PUNCT == 0) { s0 = d[2 * t]; s1 = d[2 * t + 1]; }
            else if (PUNCT == 2) {                                  // 4 in -> {d0,d1,127,d2,127,d3}
                const int g = t / 3, r = t % 3;
                s0 = r == 0 ? d[4 * g] : 127u;
                s1 = r == 0 ? d[4 * g + 1] : (r == 1 ? d[4 * g + 2] : d[4 * g + 3]);
            } else {                                                // 3 in -> {d0,127,d1,d2}
                const int g = t / 2, r = t % 2;
                s0 = r == 0 ? d[3 * g] : d[3 * g + 1];
                s1 = r == 0 ? 127u : d[3 * g + 2];
            }
            return s0 | (s1 << 8);
        };
        uint32_t wds[STEPS / 2];                                    // two steps per dword
#pragma unroll
        for (int t = 0; t < STEPS; t += 2) wds[t / 2] = soft_pair(t) | (soft_pair(t + 1) << 16);
        if constexpr (COOP) {
#pragma unroll
            for (int i = 0; i < PIECES; i++) sh->out[lane][i] = make_uint2(wds[2 * i], wds[2 * i + 1]);
            wave_lds_sync();
#pragma unroll
            for (int it = 0; it < PIECES; it++) {
                const int idx = it * 64 + lane, seg = idx / PIECES, part = idx % PIECES;
                *(uint2 *)(sp + sh->out_base[seg] + q * STEPS + 4 * part) = sh->out[seg][part];
            }
            wave_lds_sync();
        } else {
#pragma unroll
            for (int t = 0; t < STEPS / 2; t++) *(uint32_t *)(sp + my_out + q * STEPS + 2 * t) = wds[t];
        }
    }
}

template <bool COOP>
__device__ __forceinline__ void emit_by_rate(int rate, const cpx (&x)[64], double scale_d, uint16_t *sp, int64_t my_out, LpsShared *sh, int lane)
{
    // one unrolled body per (modulation, puncturing)
    switch (rate) {
    case 0: emit_symbol_lps<1, 0, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 1: emit_symbol_lps<1, 1, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 2: emit_symbol_lps<1, 2, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 3: emit_symbol_lps<2, 0, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 4: emit_symbol_lps<2, 1, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 5: emit_symbol_lps<2, 2, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 6: emit_symbol_lps<4, 0, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 7: emit_symbol_lps<4, 1, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 8: emit_symbol_lps<4, 2, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    case 9: emit_symbol_lps<6, 1, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    default: emit_symbol_lps<6, 2, COOP>(x, scale_d, sp, my_out, sh, lane); break;
    }
}

#if FOA_XCHECK      // cross-check build only: the lane-per-symbol front end (its per-symbol arithmetic above is what the quad kernel runs)
__global__ __launch_bounds__(64) void k_data_symbols_lps(const float2 *__restrict__ iq, const foa_frame_desc *__restrict__ descs,
                                                         const FrameInfo *__restrict__ info, const int32_t *__restrict__ sym2frame,
                                                         const int64_t *__restrict__ totals, const double2 *__restrict__ hinv,
                                                         uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    __shared__ LpsShared sh;
    const int lane = threadIdx.x;
    const int64_t w0 = (int64_t)blockIdx.x * 64, w = w0 + lane;
    const int64_t total = min(totals[0], totals[3]);
    if (w0 >= total) return;
    // tables first, while all 64 lanes are still here (the next wave_lds_sync orders them)
    for (int i = lane; i < 641; i += 64) sh.qam[i] = g_tab.qam_lut[i];
    // lanes past the end (or in unused slots) shadow the wave's first symbol and write nothing
    int f = w < total ? sym2frame[w] : -1;
    const bool valid = f >= 0;
    const int f0 = __builtin_amdgcn_readfirstlane(f);
    if (f0 < 0) {                                                // the wave's first slot is unused: no cooperative path
        if (!valid) return;
    }
    const int64_t wq = valid ? w : w0;
    if (!valid) f = f0 >= 0 ? f0 : 0;
    const FrameInfo fi = info[f];
    const int k = (int)(wq - fi.sym_off) + 1;                      // 1-based data symbol (SIGNAL is symbol 0)
    const foa_frame_desc d = descs[f];
    const int64_t start = d.lts1_pos + 144 + 80 * (int64_t)k;
    const RateRow rr = g_tab.rates[fi.rate];
    const int64_t my_out = fi.dec_off + (int64_t)(k - 1) * rr.dbps;
    const bool coop = __all(valid) && __all(fi.rate == __builtin_amdgcn_readfirstlane(fi.rate));

    // ---- the 64 samples of the symbol, fetched 16 at a time as coalesced 16-byte pieces through LDS ----
    cpx x[64];
    sh.in_base[lane] = start;
    sh.out_base[lane] = my_out;
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int idx = it * 64 + lane, seg = idx >> 3, part = idx & 7;
            sh.in[seg][part] = *(const float4 *)(iq + sh.in_base[seg] + 16 * r + 2 * part);
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 v = sh.in[lane][j];
            const int64_t idx = start + 16 * r + 2 * j;
            x[16 * r + 2 * j] = cmul(cpx{ (double)v.x, (double)v.y }, idx >= d.rot_start ? cpx{ d.c, d.s } : cpx{ d.c_prev, d.s_prev });
            x[16 * r + 2 * j + 1] = cmul(cpx{ (double)v.z, (double)v.w }, idx + 1 >= d.rot_start ? cpx{ d.c, d.s } : cpx{ d.c_prev, d.s_prev });
        }
        wave_lds_sync();
    }
    fft64_regs(x);

    // channel_est.cpp:77-81 on the 52 used subcarriers, then phase_tracker.cpp:83-99
    const double2 *h = hinv + (size_t)f * 64;
    cpx pe = { 0.0, 0.0 };
    {
#pragma clang fp contract(off)
        const int ps[4] = { 11, 25, 39, 53 };
        const double sgn[4] = { 1.0, 1.0, 1.0, -1.0 };
        const double pol = (double)g_tab.polarity[k % 127];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const double2 hh = h[ps[p]];
            const cpx zp = cmul(cpx{ hh.x, hh.y }, x[subcarrier_lane(ps[p])]);
            const double pil = (double)(int)(sgn[p] * pol);
            pe.x += (zp.x * pil) / 4.0;
            pe.y += (zp.y * pil) / 4.0;
        }
    }
    const cpx rot = unit_conj(pe);
    // derotated data carriers overwrite their FFT registers
#pragma unroll
    for (int di = 0; di < 48; di++) {
        const int s = data_subcarrier(di), idx = subcarrier_lane(s);
        const double2 hh = h[s];
        x[idx] = cmul(cmul(cpx{ hh.x, hh.y }, x[idx]), rot);
    }
    if (eq_tap && valid) {
#pragma unroll
        for (int di = 0; di < 48; di++) {
            const cpx zc = x[subcarrier_lane(data_subcarrier(di))];
            eq_tap[(size_t)w * 48 + di] = make_double2(zc.x, zc.y);
        }
    }
    if (coop) emit_by_rate<true>(__builtin_amdgcn_readfirstlane(fi.rate), x, rr.scale_d, sp, my_out, &sh, lane);
    else if (valid) emit_by_rate<false>(fi.rate, x, rr.scale_d, sp, my_out, &sh, lane);
}

#endif  // FOA_XCHECK

}  // namespace foa
