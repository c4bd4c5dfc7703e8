// frontend_q4.h -- data symbols, FOUR LANES (one DPP quad) PER OFDM SYMBOL, sixteen symbols per wave.
//
// Replaces, for the data symbols of every frame of a call: fft_symbols.cpp:33-79 + fft.cpp:50-59, channel_est.cpp:77-81,
// phase_tracker.cpp:83-99, modulator.cpp:108-164 / qam.h:110-125, interleaver.cpp:28-38, puncturer.cpp:78-123 -- the same
// arithmetic in the same operation order as the wave-wide fft64_lane of device_math.h, laid out so that a wave carries sixteen
// symbols.  Lane m of a quad holds the 16 samples n = m (mod 4) of its symbol:
//   * radix-4 DIF stage 1 pairs n, n+16, n+32, n+48 and stage 2 pairs n, n+4, n+8, n+12 inside every 16-block --
//     all four operands have the same n mod 4, so both stages run in the lane's own registers;
//   * stage 3 pairs n, n+1, n+2, n+3 = one operand per lane: a 4x4 transpose inside the quad (through LDS, 4 KB per
//     wave, four rounds), after which lane m holds the bins k = 16 j + 4 m + a (j, a = 0..3);
//   * equaliser, derotation and soft demapping stay per lane (12 data carriers each on average); the four pilot
//     terms live in lanes 1 and 2 and are broadcast so that every lane adds them in the reference's order;
//   * soft bytes are scattered into the symbol's depunctured order in LDS ((carrier, bit) -> position table per
//     rate), and leave in 8-byte stores, four trellis steps per lane and trip.
// Its waves are small (<= 176 VGPRs), so several share a SIMD and hide each other's latencies, and they fit next to the forward
// pass of the previous call when calls are pipelined (0.18 ms alone at config 2; HISTORY.md has the layouts it replaced).
#pragma once

#include "device_math.h"

namespace foa {

constexpr int kQ4Waves = 4;                  // waves per block: FOUR, one per SIMD (10 KB of tables + 7 KB per wave = 38 KB of LDS).
                                             // With five (a block's waves go round the SIMDs, so the fifth doubles up on one) the
                                             // kernel alone took 0.447 ms instead of 0.308 and the pipelined step 1.32 ms instead
                                             // of 1.21 (2, 3, 6 waves per block: 1.23; 8: 1.28).  The register budget does not
                                             // matter (128 ... 256 VGPRs: 1.20-1.23 ms).

struct Q4Wave {                              // LDS private to one wave
    union {
        double2 xpose[16][4][4];             // [quad][s][m]: one round of the stage-3 transpose
        uint8_t soft[16][464];               // [quad][depunctured soft byte of the symbol] (432 used; 464 = 116 dwords:
                                             //  the 16 rows start 52 q mod 64 banks apart, all distinct)
    };
};

struct Q4Shared {
    uint32_t qam[641];
    double2 tw[64];                          // exp(-2 pi j k / 64)
    uint16_t pos[kNumRates][288];            // demodulated byte (carrier * bpsc + bit) -> depunctured position
    int8_t dindex[64];                       // subcarrier index -> data carrier 0..47, -1 otherwise
    Q4Wave w[kQ4Waves];
};

// the radix-4 butterfly of fft64_lane (same association)
__device__ __forceinline__ void q4_butterfly(cpx a, cpx b, cpx c, cpx d, cpx &y0, cpx &y1, cpx &y2, cpx &y3)
{
    const cpx t0 = cadd(a, c), t1 = cadd(a, cneg(c));
    const cpx u = cadd(b, d);
    const cpx v1 = cadd(cpx{ b.y, -b.x }, cpx{ -d.y, d.x });     // (-j) b + (j) d
    const cpx v3 = cadd(cpx{ -b.y, b.x }, cpx{ d.y, -d.x });     // (j) b + (-j) d
    y0 = cadd(t0, u); y1 = cadd(t1, v1); y2 = cadd(t0, cadd(cneg(b), cneg(d))); y3 = cadd(t1, v3);
}

// Equalise + derotate this lane's data carriers, demap them, scatter the soft bytes into the symbol's depunctured order and
// store them.  BPSC > 0: bits per carrier known at compile time (rr / rate wave-uniform); 0: generic.
template <int BPSC>
__device__ __forceinline__ void q4_emit(Q4Shared &sh, Q4Wave &ws, const cpx (&X)[16], cpx rot, const double2 *__restrict__ h, const RateRow &rr, int rate,
                                        int qd, int m, bool valid, int64_t w, int64_t my_out, uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    // the symbol's depunctured soft bytes start as erasures (puncturer.cpp:94-102)
    {
        const uint4 fill = make_uint4(0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu, 0x7F7F7F7Fu);
#pragma unroll
        for (int i = 0; i < 7; i++) *(uint4 *)&ws.soft[qd][112 * m + 16 * i] = fill;
    }
    wave_lds_sync();
    const int bpsc = BPSC > 0 ? BPSC : rr.bpsc, nb = bpsc == 1 ? 1 : bpsc / 2;
    const uint16_t *pos = sh.pos[rate];
    // data-carrier index of this lane's sixteen bins, four dwords read at once (bins 16 j + 4 m + a, a = 0..3, are four consecutive
    // bytes of the table): one LDS round trip instead of sixteen
    uint32_t dj[4];
#pragma unroll
    for (int j = 0; j < 4; j++) dj[j] = *(const uint32_t *)&sh.dindex[(16 * j + 4 * m + 32) & 63];
#pragma unroll
    for (int a = 0; a < 4; a++) {
        __builtin_amdgcn_sched_barrier(0);
        // the four taps of this round first, whether or not the bin carries data: one memory round trip per round instead of
        // one per bin (inside the `data carrier?` branch every load would be waited for on the spot)
        double2 hh4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) hh4[j] = h[(16 * j + 4 * m + a + 32) & 63];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int di = (int)(int8_t)(dj[j] >> (8 * a));
            if (di < 0) continue;
            const double2 hh = hh4[j];
            const cpx zc = cmul(cmul(cpx{ hh.x, hh.y }, X[4 * a + j]), rot);
            if (eq_tap && valid) eq_tap[(size_t)w * 48 + di] = make_double2(zc.x, zc.y);
            const uint32_t li = qam_lookup(sh.qam, zc.x, rr.scale_d), lq = bpsc > 1 ? qam_lookup(sh.qam, zc.y, rr.scale_d) : 0u;
            // the carrier's positions first, all of them, then its bytes: read and written one by one, every byte waited out an LDS
            // round trip of its own (72 per lane and symbol at 64-QAM)
            uint32_t at[6];
#pragma unroll
            for (int b = 0; b < 6; b++) at[b] = b < bpsc ? pos[di * bpsc + b] : 0u;
#pragma unroll
            for (int b = 0; b < 6; b++) {
                if (b < bpsc) {
                    const uint32_t byte = b < nb ? (li >> (8 * b)) & 255u : (lq >> (8 * (b - nb))) & 255u;
                    ws.soft[qd][at[b]] = (uint8_t)byte;                          // interleaver.cpp:33-36, puncturer.cpp:112-118
                }
            }
        }
    }
    wave_lds_sync();
    // ---- the symbol's depunctured soft bytes leave as they are, two per trellis step (the forward pass forms the branch
    // metrics of viterbi.cpp:242-247 from them when it stages a chunk): 8 bytes = four steps per lane and trip ----
    if (valid) {
        const int ngroups = rr.dbps / 4;
        for (int g4 = m; g4 < ngroups; g4 += 4) *(uint2 *)(sp + my_out + 4 * g4) = *(const uint2 *)&ws.soft[qd][8 * g4];
    }
}

template <typename S>
__global__ __launch_bounds__(64 * kQ4Waves)
void k_data_symbols_q4(const S *__restrict__ iq, const foa_frame_desc *__restrict__ descs, const FrameInfo *__restrict__ info,
                       const int32_t *__restrict__ sym2frame, const SpecSym *__restrict__ spec, const int64_t *__restrict__ totals,
                       const double2 *__restrict__ hinv, uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    __shared__ Q4Shared sh;
    // (No wave priority: what matters is that these waves do not go ahead of the forward pass's; profiles/r03_ab_fwd_prio.txt.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, qd = lane >> 2, m = lane & 3;
    const int64_t total = min(totals[0], totals[3]);
    if ((int64_t)blockIdx.x * kQ4Waves * 16 >= total) return;       // the grid is sized by an upper bound: surplus blocks leave at once
    for (int i = tid; i < 641; i += 64 * kQ4Waves) sh.qam[i] = g_tab.qam_lut[i];
    static_assert(sizeof(sh.pos) == sizeof(g_tab.sym_pos) && sizeof(sh.pos) % 4 == 0 && offsetof(DeviceTables, sym_pos) % 4 == 0 &&
                  offsetof(Q4Shared, pos) % 4 == 0, "position tables are copied as dwords");
    for (int i = tid; i < (int)(sizeof(sh.pos) / 4); i += 64 * kQ4Waves) ((uint32_t *)sh.pos)[i] = ((const uint32_t *)g_tab.sym_pos)[i];
    if (tid < 64) { sh.tw[tid] = make_double2(g_tab.tw_re[tid], g_tab.tw_im[tid]); sh.dindex[tid] = g_tab.data_index[tid]; }
    __syncthreads();
    Q4Wave &ws = sh.w[wave];

    const int64_t w0 = ((int64_t)blockIdx.x * kQ4Waves + wave) * 16, w = w0 + qd;      // this quad's symbol slot
    if (w0 >= total) return;                                       // whole wave idle (wave-uniform; no block sync below)
    // The symbol's frame, and where its window comes from: window k of the frame's own alignment as a rule; for a frame that fills on
    // beyond its alignment (sym2frame <= -2, k_scan_apply) the partial vector or a vector of a later alignment: that alignment's window,
    // rotation, channel estimate and symbol count (channel_est.cpp:77-81 and phase_tracker.cpp:74-99 know nothing of frames).
    const int fq = w < total ? sym2frame[w] : -1;
    const bool valid = fq != -1;
    int f = fq >= 0 ? fq : 0, src = f, ks = 0, fresh = 64, late = -1;
    if (fq <= -2) { const SpecSym e = spec[-2 - fq]; f = e.frame; src = e.src; ks = e.k; fresh = e.fresh & 255; late = e.fresh >> 8; }
    const FrameInfo fi = info[f];
    if (late < 0) late = (fi.flags & kInfoLate) ? 1 : 0;                // (a late alignment's windows sit one symbol further on: frontend_kernels.h)
    const int rate = valid ? fi.rate : 0;
    const int kf = valid ? (int)(w - fi.sym_off) + 1 : 1;            // 1-based data symbol of the frame (SIGNAL is symbol 0): where its soft bytes go
    const int k = fq <= -2 ? ks : kf;                                // vector of alignment src: which window, which pilot polarity
    const foa_frame_desc d = descs[src];
    const int64_t start = d.lts1_pos + 144 + 80 * (int64_t)(k + late);
    const RateRow rr = g_tab.rates[rate];
    const int64_t my_out = fi.dec_off + (int64_t)(kf - 1) * rr.dbps;

    // ---- samples n = m + 4u, rotated (timing_sync.cpp:124-125) ----
    cpx x[16];
    // A data symbol's window starts 224 samples behind LTS1 and the phasor changes at most 8 behind it (rot_start), so every sample of
    // the window takes (c, s) -- unless a caller's descriptor says otherwise, which costs nothing to honour: the wave looks once.  (The
    // per-sample choice was a 64-bit compare and four v_cndmask_b32 on VCC per sample, sixteen times per lane: the select on VCC issues
    // at 16 clocks per wave instruction on this part, tools/probe_issue.hip -- a seventh of the kernel's issue time for a choice that
    // always comes out the same way.)
    if (__all(!valid || (start >= d.rot_start && fresh == 64))) {
        const cpx r = { d.c, d.s };
#pragma unroll
        for (int u = 0; u < 16; u++) {
            x[u] = valid ? rotate_sample(iq[start + m + 4 * u], r) : cpx{ 0.0, 0.0 };
        }
    } else {
        // (also the partly filled vector of fft_symbols.cpp:46-50: samples fresh .. 63 still hold the window before)
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int n = m + 4 * u;
            const int64_t idx = (n >= fresh ? start - 80 : start) + n;
            x[u] = valid ? load_rotated(iq, idx, d) : cpx{ 0.0, 0.0 };
        }
    }
    // ---- stage 1: operands u = s, s+4, s+8, s+12; twiddle exponent e = n = m + 4 s ----
#pragma unroll
    for (int s = 0; s < 4; s++) {
        cpx y0, y1, y2, y3;
        q4_butterfly(x[s], x[s + 4], x[s + 8], x[s + 12], y0, y1, y2, y3);
        const int e = m + 4 * s;
        if (e > 0) {
            const double2 t1 = sh.tw[e], t2 = sh.tw[2 * e], t3 = sh.tw[3 * e];
            y1 = cmul(y1, cpx{ t1.x, t1.y }); y2 = cmul(y2, cpx{ t2.x, t2.y }); y3 = cmul(y3, cpx{ t3.x, t3.y });
        }
        x[s] = y0; x[s + 4] = y1; x[s + 8] = y2; x[s + 12] = y3;
    }
    // ---- stage 2 inside each 16-block r: operands u = 4r + 0..3; twiddle exponent e = 4 m ----
    {
        const double2 t1 = sh.tw[4 * m], t2 = sh.tw[8 * m], t3 = sh.tw[12 * m];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            cpx y0, y1, y2, y3;
            q4_butterfly(x[4 * r], x[4 * r + 1], x[4 * r + 2], x[4 * r + 3], y0, y1, y2, y3);
            if (m > 0) { y1 = cmul(y1, cpx{ t1.x, t1.y }); y2 = cmul(y2, cpx{ t2.x, t2.y }); y3 = cmul(y3, cpx{ t3.x, t3.y }); }
            x[4 * r] = y0; x[4 * r + 1] = y1; x[4 * r + 2] = y2; x[4 * r + 3] = y3;
        }
    }
    // ---- stage 3: position 16 a + 4 s + m sits in lane m as x[4a + s]; group (a, s) goes to lane s ----
    cpx (&X)[16] = x;                                              // afterwards X[4a + j] = bin k = 16 j + 4 m + a, in place
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
        for (int s = 0; s < 4; s++) ws.xpose[qd][s][m] = make_double2(x[4 * a + s].x, x[4 * a + s].y);
        wave_lds_sync();
        const double2 i0 = ws.xpose[qd][m][0], i1 = ws.xpose[qd][m][1], i2 = ws.xpose[qd][m][2], i3 = ws.xpose[qd][m][3];
        wave_lds_sync();
        q4_butterfly(cpx{ i0.x, i0.y }, cpx{ i1.x, i1.y }, cpx{ i2.x, i2.y }, cpx{ i3.x, i3.y }, X[4 * a], X[4 * a + 1], X[4 * a + 2], X[4 * a + 3]);
    }

    __builtin_amdgcn_sched_barrier(0);                             // keep the tap loads below from crowding the FFT's registers
    // ---- channel_est.cpp:77-81 + phase_tracker.cpp:83-99 ----
    const double2 *h = hinv + (size_t)src * 64;
    cpx pe = { 0.0, 0.0 };
    {
#pragma clang fp contract(off)
        // pilots: subcarrier 11 = bin 43 (j 2, m 2, a 3), 25 = bin 57 (j 3, m 2, a 1), 39 = bin 7 (j 0, m 1, a 3), 53 = bin 21 (j 1, m 1, a 1)
        const int ps[4] = { 11, 25, 39, 53 }, xi[4] = { 14, 7, 12, 5 }, owner[4] = { 2, 2, 1, 1 };
        const double sgn[4] = { 1.0, 1.0, 1.0, -1.0 };
        const double pol = (double)g_tab.polarity[k % 127];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const double2 hh = h[ps[p]];
            const cpx mine = cmul(cpx{ hh.x, hh.y }, X[xi[p]]);    // meaningful in lane owner[p] of the quad only
            const cpx zp = { __shfl(mine.x, owner[p], 4), __shfl(mine.y, owner[p], 4) };
            const double pil = (double)(int)(sgn[p] * pol);
            pe.x += (zp.x * pil) / 4.0;
            pe.y += (zp.y * pil) / 4.0;
        }
    }
    const cpx rot = unit_conj(pe);

    // ---- soft demapping, scatter into depunctured order, branch metrics: per modulation when the wave's sixteen symbols
    // share a rate (the usual case: loop bounds and table rows are then scalar), generically otherwise ----
    const int rate_u = __builtin_amdgcn_readfirstlane(rate);
    if (__all(rate == rate_u)) {
        const RateRow ru = g_tab.rates[rate_u];
        switch (ru.bpsc) {
        case 1: q4_emit<1>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        case 2: q4_emit<2>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        case 4: q4_emit<4>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        default: q4_emit<6>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        }
    } else {
        q4_emit<0>(sh, ws, X, rot, h, rr, rate, qd, m, valid, w, my_out, sp, eq_tap);
    }
}

}  // namespace foa
