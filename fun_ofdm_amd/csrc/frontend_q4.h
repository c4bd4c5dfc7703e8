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
// The kernel is persistent and keeps three groups of symbols in flight per wave (below: "the kernel"): 256 VGPRs, two waves per SIMD,
// one beside the five waves of a forward pass when calls are pipelined (0.15 ms alone at config 2; HISTORY.md has the forms it replaced).
#pragma once

#include "device_math.h"

namespace foa {

constexpr int kQ4Waves = 4;                  // waves per block: FOUR, one per SIMD (10 KB of tables + 15 KB per wave = 72 KB of LDS).
                                             // With five (a block's waves go round the SIMDs, so the fifth doubles up on one) the
                                             // kernel alone took 0.447 ms instead of 0.308 and the pipelined step 1.32 ms instead
                                             // of 1.21 (2, 3, 6 waves per block: 1.23; 8: 1.28).  The register budget does not
                                             // matter (128 ... 256 VGPRs: 1.20-1.23 ms).

constexpr int kQ4TapRows = 4;                // channel estimates staged per wave and group (two groups' worth: one in use, one on its way): sixteen
                                             // consecutive symbols belong to one or two frames as a rule; a group with more than four (frames
                                             // of three symbols and less) takes its taps from memory

struct Q4Wave {                              // LDS private to one wave
    union {
        double2 xpose[16][4][5];             // [quad][s][m] in rows of 80 bytes, 320 per quad: one round of the stage-3 transpose.  Padded so that
                                             // neither side conflicts: the writes (ds_write_b128: groups of 8 consecutive lanes, bank = dword mod
                                             // 32) of two quads land 16 banks apart, the reads ([quad][m][j], ds_read_b128: the groups of MI355X_
                                             // MICROARCH.md's table, bank = dword mod 64) of a group's four quads on sixteen different 16-byte slots.
                                             // ([16][4][4] put a group's four quads on the same four slots: every read took four turns)
        uint8_t soft[16][464];               // [quad][depunctured soft byte of the symbol] (432 used; 464 = 116 dwords:
                                             //  the 16 rows start 52 q mod 64 banks apart, all distinct)
    };
};

struct Q4Shared {
    uint32_t qam[641];
    double2 tw[64];                          // exp(-2 pi j k / 64)
    uint16_t pos[kNumRates][288];            // demodulated byte (carrier * bpsc + bit) -> depunctured position
    int8_t dindex[64];                       // subcarrier index -> data carrier 0..47, -1 otherwise
    int8_t polarity[128];                    // phase_tracker.cpp:23-32
    RateRow rates[kNumRates];                // rates.h:52-196
    Q4Wave w[kQ4Waves];
    double2 taps[kQ4Waves][2][kQ4TapRows][64];  // per wave: the channel estimates (channel_est.cpp:53-58, one 1-KB row per alignment) of the group it
                                             // is working on, fetched by LDS-DMA while the group before is worked on
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
template <int BPSC, typename Taps>
__device__ __forceinline__ void q4_emit(Q4Shared &sh, Q4Wave &ws, const cpx (&X)[16], cpx rot, Taps h, const RateRow &rr, int rate,
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
        // the four taps of this round first, whether or not the bin carries data: one round trip per round instead of
        // one per bin (inside the `data carrier?` branch every load would be waited for on the spot)
        double2 hh4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) hh4[j] = h((16 * j + 4 * m + a + 32) & 63);
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
        // (the lane's share of the addresses below does not change from group to group: hoisted out of the caller's loop it is three more
        // registers held across it, and the compiler spills them -- a scratch reload here waits for the samples in flight)
        int ms = m, qs = qd;
        asm volatile("" : "+v"(ms), "+v"(qs));
        const int ngroups = rr.dbps / 4;
        for (int g4 = ms; g4 < ngroups; g4 += 4) *(uint2 *)(sp + my_out + 4 * g4) = *(const uint2 *)&ws.soft[qs][8 * g4];
    }
}

// ---- the kernel: persistent workgroups, three groups of symbols in flight per wave -----------------------------------------------------
//
// Rounds 1-5 launched one workgroup per 64 symbols.  Alone on the machine a wave of that kernel lived ~40 000 clocks for ~7 000 clocks
// of its own issue: the tables went into LDS once per workgroup, and then one memory round trip waited for the next -- symbol -> frame,
// frame record, descriptor, samples, four rounds of taps -- with three waves per SIMD to hide them behind.  It ran at 2.3 TB/s = 29 %
// of HBM with the vector pipes a third busy, and it is what bounds the low rates (24 trellis steps per symbol instead of 216).
// Now a workgroup stays (the grid is what fits on the machine at once), fills its tables once and walks over groups of 64 symbols g,
// g + G, g + 2 G ...; each wave keeps THREE groups in flight: while it computes group j, the samples of group j + 1 are on their way
// into registers and its channel estimates into LDS (LDS-DMA: no registers), the frame record and descriptor of group j + 2 are on
// their way, and the symbol -> frame entry of group j + 3 is.  No load is waited for in the half-iteration that issues it, and the
// equaliser reads its taps from LDS (1 KB per alignment and group instead of 1 KB per SYMBOL through the vector memory path).
struct Q4Pend {                              // frame record and descriptor of a quad's symbol, loads in flight
    int fq;                                  // sym2frame entry (-1: no symbol)
    int src;                                 // the alignment the window comes from
    int where;                               // a special symbol's vector | fresh << 16 | (late + 1) << 24; fresh 64, late -1 for a plain one
    int rate, sym_off, flags;                // FrameInfo::rate, sym_off, flags
    int2 dec_off;                            // FrameInfo::dec_off
    longlong2 d0;                            // foa_frame_desc::lts1_pos, rot_start
    double2 d1;                              // c, s
};
struct Q4Sym {                               // a quad's symbol, samples in flight
    int rate;                                // -1: no symbol
    int k;                                   // vector of alignment src | fresh << 16 | slow << 24 | late << 25 | row << 26 (slow: not every sample takes (c, s),
                                             // or the vector is a partial one; row: the channel estimate's row in the wave's staging, >= kQ4TapRows: not staged)
    int src;
    int64_t my_out;
    double c, s;
};

// the equaliser's taps of one symbol: a row of the wave's LDS staging, or -- a group with more alignments than rows -- memory
struct Q4TapsLds { const double2 *row; __device__ __forceinline__ double2 operator()(int bin) const { return row[bin]; } };
struct Q4TapsMem { const double2 *__restrict__ row; __device__ __forceinline__ double2 operator()(int bin) const { return row[bin]; } };

// channel_est.cpp:77-81 + phase_tracker.cpp:83-99, then the soft bytes: per modulation when the wave's sixteen symbols share a rate (the usual
// case: loop bounds and table rows are then scalar), generically otherwise.  UNIFORM_OK: offer the per-modulation forms (the memory-tap
// form of a group with more alignments than rows does without them).
template <bool UNIFORM_OK, typename Taps>
__device__ __forceinline__ void q4_finish(Q4Shared &sh, Q4Wave &ws, const cpx (&X)[16], Taps h, int rate, int k, int qd, int m, bool valid, int64_t w,
                                          int64_t my_out, uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    cpx pe = { 0.0, 0.0 };
    {
#pragma clang fp contract(off)
        // pilots: subcarrier 11 = bin 43 (j 2, m 2, a 3), 25 = bin 57 (j 3, m 2, a 1), 39 = bin 7 (j 0, m 1, a 3), 53 = bin 21 (j 1, m 1, a 1)
        const int ps[4] = { 11, 25, 39, 53 }, xi[4] = { 14, 7, 12, 5 }, owner[4] = { 2, 2, 1, 1 };
        const double sgn[4] = { 1.0, 1.0, 1.0, -1.0 };
        const double pol = (double)sh.polarity[k % 127];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const double2 hh = h(ps[p]);
            const cpx mine = cmul(cpx{ hh.x, hh.y }, X[xi[p]]);    // meaningful in lane owner[p] of the quad only
            const cpx zp = { __shfl(mine.x, owner[p], 4), __shfl(mine.y, owner[p], 4) };
            const double pil = (double)(int)(sgn[p] * pol);
            pe.x += (zp.x * pil) / 4.0;
            pe.y += (zp.y * pil) / 4.0;
        }
    }
    const cpx rot = unit_conj(pe);
    const int rate_u = __builtin_amdgcn_readfirstlane(rate);
    if (UNIFORM_OK && __all(rate == rate_u)) {
        const RateRow ru = sh.rates[rate_u];
        switch (ru.bpsc) {
        case 1: q4_emit<1>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        case 2: q4_emit<2>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        case 4: q4_emit<4>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        default: q4_emit<6>(sh, ws, X, rot, h, ru, rate_u, qd, m, valid, w, my_out, sp, eq_tap); break;
        }
    } else {
        const RateRow rr = sh.rates[rate];
        q4_emit<0>(sh, ws, X, rot, h, rr, rate, qd, m, valid, w, my_out, sp, eq_tap);
    }
}

// one 1-KB row of channel estimates -> LDS, 16 bytes per lane, no registers (the walk kernel's idiom, viterbi_tb.h)
__device__ __forceinline__ void q4_row_to_lds(const double2 *src_row, void *lds_row, int lane)
{
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds_row;
    const double2 *p = src_row + lane;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(p), "s"(lds0)
                 : "memory");
}

template <typename S>
__global__ __launch_bounds__(64 * kQ4Waves, 2)
void k_data_symbols_q4(const S *__restrict__ iq, const foa_frame_desc *__restrict__ descs, const FrameInfo *__restrict__ info,
                       const int32_t *__restrict__ sym2frame, const SpecSym *__restrict__ spec, const int64_t *__restrict__ totals,
                       const double2 *__restrict__ hinv, uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    __shared__ Q4Shared sh;
    // (No wave priority: what matters is that these waves do not go ahead of the forward pass's; profiles/r03_ab_fwd_prio.txt.)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), qd = lane >> 2, m = lane & 3;      // (wave: a scalar, also to the compiler)
    const int64_t total = min(totals[0], totals[3]);
    if ((int64_t)blockIdx.x * kQ4Waves * 16 >= total) return;       // the grid is sized by an upper bound: surplus blocks leave at once
    for (int i = tid; i < 641; i += 64 * kQ4Waves) sh.qam[i] = g_tab.qam_lut[i];
    static_assert(sizeof(sh.pos) == sizeof(g_tab.sym_pos) && sizeof(sh.pos) % 4 == 0 && offsetof(DeviceTables, sym_pos) % 4 == 0 &&
                  offsetof(Q4Shared, pos) % 4 == 0, "position tables are copied as dwords");
    for (int i = tid; i < (int)(sizeof(sh.pos) / 4); i += 64 * kQ4Waves) ((uint32_t *)sh.pos)[i] = ((const uint32_t *)g_tab.sym_pos)[i];
    if (tid < 64) { sh.tw[tid] = make_double2(g_tab.tw_re[tid], g_tab.tw_im[tid]); sh.dindex[tid] = g_tab.data_index[tid]; }
    if (tid < 128) sh.polarity[tid] = g_tab.polarity[tid];
    if (tid < kNumRates) sh.rates[tid] = g_tab.rates[tid];
    __syncthreads();
    Q4Wave &ws = sh.w[wave];
    double2 (*taps_use)[64] = sh.taps[wave][0], (*taps_fill)[64] = sh.taps[wave][1];      // the rows of the group in work; of the group on its way

    const int64_t stride = (int64_t)gridDim.x * kQ4Waves * 16;     // symbols between a wave's consecutive groups
    int64_t w = ((int64_t)blockIdx.x * kQ4Waves + wave) * 16 + qd;  // this quad's symbol slot in the wave's current group

    // stage 0: the symbol's entry in the symbol -> frame map
    auto map_entry = [&](int64_t ww) -> int { return ww < total ? sym2frame[ww] : -1; };
    // stage 1: the symbol's frame, and where its window comes from: window k of the frame's own alignment as a rule; for a frame that fills
    // on beyond its alignment (sym2frame <= -2, k_scan_apply) the partial vector or a vector of a later alignment: that alignment's window,
    // rotation, channel estimate and symbol count (channel_est.cpp:77-81 and phase_tracker.cpp:74-99 know nothing of frames).
    auto frame_loads = [&](int fq) -> Q4Pend {
        Q4Pend p;
        int f = fq >= 0 ? fq : 0;
        p.fq = fq; p.src = f; p.where = 64 << 16;
        if (__any(fq <= -2)) {                                      // rare: waited for on the spot
            if (fq <= -2) { const SpecSym e = spec[-2 - fq]; f = e.frame; p.src = e.src; p.where = e.k | (e.fresh & 255) << 16 | ((e.fresh >> 8) + 1) << 24; }
        }
        const FrameInfo *ip = info + f;
        p.rate = ip->rate; p.sym_off = ip->sym_off; p.flags = ip->flags;
        p.dec_off = *(const int2 *)&ip->dec_off;
        const foa_frame_desc *dp = descs + p.src;
        p.d0 = *(const longlong2 *)&dp->lts1_pos;
        p.d1 = *(const double2 *)&dp->c;
        return p;
    };
    // stage 2a: the group's channel estimates -> the wave's LDS rows.  A new row wherever a quad's alignment differs from the quad
    // before it (sixteen consecutive symbols: one or two rows as a rule); the quads of a group that needs more rows than there are
    // read their taps from memory.
    auto tap_plan = [&](int src, unsigned long long &starts) -> int {
        const int before = __shfl_up(src, 4);
        const bool first = m == 0 && (qd == 0 || src != before);
        starts = __ballot(first);
        return __popcll(starts & ((2ull << (lane | 3)) - 1)) - 1;               // rows started at or before this quad, less one
    };
    auto tap_fetch = [&](int src, unsigned long long starts, double2 (*rows)[64]) {
        for (int r = 0; r < kQ4TapRows && starts; r++) {
            const int l = __ffsll((unsigned long long)starts) - 1;
            starts &= starts - 1;
            q4_row_to_lds(hinv + (size_t)__builtin_amdgcn_readlane(src, l) * 64, &rows[r][0], lane);
        }
    };
    // stage 2b: the window's samples n = m + 4u.  A data symbol's window starts 224 samples behind LTS1 and the phasor changes at most 8
    // behind it (rot_start), so every sample of the window takes (c, s) -- unless a caller's descriptor says otherwise, or the vector is
    // the partly filled one of fft_symbols.cpp:46-50 (samples fresh .. 63 still hold the window before): the wave looks once.
    // (Loads of lanes without a symbol go to the stream's first samples: no branch, no wait per load.)
    auto sample_loads = [&](const Q4Pend &p, int64_t ww, int row, S (&raw)[16]) -> Q4Sym {
        Q4Sym y;
        const bool valid = p.fq != -1;
        const int fresh = (p.where >> 16) & 255, late_s = p.where >> 24;
        const int late = late_s > 0 ? late_s - 1 : ((p.flags & kInfoLate) ? 1 : 0);      // (a late alignment's windows sit one symbol further on: frontend_kernels.h)
        y.rate = valid ? p.rate : -1;
        const int kf = valid ? (int)(ww - p.sym_off) + 1 : 1;       // 1-based data symbol of the frame (SIGNAL is symbol 0): where its soft bytes go
        const int k = p.fq <= -2 ? (p.where & 0xFFFF) : kf;         // vector of alignment src: which window, which pilot polarity
        y.src = p.src;
        const int64_t start = p.d0.x + 144 + 80 * (int64_t)(k + late);
        y.c = p.d1.x; y.s = p.d1.y;
        const int64_t dec_off = (int64_t)(((uint64_t)(uint32_t)p.dec_off.y << 32) | (uint32_t)p.dec_off.x);
        y.my_out = dec_off + (int64_t)(kf - 1) * sh.rates[valid ? p.rate : 0].dbps;
        const bool slow = valid && (start < p.d0.y || fresh != 64);
        y.k = k | fresh << 16 | (slow ? 1 << 24 : 0) | late << 25 | min(row, kQ4TapRows) << 26;
        const S *base = iq + (valid ? start : 0) + m;
        if (__all(!valid || fresh == 64)) {
#pragma unroll
            for (int u = 0; u < 16; u++) raw[u] = base[4 * u];
        } else {
#pragma unroll
            for (int u = 0; u < 16; u++) raw[u] = base[4 * u - ((valid && m + 4 * u >= fresh) ? 80 : 0)];
        }
        return y;
    };

    // fill the pipeline: group 0's estimates and samples, group 1's map entry
    S raw[16];
    Q4Sym cur;
    int fq1 = map_entry(w + stride);
    {
        const Q4Pend p0 = frame_loads(map_entry(w));
        unsigned long long starts;
        const int row0 = tap_plan(p0.src, starts);
        tap_fetch(p0.src, starts, taps_use);
        cur = sample_loads(p0, w, row0, raw);
    }

    // The memory pipe within an iteration (it returns in order, and the one full wait is at the top, for the samples): [map entry j + 2,
    // frame record and descriptor j + 1] behind the rotation, the record consumed in the middle, behind the first two FFT stages, the
    // map entry an iteration on; [tap rows j + 1, samples j + 1] in the middle, consumed at the top of the next iteration and behind it.
    for (; w - qd < total; w += stride) {
        const bool valid = cur.rate >= 0;
        const int rate = valid ? cur.rate : 0, k = cur.k & 0xFFFF, src = cur.src, row = (cur.k >> 26) & 7;
        const int64_t my_out = cur.my_out, wsym = w;

        // ---- group j: rotate (timing_sync.cpp:124-125) ----
        cpx x[16];
        if (__all(((cur.k >> 24) & 1) == 0)) {
            const cpx r = { cur.c, cur.s };
#pragma unroll
            for (int u = 0; u < 16; u++) x[u] = valid ? rotate_sample(raw[u], r) : cpx{ 0.0, 0.0 };
        } else {
            // (The per-sample choice of phasor is a 64-bit compare and four v_cndmask_b32 on VCC per sample -- the select on VCC issues at 16
            // clocks per wave instruction on this part, tools/probe_issue.hip -- for a choice that nearly always comes out the same way.)
            const foa_frame_desc d = descs[src];
            const int fresh = (cur.k >> 16) & 255;
            const int64_t start = d.lts1_pos + 144 + 80 * (int64_t)(k + ((cur.k >> 25) & 1));
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int n = m + 4 * u;
                const int64_t idx = (n >= fresh ? start - 80 : start) + n;
                const cpx r = idx >= d.rot_start ? cpx{ d.c, d.s } : cpx{ d.c_prev, d.s_prev };
                x[u] = valid ? rotate_sample(raw[u], r) : cpx{ 0.0, 0.0 };
            }
        }
        // ---- groups j + 2 and j + 1: map entry; frame record and descriptor ----
        __builtin_amdgcn_sched_barrier(0);
        const int fq2 = map_entry(w + 2 * stride);
        const Q4Pend nxt = frame_loads(fq1);
        __builtin_amdgcn_sched_barrier(0);
        // (the twiddles depend on the lane only: left to itself the compiler reads all fifteen before the loop and keeps them in 60
        // registers, which the three groups in flight need; an LDS read per use costs four LDS cycles)
        int mt = m;
        asm volatile("" : "+v"(mt));
        // ---- stage 1: operands u = s, s+4, s+8, s+12; twiddle exponent e = n = m + 4 s ----
#pragma unroll
        for (int s = 0; s < 4; s++) {
            cpx y0, y1, y2, y3;
            q4_butterfly(x[s], x[s + 4], x[s + 8], x[s + 12], y0, y1, y2, y3);
            const int e = mt + 4 * s;
            if (e > 0) {
                const double2 t1 = sh.tw[e], t2 = sh.tw[2 * e], t3 = sh.tw[3 * e];
                y1 = cmul(y1, cpx{ t1.x, t1.y }); y2 = cmul(y2, cpx{ t2.x, t2.y }); y3 = cmul(y3, cpx{ t3.x, t3.y });
            }
            x[s] = y0; x[s + 4] = y1; x[s + 8] = y2; x[s + 12] = y3;
        }
        // ---- stage 2 inside each 16-block r: operands u = 4r + 0..3; twiddle exponent e = 4 m ----
        {
            const double2 t1 = sh.tw[4 * mt], t2 = sh.tw[8 * mt], t3 = sh.tw[12 * mt];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                cpx y0, y1, y2, y3;
                q4_butterfly(x[4 * r], x[4 * r + 1], x[4 * r + 2], x[4 * r + 3], y0, y1, y2, y3);
                if (m > 0) { y1 = cmul(y1, cpx{ t1.x, t1.y }); y2 = cmul(y2, cpx{ t2.x, t2.y }); y3 = cmul(y3, cpx{ t3.x, t3.y }); }
                x[4 * r] = y0; x[4 * r + 1] = y1; x[4 * r + 2] = y2; x[4 * r + 3] = y3;
            }
        }
        // ---- group j + 1: its channel estimates set off for the rows not in use, its samples for the registers group j's came in ----
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long starts1;
        const int row1 = tap_plan(nxt.src, starts1);
        tap_fetch(nxt.src, starts1, taps_fill);
        const Q4Sym nx = sample_loads(nxt, w + stride, row1, raw);
        __builtin_amdgcn_sched_barrier(0);
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

        __builtin_amdgcn_sched_barrier(0);
        // ---- equalise, derotate, demap: taps from the wave's staged rows, or (a group with more alignments than rows) from memory ----
        if (__all(row < kQ4TapRows)) q4_finish<true>(sh, ws, X, Q4TapsLds{ &taps_use[row][0] }, rate, k, qd, m, valid, wsym, my_out, sp, eq_tap);
        else q4_finish<false>(sh, ws, X, Q4TapsMem{ hinv + (size_t)src * 64 }, rate, k, qd, m, valid, wsym, my_out, sp, eq_tap);
        cur = nx; fq1 = fq2;
        { double2 (*t)[64] = taps_use; taps_use = taps_fill; taps_fill = t; }
    }
}

}  // namespace foa
