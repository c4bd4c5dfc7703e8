// frontend_kernels.h -- post-sync samples -> depunctured soft bytes, for gfx950 (wave64).
//
// Replaces fft_symbols::work + fft::forward (fft_symbols.cpp:33-79, fft.cpp:50-59),
// channel_est::work (channel_est.cpp:36-85), phase_tracker::work (phase_tracker.cpp:70-104) and
// the front half of ppdu::decode_header / decode_data (ppdu.cpp:168-218,223-244):
// modulator::demodulate (modulator.cpp:108-164, qam.h:110-125), interleaver::deinterleave
// (interleaver.cpp:28-38) and puncturer::depuncture (puncturer.cpp:78-123).
//
// Mapping: one wavefront per OFDM symbol, lane = FFT point.  The 64-point DFT is three radix-4
// decimation-in-frequency stages exchanged through LDS; its output stays digit-reversed in the
// lanes (lane p holds bin k = rev4(p)), and every later per-subcarrier step indexes its tables by
// the lane's subcarrier, so no reordering pass exists.  Arithmetic is fp64 like the reference
// (std::complex<double> end to end, tagged_vector.h:46): the 20 MS/s stream is float in HBM and is
// widened on load exactly as the CPU receiver widens it, so soft bytes agree with the CPU to the
// last bit except where a carrier lands within ~1e-15 of a truncation boundary of qam.h:112.
// The stage is HBM-light (8 B/sample in, <= 5.4 B/sample out) and nowhere near any roofline; it is
// written for exactness first.
#pragma once

#include "foa_common.h"

namespace foa {

__constant__ DeviceTables g_tab;

struct cpx { double x, y; };

__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return { a.x + b.x, a.y + b.y }; }
__device__ __forceinline__ cpx cneg(cpx a) { return { -a.x, -a.y }; }

// The reference multiplies with the plain four-product formula (libgcc __muldc3 fast path); keep
// the compiler from fusing it into FMAs so that rounding matches the CPU bit for bit.
__device__ __forceinline__ cpx cmul(cpx a, cpx b)
{
#pragma clang fp contract(off)
    double ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
    return { ac - bd, ad + bc };
}

// Complex division, Smith's method as in libgcc's __divdc3 for operands in normal range
// (channel_est.cpp:55-57 divides LTS_FREQ_DOMAIN[j] by the received carrier).
__device__ __forceinline__ cpx cdiv(cpx a, cpx b)
{
#pragma clang fp contract(off)
    cpx r;
    if (fabs(b.x) < fabs(b.y)) {
        double ratio = b.x / b.y, denom = b.x * ratio + b.y;
        r.x = (a.x * ratio + a.y) / denom;
        r.y = (a.y * ratio - a.x) / denom;
    } else {
        double ratio = b.y / b.x, denom = b.y * ratio + b.x;
        r.x = (a.x + a.y * ratio) / denom;
        r.y = (a.y - a.x * ratio) / denom;
    }
    return r;
}

// multiply by (-j)^q
__device__ __forceinline__ cpx rot_mj(cpx z, int q)
{
    cpx r = z;
    if (q == 1) r = { z.y, -z.x };
    else if (q == 2) r = { -z.x, -z.y };
    else if (q == 3) r = { -z.y, z.x };
    return r;
}

__device__ __forceinline__ void wave_lds_sync()
{
    // LDS traffic of one wave is ordered; this only stops the compiler from moving accesses and
    // waits for outstanding LDS operations of this wave.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Subcarrier index (reference numbering: index = bin + 32 mod 64, fft.cpp:20-24) held by lane p
// after fft64_lane().
__device__ __forceinline__ int lane_subcarrier(int p)
{
    int k = (p >> 4) | (((p >> 2) & 3) << 2) | ((p & 3) << 4);
    return (k + 32) & 63;
}
// inverse: lane that holds subcarrier index s
__host__ __device__ constexpr int subcarrier_lane(int s)
{
    int k = (s + 32) & 63;
    return ((k & 3) << 4) | (((k >> 2) & 3) << 2) | (k >> 4);
}

// 64-point forward DFT across the wave; lane n supplies x[n], lane p returns X[rev4(p)].
// lds: 64 cpx private to this wave.
__device__ __forceinline__ cpx fft64_lane(cpx v, cpx *lds, int lane)
{
#pragma unroll
    for (int st = 0; st < 3; st++) {
        const int span = 16 >> (2 * st);
        lds[lane] = v;
        wave_lds_sync();
        const int m = (lane / span) & 3;
        const int base = lane - m * span;
        cpx a = lds[base], b = lds[base + span], c = lds[base + 2 * span], d = lds[base + 3 * span];
        wave_lds_sync();
        cpx bb = rot_mj(b, m), cc = (m & 1) ? cneg(c) : c, dd = rot_mj(d, (3 * m) & 3);
        cpx y = cadd(cadd(a, cc), cadd(bb, dd));
        if (st < 2) {
            const int e = (lane & (span - 1)) * m * (16 / span);
            cpx w = { g_tab.tw_re[e], g_tab.tw_im[e] };
            y = cmul(y, w);
        }
        v = y;
    }
    return v;
}

// timing_sync.cpp:124-125 rotation + float->double widening of one window sample
__device__ __forceinline__ cpx load_rotated(const float2 *iq, int64_t idx, const foa_frame_desc &d)
{
    float2 s = iq[idx];
    cpx v = { (double)s.x, (double)s.y };
    cpx r = idx >= d.rot_start ? cpx{ d.c, d.s } : cpx{ d.c_prev, d.s_prev };
    return cmul(v, r);
}

// phase_tracker.cpp:97-98 rotates by (cos(-angle), sin(-angle)) with angle = arg(pe): that is conj(pe)/|pe|.
// Computing it as such (one sqrt, one divide) instead of atan2 + cos + sin removes ~250 fp64 instructions per
// symbol; both forms are within an ulp or two of the exact value, like the host's libm, and eight orders below
// the 1e-4 parity tolerance.  pe == 0 (no pilots at all) gives angle 0 in the reference.
__device__ __forceinline__ cpx unit_conj(cpx pe)
{
#pragma clang fp contract(off)
    const double r2 = pe.x * pe.x + pe.y * pe.y;
    if (!(r2 > 0.0)) {
        // zero, NaN or underflow: fall back to the reference's own sequence of calls
        const double angle = atan2(pe.y, pe.x);
        return cpx{ cos(-angle), sin(-angle) };
    }
    const double r = sqrt(r2);
    return cpx{ pe.x / r, -pe.y / r };
}

// phase_tracker.cpp:83-99 for one symbol: returns the derotated carrier of this lane
__device__ __forceinline__ cpx pilot_derotate(cpx z, int polarity)
{
#pragma clang fp contract(off)
    constexpr int LP[4] = { subcarrier_lane(11), subcarrier_lane(25), subcarrier_lane(39), subcarrier_lane(53) };
    const double sgn[4] = { 1.0, 1.0, 1.0, -1.0 };
    cpx pe = { 0.0, 0.0 };
#pragma unroll
    for (int p = 0; p < 4; p++) {
        double px = __shfl(z.x, LP[p]), py = __shfl(z.y, LP[p]);
        double pil = (double)(int)(sgn[p] * (double)polarity);
        pe.x += (px * pil) / 4.0;
        pe.y += (py * pil) / 4.0;
    }
    return cmul(z, unit_conj(pe));
}

// qam.h:110-125; `int pt = sym * d_scale_d` has cvttsd2si semantics on the reference's platform
__device__ __forceinline__ int trunc_to_int(double v)
{
    return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : (int)0x80000000;
}

__device__ __forceinline__ void qam_decode(double sym, int nb, double scale_d, uint8_t *bits)
{
#pragma clang fp contract(off)
    uint32_t pt = (uint32_t)trunc_to_int(sym * scale_d);
    int flip = 1, amp = 128;
    for (int i = 0; i < nb; i++) {
        int v = (int)((uint32_t)flip * pt + 128u);
        bits[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        int bit = ((int)pt < 0) ? -1 : 1;
        pt -= (uint32_t)(bit * amp);
        flip = -bit;
        amp >>= 1;
    }
}

// interleaver.h:66-75 with (48,1): index(k) = 3*(k%16) + k/16; its inverse
__device__ __forceinline__ int deinterleaved_pos(int w) { return 16 * (w % 3) + w / 3; }

// position of deinterleaved coded byte D (frame-wide numbering) in the depunctured stream
// (puncturer.cpp:94-102,112-118)
__device__ __forceinline__ int depunct_pos(int D, int punct)
{
    if (punct == 2) { const int t[4] = { 0, 1, 3, 5 }; return 6 * (D >> 2) + t[D & 3]; }
    if (punct == 1) { const int t[3] = { 0, 2, 3 }; return 4 * (D / 3) + t[D % 3]; }
    return D;
}

// ---- the lane-per-state K=7 ACS step shared by the SIGNAL decode and viterbi v1 ----
// viterbi.cpp:208-457.
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

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

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

// One data symbol's 48 derotated carriers -> the depunctured soft bytes of its dbps trellis steps, two per step (what
// puncturer::depuncture hands viterbi::conv_decode; the forward pass forms its branch metrics from them, viterbi_v3.h).
// Lane with data index di >= 0 holds carrier z.  stage: 448 B of LDS private to the wave.
// modulator.cpp:108-164 / qam.h:110-125, interleaver.cpp:28-38, puncturer.cpp:78-123.
__device__ __forceinline__ void emit_symbol_soft(cpx z, int di, const RateRow &rr, uint8_t *stage, uint16_t *sp_dst, int lane)
{
    // erasures first (puncturer.cpp:98,100,114), then scatter this carrier's soft bytes
    const int out_bytes = 2 * rr.dbps;                         // depunctured bytes of this symbol
    if (rr.punct != 0) {
        uint32_t *st32 = (uint32_t *)stage;
        for (int i = lane; i < out_bytes / 4; i += 64) st32[i] = 0x7F7F7F7Fu;
    }
    wave_lds_sync();
    if (di >= 0) {
        uint8_t bits[6];
        qam_decode(z.x, rr.numbits, rr.scale_d, bits);
        if (rr.bpsc > 1) qam_decode(z.y, rr.numbits, rr.scale_d, bits + rr.numbits);
        for (int b = 0; b < rr.bpsc; b++) {
            int c = di * rr.bpsc + b;                          // demodulated byte index within the symbol
            int dd = 48 * (c / 48) + deinterleaved_pos(c % 48);
            stage[depunct_pos(dd, rr.punct)] = bits[b];        // symbol-local: cbps is a multiple of 12
        }
    }
    wave_lds_sync();
    uint32_t *dst = (uint32_t *)sp_dst;                        // (symbols start on 8-byte boundaries: dbps is a multiple of 4)
    const uint32_t *st32 = (const uint32_t *)stage;
    for (int i = lane; i < out_bytes / 4; i += 64) dst[i] = st32[i];
}

// =================================================================================================
// K1: per alignment: LTS1 + LTS2 + SIGNAL.  Channel estimate -> hinv, SIGNAL decode -> FrameInfo.
// One wave (64 threads) per block, one block per frame.
// =================================================================================================
__global__ __launch_bounds__(64) void k_header(const float2 *__restrict__ iq, const foa_frame_desc *__restrict__ descs,
                                               const int64_t *__restrict__ ends, int64_t n_samples, int n_frames,
                                               FrameInfo *__restrict__ info, double2 *__restrict__ hinv, double2 *__restrict__ eq_tap)
{
    __shared__ cpx lds[64];
    __shared__ uint8_t dem[48];
    __shared__ uint64_t decs[24];
#if defined(FOA_HDR_PRIO) && FOA_HDR_PRIO
    __builtin_amdgcn_s_setprio(FOA_HDR_PRIO);          // (A/B only)
#endif
    const int f = blockIdx.x, lane = threadIdx.x;
    if (f >= n_frames) return;
    const foa_frame_desc d = descs[f];
    const int64_t end = min(ends[f], n_samples), p = d.lts1_pos;      // nothing is read beyond the stream, whatever the caller's ends say
    FrameInfo fi;
    fi.status = FOA_ST_HEADER_FAIL; fi.rate = -1; fi.length = 0; fi.nsym = 0; fi.sym_off = 0; fi.nsteps = 0;
    fi.soft_off = 0; fi.dec_off = 0;
    if (p < 0 || p + 208 > end) {                     // LTS or SIGNAL window cut off
        fi.status = FOA_ST_TRUNCATED;
        if (lane == 0) info[f] = fi;
        return;
    }
    const int s = lane_subcarrier(lane);
    // channel_est.cpp:44-58: est = sum over the two LTS of (LTS_FREQ_DOMAIN / Y) / 2
    cpx est = { 0.0, 0.0 };
    const cpx ref = { (double)g_tab.lts_freq[s], 0.0 };
#pragma unroll
    for (int w = 0; w < 2; w++) {
        cpx y = fft64_lane(load_rotated(iq, p + 64 * w + lane, d), lds, lane);
        cpx q = cdiv(ref, y);
        est.x += q.x / 2.0;
        est.y += q.y / 2.0;
    }
    hinv[(size_t)f * 64 + s] = make_double2(est.x, est.y);
    // SIGNAL: equalise (channel_est.cpp:77-81), pilot phase with polarity[0] (phase_tracker.cpp:74-99)
    cpx y = fft64_lane(load_rotated(iq, p + 144 + lane, d), lds, lane);
    cpx z = pilot_derotate(cmul(est, y), (int)g_tab.polarity[0]);
    const int di = g_tab.data_index[s];
    if (eq_tap && di >= 0) eq_tap[(size_t)f * 48 + di] = make_double2(z.x, z.y);   // SIGNAL tap: one row per frame
    // ppdu.cpp:171-175: BPSK demap, deinterleave
    if (di >= 0) {
        uint8_t b;
        qam_decode(z.x, 1, 128.0, &b);
        dem[deinterleaved_pos(di)] = b;
    }
    __syncthreads();
    int rate, length, nsym;
    decode_signal_bits(dem, decs, lane, rate, length, nsym);
    if (lane == 0) {
        if (rate >= 0) {
            fi.rate = rate; fi.length = length;
            if (p + 144 + 80 * (int64_t)nsym + 64 > end) { fi.status = FOA_ST_TRUNCATED; fi.nsteps = -nsym; }   // nsym still reported
            else { fi.status = FOA_ST_CRC_FAIL; fi.nsym = nsym; fi.nsteps = nsym * g_tab.rates[rate].dbps; }   // pending until the CRC is checked
        }
        info[f] = fi;
    }
}

// =================================================================================================
// K2: exclusive scans over frames -> sym_off / soft_off / dec_off / seg_off, and the symbol / segment maps.
// =================================================================================================
// words reserved per frame in the per-step buffers (bm / dec / decoded): the chain-back reads whole 48-step chunks
__host__ __device__ constexpr int64_t dec_words(int64_t nsteps) { return nsteps > 0 ? (nsteps + 48 + 63) & ~(int64_t)63 : 0; }
// chain-back segments of a frame (viterbi_v3.h): its nsteps - 6 data steps in pieces of seg_steps
__host__ __device__ constexpr int tb_segments(int nsteps, int seg_steps) { return nsteps > 6 ? (nsteps - 6 + seg_steps - 1) / seg_steps : 0; }

// Three small kernels instead of one block: a single CU's memory pipeline (one 64-line request per wave instruction)
// made the one-block version the 50 us tail of a 2 ms call.  One thread per frame throughout.
constexpr int kScanBlock = 256;

struct ScanQ { int64_t v[4]; };          // data symbols, (unused), per-step words (soft pairs / decisions / decoded), chain-back segments

__device__ __forceinline__ ScanQ scan_quantities(const FrameInfo *info, int f, int n_frames, int seg_steps)
{
    ScanQ q = { { 0, 0, 0, 0 } };
    if (f < n_frames) {
        const int nsym = info[f].nsym, nsteps = nsym > 0 ? info[f].nsteps : 0;
        q.v[0] = nsym; q.v[1] = 0; q.v[2] = dec_words(nsteps); q.v[3] = tb_segments(nsteps, seg_steps);
    }
    return q;
}

// exclusive scan over the block's threads; totals in tot[] (valid in every thread)
__device__ __forceinline__ ScanQ block_exclusive_scan(const ScanQ &q, int64_t (&tot)[4], int64_t (*part)[kScanBlock / 64])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    ScanQ r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t x = q.v[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) part[i][wv] = x;
        r.v[i] = x - q.v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int64_t run = 0;
#pragma unroll
        for (int w = 0; w < kScanBlock / 64; w++) {
            if (w == wv) r.v[i] += run;
            run += part[i][w];
        }
        tot[i] = run;
    }
    return r;
}

// pass 1: per-block sums -> blk[4][n_blocks]
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(const FrameInfo *__restrict__ info, int n_frames, int seg_steps, int64_t *__restrict__ blk)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    int64_t tot[4];
    block_exclusive_scan(scan_quantities(info, blockIdx.x * kScanBlock + threadIdx.x, n_frames, seg_steps), tot, part);
    if (threadIdx.x < 4) blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = tot[threadIdx.x];
}

// pass 2: exclusive scan of the block sums in place, grand totals -> totals[0..2] and totals[4]  (one block)
__global__ __launch_bounds__(1024) void k_scan_blocks(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    if (threadIdx.x == 0) totals[3] = sym_cap;            // read by the data-symbol kernels next to totals[0]
    __shared__ int64_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = 0; i < 4; i++) {
        int64_t carry = 0;
        for (int base = 0; base < n_blocks; base += 1024) {
            const int j = base + tid;
            const int64_t v = j < n_blocks ? blk[(size_t)i * n_blocks + j] : 0;
            int64_t x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int64_t y = __shfl_up(x, o);
                if (lane >= o) x += y;
            }
            __syncthreads();
            if (lane == 63) wsum[wv] = x;
            __syncthreads();
            int64_t before = 0, all = 0;
            for (int w = 0; w < 16; w++) {
                if (w < wv) before += wsum[w];
                all += wsum[w];
            }
            if (j < n_blocks) blk[(size_t)i * n_blocks + j] = carry + before + x - v;
            carry += all;
        }
        if (tid == 0) totals[i < 3 ? i : 4] = carry;          // totals[3] is the symbol capacity (above)
    }
}

// pass 2 for up to 64 x 64 blocks (a million frames) as ONE wave: under a forward pass a 1024-thread block waits for sixteen
// free wave slots on one CU (5.8 us alone, 100 us on average and up to 750 us in the pipelined bench), a lone wave for one
__global__ __launch_bounds__(64) void k_scan_blocks_w(int64_t *__restrict__ blk, int n_blocks, int64_t sym_cap, int64_t *__restrict__ totals)
{
    const int lane = threadIdx.x, per = (n_blocks + 63) / 64, lo = lane * per, hi = min(lo + per, n_blocks);
    if (lane == 0) totals[3] = sym_cap;
    for (int i = 0; i < 4; i++) {
        int64_t s = 0;
        for (int j = lo; j < hi; j++) s += blk[(size_t)i * n_blocks + j];
        int64_t x = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int64_t y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        int64_t run = x - s;                                   // exclusive prefix of this lane's run
        for (int j = lo; j < hi; j++) { const int64_t v = blk[(size_t)i * n_blocks + j]; blk[(size_t)i * n_blocks + j] = run; run += v; }
        if (lane == 63) totals[i < 3 ? i : 4] = x;
    }
}

// pass 3: offsets into the frame records (frames that do not fit are marked FOA_ST_NO_SPACE), and the symbol -> frame
// and segment -> frame maps (so that the data-symbol and chain-back kernels find their frame without a search;
// frames are short: <= 1368 symbols)
__global__ __launch_bounds__(kScanBlock) void k_scan_apply(FrameInfo *__restrict__ info, int n_frames, int64_t sym_cap,
                                                           int64_t dec_cap, int seg_steps, int64_t seg_cap, const int64_t *__restrict__ blk,
                                                           int32_t *__restrict__ sym2frame, int32_t *__restrict__ seg2frame)
{
    __shared__ int64_t part[4][kScanBlock / 64];
    const int f = blockIdx.x * kScanBlock + threadIdx.x;
    const ScanQ q = scan_quantities(info, f, n_frames, seg_steps);
    int64_t tot[4];
    const ScanQ ex = block_exclusive_scan(q, tot, part);
    if (f >= n_frames) return;
    const int64_t a = ex.v[0] + blk[blockIdx.x],
                  c = ex.v[2] + blk[(size_t)2 * gridDim.x + blockIdx.x], d = ex.v[3] + blk[(size_t)3 * gridDim.x + blockIdx.x];
    const int nsym = (int)q.v[0];
    info[f].seg_off = (int32_t)d;
    const int ns = (int)q.v[3];
    if (nsym > 0 && (a + nsym > sym_cap || c + q.v[2] > dec_cap)) {
        // keeps its slots in the numbering; they are marked unused (as far as the maps reach)
        info[f].status = FOA_ST_NO_SPACE; info[f].nsteps = -nsym; info[f].nsym = 0;
        for (int k = 0; k < nsym; k++) if (a + k < sym_cap) sym2frame[a + k] = -1;
        for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = -1;
        return;
    }
    info[f].sym_off = (int32_t)a; info[f].soft_off = 2 * c; info[f].dec_off = c;    // (soft_off: byte offset of the frame's soft pairs)
    for (int k = 0; k < nsym; k++) sym2frame[a + k] = f;
    for (int k = 0; k < ns; k++) if (d + k < seg_cap) seg2frame[d + k] = f;
}

// =================================================================================================
// K3: data symbols.  One wave per symbol, 4 waves per block.
// =================================================================================================
constexpr int kSymWaves = 4;
#if FOA_XCHECK      // cross-check build only: the wave-per-symbol front end

__global__ __launch_bounds__(64 * kSymWaves) void k_data_symbols(const float2 *__restrict__ iq, const foa_frame_desc *__restrict__ descs,
                                                                 const FrameInfo *__restrict__ info, const int32_t *__restrict__ sym2frame,
                                                                 const int64_t *__restrict__ totals, const double2 *__restrict__ hinv,
                                                                 uint16_t *__restrict__ sp, double2 *__restrict__ eq_tap)
{
    __shared__ cpx lds_all[kSymWaves][64];
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[kSymWaves][448];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * kSymWaves + wave;
    if (w >= totals[0] || w >= totals[3]) return;              // totals[3] = symbol capacity
    cpx *lds = lds_all[wave];
    uint8_t *stage = stage_all[wave];
    const int f = sym2frame[w];
    if (f < 0) return;
    const FrameInfo fi = info[f];
    const int k = (int)(w - fi.sym_off) + 1;                  // 1-based data symbol (SIGNAL is symbol 0)
    const foa_frame_desc d = descs[f];
    const RateRow rr = g_tab.rates[fi.rate];
    const int s = lane_subcarrier(lane);

    cpx y = fft64_lane(load_rotated(iq, d.lts1_pos + 144 + 80 * (int64_t)k + lane, d), lds, lane);
    double2 h = hinv[(size_t)f * 64 + s];
    cpx z = pilot_derotate(cmul(cpx{ h.x, h.y }, y), (int)g_tab.polarity[k % 127]);
    const int di = g_tab.data_index[s];
    if (eq_tap && di >= 0) eq_tap[(size_t)w * 48 + di] = make_double2(z.x, z.y);          // data tap: one row per symbol

    emit_symbol_soft(z, di, rr, stage, sp + fi.dec_off + (int64_t)(k - 1) * rr.dbps, lane);
}

#endif  // FOA_XCHECK

}  // namespace foa
