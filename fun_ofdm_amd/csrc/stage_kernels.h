// stage_kernels.h -- per-block entry points: the arithmetic of one reference block at a time, for the
// fun::block<I,O> adaptors of include/fun_ofdm_amd/blocks.hpp.  Inputs/outputs are in the reference's own
// element order (subcarrier index = FFT bin + 32, tagged_vector.h), so lane j = subcarrier j here.
#pragma once

#include "signal_decode.h"

namespace foa {

constexpr int kSymWaves = 4;          // waves (symbols) per block of k_stage_demap

// fft::forward (fft.cpp:50-59) on vectors of 64 complex doubles, in place: one wave per vector
__global__ __launch_bounds__(256) void k_fft_vectors(double2 *__restrict__ v, int n_vec)
{
    __shared__ cpx lds_all[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= n_vec) return;
    double2 x = v[(size_t)i * 64 + lane];
    cpx y = fft64_lane(cpx{ x.x, x.y }, lds_all[wave], lane);
    v[(size_t)i * 64 + lane_subcarrier(lane)] = make_double2(y.x, y.y);
}

// One data symbol's 48 derotated carriers -> the depunctured soft bytes of its dbps trellis steps, two per step (what
// puncturer::depuncture hands viterbi::conv_decode; the forward pass forms its branch metrics from them, viterbi_fwd.h).
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

// channel_est.cpp:44-58: est = (LTS_FREQ_DOMAIN / Y1) / 2 + (LTS_FREQ_DOMAIN / Y2) / 2, one wave per LTS pair
__global__ __launch_bounds__(64) void k_stage_chanest(const double2 *__restrict__ lts_pairs, double2 *__restrict__ hinv, int n)
{
    const int i = blockIdx.x, j = threadIdx.x;
    if (i >= n) return;
    const cpx ref = { (double)g_tab.lts_freq[j], 0.0 };
    cpx est = { 0.0, 0.0 };
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const double2 y = lts_pairs[((size_t)i * 2 + w) * 64 + j];
        const cpx q = cdiv(ref, cpx{ y.x, y.y });
        est.x += q.x / 2.0;
        est.y += q.y / 2.0;
    }
    hinv[(size_t)i * 64 + j] = make_double2(est.x, est.y);
}

// channel_est.cpp:77-81: vector[j] = m_chan_est[j] * vector[j]
__global__ __launch_bounds__(256) void k_stage_equalize(double2 *__restrict__ v, int n_vec, const double2 *__restrict__ hinv, const int32_t *__restrict__ hidx)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (i >= n_vec) return;
    const double2 h = hinv[(size_t)hidx[i] * 64 + j], y = v[(size_t)i * 64 + j];
    const cpx z = cmul(cpx{ h.x, h.y }, cpx{ y.x, y.y });
    v[(size_t)i * 64 + j] = make_double2(z.x, z.y);
}

// phase_tracker.cpp:83-99 with the pilots in natural subcarrier order (lanes 11, 25, 39, 53)
__device__ __forceinline__ cpx pilot_derotate_natural(cpx z, int polarity)
{
#pragma clang fp contract(off)
    const int LP[4] = { 11, 25, 39, 53 };
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

__global__ __launch_bounds__(256) void k_stage_phase(const double2 *__restrict__ v, const int32_t *__restrict__ symbol_count, int n_vec,
                                                     double2 *__restrict__ out48)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), j = threadIdx.x & 63;
    if (i >= n_vec) return;
    const double2 y = v[(size_t)i * 64 + j];
    const cpx z = pilot_derotate_natural(cpx{ y.x, y.y }, (int)g_tab.polarity[symbol_count[i] % 127]);
    const int di = g_tab.data_index[j];
    if (di >= 0) out48[(size_t)i * 48 + di] = make_double2(z.x, z.y);
}

// ppdu::decode_header (ppdu.cpp:168-218) on 48 carriers per header, one wave each
__global__ __launch_bounds__(64) void k_stage_header(const double2 *__restrict__ carriers48, int n, foa_frame_result *__restrict__ results)
{
    __shared__ uint8_t dem[48];
    __shared__ uint64_t decs[24];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= n) return;
    if (lane < 48) {
        uint8_t b;
        qam_decode(carriers48[(size_t)i * 48 + lane].x, 1, 128.0, &b);
        dem[deinterleaved_pos(lane)] = b;
    }
    __syncthreads();
    int rate, length, nsym;
    decode_signal_bits(dem, decs, lane, rate, length, nsym);
    if (lane == 0) {
        foa_frame_result r;
        r.status = rate >= 0 ? FOA_ST_OK : FOA_ST_HEADER_FAIL; r.rate = rate; r.length = length; r.num_symbols = nsym;
        results[i] = r;
    }
}

// viterbi::conv_decode through the kernels of the batch path (foa_conv_decode): the soft bytes of block blockIdx.y go into
// the block's region as they are, two per trellis step -- exactly what the front end hands the forward pass.
__global__ __launch_bounds__(256) void k_conv_sp(const uint8_t *__restrict__ symbols, size_t sym_stride, int T, const FrameInfo *__restrict__ info,
                                                 uint16_t *__restrict__ sp)
{
    const int b = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const uint8_t *s = symbols + (size_t)b * sym_stride + 2 * (size_t)t;
    sp[info[b].dec_off + t] = (uint16_t)(s[0] | (s[1] << 8));
}

// ... and the decoded bytes (MSB-first, as the chain-back kernels leave them) of every block, packed
__global__ __launch_bounds__(256) void k_conv_pack(const uint32_t *__restrict__ decoded, const FrameInfo *__restrict__ info, int nbytes_have, int nbytes_out,
                                                   uint8_t *__restrict__ data)
{
    const int b = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    if (x >= nbytes_out) return;
    const uint8_t *src = (const uint8_t *)(decoded + decoded_word_off(info[b].dec_off));
    data[(size_t)b * nbytes_out + x] = x < nbytes_have ? src[x] : (uint8_t)0;
}

// front half of ppdu::decode_data (ppdu.cpp:238-244) from derotated carriers: one wave per data symbol
__global__ __launch_bounds__(64 * kSymWaves) void k_stage_demap(const double2 *__restrict__ carriers, const int64_t *__restrict__ car_off,
                                                                const FrameInfo *__restrict__ info, const int32_t *__restrict__ sym2frame,
                                                                int n_sym, uint16_t *__restrict__ sp)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[kSymWaves][448];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * kSymWaves + wave;
    if (w >= n_sym) return;
    const int f = sym2frame[w];
    const FrameInfo fi = info[f];
    const int k = w - fi.sym_off;                                  // 0-based data symbol of the frame
    const RateRow rr = g_tab.rates[fi.rate];
    cpx z = { 0.0, 0.0 };
    const int di = lane < 48 ? lane : -1;
    if (di >= 0) { const double2 c = carriers[car_off[f] + (size_t)k * 48 + lane]; z = cpx{ c.x, c.y }; }
    emit_symbol_soft(z, di, rr, stage_all[wave], sp + fi.dec_off + (int64_t)k * rr.dbps, lane);
}

}  // namespace foa
