// tx_kernels.h -- frame_builder::build_frame on the device, plus the synthetic channel of SURVEY 8d.
//
// The transmit side is not part of the receive hot path; it exists so that large synthetic workloads can be created
// in HBM (80 000 frames are 2.25 GB of samples before padding) and so that device-resident loop-back tests are
// possible.  It reproduces the reference's deviations from 802.11a exactly like the oracle's fo_build_frame:
//   frame_builder.cpp:53-82   preamble table, SIGNAL + data symbols, 16-sample cyclic prefix
//   ppdu.cpp:65-165           header field, service + payload + CRC-32 (little endian), byte-wise LSB scrambler
//                             (pad bytes scrambled too, tail not zeroed), K=7 encoder, puncturing, 48-entry interleaver
//   symbol_mapper.cpp:81-119  48 data + 4 pilot carriers, polarity sequence
//   fft.cpp:68-96             index shift, unscaled inverse DFT, 1/64
// Every OFDM symbol depends only on its own dbps bits and the six bits before them (the encoder's memory), so the
// symbols of all frames are built in parallel, one thread each; the IFFT is the front end's radix-4 network applied
// to the conjugate.
#pragma once

#include "device_math.h"

namespace foa {

// ---- K1: one thread per frame: service + payload + CRC-32, scrambled -> scr[f][0 .. nbytes] (ppdu.cpp:125-153) ----
__global__ void k_tx_prepare(const uint8_t *__restrict__ payload, size_t payload_pitch, int length, int n_frames, int nbytes,
                             uint8_t *__restrict__ scr, size_t stride)
{
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t *p = payload + (size_t)f * payload_pitch;
    uint8_t *o = scr + (size_t)f * stride;
    uint32_t crc = 0xFFFFFFFFu;
    crc = g_tab.crc_table[(crc ^ 0u) & 0xFFu] ^ (crc >> 8);                 // the two zero service bytes
    crc = g_tab.crc_table[(crc ^ 0u) & 0xFFu] ^ (crc >> 8);
    for (int i = 0; i < length; i++) crc = g_tab.crc_table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    crc ^= 0xFFFFFFFFu;
    for (int i = 0; i < nbytes; i++) {
        uint32_t b = 0;
        if (i >= 2 && i < 2 + length) b = p[i - 2];
        else if (i >= 2 + length && i < 6 + length) b = (crc >> (8 * (i - 2 - length))) & 0xFFu;
        o[i] = (uint8_t)(b ^ g_tab.scramble[i % 127]);                       // one LFSR step per byte, into bit 0 (ppdu.cpp:141-147)
    }
    o[nbytes] = 0;
}

__device__ __forceinline__ uint32_t tx_bit(const uint8_t *bytes, int i) { return (bytes[i >> 3] >> (7 - (i & 7))) & 1u; }   // MSB first

// qam.h:83-97
__device__ __forceinline__ double tx_qam_encode(const uint8_t *bits, int nb, double scale_e)
{
    int pt = 0, flip = 1;
    for (int i = 0; i < nb; i++) {
        const int bit = (int)bits[i] * 2 - 1;
        pt = bit * flip + pt * 2;
        flip *= -bit;
    }
    return (double)pt * scale_e;
}

// ---- K2: one thread per (frame, symbol y): y = 0 is SIGNAL (and the preamble), y >= 1 the data symbols ----
__global__ __launch_bounds__(64) void k_tx_symbols(const uint8_t *__restrict__ scr, size_t stride, int length, int rate, int nsym, int n_frames,
                                                   double2 *__restrict__ out, size_t frame_samples)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = (int)(gid / (nsym + 1)), y = (int)(gid % (nsym + 1));
    if (f >= n_frames) return;
    double2 *o = out + (size_t)f * frame_samples;
    const RateRow rr = g_tab.rates[y == 0 ? 0 : rate];                       // SIGNAL is BPSK 1/2 (ppdu.cpp:97-108)
    uint8_t coded[432];                                                       // mother-code bits of this symbol: 2 per data bit
    uint8_t hdr[3];
    const uint8_t *bytes;
    int first;                                                                // index of the symbol's first data bit
    if (y == 0) {
        // ppdu.cpp:86-95: parity(1) rate(4) reserved(1) length(12) tail(6), MSB first
        uint32_t field = ((uint32_t)(g_tab.rates[rate].rate_field & 0xF) << 13) | ((uint32_t)length & 0xFFFu);
        if (__popc(field) & 1) field |= 131072u;
        field <<= 6;
        hdr[0] = (uint8_t)(field >> 16); hdr[1] = (uint8_t)(field >> 8); hdr[2] = (uint8_t)field;
        bytes = hdr; first = 0;
        for (int i = 0; i < 320; i++) o[i] = make_double2(g_tab.preamble_re[i], g_tab.preamble_im[i]);
    } else {
        bytes = scr + (size_t)f * stride; first = (y - 1) * rr.dbps;
    }
    // viterbi.cpp:39-62: sr = (sr << 1) | bit, outputs parity(sr & 121), parity(sr & 91); sr enters with the six bits before
    uint32_t sr = 0;
    for (int i = first - 6; i < first; i++) sr = (sr << 1) | (i >= 0 ? tx_bit(bytes, i) : 0u);
    for (int i = 0; i < rr.dbps; i++) {
        sr = (sr << 1) | tx_bit(bytes, first + i);
        coded[2 * i] = (uint8_t)(__popc(sr & 121u) & 1);
        coded[2 * i + 1] = (uint8_t)(__popc(sr & 91u) & 1);
    }
    // puncturer.cpp:43-62 (symbol-local: cbps is a multiple of the pattern), interleaver.cpp:18-26, modulator.cpp:58-104
    uint8_t tx[288];
    for (int c = 0; c < rr.cbps; c++) {
        const int src = rr.punct == 2 ? 6 * (c >> 2) + (c & 3) + ((c & 3) >= 2) + ((c & 3) >= 3) : rr.punct == 1 ? 4 * (c / 3) + (c % 3) + ((c % 3) >= 1) : c;
        const int w = c % 48;
        tx[48 * (c / 48) + 3 * (w % 16) + w / 16] = coded[src];              // interleaver.h:66-75: index(k) = 3 (k mod 16) + k / 16
    }
    const int nb = rr.numbits;
    const double power = rr.bpsc == 1 ? 1.0 : 0.5;
    const int nn = 1 << (nb - 1), sum2 = (4 * nn * nn * nn - nn) / 3;
    const double scale_e = sqrt(power * (double)nn / (double)sum2);          // qam.h:35-51
    cpx x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = cpx{ 0.0, 0.0 };
    // symbol_mapper.cpp:81-119 into fft.cpp:77-80's shifted order: time-domain input index = subcarrier index + 32 mod 64;
    // the radix-4 network is a forward DFT, so feed the conjugate and conjugate the result (inverse = conj o forward o conj)
    const double pol = (double)g_tab.polarity[y % 127];
#pragma unroll
    for (int s = 0; s < 64; s++) {
        const int di = g_tab.data_index[s];
        cpx v = { 0.0, 0.0 };
        if (di >= 0) {
            v.x = tx_qam_encode(tx + di * rr.bpsc, nb, scale_e);
            v.y = rr.bpsc > 1 ? tx_qam_encode(tx + di * rr.bpsc + nb, nb, scale_e) : 0.0;
        } else if (s == 11 || s == 25 || s == 39) {
            v.x = pol;
        } else if (s == 53) {
            v.x = -pol;
        }
        x[(s + 32) & 63] = cpx{ v.x, -v.y };
    }
    fft64_regs(x);
    // time sample n = conj(X[n]) / 64 where X[n] sits in register subcarrier_lane((n + 32) & 63); cyclic prefix = samples 48..63
    double2 *sym = o + 320 + (size_t)80 * y;
#pragma unroll
    for (int n = 0; n < 64; n++) {
        const cpx v = x[subcarrier_lane((n + 32) & 63)];
        const double2 t = make_double2(v.x / 64.0, -v.y / 64.0);
        sym[16 + n] = t;
        if (n >= 48) sym[n - 48] = t;
    }
}

// ---- K3: the synthetic channel of SURVEY 8d: frames at a fixed pitch, per-frame carrier phase (and optional constant
// frequency offset), complex white Gaussian noise, rounding to complex<float>.  One thread per output sample; the random
// numbers come from a counter hash, so the stream depends only on (seed, sample index). ----
__device__ __forceinline__ uint64_t tx_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double tx_unit(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }   // (0, 1)

__global__ void k_tx_channel(const double2 *__restrict__ frames, int64_t n_frames, int64_t frame_samples, int64_t pitch, int64_t lead,
                             double sigma, double cfo_hz, uint64_t seed, float2 *__restrict__ iq)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_frames * pitch) return;
    const int64_t f = i / pitch, o = i % pitch - lead;
    double re = 0.0, im = 0.0;
    if (o >= 0 && o < frame_samples) {
        const double2 v = frames[f * frame_samples + o];
        const double ph = 6.283185307179586476925 * tx_unit(tx_mix64(seed ^ (0x5851F42D4C957F2Dull * (uint64_t)(f + 1))));
        const double fo = cfo_hz > 0.0 ? cfo_hz * (2.0 * tx_unit(tx_mix64(seed + 0x632BE59BD9B4E019ull * (uint64_t)(f + 1))) - 1.0) : 0.0;
        const double a = ph + 6.283185307179586476925 * fo * (double)o / 20e6;
        double s, c;
        sincos(a, &s, &c);
        re = v.x * c - v.y * s; im = v.x * s + v.y * c;
    }
    // Box-Muller on two uniforms of the sample's own counters
    const double u1 = tx_unit(tx_mix64(seed + 2 * (uint64_t)i)), u2 = tx_unit(tx_mix64(~seed + 2 * (uint64_t)i + 1));
    const double r = sigma * sqrt(-2.0 * log(u1));
    double sn, cn;
    sincos(6.283185307179586476925 * u2, &sn, &cn);
    iq[i] = make_float2((float)(re + r * cn), (float)(im + r * sn));
}

}  // namespace foa
