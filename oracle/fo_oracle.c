/*
 * fo_oracle.c -- CPU restatement of the fun_ofdm receive path.  TEST INFRASTRUCTURE ONLY
 * (see fo_oracle.h for who may use it and for how it is pinned).
 *
 * Citations are reference src/ file:line.  Complex arithmetic uses C99 `double _Complex`, which
 * gcc lowers to the same libgcc routines (__muldc3 / __divdc3) as the std::complex<double>
 * operators in the reference, so the fp64 stages agree with the compiled reference bit for bit.
 */
#define _GNU_SOURCE
#include "fo_oracle.h"

#include <complex.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <semaphore.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double _Complex cplx;

static inline cplx to_c(fo_c64 v) { return CMPLX(v.re, v.im); }
static inline fo_c64 from_c(cplx v) { fo_c64 r = { creal(v), cimag(v) }; return r; }

/* ------------------------------------------------------------------------------------------ */
/* rates.h:52-196                                                                              */
/* ------------------------------------------------------------------------------------------ */
static const fo_rate_params RATE_TABLE[FO_NUM_RATES] = {
    /* field cbps dbps bpsc rate punct */
    { 0xD,  48,  24, 1,  0, 0 }, { 0xE,  48,  32, 1,  1, 1 }, { 0xF,  48,  36, 1,  2, 2 },
    { 0x5,  96,  48, 2,  3, 0 }, { 0x6,  96,  64, 2,  4, 1 }, { 0x7,  96,  72, 2,  5, 2 },
    { 0x9, 192,  96, 4,  6, 0 }, { 0xA, 192, 128, 4,  7, 1 }, { 0xB, 192, 144, 4,  8, 2 },
    { 0x1, 288, 192, 6,  9, 1 }, { 0x3, 288, 216, 6, 10, 2 },
};

int fo_rate_params_get(int rate, fo_rate_params *out)
{
    if (rate < 0 || rate >= FO_NUM_RATES) return -1;
    *out = RATE_TABLE[rate];
    return 0;
}

int fo_rate_from_field(int rate_field)
{
    for (int r = 0; r < FO_NUM_RATES; r++)
        if (RATE_TABLE[r].rate_field == rate_field) return r;
    return -1;
}

/* ppdu.cpp:40-44: ceil((16 + 8*(length+4) + 6) / dbps) in double */
int fo_num_symbols(int rate, int length)
{
    return (int)ceil((double)(16 + 8 * (length + 4) + 6) / (double)RATE_TABLE[rate].dbps);
}

/* qam.h:35-51.  NumBits = max(bpsc/2,1); power 1.0 for BPSK, 0.5 otherwise (modulator.cpp:117-157) */
static void qam_params(int rate, int *numbits, double *scale_e, double *scale_d)
{
    int bpsc = RATE_TABLE[rate].bpsc;
    int nb = bpsc == 1 ? 1 : bpsc / 2;
    double power = bpsc == 1 ? 1.0 : 0.5;
    int gain = CHAR_BIT - nb;
    int nn = 1 << (nb - 1);
    int sum2 = (4 * nn * nn * nn - nn) / 3;
    double sf = sqrt(power * (double)nn / (double)sum2);
    *numbits = nb;
    *scale_e = sf;
    *scale_d = (double)(1 << gain) / sf;
}

double fo_demod_scale(int rate)
{
    int nb; double e, d;
    qam_params(rate, &nb, &e, &d);
    return d;
}

/* ------------------------------------------------------------------------------------------ */
/* 64-point DFT (FFTW3 restated by definition; fft.cpp:30-59,68-96)                            */
/* ------------------------------------------------------------------------------------------ */
static cplx TW64[64];            /* exp(-2*pi*j*k/64) */
static fo_c64 PREAMBLE[320];
static fo_c64 LTS_FREQ[64];
static fo_c64 LTS_TIME_CONJ[64];
static pthread_once_t tables_once = PTHREAD_ONCE_INIT;

/* radix-2 decimation-in-time, sign = -1 forward / +1 backward, unscaled */
static void dft64(cplx *x, int sign)
{
    for (int i = 0; i < 64; i++) {               /* 6-bit reversal */
        int r = 0;
        for (int b = 0; b < 6; b++) r |= ((i >> b) & 1) << (5 - b);
        if (r > i) { cplx t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    for (int len = 2; len <= 64; len <<= 1) {
        int step = 64 / len;
        for (int base = 0; base < 64; base += len)
            for (int k = 0; k < len / 2; k++) {
                cplx w = TW64[k * step];
                if (sign > 0) w = conj(w);
                cplx a = x[base + k], b = x[base + k + len / 2] * w;
                x[base + k] = a + b;
                x[base + k + len / 2] = a - b;
            }
    }
}

/* the 12-significant-digit decimal rounding the literal tables of preamble.h carry */
static double r12(double v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.12g", v);
    return strtod(buf, NULL);
}

static void tables_init(void)
{
    for (int k = 0; k < 64; k++) {
        /* exact octant symmetries keep the table accurate to the last bit or so */
        double a = -2.0 * M_PI * (double)k / 64.0;
        TW64[k] = CMPLX(cos(a), sin(a));
    }
    TW64[0] = CMPLX(1, 0); TW64[16] = CMPLX(0, -1); TW64[32] = CMPLX(-1, 0); TW64[48] = CMPLX(0, 1);

    /* IEEE 802.11a-1999 17.3.3 short and long training sequences (frequency domain) */
    static const int sts_pos[12] = { -24, -20, -16, -12, -8, -4, 4, 8, 12, 16, 20, 24 };
    static const int sts_sgn[12] = {   1,  -1,   1,  -1, -1,  1, -1, -1,  1,  1,  1,  1 };
    static const signed char lts_seq[53] = {
        1, 1,-1,-1, 1, 1,-1, 1,-1, 1, 1, 1, 1, 1, 1,-1,-1, 1, 1,-1, 1,-1, 1, 1, 1, 1, 0,
        1,-1,-1, 1, 1,-1, 1,-1, 1,-1,-1,-1,-1,-1, 1, 1,-1,-1, 1,-1, 1,-1, 1, 1, 1, 1 };
    cplx s[64], l[64];
    memset(s, 0, sizeof s); memset(l, 0, sizeof l);
    double amp = sqrt(13.0 / 6.0);
    for (int i = 0; i < 12; i++) s[(sts_pos[i] + 64) % 64] = CMPLX(sts_sgn[i] * amp, sts_sgn[i] * amp);
    for (int i = 0; i < 53; i++) l[(i - 26 + 64) % 64] = lts_seq[i];
    /* preamble.h:363: index = subcarrier + 32 */
    for (int i = 0; i < 64; i++) { LTS_FREQ[i].re = creal(l[(i + 32) % 64]); LTS_FREQ[i].im = 0.0; }
    dft64(s, +1); dft64(l, +1);
    for (int i = 0; i < 64; i++) { s[i] /= 64.0; l[i] /= 64.0; }
    /* preamble.h:432: conj(LTS) in time domain, 12 significant digits */
    for (int i = 0; i < 64; i++) { LTS_TIME_CONJ[i].re = r12(creal(l[i])); LTS_TIME_CONJ[i].im = r12(-cimag(l[i])); }
    /* preamble.h:24: 10 x STS(16) | LTS[32..63] | LTS | LTS, 12 significant digits ... */
    for (int i = 0; i < 160; i++) { PREAMBLE[i].re = r12(creal(s[i % 16])); PREAMBLE[i].im = r12(cimag(s[i % 16])); }
    for (int i = 0; i < 160; i++) {
        cplx v = l[(i + 32) % 64];
        PREAMBLE[160 + i].re = r12(creal(v)); PREAMBLE[160 + i].im = r12(cimag(v));
    }
    /* ... except the two window-halved entries: [0] = STS[0]/2 (preamble.h:26) and the literal
     * -0.078 at [160] (preamble.h:197) */
    PREAMBLE[0].re = r12(creal(s[0]) / 2.0); PREAMBLE[0].im = r12(cimag(s[0]) / 2.0);
    PREAMBLE[160].re = -0.078; PREAMBLE[160].im = 0.0;
}

static void ensure_tables(void) { pthread_once(&tables_once, tables_init); }

const fo_c64 *fo_preamble_samples(void) { ensure_tables(); return PREAMBLE; }
const fo_c64 *fo_lts_freq_domain(void) { ensure_tables(); return LTS_FREQ; }
const fo_c64 *fo_lts_time_domain_conj(void) { ensure_tables(); return LTS_TIME_CONJ; }

/* phase_tracker.cpp:23-32: the 127-periodic pilot polarity = 1 - 2*x, x the all-ones-seeded
 * x^7+x^4+1 scrambler sequence (802.11a 17.3.5.9) */
static double POLARITY[127];
static pthread_once_t pol_once = PTHREAD_ONCE_INIT;
static void pol_init(void)
{
    int st = 0x7F;
    for (int i = 0; i < 127; i++) {
        int fb = ((st >> 6) ^ (st >> 3)) & 1;
        st = ((st << 1) & 0x7E) | fb;
        POLARITY[i] = fb ? -1.0 : 1.0;
    }
}
const double *fo_polarity(void) { pthread_once(&pol_once, pol_init); return POLARITY; }

/* phase_tracker.cpp:37-50 */
static const int PILOT_IDX[4] = { 11, 25, 39, 53 };
static const int PILOT_SGN[4] = { 1, 1, 1, -1 };
static int DATA_IDX[48];
static pthread_once_t didx_once = PTHREAD_ONCE_INIT;
static void didx_init(void)
{
    int n = 0;
    for (int i = 6; i <= 58; i++)
        if (i != 11 && i != 25 && i != 32 && i != 39 && i != 53) DATA_IDX[n++] = i;
}
const int *fo_data_subcarriers(void) { pthread_once(&didx_once, didx_init); return DATA_IDX; }
const int *fo_pilot_subcarriers(void) { return PILOT_IDX; }

void fo_fft64(fo_c64 *data)
{
    ensure_tables();
    cplx x[64];
    for (int i = 0; i < 64; i++) x[i] = to_c(data[i]);
    dft64(x, -1);
    for (int s = 0; s < 64; s++) data[s] = from_c(x[(s + 32) % 64]);   /* fft.cpp:20-24,54-58 */
}

void fo_ifft64(fo_c64 *data)
{
    ensure_tables();
    cplx x[64];
    for (int s = 0; s < 64; s++) x[s] = to_c(data[(s + 32) % 64]);     /* fft.cpp:77-80 */
    dft64(x, +1);
    for (int i = 0; i < 64; i++) { data[i].re = creal(x[i]) / 64.0; data[i].im = cimag(x[i]) / 64.0; }
}

/* ------------------------------------------------------------------------------------------ */
/* bit-level codec                                                                             */
/* ------------------------------------------------------------------------------------------ */
int fo_parity(unsigned int x)
{
    x ^= x >> 16; x ^= x >> 8; x ^= x >> 4; x ^= x >> 2; x ^= x >> 1;
    return (int)(x & 1u);
}

/* IEEE 802.3 CRC-32 (reflected 0xEDB88320, init and final xor 0xFFFFFFFF) */
uint32_t fo_crc32(const uint8_t *data, size_t n)
{
    static uint32_t table[256];
    static int ready = 0;
    if (!ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
            table[i] = c;
        }
        __atomic_store_n(&ready, 1, __ATOMIC_RELEASE);
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = table[(c ^ data[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* ppdu.cpp:141-147: one LFSR step per BYTE, feedback XORed into the byte (so only bit 0 changes) */
void fo_scramble(const uint8_t *in, uint8_t *out, size_t n)
{
    int state = 93;
    for (size_t x = 0; x < n; x++) {
        int fb = ((state >> 6) & 1) ^ ((state >> 3) & 1);
        out[x] = (uint8_t)(fb ^ in[x]);
        state = ((state << 1) & 0x7E) | fb;
    }
}

/* viterbi.cpp:39-62: masks {121,91} on sr = (sr<<1)|bit, bits taken MSB first */
void fo_conv_encode(const uint8_t *data, uint8_t *symbols, int data_bits)
{
    /* (the reference shifts a signed int for the whole block, viterbi.cpp:53 -- overflow after 31 bits; only the low seven
     * bits are ever used, so an unsigned register gives the same symbols without the undefined behaviour: UBSan run,
     * tools/run_sanitizers.sh) */
    unsigned sr = 0;
    int idx = 0;
    for (int i = 0; i < data_bits + 6; i++) {
        unsigned bit = (data[i / 8] >> (7 - (i % 8))) & 1u;
        sr = (sr << 1) | bit;
        symbols[idx++] = (uint8_t)fo_parity(sr & 121u);
        symbols[idx++] = (uint8_t)fo_parity(sr & 91u);
    }
}

/* viterbi.cpp:208-457 as a scalar model.  Butterfly i combines old states i and i+32 into new
 * states 2i and 2i+1; metrics are uint8 with saturating adds; decision bit = (upper <= lower);
 * renormalise (subtract the minimum) only when the NEW metric of state 0 exceeds 210. */
void fo_viterbi_forward(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics,
                        uint64_t *stats)
{
    uint8_t B0[32], B1[32], a[64], b[64];
    uint8_t *old = a, *new_ = b;
    uint64_t clips = 0, renorms = 0;
    for (int i = 0; i < 32; i++) {                 /* viterbi.cpp:86-91 */
        B0[i] = fo_parity((unsigned)((2 * i) & 121)) ? 255 : 0;
        B1[i] = fo_parity((unsigned)((2 * i) & 91)) ? 255 : 0;
    }
    for (int i = 0; i < 64; i++) old[i] = 63;      /* viterbi.cpp:71-78 */
    old[0] = 0;
    int run = 2 * (nsteps / 2);                     /* viterbi.cpp:209: an odd last step is dropped */
    for (int t = 0; t < nsteps; t++) decisions[t] = 0;   /* viterbi.cpp:193-194 */
    for (int t = 0; t < run; t++) {
        unsigned s0 = symbols[2 * t], s1 = symbols[2 * t + 1];
        uint64_t d = 0;
        for (int i = 0; i < 32; i++) {
            unsigned m = ((((s0 ^ B0[i]) + (s1 ^ B1[i]) + 1u) >> 1) >> 2) & 63u;   /* avg, >>2, &63 */
            unsigned mc = 63u - m;
            unsigned a0 = old[i] + m, a1 = old[i + 32] + mc, b0 = old[i] + mc, b1 = old[i + 32] + m;
            if (a0 > 255) { a0 = 255; clips++; }
            if (a1 > 255) { a1 = 255; clips++; }
            if (b0 > 255) { b0 = 255; clips++; }
            if (b1 > 255) { b1 = 255; clips++; }
            if (a1 <= a0) { d |= 1ull << (2 * i); new_[2 * i] = (uint8_t)a1; } else new_[2 * i] = (uint8_t)a0;
            if (b1 <= b0) { d |= 1ull << (2 * i + 1); new_[2 * i + 1] = (uint8_t)b1; } else new_[2 * i + 1] = (uint8_t)b0;
        }
        decisions[t] = d;
        if (new_[0] > 210) {                        /* viterbi.cpp:314-332,438-456 */
            uint8_t mn = 255;
            for (int i = 0; i < 64; i++) if (new_[i] < mn) mn = new_[i];
            for (int i = 0; i < 64; i++) new_[i] = (uint8_t)(new_[i] - mn);
            renorms++;
        }
        uint8_t *tmp = old; old = new_; new_ = tmp;
    }
    if (metrics) memcpy(metrics, old, 64);
    if (stats) { stats[0] = clips; stats[1] = renorms; }
}

/* viterbi.cpp:108-146 with K=7: ADDSHIFT = 2, endstate 0 */
void fo_viterbi_chainback(const uint64_t *decisions, uint8_t *data, int data_bits)
{
    unsigned e = 0;
    for (int n = data_bits - 1; n >= 0; n--) {
        unsigned k = (unsigned)((decisions[n + 6] >> (e >> 2)) & 1u);
        e = (e >> 1) | (k << 7);
        data[n >> 3] = (uint8_t)e;
    }
}

/* 0 (default, the CHECKER): fo_conv_decode runs the scalar model.  1: the SSE forward pass (identical decision words) -- set only
 * around TIMED CPU legs of bench.py (the reference-shaped chain), never while anything is being checked. */
static int g_conv_simd = 0;
void fo_set_timed_simd_viterbi(int on) { __atomic_store_n(&g_conv_simd, on ? 1 : 0, __ATOMIC_RELEASE); }
void fo_viterbi_forward_simd(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics);

void fo_conv_decode(const uint8_t *symbols, uint8_t *data, int data_bits)
{
    int nsteps = data_bits + 6;
    uint64_t *dec = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(nsteps > 0 ? nsteps : 1));
    if (__atomic_load_n(&g_conv_simd, __ATOMIC_ACQUIRE)) fo_viterbi_forward_simd(symbols, nsteps, dec, NULL);
    else fo_viterbi_forward(symbols, nsteps, dec, NULL, NULL);
    fo_viterbi_chainback(dec, data, data_bits);
    free(dec);
}

size_t fo_puncture(const uint8_t *in, size_t n, int rate, uint8_t *out)
{
    size_t k = 0;
    switch (RATE_TABLE[rate].punct) {
    case 0: memcpy(out, in, n); return n;
    case 2: for (size_t x = 0; x < n; x += 6) { out[k++] = in[x]; out[k++] = in[x + 1]; out[k++] = in[x + 3]; out[k++] = in[x + 5]; } return k;
    default: for (size_t x = 0; x < n; x += 4) { out[k++] = in[x]; out[k++] = in[x + 2]; out[k++] = in[x + 3]; } return k;
    }
}

size_t fo_depuncture(const uint8_t *in, size_t n, int rate, uint8_t *out)
{
    size_t k = 0;
    switch (RATE_TABLE[rate].punct) {
    case 0: memcpy(out, in, n); return n;
    case 2:
        for (size_t x = 0; x < n; x += 4) {
            out[k++] = in[x]; out[k++] = in[x + 1]; out[k++] = 127; out[k++] = in[x + 2]; out[k++] = 127; out[k++] = in[x + 3];
        }
        return k;
    default:
        for (size_t x = 0; x < n; x += 3) { out[k++] = in[x]; out[k++] = 127; out[k++] = in[x + 1]; out[k++] = in[x + 2]; }
        return k;
    }
}

/* interleaver.h:66-75 with (ncarriers=48, nbits=1): s = 1, so j = i = 3*(k%16) + k/16 */
static inline unsigned ileave_index(unsigned k) { return 3u * (k % 16u) + k / 16u; }

void fo_interleave(const uint8_t *in, size_t n, uint8_t *out)
{
    for (size_t x = 0; x < n; x += 48)
        for (unsigned y = 0; y < 48; y++) out[x + ileave_index(y)] = in[x + y];
}

void fo_deinterleave(const uint8_t *in, size_t n, uint8_t *out)
{
    /* map[index(i)] = i; out[s + map[t]] = in[s + t]  <=>  out[s + i] = in[s + index(i)] */
    for (size_t s = 0; s < n; s += 48)
        for (unsigned i = 0; i < 48; i++) out[s + i] = in[s + ileave_index(i)];
}

/* qam.h:83-97 */
static double qam_encode(const uint8_t *bits, int nb, double scale_e)
{
    int pt = 0, flip = 1;
    for (int i = 0; i < nb; i++) {
        int bit = (int)(signed char)bits[i] * 2 - 1;
        pt = bit * flip + pt * 2;
        flip *= -bit;
    }
    return pt * scale_e;
}

/* x86 cvttsd2si semantics of `int pt = sym * d_scale_d` (qam.h:112) */
static inline int32_t trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT32_MIN;
    return (int32_t)v;
}

/* qam.h:110-125.  amp starts at 128 for every NumBits ((1<<(nb-1)) << (8-nb)). */
static void qam_decode(double sym, int nb, double scale_d, uint8_t *bits)
{
    uint32_t pt = (uint32_t)trunc_to_int(sym * scale_d);    /* two's-complement wrap like the compiled code */
    int32_t flip = 1, amp = 128;
    for (int i = 0; i < nb; i++) {
        int32_t v = (int32_t)((uint32_t)flip * pt + 128u);
        bits[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        int32_t bit = ((int32_t)pt < 0) ? -1 : 1;
        pt -= (uint32_t)(bit * amp);
        flip = -bit;
        amp /= 2;
    }
}

size_t fo_modulate(const uint8_t *bits, size_t n, int rate, fo_c64 *out)
{
    int nb; double se, sd;
    qam_params(rate, &nb, &se, &sd);
    int bpsc = RATE_TABLE[rate].bpsc;
    size_t count = n / (size_t)bpsc;
    for (size_t x = 0; x < count; x++) {
        if (bpsc == 1) { out[x].re = qam_encode(bits + x, 1, se); out[x].im = 0.0; }
        else { out[x].re = qam_encode(bits + x * bpsc, nb, se); out[x].im = qam_encode(bits + x * bpsc + nb, nb, se); }
    }
    return count;
}

size_t fo_demodulate(const fo_c64 *in, size_t n, int rate, uint8_t *out)
{
    int nb; double se, sd;
    qam_params(rate, &nb, &se, &sd);
    int bpsc = RATE_TABLE[rate].bpsc;
    for (size_t s = 0; s < n; s++) {
        qam_decode(in[s].re, nb, sd, out + s * bpsc);
        if (bpsc > 1) qam_decode(in[s].im, nb, sd, out + s * bpsc + nb);
    }
    return n * (size_t)bpsc;
}

/* ------------------------------------------------------------------------------------------ */
/* PPDU                                                                                        */
/* ------------------------------------------------------------------------------------------ */
void fo_encode_header(int rate, int length, fo_c64 *out48)
{
    unsigned field = ((unsigned)(RATE_TABLE[rate].rate_field & 0xF) << 13) | ((unsigned)length & 0xFFFu);
    if (fo_parity(field) == 1) field |= 131072u;
    field <<= 6;
    uint8_t bytes[4] = { (uint8_t)(field >> 16), (uint8_t)(field >> 8), (uint8_t)field, 0 };
    uint8_t sym[48], il[48];
    fo_conv_encode(bytes, sym, 18);
    fo_interleave(sym, 48, il);
    fo_modulate(il, 48, 0, out48);
}

int fo_decode_header(const fo_c64 *in48, int *rate, int *length, int *num_symbols)
{
    uint8_t dem[48], dei[48], hb[4] = { 0, 0, 0, 0 };
    fo_demodulate(in48, 48, 0, dem);
    fo_deinterleave(dem, 48, dei);
    fo_conv_decode(dei, hb, 18);
    unsigned field = ((unsigned)hb[0] << 16) | ((unsigned)hb[1] << 8) | hb[2];
    if (fo_parity(field) == 1) return 0;
    int rf = (int)((field >> 19) & 0xFu);
    int len = (int)((field >> 6) & 0xFFFu);
    int r = fo_rate_from_field(rf);
    if (r < 0) return 0;
    *rate = r; *length = len; *num_symbols = fo_num_symbols(r, len);
    return 1;
}

size_t fo_encode_data(const uint8_t *payload, int length, int rate, fo_c64 *out)
{
    const fo_rate_params *rp = &RATE_TABLE[rate];
    int nsym = fo_num_symbols(rate, length);
    int nbits = nsym * rp->dbps, nbytes = nbits / 8;
    uint8_t *data = (uint8_t *)calloc((size_t)nbytes + 8, 1);
    uint8_t *scr = (uint8_t *)calloc((size_t)nbytes + 8, 1);
    memcpy(data + 2, payload, (size_t)length);
    uint32_t crc = fo_crc32(data, (size_t)(2 + length));
    data[2 + length] = (uint8_t)crc; data[3 + length] = (uint8_t)(crc >> 8);
    data[4 + length] = (uint8_t)(crc >> 16); data[5 + length] = (uint8_t)(crc >> 24);
    fo_scramble(data, scr, (size_t)nbytes);              /* byte nbytes stays 0 */
    uint8_t *enc = (uint8_t *)calloc((size_t)nbits * 2 + 16, 1);
    fo_conv_encode(scr, enc, nbits - 6);
    uint8_t *pun = (uint8_t *)malloc((size_t)nbits * 2 + 16);
    size_t np = fo_puncture(enc, (size_t)nbits * 2, rate, pun);
    uint8_t *il = (uint8_t *)malloc(np + 48);
    fo_interleave(pun, np, il);
    size_t n = fo_modulate(il, np, rate, out);
    free(data); free(scr); free(enc); free(pun); free(il);
    return n;
}

/* Scratch of one worker of the timed CPU baseline (fo_pool, below): every buffer fo_decode_data and fo_decode_alignment_f32 would
 * otherwise malloc per frame, sized for the longest frame (1369 symbols of 24 bits at 6 Mbps .. 152 symbols of 216 at 54 Mbps: at most
 * 4095 + 6 bytes = 32 856 + padding bits, i.e. 33 048 trellis steps).  With a scratch the Viterbi forward pass is the SIMD one. */
#define FO_MAX_STEPS 33264
#define FO_MAX_SYMS 1372
struct fo_scratch {
    uint8_t dem[FO_MAX_SYMS * 288 + 64], dei[FO_MAX_SYMS * 288 + 64], dep[2 * FO_MAX_STEPS + 64];
    uint8_t dec[FO_MAX_STEPS / 8 + 16], des[FO_MAX_STEPS / 8 + 16];
    uint64_t decisions[FO_MAX_STEPS + 8];
    fo_c64 car[FO_MAX_SYMS * 48];
};

void fo_viterbi_forward_simd(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics);

static int decode_data_impl(const fo_c64 *in, int rate, int length, uint8_t *payload, uint8_t *soft, uint8_t *decoded_out, struct fo_scratch *sc)
{
    const fo_rate_params *rp = &RATE_TABLE[rate];
    int nsym = fo_num_symbols(rate, length);
    int nbits = nsym * rp->dbps, nbytes = nbits / 8;
    size_t ncar = (size_t)nsym * 48, ncoded = ncar * (size_t)rp->bpsc;
    uint8_t *dem = sc ? sc->dem : (uint8_t *)malloc(ncoded + 48), *dei = sc ? sc->dei : (uint8_t *)malloc(ncoded + 48);
    uint8_t *dep = sc ? sc->dep : (uint8_t *)malloc((size_t)nbits * 2 + 48);
    fo_demodulate(in, ncar, rate, dem);
    fo_deinterleave(dem, ncoded, dei);
    size_t nd = fo_depuncture(dei, ncoded, rate, dep);
    if (soft) memcpy(soft, dep, nd);
    uint8_t *dec = sc ? sc->dec : (uint8_t *)calloc((size_t)nbytes + 8, 1), *des = sc ? sc->des : (uint8_t *)calloc((size_t)nbytes + 8, 1);
    if (sc) {
        memset(dec, 0, (size_t)nbytes + 8); memset(des, 0, (size_t)nbytes + 8);
        fo_viterbi_forward_simd(dep, nbits, sc->decisions, NULL);          /* = fo_conv_decode with the SIMD forward pass */
        fo_viterbi_chainback(sc->decisions, dec, nbits - 6);
    } else {
        fo_conv_decode(dep, dec, nbits - 6);
    }
    fo_scramble(dec, des, (size_t)nbytes);
    if (decoded_out) memcpy(decoded_out, des, (size_t)nbytes);
    uint32_t crc = fo_crc32(des, (size_t)(2 + length));
    uint32_t given = (uint32_t)des[2 + length] | ((uint32_t)des[3 + length] << 8) |
                     ((uint32_t)des[4 + length] << 16) | ((uint32_t)des[5 + length] << 24);
    int ok = given == crc;
    if (ok && payload) memcpy(payload, des + 2, (size_t)length);
    if (!sc) { free(dem); free(dei); free(dep); free(dec); free(des); }
    return ok;
}

int fo_decode_data(const fo_c64 *in, int rate, int length, uint8_t *payload, uint8_t *soft, uint8_t *decoded_out)
{
    return decode_data_impl(in, rate, length, payload, soft, decoded_out, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* TX                                                                                          */
/* ------------------------------------------------------------------------------------------ */
size_t fo_symbol_map(const fo_c64 *in, size_t n48, fo_c64 *out)
{
    const double *pol = fo_polarity();
    const int *didx = fo_data_subcarriers();
    size_t nsym = n48 / 48;
    for (size_t y = 0; y < nsym; y++) {
        fo_c64 *o = out + y * 64;
        for (int i = 0; i < 64; i++) { o[i].re = 0.0; o[i].im = 0.0; }
        for (int i = 0; i < 48; i++) o[didx[i]] = in[y * 48 + i];
        for (int p = 0; p < 4; p++) { o[PILOT_IDX[p]].re = PILOT_SGN[p] * pol[y % 127]; o[PILOT_IDX[p]].im = 0.0 * pol[y % 127]; }
    }
    return nsym * 64;
}

size_t fo_frame_samples(int rate, int length) { return 320 + 80 * (size_t)(fo_num_symbols(rate, length) + 1); }

size_t fo_build_frame(const uint8_t *payload, int length, int rate, fo_c64 *out)
{
    ensure_tables();
    int nsym = fo_num_symbols(rate, length);
    size_t ncar = (size_t)(nsym + 1) * 48;
    fo_c64 *car = (fo_c64 *)malloc(sizeof(fo_c64) * ncar);
    fo_c64 *bins = (fo_c64 *)malloc(sizeof(fo_c64) * (size_t)(nsym + 1) * 64);
    fo_encode_header(rate, length, car);
    fo_encode_data(payload, length, rate, car + 48);
    fo_symbol_map(car, ncar, bins);
    memcpy(out, PREAMBLE, sizeof PREAMBLE);
    for (int y = 0; y <= nsym; y++) {
        fo_c64 *b = bins + (size_t)y * 64, *o = out + 320 + (size_t)y * 80;
        fo_ifft64(b);
        memcpy(o, b + 48, 16 * sizeof(fo_c64));
        memcpy(o + 16, b, 64 * sizeof(fo_c64));
    }
    free(car); free(bins);
    return 320 + 80 * (size_t)(nsym + 1);
}

/* ------------------------------------------------------------------------------------------ */
/* frame_detector (frame_detector.cpp:41-93, circular_accumulator.h:88-95)                     */
/* ------------------------------------------------------------------------------------------ */
struct fo_frame_detector {
    cplx corr_ring[16]; cplx corr_sum; int corr_idx;
    double pow_ring[16]; double pow_sum; int pow_idx;
    int plateau_length, plateau_flag;
    cplx carry[16];
};

fo_frame_detector *fo_frame_detector_new(void) { return (fo_frame_detector *)calloc(1, sizeof(fo_frame_detector)); }
void fo_frame_detector_free(fo_frame_detector *d) { free(d); }

void fo_frame_detector_work(fo_frame_detector *d, const fo_c64 *in, size_t n, fo_tagged_sample *out)
{
    if (n == 0) return;
    for (size_t x = 0; x < n; x++) {
        out[x].tag = FO_NONE; out[x]._pad = 0;
        cplx cur = to_c(in[x]);
        cplx delayed = x < 16 ? d->carry[x] : to_c(in[x - 16]);
        cplx c = cur * conj(delayed);
        if (creal(c) != creal(c) || cimag(c) != cimag(c)) c = 0;
        d->corr_sum -= d->corr_ring[d->corr_idx];
        d->corr_sum += c;
        d->corr_ring[d->corr_idx++] = c;
        if (d->corr_idx >= 16) d->corr_idx = 0;
        double p = in[x].re * in[x].re + in[x].im * in[x].im;
        if (p != p) p = 0;
        d->pow_sum -= d->pow_ring[d->pow_idx];
        d->pow_sum += p;
        d->pow_ring[d->pow_idx++] = p;
        if (d->pow_idx >= 16) d->pow_idx = 0;
        double corr = cabs(d->corr_sum) / d->pow_sum;
        if (corr > 0.9) {
            d->plateau_length++;
            if (d->plateau_length == 16) { out[x].tag = FO_STS_START; d->plateau_flag = 1; }
        } else {
            if (d->plateau_flag) { out[x].tag = FO_STS_END; d->plateau_flag = 0; }
            d->plateau_length = 0;
        }
        out[x].sample = in[x];
    }
    for (int i = 0; i < 16; i++) d->carry[i] = to_c(in[n - 16 + i]);
}

/* ------------------------------------------------------------------------------------------ */
/* timing_sync (timing_sync.cpp:51-139)                                                        */
/* ------------------------------------------------------------------------------------------ */
struct fo_timing_sync {
    double phase_offset, phase_acc;
    fo_tagged_sample carry[160];
    fo_tagged_sample *work; size_t work_cap;
    /* bookkeeping for fo_find_alignments: events seen in the last work() call */
    int64_t ev_pos[64]; double ev_phase[64]; int64_t ev_x[64]; double ev_prev[64]; int n_ev;
};

fo_timing_sync *fo_timing_sync_new(void)
{
    fo_timing_sync *t = (fo_timing_sync *)calloc(1, sizeof(fo_timing_sync));
    return t;   /* carry-over: 160 zero samples tagged NONE (timing_sync.cpp:25-29) */
}
void fo_timing_sync_free(fo_timing_sync *t) { if (t) { free(t->work); free(t); } }
double fo_timing_sync_phase_acc(const fo_timing_sync *t) { return t->phase_acc; }

typedef struct { double v; int p; } peak_t;
static int peak_desc(const void *a, const void *b)
{
    const peak_t *x = (const peak_t *)a, *y = (const peak_t *)b;   /* sort ascending then reverse */
    if (x->v != y->v) return x->v < y->v ? 1 : -1;
    return x->p < y->p ? 1 : (x->p > y->p ? -1 : 0);
}

void fo_timing_sync_work(fo_timing_sync *t, const fo_tagged_sample *in, size_t n, fo_tagged_sample *out)
{
    if (n == 0) return;
    ensure_tables();
    if (t->work_cap < n + 160) { free(t->work); t->work_cap = n + 160; t->work = (fo_tagged_sample *)malloc(sizeof(fo_tagged_sample) * t->work_cap); }
    fo_tagged_sample *w = t->work;
    memcpy(w, t->carry, sizeof t->carry);
    memcpy(w + 160, in, sizeof(fo_tagged_sample) * n);
    t->n_ev = 0;
    for (size_t x = 0; x < n; x++) {
        if (w[x].tag == FO_STS_END) {
            peak_t peaks[96]; int np = 0;
            for (size_t p = x; p < x + 160 - 64; p++) {
                cplx corr = 0; double power = 0;
                for (int s = 0; s < 64; s++) {
                    corr += to_c(w[p + s].sample) * to_c(LTS_TIME_CONJ[s]);
                    power += w[p + s].sample.re * w[p + s].sample.re + w[p + s].sample.im * w[p + s].sample.im;
                }
                double cn = cabs(corr) / power;
                if (cn > 0.9) { peaks[np].v = cn; peaks[np].p = (int)p; np++; }
            }
            qsort(peaks, (size_t)np, sizeof(peak_t), peak_desc);
            /* only s = 0 is ever examined (s += 5 while s < min(np,3)); t runs over the top 5 */
            if (np > 0) {
                int lim = np < 5 ? np : 5;
                for (int k = 0; k < lim; k++) {
                    if (abs(peaks[0].p - peaks[k].p) == 64) {
                        int lts_offset = (peaks[0].p < peaks[k].p ? peaks[0].p : peaks[k].p) - 32;
                        if (lts_offset < 0) break;
                        w[lts_offset + 24].tag = FO_LTS1;
                        w[lts_offset + 24 + 64].tag = FO_LTS2;
                        double prev = t->phase_acc;
                        t->phase_offset = atan2(0.0, 0.0) / 64.0;    /* the loop at :109 never runs */
                        cplx v = to_c(w[lts_offset + 32 + 128 - 1].sample) * to_c(LTS_TIME_CONJ[63]);
                        t->phase_acc = atan2(cimag(v), creal(v));
                        if (t->n_ev < 64) {
                            t->ev_pos[t->n_ev] = (int64_t)lts_offset + 24; t->ev_x[t->n_ev] = (int64_t)x;
                            t->ev_phase[t->n_ev] = t->phase_acc; t->ev_prev[t->n_ev] = prev; t->n_ev++;
                        }
                        break;
                    }
                }
            }
        }
        t->phase_acc += t->phase_offset;
        while (t->phase_acc > 2.0 * M_PI) t->phase_acc -= 2.0 * M_PI;
        while (t->phase_acc < -2.0 * M_PI) t->phase_acc += 2.0 * M_PI;
        cplx pc = CMPLX(cos(t->phase_acc), sin(t->phase_acc));
        w[x].sample = from_c(to_c(w[x].sample) * pc);
    }
    memcpy(out, w, sizeof(fo_tagged_sample) * n);
    memcpy(t->carry, w + n, sizeof t->carry);
}

/* ------------------------------------------------------------------------------------------ */
/* fft_symbols (fft_symbols.cpp:33-79)                                                         */
/* ------------------------------------------------------------------------------------------ */
struct fo_fft_symbols { fo_tagged_vec64 cur; int offset; };

fo_fft_symbols *fo_fft_symbols_new(void) { return (fo_fft_symbols *)calloc(1, sizeof(fo_fft_symbols)); }
void fo_fft_symbols_free(fo_fft_symbols *f) { free(f); }

size_t fo_fft_symbols_work(fo_fft_symbols *f, const fo_tagged_sample *in, size_t n, fo_tagged_vec64 *out)
{
    size_t k = 0;
    for (size_t x = 0; x < n; x++) {
        if (in[x].tag == FO_LTS1) {
            if (f->offset > 15) out[k++] = f->cur;
            f->cur.tag = FO_LTS_START;
            f->offset = 16;
        }
        if (in[x].tag == FO_LTS2) f->offset = 16;
        if (f->offset > 15) f->cur.samples[f->offset - 16] = in[x].sample;
        f->offset++;
        if (f->offset == 80) { out[k++] = f->cur; f->cur.tag = FO_NONE; f->offset = 0; }
    }
    for (size_t i = 0; i < k; i++) fo_fft64(out[i].samples);
    return k;
}

/* ------------------------------------------------------------------------------------------ */
/* channel_est (channel_est.cpp:36-85)                                                         */
/* ------------------------------------------------------------------------------------------ */
struct fo_channel_est { fo_c64 est[64]; int lts_flag, frame_start; };

fo_channel_est *fo_channel_est_new(void)
{
    fo_channel_est *c = (fo_channel_est *)calloc(1, sizeof(fo_channel_est));
    for (int j = 0; j < 64; j++) c->est[j].re = 1.0;
    return c;
}
void fo_channel_est_free(fo_channel_est *c) { free(c); }
const fo_c64 *fo_channel_est_state(const fo_channel_est *c) { return c->est; }

size_t fo_channel_est_work(fo_channel_est *c, const fo_tagged_vec64 *in, size_t n, fo_tagged_vec64 *out)
{
    ensure_tables();
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        if (in[i].tag == FO_LTS_START) {
            c->lts_flag = 1;
            for (int j = 0; j < 64; j++) { c->est[j].re = 0.0; c->est[j].im = 0.0; }
        }
        if (c->lts_flag > 0) {
            for (int j = 0; j < 64; j++) {
                cplx ref = to_c(LTS_FREQ[j]), rec = to_c(in[i].samples[j]);
                cplx q = ref / rec;
                cplx h = CMPLX(creal(q) / 2.0, cimag(q) / 2.0);
                c->est[j] = from_c(to_c(c->est[j]) + h);
            }
            c->lts_flag++;
            if (c->lts_flag == 3) { c->lts_flag = 0; c->frame_start = 1; }
        } else {
            fo_tagged_vec64 *o = &out[k++];
            o->tag = FO_NONE; o->_pad = 0;
            if (c->frame_start) { o->tag = FO_START_OF_FRAME; c->frame_start = 0; }
            for (int j = 0; j < 64; j++) o->samples[j] = from_c(to_c(c->est[j]) * to_c(in[i].samples[j]));
        }
    }
    return k;
}

/* ------------------------------------------------------------------------------------------ */
/* phase_tracker (phase_tracker.cpp:70-104)                                                    */
/* ------------------------------------------------------------------------------------------ */
struct fo_phase_tracker { int symbol_count; };

fo_phase_tracker *fo_phase_tracker_new(void) { return (fo_phase_tracker *)calloc(1, sizeof(fo_phase_tracker)); }
void fo_phase_tracker_free(fo_phase_tracker *p) { free(p); }

void fo_phase_tracker_work(fo_phase_tracker *pt, const fo_tagged_vec64 *in, size_t n, fo_tagged_vec48 *out)
{
    const double *pol = fo_polarity();
    const int *didx = fo_data_subcarriers();
    for (size_t i = 0; i < n; i++) {
        if (in[i].tag == FO_START_OF_FRAME) pt->symbol_count = 0;
        cplx pe = 0;
        for (int p = 0; p < 4; p++) {
            int pilot = (int)(PILOT_SGN[p] * pol[pt->symbol_count % 127]);
            cplx ref = CMPLX((double)pilot, 0.0);
            cplx rec = to_c(in[i].samples[PILOT_IDX[p]]);
            cplx pr = rec * conj(ref);
            pe += CMPLX(creal(pr) / 4.0, cimag(pr) / 4.0);
        }
        double angle = atan2(cimag(pe), creal(pe));
        cplx rot = CMPLX(cos(-angle), sin(-angle));
        for (int s = 0; s < 48; s++) out[i].samples[s] = from_c(to_c(in[i].samples[didx[s]]) * rot);
        out[i].tag = in[i].tag; out[i]._pad = 0;
        pt->symbol_count++;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* payload lists                                                                               */
/* ------------------------------------------------------------------------------------------ */
struct fo_payloads { uint8_t *bytes; size_t nbytes, cap; size_t *off; size_t *len; size_t count, ccap; };

fo_payloads *fo_payloads_new(void) { return (fo_payloads *)calloc(1, sizeof(fo_payloads)); }
void fo_payloads_free(fo_payloads *p) { if (p) { free(p->bytes); free(p->off); free(p->len); free(p); } }
void fo_payloads_clear(fo_payloads *p) { p->nbytes = 0; p->count = 0; }
size_t fo_payloads_count(const fo_payloads *p) { return p->count; }
size_t fo_payloads_len(const fo_payloads *p, size_t i) { return p->len[i]; }
const uint8_t *fo_payloads_data(const fo_payloads *p, size_t i) { return p->bytes + p->off[i]; }

static void payloads_push(fo_payloads *p, const uint8_t *d, size_t n)
{
    if (p->nbytes + n + 1 > p->cap) { p->cap = (p->nbytes + n + 1) * 2; p->bytes = (uint8_t *)realloc(p->bytes, p->cap); }
    if (p->count + 1 > p->ccap) {
        p->ccap = (p->count + 1) * 2;
        p->off = (size_t *)realloc(p->off, p->ccap * sizeof(size_t));
        p->len = (size_t *)realloc(p->len, p->ccap * sizeof(size_t));
    }
    memcpy(p->bytes + p->nbytes, d, n);
    p->off[p->count] = p->nbytes; p->len[p->count] = n; p->count++; p->nbytes += n;
}

/* ------------------------------------------------------------------------------------------ */
/* frame_decoder (frame_decoder.cpp:45-91, frame_decoder.h:29-65)                              */
/* ------------------------------------------------------------------------------------------ */
struct fo_frame_decoder {
    int sample_count, samples_copied, rate, length;
    fo_c64 *samples; size_t cap;
    uint8_t payload[4096];
    uint64_t stats[4];
};

fo_frame_decoder *fo_frame_decoder_new(void)
{
    fo_frame_decoder *d = (fo_frame_decoder *)calloc(1, sizeof(fo_frame_decoder));
    d->cap = 100000;                                    /* frame_decoder.h:44 */
    d->samples = (fo_c64 *)calloc(d->cap, sizeof(fo_c64));
    return d;
}
void fo_frame_decoder_free(fo_frame_decoder *d) { if (d) { free(d->samples); free(d); } }
const uint64_t *fo_frame_decoder_stats(const fo_frame_decoder *d) { return d->stats; }

void fo_frame_decoder_work(fo_frame_decoder *d, const fo_tagged_vec48 *in, size_t n, fo_payloads *out)
{
    if (n == 0) return;
    fo_payloads_clear(out);
    for (size_t x = 0; x < n; x++) {
        if (d->samples_copied < d->sample_count) {
            memcpy(d->samples + d->samples_copied, in[x].samples, 48 * sizeof(fo_c64));
            d->samples_copied += 48;
        }
        if (d->samples_copied >= d->sample_count && d->sample_count != 0) {
            if (fo_decode_data(d->samples, d->rate, d->length, d->payload, NULL, NULL)) {
                payloads_push(out, d->payload, (size_t)d->length);
                d->stats[2]++;
            } else d->stats[3]++;
            d->sample_count = 0;
        }
        if (in[x].tag == FO_START_OF_FRAME) {
            int rate, length, nsym;
            if (!fo_decode_header(in[x].samples, &rate, &length, &nsym)) { d->stats[1]++; continue; }
            d->stats[0]++;
            d->rate = rate; d->length = length; d->sample_count = nsym * 48; d->samples_copied = 0;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* receiver_chain (receiver_chain.cpp:29-126)                                                  */
/* ------------------------------------------------------------------------------------------ */
struct fo_receiver_chain {
    fo_frame_detector *fd; fo_timing_sync *ts; fo_fft_symbols *fs; fo_channel_est *ce; fo_phase_tracker *pt; fo_frame_decoder *dec;
    /* input_buffer / output_buffer of each block; sizes n_* */
    fo_c64 *fd_in; size_t fd_in_n, fd_in_cap;
    fo_tagged_sample *fd_out, *ts_in, *ts_out, *fs_in; size_t fd_out_n, ts_in_n, ts_out_n, fs_in_n, a_cap[4];
    fo_tagged_vec64 *fs_out, *ce_in, *ce_out, *pt_in; size_t fs_out_n, ce_in_n, ce_out_n, pt_in_n, b_cap[4];
    fo_tagged_vec48 *pt_out, *dec_in; size_t pt_out_n, dec_in_n, c_cap[2];
    fo_payloads *dec_out;
    int threaded;
    pthread_t th[6]; sem_t wake[6], done[6]; int idx[6];
};

static void *grow(void *p, size_t *cap, size_t need, size_t elem)
{
    if (*cap >= need) return p;
    *cap = need * 2;
    return realloc(p, *cap * elem);
}

static void chain_run_block(fo_receiver_chain *c, int i)
{
    switch (i) {
    case 0:
        if (c->fd_in_n) { c->fd_out = grow(c->fd_out, &c->a_cap[0], c->fd_in_n, sizeof(fo_tagged_sample)); fo_frame_detector_work(c->fd, c->fd_in, c->fd_in_n, c->fd_out); c->fd_out_n = c->fd_in_n; }
        break;
    case 1:
        if (c->ts_in_n) { c->ts_out = grow(c->ts_out, &c->a_cap[2], c->ts_in_n, sizeof(fo_tagged_sample)); fo_timing_sync_work(c->ts, c->ts_in, c->ts_in_n, c->ts_out); c->ts_out_n = c->ts_in_n; }
        break;
    case 2:
        if (c->fs_in_n) { c->fs_out = grow(c->fs_out, &c->b_cap[0], c->fs_in_n / 64 + 4, sizeof(fo_tagged_vec64)); c->fs_out_n = fo_fft_symbols_work(c->fs, c->fs_in, c->fs_in_n, c->fs_out); }
        break;
    case 3:
        if (c->ce_in_n) { c->ce_out = grow(c->ce_out, &c->b_cap[2], c->ce_in_n, sizeof(fo_tagged_vec64)); c->ce_out_n = fo_channel_est_work(c->ce, c->ce_in, c->ce_in_n, c->ce_out); }
        break;
    case 4:
        if (c->pt_in_n) { c->pt_out = grow(c->pt_out, &c->c_cap[0], c->pt_in_n, sizeof(fo_tagged_vec48)); fo_phase_tracker_work(c->pt, c->pt_in, c->pt_in_n, c->pt_out); c->pt_out_n = c->pt_in_n; }
        break;
    case 5:
        if (c->dec_in_n) fo_frame_decoder_work(c->dec, c->dec_in, c->dec_in_n, c->dec_out);
        break;
    }
}

struct chain_thread_arg { fo_receiver_chain *c; int i; };

static void *chain_thread(void *arg)
{
    struct chain_thread_arg *a = (struct chain_thread_arg *)arg;
    fo_receiver_chain *c = a->c; int i = a->i;
    free(a);
    for (;;) {                                  /* receiver_chain.cpp:78-95 */
        sem_wait(&c->wake[i]);
        if (c->threaded < 0) break;
        chain_run_block(c, i);
        sem_post(&c->done[i]);
    }
    return NULL;
}

static fo_receiver_chain *chain_new(int threaded)
{
    fo_receiver_chain *c = (fo_receiver_chain *)calloc(1, sizeof(fo_receiver_chain));
    c->fd = fo_frame_detector_new(); c->ts = fo_timing_sync_new(); c->fs = fo_fft_symbols_new();
    c->ce = fo_channel_est_new(); c->pt = fo_phase_tracker_new(); c->dec = fo_frame_decoder_new();
    c->dec_out = fo_payloads_new();
    c->threaded = threaded;
    if (threaded)
        for (int i = 0; i < 6; i++) {
            sem_init(&c->wake[i], 0, 0); sem_init(&c->done[i], 0, 0);
            struct chain_thread_arg *a = (struct chain_thread_arg *)malloc(sizeof *a);
            a->c = c; a->i = i;
            pthread_create(&c->th[i], NULL, chain_thread, a);
        }
    return c;
}

fo_receiver_chain *fo_receiver_chain_new(void) { return chain_new(0); }
fo_receiver_chain *fo_receiver_chain_new_threaded(void) { return chain_new(1); }

void fo_receiver_chain_free(fo_receiver_chain *c)
{
    if (!c) return;
    if (c->threaded) {
        c->threaded = -1;
        for (int i = 0; i < 6; i++) sem_post(&c->wake[i]);
        for (int i = 0; i < 6; i++) pthread_join(c->th[i], NULL);
    }
    fo_frame_detector_free(c->fd); fo_timing_sync_free(c->ts); fo_fft_symbols_free(c->fs);
    fo_channel_est_free(c->ce); fo_phase_tracker_free(c->pt); fo_frame_decoder_free(c->dec);
    fo_payloads_free(c->dec_out);
    free(c->fd_in); free(c->fd_out); free(c->ts_in); free(c->ts_out); free(c->fs_in);
    free(c->fs_out); free(c->ce_in); free(c->ce_out); free(c->pt_in); free(c->pt_out); free(c->dec_in);
    free(c);
}

const uint64_t *fo_receiver_chain_decoder_stats(const fo_receiver_chain *c) { return fo_frame_decoder_stats(c->dec); }

#define SWAP_BUF(T, a, an, acap, b, bn, bcap) do { T *tp = a; a = b; b = tp; size_t tn = an; an = bn; bn = tn; size_t tc = acap; acap = bcap; bcap = tc; } while (0)

const fo_payloads *fo_receiver_chain_process_samples(fo_receiver_chain *c, const fo_c64 *in, size_t n)
{
    /* receiver_chain.cpp:109: the argument becomes frame_detector's input_buffer */
    c->fd_in = grow(c->fd_in, &c->fd_in_cap, n ? n : 1, sizeof(fo_c64));
    memcpy(c->fd_in, in, n * sizeof(fo_c64));
    c->fd_in_n = n;
    if (c->threaded) {
        for (int i = 0; i < 6; i++) sem_post(&c->wake[i]);
        for (int i = 0; i < 6; i++) sem_wait(&c->done[i]);
    } else {
        for (int i = 0; i < 6; i++) chain_run_block(c, i);
    }
    /* receiver_chain.cpp:118-122: swap each output_buffer with the next block's input_buffer */
    SWAP_BUF(fo_tagged_sample, c->ts_in, c->ts_in_n, c->a_cap[1], c->fd_out, c->fd_out_n, c->a_cap[0]);
    SWAP_BUF(fo_tagged_sample, c->fs_in, c->fs_in_n, c->a_cap[3], c->ts_out, c->ts_out_n, c->a_cap[2]);
    SWAP_BUF(fo_tagged_vec64, c->ce_in, c->ce_in_n, c->b_cap[1], c->fs_out, c->fs_out_n, c->b_cap[0]);
    SWAP_BUF(fo_tagged_vec64, c->pt_in, c->pt_in_n, c->b_cap[3], c->ce_out, c->ce_out_n, c->b_cap[2]);
    SWAP_BUF(fo_tagged_vec48, c->dec_in, c->dec_in_n, c->c_cap[1], c->pt_out, c->pt_out_n, c->c_cap[0]);
    return c->dec_out;
}

/* ------------------------------------------------------------------------------------------ */
/* the hot path in isolation                                                                   */
/* ------------------------------------------------------------------------------------------ */
static void decode_alignment_impl(const float *iq, int64_t end, const fo_frame_desc *d, uint8_t *psdu,
                                  fo_frame_result *res, fo_c64 *hinv, fo_c64 *eq, uint8_t *soft, fo_c64 *fftout, struct fo_scratch *sc)
{
    ensure_tables();
    res->status = FO_ST_HEADER_FAIL; res->rate = -1; res->length = 0; res->num_symbols = 0;
    int64_t p = d->lts1_pos;
    cplx rot = CMPLX(d->c, d->s), rot_prev = CMPLX(d->c_prev, d->s_prev);
    fo_tagged_vec64 v;
    fo_channel_est ce_obj; fo_phase_tracker pt_obj;       /* (as fo_channel_est_new / fo_phase_tracker_new leave them) */
    memset(&ce_obj, 0, sizeof ce_obj); memset(&pt_obj, 0, sizeof pt_obj);
    for (int j = 0; j < 64; j++) ce_obj.est[j].re = 1.0;
    fo_channel_est *ce = &ce_obj;
    fo_phase_tracker *pt = &pt_obj;
    fo_tagged_vec64 eqv; fo_tagged_vec48 dv;
    fo_c64 *car = NULL;
    int rate = -1, length = 0, nsym = 0;
    /* windows: LTS1 [p,p+64), LTS2 [p+64,p+128), SIGNAL [p+144,..), data k at p+224+80k (fft_symbols.cpp:41-73) */
    for (int sym = -2; ; sym++) {
        int64_t start = sym == -2 ? p : (sym == -1 ? p + 64 : p + 144 + 80 * (int64_t)sym);
        if (start + 64 > end) { res->status = FO_ST_TRUNCATED; break; }
        for (int i = 0; i < 64; i++) {
            int64_t idx = start + i;
            cplx smp = CMPLX((double)iq[2 * idx], (double)iq[2 * idx + 1]);
            v.samples[i] = from_c(smp * (idx >= d->rot_start ? rot : rot_prev));     /* timing_sync.cpp:124-125 */
        }
        v.tag = sym == -2 ? FO_LTS_START : FO_NONE;
        fo_fft64(v.samples);
        if (fftout) memcpy(fftout + (size_t)(sym + 2) * 64, v.samples, sizeof v.samples);
        if (fo_channel_est_work(ce, &v, 1, &eqv) == 0) continue;
        if (sym == 0 && hinv) memcpy(hinv, ce->est, sizeof ce->est);
        fo_phase_tracker_work(pt, &eqv, 1, &dv);
        if (eq && (sym == 0 || rate >= 0)) memcpy(eq + (size_t)sym * 48, dv.samples, sizeof dv.samples);
        if (sym == 0) {
            if (!fo_decode_header(dv.samples, &rate, &length, &nsym)) { rate = -1; break; }
            res->rate = rate; res->length = length; res->num_symbols = nsym;
            car = sc ? sc->car : (fo_c64 *)malloc(sizeof(fo_c64) * 48 * (size_t)nsym);
        } else {
            memcpy(car + (size_t)(sym - 1) * 48, dv.samples, sizeof dv.samples);
            if (sym == nsym) {
                int ok = decode_data_impl(car, rate, length, psdu, soft, NULL, sc);
                res->status = ok ? FO_ST_OK : FO_ST_CRC_FAIL;
                break;
            }
        }
    }
    if (!sc) free(car);
}

void fo_decode_alignment_f32(const float *iq, int64_t end, const fo_frame_desc *d, uint8_t *psdu,
                             fo_frame_result *res, fo_c64 *hinv, fo_c64 *eq, uint8_t *soft, fo_c64 *fftout)
{
    decode_alignment_impl(iq, end, d, psdu, res, hinv, eq, soft, fftout, NULL);
}

size_t fo_find_alignments_f32(const float *iq, int64_t n, fo_frame_desc *out, size_t cap)
{
    const size_t chunk = 4096;                     /* receiver.h:16 */
    fo_frame_detector *fd = fo_frame_detector_new();
    fo_timing_sync *ts = fo_timing_sync_new();
    fo_c64 *buf = (fo_c64 *)malloc(sizeof(fo_c64) * chunk);
    fo_tagged_sample *a = (fo_tagged_sample *)malloc(sizeof(fo_tagged_sample) * chunk);
    fo_tagged_sample *b = (fo_tagged_sample *)malloc(sizeof(fo_tagged_sample) * chunk);
    size_t k = 0;
    /* feed the stream plus one flushing chunk of zeros; timing_sync's output lags its input by 160 */
    for (int64_t base = 0; base < n + (int64_t)chunk; base += (int64_t)chunk) {
        for (size_t i = 0; i < chunk; i++) {
            int64_t idx = base + (int64_t)i;
            buf[i].re = idx < n ? (double)iq[2 * idx] : 0.0;
            buf[i].im = idx < n ? (double)iq[2 * idx + 1] : 0.0;
        }
        fo_frame_detector_work(fd, buf, chunk, a);
        fo_timing_sync_work(ts, a, chunk, b);
        for (int e = 0; e < ts->n_ev && k < cap; e++) {
            /* work-buffer index w corresponds to stream index base - 160 + w */
            fo_frame_desc *d = &out[k];
            d->lts1_pos = base - 160 + ts->ev_pos[e];
            d->rot_start = base - 160 + ts->ev_x[e];
            d->c = cos(ts->ev_phase[e]); d->s = sin(ts->ev_phase[e]);
            d->c_prev = cos(ts->ev_prev[e]); d->s_prev = sin(ts->ev_prev[e]);
            if (d->lts1_pos >= 0 && d->lts1_pos < n) k++;
        }
    }
    free(buf); free(a); free(b);
    fo_frame_detector_free(fd); fo_timing_sync_free(ts);
    return k;
}

struct batch_job {
    const float *iq; const fo_frame_desc *descs; const int64_t *ends; size_t n_frames;
    uint8_t *psdu; size_t slot; fo_frame_result *res; size_t next; pthread_mutex_t mu;
};

static void *batch_worker(void *arg)
{
    struct batch_job *j = (struct batch_job *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        size_t lo = j->next, hi = lo + 8 < j->n_frames ? lo + 8 : j->n_frames;
        j->next = hi;
        pthread_mutex_unlock(&j->mu);
        if (lo >= hi) break;
        for (size_t f = lo; f < hi; f++)
            fo_decode_alignment_f32(j->iq, j->ends[f], &j->descs[f], j->psdu + f * j->slot, &j->res[f], NULL, NULL, NULL, NULL);
    }
    return NULL;
}

static void v2_fixup(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_frames, uint8_t *psdu, size_t slot_bytes,
                     fo_frame_result *res);

void fo_decode_batch_f32(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends,
                         size_t n_frames, uint8_t *psdu, size_t slot_bytes, fo_frame_result *res, int threads)
{
    struct batch_job j = { iq, descs, ends, n_frames, psdu, slot_bytes, res, 0, PTHREAD_MUTEX_INITIALIZER };
    if (threads <= 1) batch_worker(&j);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, batch_worker, &j);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
        free(th);
    }
    v2_fixup(iq, n, descs, ends, n_frames, psdu, slot_bytes, res);      /* alignments cut short by the next one: decided with what follows them */
}

/* ------------------------------------------------------------------------------------------ */
/* the timed CPU baseline (bench.py `cpu_baseline`): the same per-alignment path, made fast the */
/* way the reference's own CPU path is fast                                                    */
/* ------------------------------------------------------------------------------------------ */
/* viterbi.cpp:208-457 keeps its 64 uint8 path metrics in four SSE registers; fo_viterbi_forward above is the readable
 * scalar model of it and stays the CHECKER.  What is TIMED as "the CPU" must not be an order of magnitude slower than the
 * reference's decoder, so this is the same recursion on sixteen butterflies per instruction, written from the scalar model
 * (tests/test_oracle_golden.py asserts equal decision words and metrics on random, saturating and erased inputs; where the
 * reference's compiled decoder is present, tests/test_oracle_vs_ref.py compares against it too).
 * Layout: v[0..3] = metrics of states 0-15, 16-31, 32-47, 48-63.  Butterfly i (0..31) reads states i and i+32 and writes 2i
 * and 2i+1: for i = 0..15 that is (v0, v2) -> interleaved into new v0, v1; for i = 16..31 (v1, v3) -> new v2, v3. */
#if defined(__x86_64__) || defined(__i386__)
#include <smmintrin.h>

void fo_viterbi_forward_simd(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics)
{
    uint8_t b0[32], b1[32], init[64];
    for (int i = 0; i < 32; i++) {                 /* viterbi.cpp:86-91 */
        b0[i] = fo_parity((unsigned)((2 * i) & 121)) ? 255 : 0;
        b1[i] = fo_parity((unsigned)((2 * i) & 91)) ? 255 : 0;
    }
    for (int i = 0; i < 64; i++) init[i] = 63;     /* viterbi.cpp:71-78 */
    init[0] = 0;
    const __m128i B0a = _mm_loadu_si128((const __m128i *)b0), B0b = _mm_loadu_si128((const __m128i *)(b0 + 16));
    const __m128i B1a = _mm_loadu_si128((const __m128i *)b1), B1b = _mm_loadu_si128((const __m128i *)(b1 + 16));
    const __m128i k63 = _mm_set1_epi8(63);
    __m128i v0 = _mm_loadu_si128((const __m128i *)init), v1 = _mm_loadu_si128((const __m128i *)(init + 16)),
            v2 = _mm_loadu_si128((const __m128i *)(init + 32)), v3 = _mm_loadu_si128((const __m128i *)(init + 48));
    const int run = 2 * (nsteps / 2);               /* viterbi.cpp:209: an odd last step is dropped */
    for (int t = run; t < nsteps; t++) decisions[t] = 0;
    for (int t = 0; t < run; t++) {
        const __m128i s0 = _mm_set1_epi8((char)symbols[2 * t]), s1 = _mm_set1_epi8((char)symbols[2 * t + 1]);
        /* ((s0 ^ B0) + (s1 ^ B1) + 1) >> 1 is the rounding average; then >> 2, & 63 */
        const __m128i ma = _mm_and_si128(_mm_srli_epi16(_mm_avg_epu8(_mm_xor_si128(s0, B0a), _mm_xor_si128(s1, B1a)), 2), k63);
        const __m128i mb = _mm_and_si128(_mm_srli_epi16(_mm_avg_epu8(_mm_xor_si128(s0, B0b), _mm_xor_si128(s1, B1b)), 2), k63);
        const __m128i ca = _mm_sub_epi8(k63, ma), cb = _mm_sub_epi8(k63, mb);
        /* butterflies 0..15 */
        __m128i a0 = _mm_adds_epu8(v0, ma), a1 = _mm_adds_epu8(v2, ca), c0 = _mm_adds_epu8(v0, ca), c1 = _mm_adds_epu8(v2, ma);
        __m128i na = _mm_min_epu8(a0, a1), nc = _mm_min_epu8(c0, c1);
        __m128i da = _mm_cmpeq_epi8(na, a1), dc = _mm_cmpeq_epi8(nc, c1);          /* upper <= lower */
        const __m128i n0 = _mm_unpacklo_epi8(na, nc), n1 = _mm_unpackhi_epi8(na, nc);
        uint64_t d = (uint64_t)(unsigned)_mm_movemask_epi8(_mm_unpacklo_epi8(da, dc)) | (uint64_t)(unsigned)_mm_movemask_epi8(_mm_unpackhi_epi8(da, dc)) << 16;
        /* butterflies 16..31 */
        a0 = _mm_adds_epu8(v1, mb); a1 = _mm_adds_epu8(v3, cb); c0 = _mm_adds_epu8(v1, cb); c1 = _mm_adds_epu8(v3, mb);
        na = _mm_min_epu8(a0, a1); nc = _mm_min_epu8(c0, c1);
        da = _mm_cmpeq_epi8(na, a1); dc = _mm_cmpeq_epi8(nc, c1);
        const __m128i n2 = _mm_unpacklo_epi8(na, nc), n3 = _mm_unpackhi_epi8(na, nc);
        d |= (uint64_t)(unsigned)_mm_movemask_epi8(_mm_unpacklo_epi8(da, dc)) << 32 | (uint64_t)(unsigned)_mm_movemask_epi8(_mm_unpackhi_epi8(da, dc)) << 48;
        decisions[t] = d;
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        if ((unsigned)(_mm_extract_epi8(v0, 0) & 255) > 210u) {                    /* viterbi.cpp:314-332,438-456 */
            __m128i m = _mm_min_epu8(_mm_min_epu8(v0, v1), _mm_min_epu8(v2, v3));
            m = _mm_min_epu8(m, _mm_srli_epi16(m, 8));                              /* low byte of every 16-bit lane: min of the pair; high byte 0 */
            m = _mm_minpos_epu16(_mm_and_si128(m, _mm_set1_epi16(0x00FF)));
            const __m128i mn = _mm_set1_epi8((char)(_mm_extract_epi16(m, 0) & 255));
            v0 = _mm_sub_epi8(v0, mn); v1 = _mm_sub_epi8(v1, mn); v2 = _mm_sub_epi8(v2, mn); v3 = _mm_sub_epi8(v3, mn);
        }
    }
    if (metrics) {
        _mm_storeu_si128((__m128i *)metrics, v0); _mm_storeu_si128((__m128i *)(metrics + 16), v1);
        _mm_storeu_si128((__m128i *)(metrics + 32), v2); _mm_storeu_si128((__m128i *)(metrics + 48), v3);
    }
}
const char *fo_viterbi_simd_kind(void) { return "sse4.1, 16 butterflies per instruction"; }
#else
void fo_viterbi_forward_simd(const uint8_t *symbols, int nsteps, uint64_t *decisions, uint8_t *metrics)
{
    fo_viterbi_forward(symbols, nsteps, decisions, metrics, NULL);
}
const char *fo_viterbi_simd_kind(void) { return "scalar (no SSE4.1 on this host)"; }
#endif

/* A pool of pre-spawned workers, each with its own scratch (no allocation per frame), frames handed out eight at a time
 * from one atomic counter.  fo_pool_decode blocks until the batch is done and may be called any number of times. */
struct fo_pool {
    int threads;
    pthread_t *th;
    struct fo_scratch **scratch;
    pthread_mutex_t mu;
    pthread_cond_t go, done;
    unsigned long epoch;
    int running, quit;
    /* the batch in hand */
    const float *iq; const fo_frame_desc *descs; const int64_t *ends; size_t n_frames;
    uint8_t *psdu; size_t slot; fo_frame_result *res;
    size_t next;          /* (atomic) */
};

struct pool_arg { fo_pool *p; int i; };

static void pool_run(fo_pool *p, struct fo_scratch *sc)
{
    for (;;) {
        const size_t lo = __atomic_fetch_add(&p->next, 8, __ATOMIC_RELAXED);
        if (lo >= p->n_frames) break;
        const size_t hi = lo + 8 < p->n_frames ? lo + 8 : p->n_frames;
        for (size_t f = lo; f < hi; f++)
            decode_alignment_impl(p->iq, p->ends[f], &p->descs[f], p->psdu + f * p->slot, &p->res[f], NULL, NULL, NULL, NULL, sc);
    }
}

static void *pool_worker(void *arg)
{
    struct pool_arg *a = (struct pool_arg *)arg;
    fo_pool *p = a->p; const int i = a->i;
    free(a);
    unsigned long seen = 0;
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (!p->quit && p->epoch == seen) pthread_cond_wait(&p->go, &p->mu);
        if (p->quit) { pthread_mutex_unlock(&p->mu); break; }
        seen = p->epoch;
        pthread_mutex_unlock(&p->mu);
        pool_run(p, p->scratch[i]);
        pthread_mutex_lock(&p->mu);
        if (--p->running == 0) pthread_cond_signal(&p->done);
        pthread_mutex_unlock(&p->mu);
    }
    return NULL;
}

fo_pool *fo_pool_new(int threads)
{
    ensure_tables();
    if (threads < 1) threads = 1;
    fo_pool *p = (fo_pool *)calloc(1, sizeof(fo_pool));
    p->threads = threads;
    p->th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    p->scratch = (struct fo_scratch **)calloc((size_t)threads, sizeof(struct fo_scratch *));
    pthread_mutex_init(&p->mu, NULL); pthread_cond_init(&p->go, NULL); pthread_cond_init(&p->done, NULL);
    for (int i = 0; i < threads; i++) {
        p->scratch[i] = (struct fo_scratch *)malloc(sizeof(struct fo_scratch));
        if (i == 0) continue;                    /* worker 0 is the calling thread */
        struct pool_arg *a = (struct pool_arg *)malloc(sizeof *a);
        a->p = p; a->i = i;
        pthread_create(&p->th[i], NULL, pool_worker, a);
    }
    return p;
}

void fo_pool_free(fo_pool *p)
{
    if (!p) return;
    pthread_mutex_lock(&p->mu); p->quit = 1; pthread_cond_broadcast(&p->go); pthread_mutex_unlock(&p->mu);
    for (int i = 1; i < p->threads; i++) pthread_join(p->th[i], NULL);
    for (int i = 0; i < p->threads; i++) free(p->scratch[i]);
    free(p->scratch); free(p->th);
    pthread_mutex_destroy(&p->mu); pthread_cond_destroy(&p->go); pthread_cond_destroy(&p->done);
    free(p);
}

int fo_pool_threads(const fo_pool *p) { return p->threads; }

void fo_pool_decode(fo_pool *p, const float *iq, const fo_frame_desc *descs, const int64_t *ends, size_t n_frames,
                    uint8_t *psdu, size_t slot_bytes, fo_frame_result *res)
{
    pthread_mutex_lock(&p->mu);
    p->iq = iq; p->descs = descs; p->ends = ends; p->n_frames = n_frames; p->psdu = psdu; p->slot = slot_bytes; p->res = res;
    __atomic_store_n(&p->next, 0, __ATOMIC_RELAXED);
    p->running = p->threads - 1;
    p->epoch++;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    pool_run(p, p->scratch[0]);
    pthread_mutex_lock(&p->mu);
    while (p->running > 0) pthread_cond_wait(&p->done, &p->mu);
    pthread_mutex_unlock(&p->mu);
    int64_t n = 0;                                        /* (the stream is at least as long as the furthest end handed over) */
    for (size_t f = 0; f < n_frames; f++) if (ends[f] > n) n = ends[f];
    v2_fixup(iq, n, descs, ends, n_frames, psdu, slot_bytes, res);
}

/* ------------------------------------------------------------------------------------------ */
/* the blocks behind timing_sync fed with a GIVEN list of alignments, and the batch restatement */
/* of what they do with it (round 4: the partial-vector flush of fft_symbols.cpp:41-50)          */
/* ------------------------------------------------------------------------------------------ */
/* fo_chain_from_tags_f32: the tagged, rotated sample stream timing_sync would hand on for these alignments -- LTS1 at lts1_pos, LTS2
 * 64 later (a later alignment's tag overwrites an earlier one at the same sample, as in timing_sync.cpp:105-106), every sample
 * multiplied by the phasor in force at its index (timing_sync.cpp:121-125: (c, s) of the latest alignment whose rot_start it has
 * reached, (c_prev, s_prev) of the first one before that) -- through fft_symbols, channel_est, phase_tracker and frame_decoder in one
 * call each.  The ground truth for what a batch decoder must deliver for the same descriptors. */
void fo_chain_from_tags_f32(const float *iq, int64_t n, const fo_frame_desc *descs, size_t n_al, fo_payloads *out)
{
    ensure_tables();
    fo_tagged_sample *ts = (fo_tagged_sample *)calloc((size_t)(n > 0 ? n : 1), sizeof(fo_tagged_sample));
    size_t a = 0;                                       /* alignments whose rot_start has been reached */
    cplx rot = n_al ? CMPLX(descs[0].c_prev, descs[0].s_prev) : CMPLX(1.0, 0.0);
    for (int64_t i = 0; i < n; i++) {
        while (a < n_al && descs[a].rot_start <= i) { rot = CMPLX(descs[a].c, descs[a].s); a++; }
        ts[i].sample = from_c(CMPLX((double)iq[2 * i], (double)iq[2 * i + 1]) * rot);
        ts[i].tag = FO_NONE;
    }
    for (size_t j = 0; j < n_al; j++) {
        const int64_t p = descs[j].lts1_pos;
        if (p >= 0 && p < n) ts[p].tag = FO_LTS1;
        if (p + 64 >= 0 && p + 64 < n) ts[p + 64].tag = FO_LTS2;
    }
    fo_fft_symbols *fs = fo_fft_symbols_new(); fo_channel_est *ce = fo_channel_est_new();
    fo_phase_tracker *pt = fo_phase_tracker_new(); fo_frame_decoder *dec = fo_frame_decoder_new();
    fo_tagged_vec64 *v = (fo_tagged_vec64 *)malloc(sizeof(fo_tagged_vec64) * ((size_t)n / 64 + n_al + 8));
    const size_t nv = fo_fft_symbols_work(fs, ts, (size_t)n, v);
    fo_tagged_vec64 *e = (fo_tagged_vec64 *)malloc(sizeof(fo_tagged_vec64) * (nv + 1));
    const size_t ne = fo_channel_est_work(ce, v, nv, e);
    fo_tagged_vec48 *d = (fo_tagged_vec48 *)malloc(sizeof(fo_tagged_vec48) * (ne + 1));
    fo_phase_tracker_work(pt, e, ne, d);
    fo_payloads_clear(out);
    if (ne) fo_frame_decoder_work(dec, d, ne, out);
    free(ts); free(v); free(e); free(d);
    fo_fft_symbols_free(fs); fo_channel_est_free(ce); fo_phase_tracker_free(pt); fo_frame_decoder_free(dec);
}

/* What those blocks do, restated per alignment (the shape a batch decoder has).  Alignment j, LTS1 at p, ends at e = ends[j] (clamped
 * to the stream) and is LINKED to alignment j+1 iff e is that alignment's LTS1 -- i.e. iff the stream goes on into it; an alignment
 * whose samples end anywhere else is where the stream ends as far as it and its predecessors are concerned.  fft_symbols emits for it:
 * the two LTS vectors, then the vectors of the complete symbol windows [p + 144 + 80 k, + 64), k = 0 (SIGNAL) .. K - 1, that end by e,
 * and -- when it is linked and the window in progress has got past its cyclic prefix (m_offset > 15, fft_symbols.cpp:46) -- ONE MORE:
 * the first (e - p - 128) mod 80 - 16 samples of window K, the rest of the vector still holding window K - 1 (the LTS2 window for
 * K = 0).  channel_est equalises all of them with alignment j's estimate and phase_tracker counts them on (k = 0 is START_OF_FRAME).
 * These vectors of all linked alignments, in stream order, are ONE sequence: frame_decoder copies the nsym vectors that follow a valid
 * SIGNAL into its frame wherever they come from -- the partial vector, the next alignment's SIGNAL, its data symbols if that SIGNAL is
 * invalid -- and drops the frame when a VALID SIGNAL arrives before the last of them (frame_decoder.cpp:52-88).
 * Status of alignment j: HEADER_FAIL / OK / CRC_FAIL as the blocks decide; TRUNCATED = the samples ran out (an unlinked end) before its
 * LTS windows, its SIGNAL vector or its frame's last vector; SUPERSEDED = the stream went on into a later alignment that took it over:
 * its LTS or SIGNAL window was cut by the next LTS1 (no vector at all), or a valid SIGNAL arrived before its frame's last vector.
 * Pile-ups: an alignment whose second LTS window is cut (e < p + 128) gives no vector -- what fft_symbols pushes of it is swallowed by
 * channel_est as an LTS vector of an estimate the next LTS_START zeroes (channel_est.cpp:44-50); an alignment less than 64 samples behind
 * another is LATE: see v2_al_compute.  With that the restatement equals the blocks for every tag layout (tests/manual/stress_tags.py places
 * tags down to one sample apart). */
typedef struct { int32_t K, has_part, fresh, nvec, valid, rate, length, nsym, link, dead, done, late; fo_c64 est[64]; } v2_al;

/* The stream: complex<float> samples that the descriptor's phasors rotate (timing_sync.cpp:124-125), or (f64) complex<double> samples
 * that timing_sync has rotated already -- its own output_buffer, as the fused stage block of blocks.hpp receives it. */
typedef struct { const void *iq; int f64; } v2_stream;
static inline cplx v2_sample(const v2_stream *st, int64_t idx, const fo_frame_desc *d)
{
    if (st->f64) { const double *q = (const double *)st->iq; return CMPLX(q[2 * idx], q[2 * idx + 1]); }
    const float *q = (const float *)st->iq;
    const cplx smp = CMPLX((double)q[2 * idx], (double)q[2 * idx + 1]);
    return smp * (idx >= d->rot_start ? CMPLX(d->c, d->s) : CMPLX(d->c_prev, d->s_prev));
}

static void v2_vector(const v2_stream *iq, const fo_frame_desc *d, const v2_al *a, int k, fo_tagged_vec48 *out)
{
    const int64_t w0 = d->lts1_pos + 80 * a->late + 144 + 80 * (int64_t)k;
    const int partial = a->has_part && k == a->K;
    fo_tagged_vec64 v, eq;
    for (int i = 0; i < 64; i++) {
        const int64_t idx = (partial && i >= a->fresh) ? w0 - 80 + i : w0 + i;
        v.samples[i] = from_c(v2_sample(iq, idx, d));
    }
    v.tag = FO_NONE; v._pad = 0;
    fo_fft64(v.samples);
    fo_channel_est ce;
    memset(&ce, 0, sizeof ce);
    memcpy(ce.est, a->est, sizeof ce.est);
    fo_channel_est_work(&ce, &v, 1, &eq);
    fo_phase_tracker pt; pt.symbol_count = k;
    fo_phase_tracker_work(&pt, &eq, 1, out);
}

/* An LTS1 less than 64 samples behind an earlier one of the same stream (linked all the way): see v2_al_compute */
static int v2_is_late(const fo_frame_desc *descs, const int64_t *ends, int64_t n, size_t j)
{
    const int64_t p = descs[j].lts1_pos;
    for (size_t i = j; i-- > 0;) {
        const int64_t ei = ends ? ends[i] : descs[i + 1].lts1_pos;
        const int64_t dp = p - descs[i].lts1_pos;
        if (ei != descs[i + 1].lts1_pos || ei > n || dp < 0 || dp >= 64) return 0;
        if (dp > 0) return 1;
    }
    return 0;
}

/* what alignment j contributes: extent, vectors, channel estimate, SIGNAL outcome (n_tot = alignments in descs, context included) */
static void v2_al_compute(const v2_stream *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_tot, size_t j, v2_al *a)
{
    if (a->done) return;
    memset(a, 0, sizeof *a);
    a->done = 1;
    const fo_frame_desc *d = &descs[j];
    const int64_t p = d->lts1_pos;
    const int64_t e_raw = ends ? ends[j] : (j + 1 < n_tot ? descs[j + 1].lts1_pos : n);
    const int64_t e = e_raw < n ? e_raw : n;
    a->link = j + 1 < n_tot && e_raw == descs[j + 1].lts1_pos && e_raw <= n;
    /* An LTS1 less than 64 samples behind an earlier one: that alignment's LTS2 tag (timing_sync.cpp:105-106) falls inside this one's first
     * LTS window and restarts the vector there (fft_symbols.cpp:53-56), this alignment's own LTS2 tag restarts it again, and the first
     * complete vector -- still tagged LTS_START -- is the window at p + 64; the next one, 80 samples on, is taken as the second LTS vector
     * (channel_est.cpp:44-58), and START_OF_FRAME goes to the window behind that: everything sits one symbol LATE. */
    a->late = v2_is_late(descs, ends, n, j);
    const int64_t q = p + 80 * a->late;                               /* where an undisturbed alignment with the same vector grid would sit */
    if (p < 0 || e < q + 128) { a->dead = 1; return; }                /* LTS cut off: no estimate, no vectors */
    a->K = e >= q + 208 ? (int32_t)((e - (q + 208)) / 80 + 1) : 0;
    const int mo = (int)((e - (q + 128)) % 80);
    a->has_part = a->link && mo > 15;
    a->fresh = a->has_part ? mo - 16 : 0;
    a->nvec = a->K + a->has_part;
    /* channel_est.cpp:44-58 on the two LTS vectors */
    fo_channel_est ce;
    memset(&ce, 0, sizeof ce);
    for (int w = 0; w < 2; w++) {
        fo_tagged_vec64 v, dummy;
        const int64_t w0 = w == 0 ? (a->late ? p + 64 : p) : q + 64;
        for (int i = 0; i < 64; i++) v.samples[i] = from_c(v2_sample(iq, w0 + i, d));
        v.tag = w == 0 ? FO_LTS_START : FO_NONE; v._pad = 0;
        fo_fft64(v.samples);
        fo_channel_est_work(&ce, &v, 1, &dummy);
    }
    memcpy(a->est, ce.est, sizeof a->est);
    if (a->nvec == 0) return;                                         /* not even part of a SIGNAL vector: no START_OF_FRAME from this alignment */
    fo_tagged_vec48 sig;
    v2_vector(iq, d, a, 0, &sig);
    a->valid = fo_decode_header(sig.samples, &a->rate, &a->length, &a->nsym);
}

/* the outcome of alignment j (al: one record per alignment of descs, zeroed before the first call: filled as the walk needs them) */
static void v2_resolve(const v2_stream *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_tot, v2_al *al, size_t j, uint8_t *psdu,
                       fo_frame_result *res)
{
    v2_al_compute(iq, n, descs, ends, n_tot, j, &al[j]);
    const v2_al *a = &al[j];
    res->status = FO_ST_TRUNCATED; res->rate = -1; res->length = 0; res->num_symbols = 0;
    if (a->dead || a->nvec == 0) { if (a->link) res->status = FO_ST_SUPERSEDED; return; }
    if (!a->valid) { res->status = FO_ST_HEADER_FAIL; return; }
    res->rate = a->rate; res->length = a->length; res->num_symbols = a->nsym;
    /* the frame's vectors 1 .. nsym of the sequence behind its SIGNAL: its own (k = 1 .. nvec - 1), then those of the alignments it is linked to */
    int64_t c = a->nvec - 1;
    size_t g = j;
    while (c < a->nsym) {
        if (!al[g].link) return;                                      /* the samples end first: TRUNCATED */
        g++;
        v2_al_compute(iq, n, descs, ends, n_tot, g, &al[g]);
        if (al[g].nvec == 0) continue;
        if (al[g].valid && c + 1 < a->nsym) { res->status = FO_ST_SUPERSEDED; return; }     /* a valid SIGNAL before the frame's last vector */
        c += al[g].nvec;
    }
    fo_c64 *car = (fo_c64 *)malloc(sizeof(fo_c64) * 48 * (size_t)a->nsym);
    size_t src = j;
    int64_t k = 1;                                                    /* next vector of alignment src */
    for (int64_t v = 1; v <= a->nsym; v++) {
        while (k >= al[src].nvec) { src++; k = 0; }                   /* (alignments without vectors are skipped over) */
        fo_tagged_vec48 dv;
        v2_vector(iq, &descs[src], &al[src], (int)k, &dv);
        memcpy(car + (size_t)(v - 1) * 48, dv.samples, sizeof dv.samples);
        k++;
    }
    const int ok = fo_decode_data(car, a->rate, a->length, psdu, NULL, NULL);
    res->status = ok ? FO_ST_OK : FO_ST_CRC_FAIL;
    free(car);
}

void fo_decode_batch_v2_f32(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_al, size_t n_ctx, uint8_t *psdu,
                            size_t slot_bytes, fo_frame_result *res)
{
    ensure_tables();
    const size_t n_tot = n_al + n_ctx;
    v2_al *al = (v2_al *)calloc(n_tot ? n_tot : 1, sizeof(v2_al));
    const v2_stream st = { iq, 0 };
    for (size_t j = 0; j < n_al; j++) v2_resolve(&st, n, descs, ends, n_tot, al, j, psdu + j * slot_bytes, &res[j]);
    free(al);
}

void fo_decode_batch_v2_f64(const double *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_al, size_t n_ctx, uint8_t *psdu,
                            size_t slot_bytes, fo_frame_result *res)
{
    ensure_tables();
    const size_t n_tot = n_al + n_ctx;
    v2_al *al = (v2_al *)calloc(n_tot ? n_tot : 1, sizeof(v2_al));
    const v2_stream st = { iq, 1 };
    for (size_t j = 0; j < n_al; j++) v2_resolve(&st, n, descs, ends, n_tot, al, j, psdu + j * slot_bytes, &res[j]);
    free(al);
}

/* After a pass that decoded every alignment on its own (fo_decode_batch_f32, fo_pool_decode): the alignments whose outcome depends
 * on the alignments around them -- linked and cut short by the next LTS1, or one symbol late behind a pile-up -- are decided again by the
 * rules above.  Everything else
 * (a complete SIGNAL window with an invalid header, a frame that fits in front of the next LTS1, an unlinked end) comes out the same
 * either way. */
static void v2_fixup(const float *iq, int64_t n, const fo_frame_desc *descs, const int64_t *ends, size_t n_frames, uint8_t *psdu, size_t slot_bytes,
                     fo_frame_result *res)
{
    v2_al *al = NULL;
    for (size_t j = 0; j < n_frames; j++) {
        const int cut = j + 1 < n_frames && res[j].status == FO_ST_TRUNCATED && ends[j] == descs[j + 1].lts1_pos && ends[j] <= n;
        if (!cut && !v2_is_late(descs, ends, n, j)) continue;
        if (!al) al = (v2_al *)calloc(n_frames, sizeof(v2_al));
        const v2_stream st = { iq, 0 };
        memset(psdu + j * slot_bytes, 0, slot_bytes);                     /* (what the alignment gave on its own does not count) */
        v2_resolve(&st, n, descs, ends, n_frames, al, j, psdu + j * slot_bytes, &res[j]);
    }
    free(al);
}
