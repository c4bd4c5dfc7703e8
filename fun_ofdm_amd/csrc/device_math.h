// device_math.h -- device-side arithmetic shared by the kernels of every unit: the reference's fp64 complex operations with
// its roundings, the 64-point DFT across a wave, the pilot phase fit, the soft demapper, the interleaver / puncturer maps.
//
// Arithmetic is fp64 like the reference (std::complex<double> end to end, tagged_vector.h:46): the 20 MS/s stream is float in
// HBM and is widened on load exactly as the CPU receiver widens it, so soft bytes agree with the CPU to the last bit except
// where a carrier lands within ~1e-15 of a truncation boundary of qam.h:112.
#pragma once

#include "foa_common.h"

namespace foa {

// The constant tables of this translation unit (each unit that holds kernels has its own copy; see foa_common.h).
static __constant__ DeviceTables g_tab;

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
// Three radix-4 decimation-in-frequency stages exchanged through LDS; the output stays digit-reversed in the lanes and
// every later per-subcarrier step indexes its tables by the lane's subcarrier, so no reordering pass exists.
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

// The same three stages on 64 values held by ONE lane (the transmit side's inverse DFT, tx_kernels.h); X[k] ends at index
// rev4(k) (see lane_subcarrier / subcarrier_lane).  Same association as fft64_lane.
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

// A stream sample as the arithmetic takes it: complex<float> widened exactly as the CPU receiver widens it; complex<double> as it is
// (the fused stage block of blocks.hpp hands over the doubles timing_sync produced).
__device__ __forceinline__ cpx widen(float2 v) { return cpx{ (double)v.x, (double)v.y }; }
__device__ __forceinline__ cpx widen(double2 v) { return cpx{ v.x, v.y }; }

template <typename S> struct sample_is_rotated { static constexpr bool value = false; };
template <> struct sample_is_rotated<double2> { static constexpr bool value = true; };     // timing_sync's own output: taken as it is

// timing_sync.cpp:124-125 rotation (+ widening) of one window sample
template <typename S>
__device__ __forceinline__ cpx rotate_sample(S raw, cpx r)
{
    if constexpr (sample_is_rotated<S>::value) return widen(raw);
    else return cmul(widen(raw), r);
}
template <typename S>
__device__ __forceinline__ cpx load_rotated(const S *iq, int64_t idx, const foa_frame_desc &d)
{
    cpx r = idx >= d.rot_start ? cpx{ d.c, d.s } : cpx{ d.c_prev, d.s_prev };
    return rotate_sample(iq[idx], r);
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

// phase_tracker.cpp:83-99 for one symbol held across a wave by fft64_lane: returns the derotated carrier of this lane
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

// qam.h:110-125 through the table (DeviceTables::qam_lut, copied to LDS by its users): the three soft bytes of one axis in one dword
__device__ __forceinline__ uint32_t qam_lookup(const uint32_t *qam, double sym, double scale_d)
{
#pragma clang fp contract(off)
    const int pt = trunc_to_int(sym * scale_d);
    return qam[min(max(pt, -320), 320) + 320];
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

// words reserved per frame in the per-step buffers (soft pairs / decisions / decoded): the chain-back reads whole 48-step chunks
__host__ __device__ constexpr int64_t dec_words(int64_t nsteps) { return nsteps > 0 ? (nsteps + 48 + 63) & ~(int64_t)63 : 0; }
// The decoded bits of a frame (one per data step, 32 to a word) live at word dec_off / 16 of the `decoded` buffer: half a bit of room per
// step -- the frame's words, the tail of the last 96-step unit the walk writes whole, and the 16-byte pieces the finish moves --
// where one word per step (rounds 1-5) was 4 of the 14 bytes of work-set capacity a trellis step cost.  dec_off is a multiple of 64:
// 16-byte aligned.
__host__ __device__ constexpr int64_t decoded_word_off(int64_t dec_off) { return dec_off >> 4; }
__host__ __device__ constexpr size_t decoded_words_for(size_t dec_cap) { return dec_cap / 16 + 64; }
// chain-back segments of a frame (viterbi_tb.h): its nsteps - 6 data steps in pieces of seg_steps
__host__ __device__ constexpr int tb_segments(int nsteps, int seg_steps) { return nsteps > 6 ? (nsteps - 6 + seg_steps - 1) / seg_steps : 0; }

}  // namespace foa
