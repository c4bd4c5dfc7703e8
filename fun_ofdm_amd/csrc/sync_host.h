// sync_host.h -- host-side frame_detector + timing_sync, producing alignment descriptors.
//
// The reference's two pre-sync blocks (frame_detector.cpp:41-93 with circular_accumulator.h:88-95, and
// timing_sync.cpp:51-139) sit in front of the hot path inside receiver_chain::process_samples().
// Until they run on the device this restatement supplies what the device path needs
// from them: where each LTS1 tag lands and which constant phasor timing_sync applies from where.
// It never touches sample values (the rotation itself is applied in-kernel), so it only has to
// reproduce the blocks' DECISIONS; it does that with the same operation order in fp64.
//
// Both blocks are chunk-size independent (their carry-overs make a chunked run equal to a one-shot run) up to one line of
// timing_sync (`if(lts_offset < 0) break;`, relative to a call's buffer): that one is decided here for the call size of the
// reference's own receiver (4096; set_call), whatever the sizes push() is called with.
#pragma once

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "../../include/fun_ofdm_amd.h"

namespace foa {

static inline void phasor(double phase, double &c, double &s)
{
#if defined(__GLIBC__)
    ::sincos(phase, &s, &c);
#else
    c = std::cos(phase); s = std::sin(phase);
#endif
}

// preamble.h:432: conj of the 64-sample long training symbol, as printed with 12 significant digits
inline void make_lts_time_conj(std::complex<double> *out)
{
    static const signed char L[53] = { 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                                       1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1 };
    auto round12 = [](double v) {
        if (std::fabs(v) < 1e-15) return 0.0;                   // exact zeros of the table (sums that cancel up to rounding)
        char buf[64];
        snprintf(buf, sizeof buf, "%.12g", v);
        return strtod(buf, nullptr);
    };
    for (int n = 0; n < 64; n++) {
        // inverse DFT of L(-26..26) / 64, summed in exact-angle form (k*n mod 64 keeps arguments small)
        long double re = 0, im = 0;
        for (int i = 0; i < 53; i++) {
            int k = i - 26;
            int e = ((k * n) % 64 + 64) % 64;
            long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double)e / 64.0L;
            re += L[i] * cosl(a);
            im += L[i] * sinl(a);
        }
        out[n] = std::complex<double>(round12((double)(re / 64.0L)), round12((double)(-im / 64.0L)));
    }
}

// preamble.h:24: the 320 preamble samples as the literal table holds them -- ten short symbols, the long symbol's
// last 32 samples, two long symbols; 12 significant digits, [0] halved by the window, [160] = -0.078 (not -0.078125).
inline void make_preamble(std::complex<double> *out)
{
    auto round12 = [](double v) {
        if (std::fabs(v) < 1e-15) return 0.0;                   // exact zeros of the table (sums that cancel up to rounding)
        char buf[64];
        snprintf(buf, sizeof buf, "%.12g", v);
        return strtod(buf, nullptr);
    };
    // 802.11a-1999 17.3.3: S(-26..26) = sqrt(13/6) (1+j) x {+-1 at multiples of 4}
    static const int sk[12] = { -24, -20, -16, -12, -8, -4, 4, 8, 12, 16, 20, 24 };
    static const int sv[12] = { 1, -1, 1, -1, -1, 1, -1, -1, 1, 1, 1, 1 };
    const long double amp = sqrtl(13.0L / 6.0L);
    std::complex<double> sts[16], ltc[64];
    for (int n = 0; n < 16; n++) {
        long double re = 0, im = 0;
        for (int i = 0; i < 12; i++) {
            const int e = ((sk[i] * n) % 64 + 64) % 64;
            const long double a = 2.0L * 3.141592653589793238462643383279502884L * (long double)e / 64.0L;
            // (1 + j) (cos a + j sin a)
            re += sv[i] * (cosl(a) - sinl(a));
            im += sv[i] * (cosl(a) + sinl(a));
        }
        sts[n] = std::complex<double>((double)(amp * re / 64.0L), (double)(amp * im / 64.0L));
    }
    make_lts_time_conj(ltc);                                     // conj of the long symbol, already rounded
    for (int i = 0; i < 160; i++) out[i] = std::complex<double>(round12(sts[i % 16].real()), round12(sts[i % 16].imag()));
    for (int i = 0; i < 160; i++) out[160 + i] = std::conj(ltc[(32 + i) % 64]);
    out[0] = std::complex<double>(round12(sts[0].real() / 2.0), round12(sts[0].imag() / 2.0));
    out[160] = std::complex<double>(-0.078, 0.0);
}

// Index of stream sample xs in the working buffer of the reference call that walks over it (timing_sync.cpp:57-69: a call's buffer
// is the 160 samples before it + its own `call` samples, and it walks the first `call` of them); call = 0: one call for the whole stream.
constexpr int64_t kSyncCallDefault = 4096;             // receiver.h:16 NUM_RX_SAMPLES
#if defined(__HIPCC__)
__host__ __device__
#endif
inline int64_t sync_call_index(int64_t xs, int64_t call)
{
    const int64_t v = xs + 160;
    return call > 0 ? ((v % call) + call) % call : v;
}

class SyncHost {
public:
    SyncHost() : lts_conj_(64)
    {
        make_lts_time_conj(lts_conj_.data());
        hist_.assign(kCarry, Tagged{});
    }

    // Feed n samples (interleaved re,im; T = float or double).  Alignments completed by this chunk are
    // appended to out with stream-absolute positions.
    template <typename T>
    void push(const T *iq, size_t n, std::vector<foa_frame_desc> &out)
    {
        if (n == 0) return;
        // hist_ holds the 160 most recent tagged samples that timing_sync has not walked over yet
        const size_t base = hist_.size();
        hist_.resize(base + n);
        for (size_t i = 0; i < n; i++) {
            std::complex<double> x((double)iq[2 * i], (double)iq[2 * i + 1]);
            hist_[base + i].v = x;
            hist_[base + i].tag = detect(x);
        }
        // timing_sync.cpp:66-126: walk x over the first n entries; look-ahead reaches x+159
        const int64_t origin = consumed_ - (int64_t)kCarry;      // stream index of hist_[0]
        for (size_t x = 0; x < n; x++) {
            if (hist_[x].tag != kStsEnd) continue;
            std::vector<std::pair<double, int>> peaks;
            for (size_t p = x; p < x + kCarry - 64; p++) {
                std::complex<double> corr(0, 0);
                double power = 0;
                for (int s = 0; s < 64; s++) {
                    corr += hist_[p + s].v * lts_conj_[s];
                    power += std::norm(hist_[p + s].v);
                }
                double cn = std::abs(corr) / power;
                if (cn > 0.9) peaks.push_back(std::make_pair(cn, (int)p));
            }
            std::sort(peaks.begin(), peaks.end());
            std::reverse(peaks.begin(), peaks.end());
            // timing_sync.cpp:91-95: s advances by 5 while s < min(size,3), so only the strongest peak is
            // paired, against the five strongest
            if (peaks.empty()) continue;
            const int lim = std::min((int)peaks.size(), 5);
            for (int t = 0; t < lim; t++) {
                if (std::abs(peaks[0].second - peaks[t].second) != 64) continue;
                const int lts_offset = std::min(peaks[0].second, peaks[t].second) - 32;       // index into hist_; may be negative here
                // timing_sync.cpp:99 `if(lts_offset < 0) break;` is relative to the working buffer of the CALL that examines this
                // STS_END (160 carried-over samples + the call's own): the reference drops an alignment whose LTS guard interval
                // would start before that buffer.  Which sample a call's buffer starts with depends on how the stream is cut into
                // calls -- the reference's receiver cuts it every NUM_RX_SAMPLES = 4096 (receiver.h:16) -- not on how it is pushed
                // here: x's index in that buffer is (stream index + 160) mod call.
                const int64_t x_in = sync_call_index(origin + (int64_t)x, call_);
                if (x_in + (int64_t)lts_offset - (int64_t)x < 0) break;
                if (lts_offset + 24 >= 0) hist_[lts_offset + 24].tag = kLts1;           // may overwrite a pending STS_END (one behind x is never read again)
                hist_[lts_offset + 24 + 64].tag = kLts2;
                const std::complex<double> v = hist_[lts_offset + 32 + 128 - 1].v * lts_conj_[63];
                const double prev = phase_acc_;
                phase_acc_ = std::arg(v);                          // m_phase_offset stays 0 (dead loop :109)
                // the per-sample wrap of timing_sync.cpp:116-117 cannot trigger: |arg| <= pi
                foa_frame_desc d;
                d.lts1_pos = origin + lts_offset + 24;
                d.rot_start = origin + (int64_t)x;
                // cos and sin of ONE argument as the reference's build computes them: g++ -O3 turns the pair of calls in timing_sync.cpp:124
                // into one sincos(), whose cosine is not always the bit pattern cos() returns (glibc; 1 ulp apart on 2 of 4 685 alignments of
                // round 4's collision streams, where this file's separate calls -- clang keeps them separate -- were caught against the oracle,
                // which is compiled like the reference and equals its compiled timing_sync bit for bit on those streams)
                phasor(phase_acc_, d.c, d.s);
                phasor(prev, d.c_prev, d.s_prev);
                out.push_back(d);
                break;
            }
        }
        hist_.erase(hist_.begin(), hist_.begin() + n);
        consumed_ += (int64_t)n;
    }

    // The call size of the reference's receiver to decide by (timing_sync.cpp:99, above); 0: as ONE call over the whole stream would.
    void set_call(int64_t call) { call_ = call; }

    // Everything up to this stream index has been walked by timing_sync (alignments whose STS_END lies
    // before it have been reported).
    int64_t settled() const { return consumed_ - (int64_t)kCarry; }
    int64_t consumed() const { return consumed_; }

private:
    enum { kNone = 0, kStsStart = 1, kStsEnd = 2, kLts1 = 4, kLts2 = 5 };
    static constexpr size_t kCarry = 160;
    int64_t call_ = kSyncCallDefault;
    struct Tagged { std::complex<double> v{ 0, 0 }; int tag = 0; };

    // frame_detector.cpp:51-84, one sample
    int detect(const std::complex<double> &x)
    {
        std::complex<double> c = x * std::conj(delay_[dpos_]);
        delay_[dpos_] = x;
        dpos_ = (dpos_ + 1) & 15;
        if (c != c) c = 0;                                         // circular_accumulator.h:90
        corr_sum_ -= corr_ring_[ridx_];
        corr_sum_ += c;
        corr_ring_[ridx_] = c;
        double pw = std::norm(x);
        if (pw != pw) pw = 0;
        pow_sum_ -= pow_ring_[ridx_];
        pow_sum_ += pw;
        pow_ring_[ridx_] = pw;
        ridx_ = (ridx_ + 1) & 15;
        const double corr = std::abs(corr_sum_) / pow_sum_;
        int tag = kNone;
        if (corr > 0.9) {
            if (++plateau_ == 16) { tag = kStsStart; flag_ = true; }
        } else {
            if (flag_) { tag = kStsEnd; flag_ = false; }
            plateau_ = 0;
        }
        return tag;
    }

    std::vector<std::complex<double>> lts_conj_;
    std::vector<Tagged> hist_;
    int64_t consumed_ = 0;
    double phase_acc_ = 0;
    // frame_detector state
    std::complex<double> delay_[16] = {};
    std::complex<double> corr_ring_[16] = {};
    std::complex<double> corr_sum_{ 0, 0 };
    double pow_ring_[16] = {};
    double pow_sum_ = 0;
    int dpos_ = 0, ridx_ = 0, plateau_ = 0;
    bool flag_ = false;
};

}  // namespace foa
