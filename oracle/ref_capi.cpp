// ref_capi.cpp -- extern "C" handles onto the REAL reference, for pinning the oracle.
//
// TEST INFRASTRUCTURE, dev container only.  This file is ours; it #includes the reference's headers
// from $(REF)/src (default /root/reference/src) and is linked with the reference's own self-contained
// translation units compiled where they lie (see oracle/Makefile: viterbi, parity, interleaver,
// puncturer, modulator, channel_est, phase_tracker, frame_detector, timing_sync, symbol_mapper).
// No reference source is copied into this repository and no stand-in for FFTW3/Boost/UHD is written:
// the reference files that need those libraries (fft, fft_symbols, ppdu, frame_decoder,
// frame_builder, receiver_chain, usrp, receiver, transmitter) are simply not part of this build.
// Output goes to oracle/_ref/ only (git-ignored; it does travel to the GPU box but nothing there
// needs it).
#include <complex>
#include <cstring>
#include <vector>
#include <stdint.h>

#include "viterbi.h"
#include "parity.h"
#include "interleaver.h"
#include "puncturer.h"
#include "modulator.h"
#include "rates.h"
#include "preamble.h"
#include "symbol_mapper.h"
#include "frame_detector.h"
#include "timing_sync.h"
#include "channel_est.h"
#include "phase_tracker.h"

typedef std::complex<double> cd;

extern "C" {

int ref_sizeof_tagged_sample() { return (int)sizeof(fun::tagged_sample); }
int ref_sizeof_tagged_vector64() { return (int)sizeof(fun::tagged_vector<64>); }
int ref_sizeof_tagged_vector48() { return (int)sizeof(fun::tagged_vector<48>); }

int ref_parity(int x) { return fun::parity(x); }

void ref_conv_encode(const unsigned char *data, unsigned char *symbols, int data_bits)
{
    fun::viterbi v;
    v.conv_encode(const_cast<unsigned char *>(data), symbols, data_bits);
}

void ref_conv_decode(const unsigned char *symbols, unsigned char *data, int data_bits)
{
    fun::viterbi v;
    v.conv_decode(const_cast<unsigned char *>(symbols), data, data_bits);
}

void ref_interleave(const unsigned char *in, size_t n, unsigned char *out)
{
    std::vector<unsigned char> r = fun::interleaver::interleave(std::vector<unsigned char>(in, in + n));
    memcpy(out, r.data(), r.size());
}

void ref_deinterleave(const unsigned char *in, size_t n, unsigned char *out)
{
    std::vector<unsigned char> r = fun::interleaver::deinterleave(std::vector<unsigned char>(in, in + n));
    memcpy(out, r.data(), r.size());
}

size_t ref_puncture(const unsigned char *in, size_t n, int rate, unsigned char *out)
{
    std::vector<unsigned char> r = fun::puncturer::puncture(std::vector<unsigned char>(in, in + n), fun::RateParams((fun::Rate)rate));
    memcpy(out, r.data(), r.size());
    return r.size();
}

size_t ref_depuncture(const unsigned char *in, size_t n, int rate, unsigned char *out)
{
    std::vector<unsigned char> r = fun::puncturer::depuncture(std::vector<unsigned char>(in, in + n), fun::RateParams((fun::Rate)rate));
    memcpy(out, r.data(), r.size());
    return r.size();
}

size_t ref_modulate(const unsigned char *bits, size_t n, int rate, cd *out)
{
    std::vector<cd> r = fun::modulator::modulate(std::vector<unsigned char>(bits, bits + n), (fun::Rate)rate);
    memcpy(out, r.data(), r.size() * sizeof(cd));
    return r.size();
}

size_t ref_demodulate(const cd *in, size_t n, int rate, unsigned char *out)
{
    std::vector<unsigned char> r = fun::modulator::demodulate(std::vector<cd>(in, in + n), (fun::Rate)rate);
    memcpy(out, r.data(), r.size());
    return r.size();
}

size_t ref_symbol_map(const cd *in, size_t n48, cd *out)
{
    fun::symbol_mapper m;
    std::vector<cd> r = m.map(std::vector<cd>(in, in + n48));
    memcpy(out, r.data(), r.size() * sizeof(cd));
    return r.size();
}

void ref_rate_params(int rate, int *out5, double *rel_rate)
{
    fun::RateParams rp((fun::Rate)rate);
    out5[0] = rp.rate_field; out5[1] = rp.cbps; out5[2] = rp.dbps; out5[3] = rp.bpsc; out5[4] = (int)rp.rate;
    *rel_rate = rp.rel_rate;
}

int ref_rate_from_field(int field)
{
    for (size_t i = 0; i < fun::VALID_RATES.size(); i++)
        if (fun::VALID_RATES[i] == field) return (int)fun::RateParams::FromRateField((unsigned char)field).rate;
    return -1;
}

void ref_preamble_samples(cd *out) { memcpy(out, fun::PREAMBLE_SAMPLES, 320 * sizeof(cd)); }
void ref_lts_freq_domain(cd *out) { memcpy(out, fun::LTS_FREQ_DOMAIN, 64 * sizeof(cd)); }
void ref_lts_time_domain_conj(cd *out) { memcpy(out, fun::LTS_TIME_DOMAIN_CONJ, 64 * sizeof(cd)); }

// ---- blocks: set input_buffer, call work(), copy output_buffer ----
void *ref_frame_detector_new() { return new fun::frame_detector(); }
void ref_frame_detector_free(void *p) { delete (fun::frame_detector *)p; }
void ref_frame_detector_work(void *p, const cd *in, size_t n, fun::tagged_sample *out)
{
    fun::frame_detector *b = (fun::frame_detector *)p;
    b->input_buffer.assign(in, in + n);
    b->work();
    memcpy(out, b->output_buffer.data(), b->output_buffer.size() * sizeof(fun::tagged_sample));
}

void *ref_timing_sync_new() { return new fun::timing_sync(); }
void ref_timing_sync_free(void *p) { delete (fun::timing_sync *)p; }
void ref_timing_sync_work(void *p, const fun::tagged_sample *in, size_t n, fun::tagged_sample *out)
{
    fun::timing_sync *b = (fun::timing_sync *)p;
    b->input_buffer.assign(in, in + n);
    b->work();
    memcpy(out, b->output_buffer.data(), b->output_buffer.size() * sizeof(fun::tagged_sample));
}

void *ref_channel_est_new() { return new fun::channel_est(); }
void ref_channel_est_free(void *p) { delete (fun::channel_est *)p; }
size_t ref_channel_est_work(void *p, const fun::tagged_vector<64> *in, size_t n, fun::tagged_vector<64> *out)
{
    fun::channel_est *b = (fun::channel_est *)p;
    b->input_buffer.assign(in, in + n);
    b->work();
    memcpy(out, b->output_buffer.data(), b->output_buffer.size() * sizeof(fun::tagged_vector<64>));
    return b->output_buffer.size();
}

void *ref_phase_tracker_new() { return new fun::phase_tracker(); }
void ref_phase_tracker_free(void *p) { delete (fun::phase_tracker *)p; }
size_t ref_phase_tracker_work(void *p, const fun::tagged_vector<64> *in, size_t n, fun::tagged_vector<48> *out)
{
    fun::phase_tracker *b = (fun::phase_tracker *)p;
    b->input_buffer.assign(in, in + n);
    b->work();
    memcpy(out, b->output_buffer.data(), b->output_buffer.size() * sizeof(fun::tagged_vector<48>));
    return b->output_buffer.size();
}

}  // extern "C"
