// test_adaptors.cpp -- the C++ drop-in surface (include/fun_ofdm_amd/blocks.hpp) against the oracle.
// Build/run: tests/test_gpu_cpp_adaptors.py (needs a GPU).  Exit code 0 = all checks passed.
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "fun_ofdm_amd/blocks.hpp"
extern "C" {
#include "fo_oracle.h"
}

typedef std::vector<std::vector<unsigned char> > payloads_t;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

// fun_amd::receiver takes a plain function pointer like fun::receiver does (src/receiver.h:58)
static payloads_t g_rx_packets;
static std::atomic<int> g_rx_calls(0);
static void rx_callback(std::vector<std::vector<unsigned char> > packets)
{
    g_rx_calls++;
    for (auto &p : packets) g_rx_packets.push_back(p);
}

static unsigned long long rng_state = 88172645463325252ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

static double rel_err(const std::complex<double> *a, const fo_c64 *b, int n, const bool *use = nullptr)
{
    double num = 0, den = 1e-300;
    for (int i = 0; i < n; i++) {
        if (use && !use[i]) continue;
        num = std::max(num, std::abs(a[i] - std::complex<double>(b[i].re, b[i].im)));
        den = std::max(den, std::hypot(b[i].re, b[i].im));
    }
    return num / den;
}

int main()
{
    static_assert(sizeof(fun::tagged_sample) == sizeof(fo_tagged_sample), "tagged_sample layout");
    static_assert(sizeof(fun::tagged_vector<64>) == sizeof(fo_tagged_vec64), "tagged_vector<64> layout");
    static_assert(sizeof(fun::tagged_vector<48>) == sizeof(fo_tagged_vec48), "tagged_vector<48> layout");
    // ---- a stream: mixed rates and lengths, AWGN ~25 dB, all values float-representable ----
    const int rates[] = { 10, 0, 5, 8, 3, 9, 10, 2 };
    const int lens[] = { 1024, 100, 300, 1500, 57, 700, 33, 211 };
    std::vector<std::complex<double> > stream;
    payloads_t sent;
    for (int k = 0; k < 8; k++) {
        std::vector<unsigned char> pay(lens[k]);
        for (auto &b : pay) b = (unsigned char)(urand() * 256);
        std::vector<fo_c64> fr(fo_frame_samples(rates[k], lens[k]));
        fo_build_frame(pay.data(), lens[k], rates[k], fr.data());
        const double ph = urand() * 6.283185307179586;
        const std::complex<double> rot(cos(ph), sin(ph));
        stream.insert(stream.end(), 150 + (size_t)(urand() * 500), std::complex<double>(0, 0));
        for (auto &s : fr) stream.push_back(std::complex<double>(s.re, s.im) * rot);
        sent.push_back(pay);
    }
    stream.insert(stream.end(), 700, std::complex<double>(0, 0));
    const double sigma = sqrt(0.0124 / 2 / pow(10.0, 2.5));
    for (auto &s : stream) s = std::complex<double>((float)(s.real() + sigma * nrand()), (float)(s.imag() + sigma * nrand()));
    const size_t chunk = 4096;
    while (stream.size() % chunk) stream.push_back(std::complex<double>(0, 0));
    for (int i = 0; i < 8 * (int)chunk; i++) stream.push_back(std::complex<double>(0, 0));     // flush the reference's 5-call latency

    // ---- expected: the oracle's receiver_chain ----
    payloads_t want;
    {
        fo_receiver_chain *rc = fo_receiver_chain_new();
        for (size_t x = 0; x < stream.size(); x += chunk) {
            const fo_payloads *p = fo_receiver_chain_process_samples(rc, reinterpret_cast<const fo_c64 *>(&stream[x]), chunk);
            for (size_t i = 0; i < fo_payloads_count(p); i++) want.push_back(std::vector<unsigned char>(fo_payloads_data(p, i), fo_payloads_data(p, i) + fo_payloads_len(p, i)));
        }
        fo_receiver_chain_free(rc);
    }
    // the reference's sync misses a frame now and then (or a CRC fails); what it does return must be what was sent
    CHECK(want.size() >= 6, "oracle decoded only %zu of 8 frames", want.size());
    for (auto &w : want) CHECK(std::find(sent.begin(), sent.end(), w) != sent.end(), "oracle returned a payload that was not sent");

    // ---- 1. per-block adaptors fed by the (oracle's) pre-sync blocks, against the oracle's own blocks ----
    {
        fo_frame_detector *fd = fo_frame_detector_new();
        fo_timing_sync *ts = fo_timing_sync_new();
        fo_fft_symbols *ofs = fo_fft_symbols_new();
        fo_channel_est *oce = fo_channel_est_new();
        fo_phase_tracker *opt = fo_phase_tracker_new();
        fun_amd::fft_symbols fs;
        fun_amd::channel_est ce;
        fun_amd::phase_tracker pt;
        fun_amd::frame_decoder dec;
        payloads_t got;
        bool used[64];
        for (int j = 0; j < 64; j++) used[j] = !(j < 6 || j > 58 || j == 32);
        double e_fft = 0, e_eq = 0, e_pt = 0;
        std::vector<fo_tagged_sample> a(chunk), b(chunk);
        std::vector<fo_tagged_vec64> ov(chunk / 64 + 4), oe(chunk / 64 + 4);
        std::vector<fo_tagged_vec48> op(chunk / 64 + 4);
        for (size_t x = 0; x < stream.size(); x += chunk) {
            fo_frame_detector_work(fd, reinterpret_cast<const fo_c64 *>(&stream[x]), chunk, a.data());
            fo_timing_sync_work(ts, a.data(), chunk, b.data());
            fs.input_buffer.resize(chunk);
            for (size_t i = 0; i < chunk; i++) { fs.input_buffer[i].sample = std::complex<double>(b[i].sample.re, b[i].sample.im); fs.input_buffer[i].tag = (fun::vector_tag)b[i].tag; }
            fs.work();
            size_t nv = fo_fft_symbols_work(ofs, b.data(), chunk, ov.data());
            CHECK(nv == fs.output_buffer.size(), "fft_symbols: %zu vs %zu vectors", fs.output_buffer.size(), nv);
            for (size_t i = 0; i < nv && i < fs.output_buffer.size(); i++) {
                CHECK((int)fs.output_buffer[i].tag == ov[i].tag, "fft_symbols tag");
                e_fft = std::max(e_fft, rel_err(fs.output_buffer[i].samples, ov[i].samples, 64));
            }
            ce.input_buffer.swap(fs.output_buffer);
            ce.work();
            size_t ne = fo_channel_est_work(oce, ov.data(), nv, oe.data());
            CHECK(ne == ce.output_buffer.size(), "channel_est: %zu vs %zu vectors", ce.output_buffer.size(), ne);
            for (size_t i = 0; i < ne && i < ce.output_buffer.size(); i++) {
                CHECK((int)ce.output_buffer[i].tag == oe[i].tag, "channel_est tag");
                e_eq = std::max(e_eq, rel_err(ce.output_buffer[i].samples, oe[i].samples, 64, used));
            }
            pt.input_buffer.swap(ce.output_buffer);
            pt.work();
            fo_phase_tracker_work(opt, oe.data(), ne, op.data());
            for (size_t i = 0; i < ne && i < pt.output_buffer.size(); i++) {
                CHECK((int)pt.output_buffer[i].tag == op[i].tag, "phase_tracker tag");
                e_pt = std::max(e_pt, rel_err(pt.output_buffer[i].samples, op[i].samples, 48));
            }
            dec.input_buffer.swap(pt.output_buffer);
            dec.work();
            if (!dec.input_buffer.empty()) for (auto &p : dec.output_buffer) got.push_back(p);
            fs.output_buffer.clear(); ce.output_buffer.clear(); pt.output_buffer.clear();
        }
        printf("block adaptors: %zu payloads, max rel err fft %.2e eq %.2e phase %.2e\n", got.size(), e_fft, e_eq, e_pt);
        CHECK(got == want, "block-adaptor chain payloads differ (%zu vs %zu)", got.size(), want.size());
        CHECK(e_fft < 1e-12 && e_eq < 1e-9 && e_pt < 1e-9, "intermediates out of tolerance");
        fo_frame_detector_free(fd); fo_timing_sync_free(ts); fo_fft_symbols_free(ofs); fo_channel_est_free(oce); fo_phase_tracker_free(opt);
    }

    // ---- 2. receiver_chain::process_samples ----
    for (size_t cs : { (size_t)4096, (size_t)1000, (size_t)16384 }) {
        fun_amd::receiver_chain rc;
        payloads_t got;
        int first_call = -1, call = 0;
        for (size_t x = 0; x < stream.size(); x += cs, call++) {
            const size_t n = std::min(cs, stream.size() - x);
            payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + n));
            if (!r.empty() && first_call < 0) first_call = call;
            for (auto &p : r) got.push_back(p);
        }
        printf("receiver_chain chunk %zu: %zu payloads, first in call %d\n", cs, got.size(), first_call);
        CHECK(got == want, "process_samples payloads differ (chunk %zu: %zu vs %zu)", cs, got.size(), want.size());
    }
    // ---- 2b. receiver_chain in asynchronous batches: the same payloads, in order, a few calls later ----
    for (int k : { 1, 3, 8 }) {
        fun_amd::receiver_chain rc(0, k);
        rc.prepare();                                 // (the handle now, not under the first call; idempotent)
        rc.prepare();
        payloads_t got;
        int calls_with_output = 0;
        for (size_t x = 0; x < stream.size(); x += chunk) {
            payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + chunk));
            if (!r.empty()) calls_with_output++;
            for (auto &p : r) got.push_back(p);
        }
        payloads_t rest = rc.flush();
        for (auto &p : rest) got.push_back(p);
        printf("receiver_chain async every %d calls: %zu payloads (%zu from flush), %d calls returned some\n", k, got.size(), rest.size(), calls_with_output);
        CHECK(got == want, "asynchronous process_samples payloads differ (batch %d: %zu vs %zu)", k, got.size(), want.size());
    }

    // ---- 3. fun_amd::receiver: source -> process_samples -> callback on every call, pause()/resume() ----
    {
        fun_amd::vector_source src(stream);
        fun_amd::receiver rx(rx_callback, &src);
        rx.pause();                                   // returns with the loop parked between two iterations
        const int at_pause = g_rx_calls;
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        CHECK(g_rx_calls == at_pause, "receiver kept running while paused (%d -> %d calls)", at_pause, (int)g_rx_calls);
        CHECK(!rx.finished(), "receiver finished while paused");
        rx.resume();
        rx.wait_finished();
        const int expect_calls = (int)((stream.size() + 4095) / 4096) + 1;      // one call per 4096 samples + the closing silence
        printf("receiver: %d callbacks, %zu payloads\n", (int)g_rx_calls, g_rx_packets.size());
        CHECK(g_rx_calls == expect_calls, "callback count %d, expected %d (it is called after every process_samples)", (int)g_rx_calls, expect_calls);
        CHECK(g_rx_packets == want, "receiver payloads differ (%zu vs %zu)", g_rx_packets.size(), want.size());
    }
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
