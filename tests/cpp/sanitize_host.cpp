// sanitize_host.cpp -- the host side of the drop-in (blocks.hpp: receiver_chain in both modes, receiver + sources, and the
// streaming pre-sync of sync_host.h) and the oracle's threaded chain, run on the CPU under ASan+UBSan / TSan.
// Linked against tests/cpp/stub_abi.cpp instead of the GPU library.  tools/run_sanitizers.sh builds and runs it.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "fun_ofdm_amd/blocks.hpp"
extern "C" {
#include "fo_oracle.h"
}

typedef std::vector<std::vector<unsigned char> > payloads_t;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

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

int main()
{
    // mixed rates and lengths, gaps from zero to a few hundred samples, AWGN 25 dB, float-representable values
    const int rates[] = { 10, 0, 5, 8, 3, 9, 10, 2, 6, 1, 4, 7 };
    const int lens[] = { 1024, 100, 300, 1500, 57, 700, 33, 211, 0, 4095, 1, 640 };
    std::vector<std::complex<double> > stream(300);
    for (int k = 0; k < 12; k++) {
        std::vector<unsigned char> pay(lens[k] ? lens[k] : 1);
        for (auto &b : pay) b = (unsigned char)(urand() * 256);
        std::vector<fo_c64> fr(fo_frame_samples(rates[k], lens[k]));
        fo_build_frame(pay.data(), lens[k], rates[k], fr.data());
        const double ph = urand() * 6.283185307179586;
        for (auto &x : fr) stream.push_back(std::complex<double>(x.re, x.im) * std::complex<double>(cos(ph), sin(ph)));
        stream.resize(stream.size() + (size_t)(urand() * 3) * 250);
    }
    stream.resize(stream.size() + 700);
    const double sigma = sqrt(0.0124 / 2 / pow(10.0, 2.5));
    for (auto &x : stream) x = std::complex<double>((float)(x.real() + sigma * nrand()), (float)(x.imag() + sigma * nrand()));

    // what the reference-shaped chain delivers: the oracle's receiver_chain, six block threads + caller (TSan sees them)
    payloads_t want;
    {
        fo_receiver_chain *rc = fo_receiver_chain_new_threaded();
        std::vector<fo_c64> chunk(4096);
        for (size_t x = 0; x < stream.size() + 7 * 4096; x += 4096) {
            for (size_t i = 0; i < 4096; i++) {
                const std::complex<double> v = x + i < stream.size() ? stream[x + i] : std::complex<double>(0, 0);
                chunk[i].re = v.real(); chunk[i].im = v.imag();
            }
            const fo_payloads *p = fo_receiver_chain_process_samples(rc, chunk.data(), 4096);
            for (size_t k = 0; k < fo_payloads_count(p); k++) want.push_back(std::vector<unsigned char>(fo_payloads_data(p, k), fo_payloads_data(p, k) + fo_payloads_len(p, k)));
        }
        fo_receiver_chain_free(rc);
    }
    printf("oracle chain (threaded): %zu payloads from %zu samples\n", want.size(), stream.size());
    CHECK(want.size() >= 9, "the stream should deliver most of its 12 frames");

    for (size_t cs : { (size_t)4096, (size_t)1000, (size_t)177, (size_t)16384 }) {
        fun_amd::receiver_chain rc;
        payloads_t got;
        for (size_t x = 0; x < stream.size(); x += cs) {
            const size_t n = std::min(cs, stream.size() - x);
            payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + n));
            for (auto &p : r) got.push_back(p);
        }
        CHECK(got == want, "process_samples payloads differ (chunk %zu: %zu vs %zu)", cs, got.size(), want.size());
    }
    for (int k : { 1, 3, 8 }) {
        fun_amd::receiver_chain rc(0, k);
        rc.set_reference_call_size(4096);
        payloads_t got;
        for (size_t x = 0; x < stream.size(); x += 4096) {
            const size_t n = std::min((size_t)4096, stream.size() - x);
            payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + n));
            for (auto &p : r) got.push_back(p);
        }
        payloads_t rest = rc.flush();
        for (auto &p : rest) got.push_back(p);
        CHECK(got == want, "asynchronous process_samples payloads differ (batch %d: %zu vs %zu)", k, got.size(), want.size());
    }
    {   // device mode of the chain (the engine behind foa_stream_* is the stub's; the wrapper's buffers and loops are real)
        fun_amd::receiver_chain rc(0, 0, 65536, 2);
        payloads_t got;
        for (size_t x = 0; x < stream.size(); x += 4096) {
            const size_t n = std::min((size_t)4096, stream.size() - x);
            payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + n));
            for (auto &p : r) got.push_back(p);
        }
        payloads_t rest = rc.flush();
        for (auto &p : rest) got.push_back(p);
        CHECK(got == want, "device-mode process_samples payloads differ (%zu vs %zu)", got.size(), want.size());
    }
    {   // ... and its device-list mode (foa_shard_* behind it), a second stream after a flush included
        fun_amd::receiver_chain rc(std::vector<int>{ 0, 0, 1 }, 65536, 2);
        for (int round = 0; round < 2; round++) {
            payloads_t got;
            for (size_t x = 0; x < stream.size(); x += 10000) {
                const size_t n = std::min((size_t)10000, stream.size() - x);
                payloads_t r = rc.process_samples(std::vector<std::complex<double> >(stream.begin() + x, stream.begin() + x + n));
                for (auto &p : r) got.push_back(p);
            }
            payloads_t rest = rc.flush();
            for (auto &p : rest) got.push_back(p);
            CHECK(got == want, "device-list process_samples payloads differ (round %d: %zu vs %zu)", round, got.size(), want.size());
        }
    }
    for (int async_calls : { 0, 4 }) {
        g_rx_packets.clear();
        g_rx_calls = 0;
        fun_amd::vector_source src(stream);
        fun_amd::receiver rx(rx_callback, &src, 0, 4096, async_calls);
        rx.pause();
        const int at_pause = g_rx_calls;
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        CHECK(g_rx_calls == at_pause, "receiver kept running while paused");
        rx.resume();
        rx.wait_finished();
        CHECK(g_rx_packets == want, "receiver payloads differ (async %d: %zu vs %zu)", async_calls, g_rx_packets.size(), want.size());
    }
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
