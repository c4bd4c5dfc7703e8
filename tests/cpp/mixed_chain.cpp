// mixed_chain.cpp -- the reference's OWN frame_detector and timing_sync in front of this repository's blocks, wired the way
// fun::receiver_chain wires its blocks: one thread per block and call, every block on another generation of the data, the buffers
// of neighbours swapped between calls (src/receiver_chain.cpp:29-51 add_block, :78-95 run_block, :106-126 process_samples).
//
//   chain A   fun::frame_detector -> fun::timing_sync -> fun_amd::fft_symbols -> fun_amd::channel_est -> fun_amd::phase_tracker
//             -> fun_amd::frame_decoder                                 (four per-stage adaptors, include/fun_ofdm_amd/blocks.hpp)
//   chain B   fun::frame_detector -> fun::timing_sync -> fun_amd::rx_backend        (the fused stage block: one device call per work())
//
// Both must return, call by call concatenated, the ordered payload list of the oracle's receiver_chain on the same stream.
// Two builds of this file:
//   -DFOA_REFERENCE_HEADERS -std=c++11 -I/root/reference/src   (dev container): the reference's block.h, tagged_vector.h,
//        frame_detector.h and timing_sync.h are included FIRST, so blocks.hpp uses the reference's types as they are -- the documented
//        integration mode -- and the two pre-sync blocks are the reference's classes themselves (objects of
//        oracle/_ref/libfun_ofdm_ref.so).  Linked against tests/cpp/stub_abi.cpp (no GPU there: the C ABI answered by the oracle).
//   without (the GPU box, where the reference tree does not exist): blocks.hpp's layout-identical declarations; the two pre-sync blocks are
//        the same compiled reference objects behind the C handles of oracle/ref_capi.cpp, wrapped as fun::block-s.  Linked against
//        the real libfun_ofdm_amd.so.
// TEST INFRASTRUCTURE (uses the oracle and the partial reference build).  Exit code 0 = all checks passed.
#ifdef FOA_REFERENCE_HEADERS
#include "block.h"
#include "tagged_vector.h"
#include "frame_detector.h"
#include "timing_sync.h"
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "fun_ofdm_amd/blocks.hpp"
extern "C" {
#include "fo_oracle.h"
}

typedef std::vector<std::vector<unsigned char> > payloads_t;
typedef std::complex<double> cd;
static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

// the adaptors ARE blocks of the plug-in contract in force (the reference's own when its headers came first)
static_assert(std::is_base_of<fun::block_base, fun_amd::fft_symbols>::value && std::is_base_of<fun::block<fun::tagged_sample, fun::tagged_vector<64> >, fun_amd::fft_symbols>::value,
              "fun_amd::fft_symbols is a fun::block<tagged_sample, tagged_vector<64>>");
static_assert(std::is_base_of<fun::block<fun::tagged_vector<48>, std::vector<unsigned char> >, fun_amd::frame_decoder>::value, "fun_amd::frame_decoder");
static_assert(std::is_base_of<fun::block<fun::tagged_sample, std::vector<unsigned char> >, fun_amd::rx_backend>::value, "fun_amd::rx_backend");
static_assert(sizeof(fun::tagged_sample) == 24 && sizeof(fun::tagged_vector<64>) == 1032 && sizeof(fun::tagged_vector<48>) == 776, "src/tagged_vector.h layouts");

#ifdef FOA_REFERENCE_HEADERS
typedef fun::frame_detector ref_frame_detector;
typedef fun::timing_sync ref_timing_sync;
#else
// the compiled reference blocks behind oracle/ref_capi.cpp's handles
extern "C" {
void *ref_frame_detector_new();
void ref_frame_detector_free(void *);
void ref_frame_detector_work(void *, const cd *in, size_t n, fun::tagged_sample *out);
void *ref_timing_sync_new();
void ref_timing_sync_free(void *);
void ref_timing_sync_work(void *, const fun::tagged_sample *in, size_t n, fun::tagged_sample *out);
int ref_sizeof_tagged_sample();
}
class ref_frame_detector : public fun::block<cd, fun::tagged_sample> {
public:
    ref_frame_detector() : block("frame_detector"), h_(ref_frame_detector_new()) {}
    ~ref_frame_detector() { ref_frame_detector_free(h_); }
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(input_buffer.size());
        ref_frame_detector_work(h_, input_buffer.data(), input_buffer.size(), output_buffer.data());
    }
private:
    void *h_;
};
class ref_timing_sync : public fun::block<fun::tagged_sample, fun::tagged_sample> {
public:
    ref_timing_sync() : block("timing_sync"), h_(ref_timing_sync_new()) {}
    ~ref_timing_sync() { ref_timing_sync_free(h_); }
    virtual void work()
    {
        if (input_buffer.size() == 0) return;
        output_buffer.resize(input_buffer.size());
        ref_timing_sync_work(h_, input_buffer.data(), input_buffer.size(), output_buffer.data());
    }
private:
    void *h_;
};
#endif

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (double)(rng_state >> 11) / 9007199254740992.0; }
static double nrand() { double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

// run every block's work() on a thread of its own, like receiver_chain::run_block does between its semaphores
static void run_all(const std::vector<fun::block_base *> &blocks)
{
    std::vector<std::thread> th;
    for (size_t i = 0; i < blocks.size(); i++) th.push_back(std::thread([&blocks, i] { blocks[i]->work(); }));
    for (size_t i = 0; i < th.size(); i++) th[i].join();
}

int main()
{
    // ---- a stream: mixed rates and lengths, a carrier frequency offset per frame, AWGN 25 dB, all values float-representable ----
    const int rates[] = { 10, 0, 5, 8, 3, 9, 10, 2, 6, 10 };
    const int lens[] = { 1024, 100, 300, 1500, 57, 700, 33, 211, 480, 64 };
    std::vector<cd> stream;
    payloads_t sent;
    for (int k = 0; k < 10; k++) {
        std::vector<unsigned char> pay((size_t)lens[k]);
        for (size_t b = 0; b < pay.size(); b++) pay[b] = (unsigned char)(urand() * 256);
        std::vector<fo_c64> fr(fo_frame_samples(rates[k], lens[k]));
        fo_build_frame(pay.data(), lens[k], rates[k], fr.data());
        const double ph = urand() * 6.283185307179586, cfo = (2.0 * urand() - 1.0) * 4000.0;
        stream.insert(stream.end(), (size_t)(k == 4 ? 0 : 150 + urand() * 500), cd(0, 0));       // (one pair of frames back to back)
        for (size_t i = 0; i < fr.size(); i++) {
            const double a = ph + 6.283185307179586 * cfo * (double)i / 20e6;
            stream.push_back(cd(fr[i].re, fr[i].im) * cd(cos(a), sin(a)));
        }
        sent.push_back(pay);
    }
    stream.insert(stream.end(), 700, cd(0, 0));
    const double sigma = sqrt(0.0124 / 2 / pow(10.0, 2.5));
    for (size_t i = 0; i < stream.size(); i++) stream[i] = cd((float)(stream[i].real() + sigma * nrand()), (float)(stream[i].imag() + sigma * nrand()));
    const size_t chunk = 4096;                                  // receiver.h:16 NUM_RX_SAMPLES
    while (stream.size() % chunk) stream.push_back(cd(0, 0));
    for (int i = 0; i < 8 * (int)chunk; i++) stream.push_back(cd(0, 0));     // lets the chains' call latency run out

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
    CHECK(want.size() >= 8, "oracle decoded only %zu of 10 frames", want.size());

    // ---- chain A: the four per-stage adaptors behind the reference's pre-sync blocks ----
    {
        ref_frame_detector fd; ref_timing_sync ts;
        fun_amd::fft_symbols fs; fun_amd::channel_est ce; fun_amd::phase_tracker pt; fun_amd::frame_decoder dec;
        std::vector<fun::block_base *> blocks;                  // receiver_chain.cpp:45-50, in this order
        blocks.push_back(&fd); blocks.push_back(&ts); blocks.push_back(&fs); blocks.push_back(&ce); blocks.push_back(&pt); blocks.push_back(&dec);
        payloads_t got;
        for (size_t x = 0; x < stream.size(); x += chunk) {
            std::vector<cd> samples(stream.begin() + x, stream.begin() + x + chunk);
            fd.input_buffer.swap(samples);                      // receiver_chain.cpp:109
            run_all(blocks);
            ts.input_buffer.swap(fd.output_buffer);             // receiver_chain.cpp:118-122
            fs.input_buffer.swap(ts.output_buffer);
            ce.input_buffer.swap(fs.output_buffer);
            pt.input_buffer.swap(ce.output_buffer);
            dec.input_buffer.swap(pt.output_buffer);
            // (a block whose input was empty leaves its output_buffer as it was, frame_decoder.cpp:47-48: take what a call really produced)
            if (!dec.output_buffer.empty()) { for (size_t i = 0; i < dec.output_buffer.size(); i++) got.push_back(dec.output_buffer[i]); dec.output_buffer.clear(); }
        }
        printf("chain A (reference pre-sync + four adaptors): %zu payloads, oracle chain %zu\n", got.size(), want.size());
        CHECK(got == want, "chain A payload list differs (%zu vs %zu)", got.size(), want.size());
    }
    // ---- chain B: the fused stage block behind the reference's pre-sync blocks ----
    {
        ref_frame_detector fd; ref_timing_sync ts;
        fun_amd::rx_backend be;
        std::vector<fun::block_base *> blocks;
        blocks.push_back(&fd); blocks.push_back(&ts); blocks.push_back(&be);
        payloads_t got;
        int calls_with_output = 0;
        for (size_t x = 0; x < stream.size(); x += chunk) {
            std::vector<cd> samples(stream.begin() + x, stream.begin() + x + chunk);
            fd.input_buffer.swap(samples);
            run_all(blocks);
            ts.input_buffer.swap(fd.output_buffer);
            be.input_buffer.swap(ts.output_buffer);
            if (!be.output_buffer.empty()) { calls_with_output++; for (size_t i = 0; i < be.output_buffer.size(); i++) got.push_back(be.output_buffer[i]); be.output_buffer.clear(); }
        }
        printf("chain B (reference pre-sync + fun_amd::rx_backend): %zu payloads in %d calls, oracle chain %zu\n", got.size(), calls_with_output, want.size());
        CHECK(got == want, "chain B payload list differs (%zu vs %zu)", got.size(), want.size());
    }
    // ---- and fun_amd::receiver_chain (same signature as fun::receiver_chain::process_samples) under the same types ----
    for (int mode = 0; mode < 2; mode++) {
        fun_amd::receiver_chain rc(0, mode == 0 ? 0 : 3);
        payloads_t got;
        for (size_t x = 0; x < stream.size(); x += chunk) {
            payloads_t r = rc.process_samples(std::vector<cd>(stream.begin() + x, stream.begin() + x + chunk));
            for (size_t i = 0; i < r.size(); i++) got.push_back(r[i]);
        }
        payloads_t rest = rc.flush();
        for (size_t i = 0; i < rest.size(); i++) got.push_back(rest[i]);
        CHECK(got == want, "fun_amd::receiver_chain (%s) payload list differs (%zu vs %zu)", mode ? "asynchronous" : "synchronous", got.size(), want.size());
    }
#ifdef FOA_REFERENCE_HEADERS
    printf("built against the reference's own block.h / tagged_vector.h / frame_detector.h / timing_sync.h (BUFFER_MAX %d)\n", (int)BUFFER_MAX);
#else
    CHECK(ref_sizeof_tagged_sample() == (int)sizeof(fun::tagged_sample), "the compiled reference's tagged_sample is %d bytes", ref_sizeof_tagged_sample());
#endif
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
