// test_ordering.cpp -- device-side ordering at the C boundary (include/fun_ofdm_amd.h: foa_rx_after, foa_rx_record_consumed,
// foa_rx_record_done, foa_rx_decode_frames_dev_after, foa_rx_sync_dev_begin_after) against the oracle.
//
// A capture loop the way a host with a device-side producer runs it, with NO host synchronisation between producing the samples and
// decoding them: every round a second handle's transmit side makes a fresh noisy capture (foa_tx_channel_dev), the CALLER'S OWN
// normal-priority stream poisons the receive buffer and copies the capture into it, poisons the output slots, and records an event; the
// decode call waits for that event on the device; the caller's stream waits for "consumed" before it touches the receive buffer again,
// and a second caller stream waits for "done" before it fetches the PSDUs.  The host only ever queues.  Every round's PSDUs and results
// must equal the oracle's decode of that round's capture (fo_decode_batch_f32: fft_symbols.cpp:33-79 .. ppdu.cpp:168-295 restated).
// The library's lanes run below the normal priority and its stitch / copy streams above it, so without these calls the producer is
// overtaken in both directions; the control pass at the end (plain calls, no events) shows how often -- printed, not asserted.
//
// Build / run: tests/test_gpu_cpp_adaptors.py::test_device_side_ordering (hipcc; needs a GPU).  Exit code 0 = all checks passed.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fun_ofdm_amd.h"
extern "C" {
#include "fo_oracle.h"
}

static int failures = 0;
#define CHECK(cond, ...) do { if (!(cond)) { failures++; if (failures < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } } while (0)
#define HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { printf("%s: %s\n", #expr, hipGetErrorString(e_)); exit(2); } } while (0)
#define FOA(expr) do { int rc_ = (expr); if (rc_ != 0) { printf("%s: %d %s\n", #expr, rc_, foa_last_error()); exit(2); } } while (0)

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned)(rng_state >> 32); }

static const int kSlot = 1024, kFrames = 48, kPool = 4, kLead = 300;

struct Capture {                       // one pool entry: a set of frames, what the oracle decodes from its noisy capture
    int rate, length;
    size_t frame_samples, pitch, n_samples, m;            // m: alignments the pre-sync finds
    double *d_frames;
    foa_frame_desc *d_descs; int64_t *d_ends;
    uint64_t seed;
    std::vector<foa_frame_desc> descs; std::vector<int64_t> ends;
    std::vector<uint8_t> want_psdu; std::vector<foa_frame_result> want_res;
};

int main()
{
    static_assert(sizeof(foa_frame_desc) == sizeof(fo_frame_desc) && sizeof(foa_frame_result) == sizeof(fo_frame_result), "descriptor layouts");
    foa_rx *rx = nullptr, *gen = nullptr;
    FOA(foa_rx_create(&rx, 0));
    FOA(foa_rx_create(&gen, 0));
    FOA(foa_rx_set_option(gen, "pipeline", 0));                        // the producer: everything on its one stream, foa_rx_stream(gen)
    hipStream_t gs = (hipStream_t)foa_rx_stream(gen), cs, cs2;
    HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));          // the caller's own streams: NORMAL priority
    HIP(hipStreamCreateWithFlags(&cs2, hipStreamNonBlocking));

    // ---- the pool: four captures of different rates / lengths, their descriptors and the oracle's answers ----
    const int rates[kPool] = { 10, 5, 0, 8 }, lens[kPool] = { 1024, 700, 180, 1000 };
    Capture cap[kPool];
    size_t max_samples = 0;
    const size_t dcap = 4096;
    for (int p = 0; p < kPool; p++) {
        Capture &c = cap[p];
        c.rate = rates[p]; c.length = lens[p]; c.seed = 1000 + 17 * p;
        std::vector<uint8_t> pay((size_t)kFrames * c.length);
        for (auto &b : pay) b = (uint8_t)rnd();
        uint8_t *d_pay;
        HIP(hipMalloc((void **)&d_pay, pay.size()));
        HIP(hipMemcpy(d_pay, pay.data(), pay.size(), hipMemcpyHostToDevice));
        FOA(foa_tx_build_frames_dev(gen, nullptr, 0, c.length, c.rate, 0, nullptr, &c.frame_samples));
        c.pitch = c.frame_samples + 2 * kLead + 37 * p;
        c.n_samples = (size_t)kFrames * c.pitch;
        max_samples = c.n_samples > max_samples ? c.n_samples : max_samples;
        HIP(hipMalloc((void **)&c.d_frames, (size_t)kFrames * c.frame_samples * 16));
        FOA(foa_tx_build_frames_dev(gen, d_pay, (size_t)c.length, c.length, c.rate, kFrames, c.d_frames, &c.frame_samples));
        FOA(foa_rx_sync(gen));
        HIP(hipFree(d_pay));
        HIP(hipMalloc((void **)&c.d_descs, dcap * sizeof(foa_frame_desc)));
        HIP(hipMalloc((void **)&c.d_ends, dcap * 8));
    }
    float *d_iq, *d_gen[2];
    HIP(hipMalloc((void **)&d_iq, max_samples * 8));
    for (auto &g : d_gen) HIP(hipMalloc((void **)&g, max_samples * 8));
    for (int p = 0; p < kPool; p++) {
        Capture &c = cap[p];
        FOA(foa_tx_channel_dev(gen, c.d_frames, kFrames, c.frame_samples, c.pitch, kLead, 22.0, 0.0, c.seed, d_gen[0]));
        FOA(foa_rx_sync(gen));
        FOA(foa_rx_sync_dev(rx, d_gen[0], c.n_samples, c.d_descs, c.d_ends, dcap, &c.m));
        CHECK(c.m >= (size_t)kFrames - 4 && c.m < 4 * (size_t)kFrames, "capture %d: %zu alignments", p, c.m);      // (the detector may miss a frame at 22 dB, and tag noise)
        std::vector<float> iq(2 * c.n_samples);
        c.descs.resize(c.m); c.ends.resize(c.m);
        HIP(hipMemcpy(iq.data(), d_gen[0], c.n_samples * 8, hipMemcpyDeviceToHost));
        HIP(hipMemcpy(c.descs.data(), c.d_descs, c.m * sizeof(foa_frame_desc), hipMemcpyDeviceToHost));
        HIP(hipMemcpy(c.ends.data(), c.d_ends, c.m * 8, hipMemcpyDeviceToHost));
        c.want_psdu.assign(c.m * kSlot, 0); c.want_res.resize(c.m);
        fo_decode_batch_f32(iq.data(), (int64_t)c.n_samples, (const fo_frame_desc *)c.descs.data(), c.ends.data(), c.m, c.want_psdu.data(), kSlot,
                            (fo_frame_result *)c.want_res.data(), 8);
        size_t ok = 0;
        for (auto &r : c.want_res) ok += r.status == FOA_ST_OK;
        CHECK(ok >= (size_t)kFrames - 8, "capture %d: the oracle decodes %zu of %d frames", p, ok, kFrames);
    }
    size_t max_m = 0;
    for (auto &c : cap) max_m = c.m > max_m ? c.m : max_m;

    uint8_t *d_psdu[3]; foa_frame_result *d_res[3];
    for (int i = 0; i < 3; i++) { HIP(hipMalloc((void **)&d_psdu[i], max_m * kSlot)); HIP(hipMalloc((void **)&d_res[i], max_m * sizeof(foa_frame_result))); }
    hipEvent_t eg, ready, consumed, copied[2], done[3], fetched[3];
    HIP(hipEventCreateWithFlags(&eg, hipEventDisableTiming)); HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    HIP(hipEventCreateWithFlags(&consumed, hipEventDisableTiming));
    for (auto &e : copied) HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : done) HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &e : fetched) HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));

    // mode 0: pipelined calls (the default) with the events; 1: calls in line (option "pipeline" 0) with the events; 2: the pre-sync inside
    // the loop as well (foa_rx_sync_dev_begin_after; its _end is the library's own wait for the count); 3: the control -- plain calls, no events
    const int rounds_of[4] = { 200, 60, 60, 60 };
    for (int mode = 0; mode < 4; mode++) {
        const int R = rounds_of[mode];
        FOA(foa_rx_sync(rx));
        HIP(hipDeviceSynchronize());
        FOA(foa_rx_set_option(rx, "pipeline", mode == 1 ? 0 : 1));
        uint8_t *h_psdu; foa_frame_result *h_res;
        HIP(hipHostMalloc((void **)&h_psdu, (size_t)R * max_m * kSlot, hipHostMallocDefault));
        HIP(hipHostMalloc((void **)&h_res, (size_t)R * max_m * sizeof(foa_frame_result), hipHostMallocDefault));
        memset(h_psdu, 0x55, (size_t)R * max_m * kSlot);
        memset(h_res, 0x55, (size_t)R * max_m * sizeof(foa_frame_result));
        std::vector<int> which(R);
        std::vector<size_t> found(R, 0);
        foa_frame_desc *d_descs2; int64_t *d_ends2;
        HIP(hipMalloc((void **)&d_descs2, dcap * sizeof(foa_frame_desc)));
        HIP(hipMalloc((void **)&d_ends2, dcap * 8));
        for (int r = 0; r < R; r++) {
            const int p = (int)(rnd() % kPool), g = r % 2, o = r % 3;
            which[r] = p;
            Capture &c = cap[p];
            // the producer: a fresh capture (same seed as the pool entry: the same samples, made again) on the second handle's stream
            if (r >= 2) HIP(hipStreamWaitEvent(gs, copied[g], 0));
            FOA(foa_tx_channel_dev(gen, c.d_frames, kFrames, c.frame_samples, c.pitch, kLead, 22.0, 0.0, c.seed, d_gen[g]));
            HIP(hipEventRecord(eg, gs));
            // the caller's stream: poison, then fill, the receive buffer and the output slots
            HIP(hipStreamWaitEvent(cs, eg, 0));
            if (r >= 1 && mode != 3) HIP(hipStreamWaitEvent(cs, consumed, 0));
            HIP(hipMemsetAsync(d_iq, 0xFF, max_samples * 8, cs));
            HIP(hipMemcpyAsync(d_iq, d_gen[g], c.n_samples * 8, hipMemcpyDeviceToDevice, cs));
            HIP(hipEventRecord(copied[g], cs));
            if (r >= 3 && mode != 3) HIP(hipStreamWaitEvent(cs, fetched[o], 0));
            HIP(hipMemsetAsync(d_psdu[o], 0xEE, max_m * kSlot, cs));
            HIP(hipMemsetAsync(d_res[o], 0xEE, max_m * sizeof(foa_frame_result), cs));
            HIP(hipEventRecord(ready, cs));
            // the decode: waits for `ready` on the device
            const foa_frame_desc *dd = c.d_descs; const int64_t *de = c.d_ends;
            size_t m = c.m;
            if (mode == 2) {
                FOA(foa_rx_sync_dev_begin_after(rx, ready, d_iq, c.n_samples, d_descs2, d_ends2, dcap));
                FOA(foa_rx_sync_dev_end(rx, &m));
                found[r] = m;
                dd = d_descs2; de = d_ends2;
                FOA(foa_rx_decode_frames_dev(rx, d_iq, c.n_samples, dd, de, m, d_psdu[o], kSlot, d_res[o]));     // (its inputs are complete: the pre-sync read them)
            } else if (mode == 3) {
                FOA(foa_rx_decode_frames_dev(rx, d_iq, c.n_samples, dd, de, m, d_psdu[o], kSlot, d_res[o]));
            } else {
                FOA(foa_rx_decode_frames_dev_after(rx, ready, d_iq, c.n_samples, dd, de, m, d_psdu[o], kSlot, d_res[o]));
            }
            if (mode != 3) {
                FOA(foa_rx_record_consumed(rx, consumed));
                FOA(foa_rx_record_done(rx, done[o]));
                HIP(hipStreamWaitEvent(cs2, done[o], 0));
            }
            const size_t mm = m < max_m ? m : max_m;
            HIP(hipMemcpyAsync(h_psdu + (size_t)r * max_m * kSlot, d_psdu[o], mm * kSlot, hipMemcpyDeviceToHost, cs2));
            HIP(hipMemcpyAsync(h_res + (size_t)r * max_m, d_res[o], mm * sizeof(foa_frame_result), hipMemcpyDeviceToHost, cs2));
            HIP(hipEventRecord(fetched[o], cs2));
        }
        FOA(foa_rx_sync(rx));
        HIP(hipDeviceSynchronize());
        int bad_rounds = 0;
        for (int r = 0; r < R; r++) {
            const Capture &c = cap[which[r]];
            bool same = mode != 2 || found[r] == c.m;
            same = same && memcmp(h_res + (size_t)r * max_m, c.want_res.data(), c.m * sizeof(foa_frame_result)) == 0;
            for (size_t a = 0; same && a < c.m; a++)
                if (c.want_res[a].status == FOA_ST_OK) same = memcmp(h_psdu + ((size_t)r * max_m + a) * kSlot, c.want_psdu.data() + a * kSlot, (size_t)c.want_res[a].length) == 0;
            bad_rounds += !same;
        }
        const char *names[4] = { "pipelined calls, device-side events", "calls in line, device-side events", "pre-sync + decode per round, device-side events",
                                 "control: plain calls, nothing orders the caller's streams" };
        printf("%-62s %3d rounds, %d differ from the oracle\n", names[mode], R, bad_rounds);
        if (mode != 3) CHECK(bad_rounds == 0, "mode %d: %d of %d rounds differ", mode, bad_rounds, R);
        HIP(hipHostFree(h_psdu)); HIP(hipHostFree(h_res));
        HIP(hipFree(d_descs2)); HIP(hipFree(d_ends2));
    }
    // argument checks
    CHECK(foa_rx_after(rx, nullptr) == FOA_E_INVALID && foa_rx_record_done(nullptr, ready) == FOA_E_INVALID && foa_rx_record_consumed(rx, nullptr) == FOA_E_INVALID,
          "NULL arguments must be refused");
    foa_rx_destroy(rx);
    foa_rx_destroy(gen);
    printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
    return failures ? 1 : 0;
}
