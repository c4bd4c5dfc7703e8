// foa_sim -- offline receiver over a raw I/Q file (the shape of the reference's examples/test_sim.cpp
// with the simulated channel replaced by a capture).
//
//   foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--device-batch B [--narrow-threads T] [--devices D0,D1,..]]
//                     [--preload [--warm-batches W] [--pace MSPS] [--latency PITCH LEAD SAMPLES]] [--out FILE]
//
// Feeds the file through fun_amd::receiver (sample source -> receiver_chain::process_samples -> callback) in chunks of N
// samples (default 4096, the reference's NUM_RX_SAMPLES) and writes every received PSDU to FILE (default: stdout summary
// only) as a record: 4-byte little-endian length, then the bytes.  --async K: decode in asynchronous batches submitted
// every K calls (fun_amd::receiver_chain's streaming mode) instead of synchronously in every call.  --device-batch B: the
// whole of process_samples() on the device in batches of B samples (pre-sync kernels included; the mode for rates far
// above real time), T helper threads narrowing large calls to float; --devices D0,D1,..: those batches dealt over several devices (batch k on
// the (k mod n)-th of the list; a device may be listed twice), payloads in stream order all the same.  --preload: read the capture into memory first (as
// the complex<double> chunks process_samples() takes) and time the receive loop alone: "x.y Msamples/s through
// process_samples" is then the rate of the drop-in API itself, without the file read and the float -> double widening of
// this program's own source.  With --preload: --warm-batches W (default 2 in device mode) feeds W batches of silence before the clock
// starts, so that the timed loop meets a running pipeline (threads on their cores, every buffer touched) rather than a cold one;
// --pace MSPS hands the chunks over at MSPS million samples per second of wall-clock time instead of as fast as they are taken (a
// radio's pace: 20; the engine's latency under a given load); --latency PITCH LEAD SAMPLES reports how long after the call that
// delivered a frame's last sample its payload came back (frame k of the capture occupies samples [k*PITCH + LEAD, k*PITCH + LEAD +
// SAMPLES) and carries k in its first four payload bytes, little-endian: tools/bench_latency.py builds such captures).  --longest N (device
// mode, --preload): the longest frame the stream will hold, in samples + 192 (option "stream_longest"): every batch then re-synchronises
// N + 2048 samples of carry instead of 112 640 -- most of a small batch's cost (a frame's latency no longer depends on it: a frame is
// decoded by the first batch that holds its last sample).
//
// build:  g++ -O2 -std=c++17 examples/foa_sim.cpp -Iinclude -Lfun_ofdm_amd/csrc -lfun_ofdm_amd -lpthread -o foa_sim
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

#include <fun_ofdm_amd/blocks.hpp>
#if defined(__GLIBC__)
#include <malloc.h>
#endif

static std::FILE *g_out = nullptr;
static size_t g_packets = 0, g_bytes = 0, g_calls = 0;

// --latency: call_time[c] = when the c-th timed process_samples() call was made; a frame's latency = now - the time of the call
// that delivered its last sample
static std::vector<double> g_call_time;
static std::vector<double> g_latency_ms;
static long long g_lat_pitch = 0, g_lat_lead = 0, g_lat_samples = 0;
static int g_lat_chunk = 0;
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void on_packets(std::vector<std::vector<unsigned char> > packets)
{
    g_calls++;
    const double t = g_lat_pitch ? now_s() : 0.0;
    for (const auto &p : packets) {
        if (g_lat_pitch && p.size() >= 4) {
            const long long k = (long long)p[0] | (long long)p[1] << 8 | (long long)p[2] << 16 | (long long)p[3] << 24;
            const long long last = k * g_lat_pitch + g_lat_lead + g_lat_samples - 1;
            const size_t call = (size_t)(last / g_lat_chunk);
            if (call < g_call_time.size()) g_latency_ms.push_back((t - g_call_time[call]) * 1e3);
        }
        g_packets++;
        g_bytes += p.size();
        if (g_out) {
            const unsigned n = (unsigned)p.size();
            const unsigned char len[4] = { (unsigned char)n, (unsigned char)(n >> 8), (unsigned char)(n >> 16), (unsigned char)(n >> 24) };
            std::fwrite(len, 1, 4, g_out);
            std::fwrite(p.data(), 1, p.size(), g_out);
        }
    }
}

int main(int argc, char **argv)
{
#if defined(__GLIBC__)
    // A caller that hands process_samples() megabyte-sized vectors at gigabytes per second must not have its allocator give every one of
    // them back to the kernel: glibc serves requests above its mmap threshold (128 KB until the first such block is freed, then whatever
    // that block's size was) with a mapping of their own, and freeing 3.5 GB of those costs 0.2-0.3 s of munmap -- page by page, under the
    // process's mmap lock -- which is where calls of >= 65 536 samples spent their time in round 3 (0.64-0.70 Gsample/s against 4 with
    // calls of 4096 samples, whose 64 KB vectors come from the heap).  A long-running receiver reaches the same state by itself (the
    // threshold adapts to the sizes it frees); --preload allocates every chunk before it frees the first, so it says so up front.
    mallopt(M_MMAP_THRESHOLD, 32 * 1024 * 1024);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
#endif
    std::string path, format = "fc32", out;
    int chunk = 4096, device = 0, async_calls = 0, narrow_threads = 0;
    size_t device_batch = 0;
    bool preload = false;
    std::vector<int> devices;                   // --devices 0,1,..: device mode over several devices (batch k on devices[k mod n])
    int warm_batches = -1;
    long long longest = 0;                      // --longest N: option "stream_longest" (device mode: the latency knob)
    double pace_msps = 0.0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--format" && i + 1 < argc) format = argv[++i];
        else if (a == "--chunk" && i + 1 < argc) chunk = std::atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--async" && i + 1 < argc) async_calls = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (a == "--device-batch" && i + 1 < argc) device_batch = (size_t)std::atoll(argv[++i]);
        else if (a == "--narrow-threads" && i + 1 < argc) narrow_threads = std::atoi(argv[++i]);
        else if (a == "--preload") preload = true;
        else if (a == "--devices" && i + 1 < argc) { for (const char *q = argv[++i]; *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; } }
        else if (a == "--warm-batches" && i + 1 < argc) warm_batches = std::atoi(argv[++i]);
        else if (a == "--pace" && i + 1 < argc) pace_msps = std::atof(argv[++i]);
        else if (a == "--longest" && i + 1 < argc) longest = std::atoll(argv[++i]);
        else if (a == "--latency" && i + 3 < argc) { g_lat_pitch = std::atoll(argv[++i]); g_lat_lead = std::atoll(argv[++i]); g_lat_samples = std::atoll(argv[++i]); }
        else if (path.empty() && a[0] != '-') path = a;
        else { std::fprintf(stderr, "usage: foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--device-batch B] [--narrow-threads T] [--preload] [--out FILE]\n"); return 2; }
    }
    if (path.empty() || chunk <= 0) { std::fprintf(stderr, "usage: foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--out FILE]\n"); return 2; }
    try {
        if (!out.empty()) {
            g_out = std::fopen(out.c_str(), "wb");
            if (!g_out) { std::fprintf(stderr, "cannot open %s\n", out.c_str()); return 1; }
        }
        fun_amd::file_source src(path, format);
        if (preload) {
            // the caller's side of the boundary, prepared up front: one complex<double> vector per process_samples() call
            std::vector<std::vector<std::complex<double> > > chunks;
            std::vector<std::complex<double> > buf;
            size_t total = 0;
            while (src.get_samples(chunk, buf)) { chunks.push_back(buf); total += buf.size(); }
            std::unique_ptr<fun_amd::receiver_chain> chain_p(devices.empty() ? new fun_amd::receiver_chain(device, async_calls, device_batch, narrow_threads)
                                                                             : new fun_amd::receiver_chain(devices, device_batch, narrow_threads));
            fun_amd::receiver_chain &chain = *chain_p;
            if (longest > 0) chain.set_stream_longest(longest);
            chain.prepare();                                                           // the handle and the engine come into being outside the timed loop
            chain.process_samples(std::vector<std::complex<double> >(512));            // (and a first call: 512 samples of silence)
            // ... and the engine's pipeline: W batches' worth of the capture's own first samples (a copy), then silence until their payloads
            // have all come back, before the clock starts -- so that the timed loop meets threads that are on their cores, buffers that have been
            // touched and kernels that have been launched before (silence alone decodes nothing: the first real batches then paid ~15 ms of
            // first-launch costs inside the timed region; single runs varied 3.1-4.4 Gsample/s cold).  At least two batches and 8 Mi samples:
            // small batches go round four lanes, twelve staging slots and six work sets.
            if (warm_batches < 0) warm_batches = device_batch > 0 ? (int)std::min<size_t>(128, std::max<size_t>(2, ((size_t)8 << 20) / device_batch)) : 0;
            if (warm_batches > 0 && device_batch > 0 && !chunks.empty()) {
                const size_t warm = std::min((size_t)warm_batches * device_batch, total);
                size_t fed_w = 0;
                for (size_t c = 0; fed_w < warm && c < chunks.size(); c++) { fed_w += chunks[c].size(); chain.process_samples(std::vector<std::complex<double> >(chunks[c])); }
                const size_t quiet = 131072 + 2 * device_batch;      // past the longest frame and through the batches in flight
                size_t fed_q = 0;
                for (; fed_q < quiet; fed_q += (size_t)chunk) chain.process_samples(std::vector<std::complex<double> >((size_t)chunk));
                // the capture proper must start on a multiple of the reference's call size: timing_sync.cpp:99 is decided by absolute stream
                // index (include/fun_ofdm_amd.h), and the payload list is compared with a decode of the capture from index 0
                const size_t pad = (4096 - (512 + fed_w + fed_q) % 4096) % 4096;      // (512: the priming call above)
                if (pad) chain.process_samples(std::vector<std::complex<double> >(pad));
                for (int idle = 0; idle < 200;) {                    // until nothing has come back for a while (empty calls only poll)
                    std::this_thread::sleep_for(std::chrono::microseconds(200));
                    idle = chain.process_samples(std::vector<std::complex<double> >()).empty() ? idle + 1 : 0;
                }
            }
            g_lat_chunk = chunk;
            g_call_time.reserve(chunks.size() + 2);
            const auto t0 = std::chrono::steady_clock::now();
            const double t0s = now_s();
            size_t fed = 0;
            for (auto &c : chunks) {
                if (pace_msps > 0.0) { const double due = t0s + (double)fed / (pace_msps * 1e6); while (now_s() < due) {} }
                fed += c.size();
                if (g_lat_pitch) g_call_time.push_back(now_s());
                on_packets(chain.process_samples(std::move(c)));
            }
            on_packets(chain.process_samples(std::vector<std::complex<double> >(512)));   // silence lets the pre-sync settle
            on_packets(chain.flush());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%.1f Msamples/s through process_samples (%zu samples in %.4f s, %zu calls of %d)\n", total / dt / 1e6, total, dt, chunks.size(), chunk);
            if (g_lat_pitch && !g_latency_ms.empty()) {
                std::sort(g_latency_ms.begin(), g_latency_ms.end());
                const size_t m = g_latency_ms.size();
                std::printf("payload latency ms: p50 %.3f p90 %.3f p99 %.3f max %.3f (%zu payloads; from the call that delivered a frame's last sample to the call that returned its payload)\n",
                            g_latency_ms[m / 2], g_latency_ms[m * 9 / 10], g_latency_ms[std::min(m - 1, m * 99 / 100)], g_latency_ms[m - 1], m);
            }
        } else {
            if (devices.empty()) {
                fun_amd::receiver rx(on_packets, &src, device, chunk, async_calls, device_batch, narrow_threads);
                rx.wait_finished();
            } else {
                fun_amd::receiver rx(on_packets, &src, devices, chunk, device_batch, narrow_threads);
                rx.wait_finished();
            }
        }
        if (g_out) std::fclose(g_out);
        std::printf("%zu packets, %zu bytes, %zu calls of %d samples\n", g_packets, g_bytes, g_calls, chunk);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "foa_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
