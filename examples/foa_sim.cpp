// foa_sim -- offline receiver over a raw I/Q file (the shape of the reference's examples/test_sim.cpp
// with the simulated channel replaced by a capture).
//
//   foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--device-batch B [--narrow-threads T]]
//                     [--preload] [--out FILE]
//
// Feeds the file through fun_amd::receiver (sample source -> receiver_chain::process_samples -> callback) in chunks of N
// samples (default 4096, the reference's NUM_RX_SAMPLES) and writes every received PSDU to FILE (default: stdout summary
// only) as a record: 4-byte little-endian length, then the bytes.  --async K: decode in asynchronous batches submitted
// every K calls (fun_amd::receiver_chain's streaming mode) instead of synchronously in every call.  --device-batch B: the
// whole of process_samples() on the device in batches of B samples (pre-sync kernels included; the mode for rates far
// above real time), T helper threads narrowing large calls to float.  --preload: read the capture into memory first (as
// the complex<double> chunks process_samples() takes) and time the receive loop alone: "x.y Msamples/s through
// process_samples" is then the rate of the drop-in API itself, without the file read and the float -> double widening of
// this program's own source.
//
// build:  g++ -O2 -std=c++17 examples/foa_sim.cpp -Iinclude -Lfun_ofdm_amd/csrc -lfun_ofdm_amd -lpthread -o foa_sim
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <fun_ofdm_amd/blocks.hpp>

static std::FILE *g_out = nullptr;
static size_t g_packets = 0, g_bytes = 0, g_calls = 0;

static void on_packets(std::vector<std::vector<unsigned char> > packets)
{
    g_calls++;
    for (const auto &p : packets) {
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
    // the host process's part of the set-up (include/fun_ofdm_amd.h, foa_recommended_hw_queues): the HIP runtime fixes its hardware queues
    // when it starts, i.e. before the library is first called
    ::setenv("GPU_MAX_HW_QUEUES", std::to_string(foa_recommended_hw_queues()).c_str(), 0);
    std::string path, format = "fc32", out;
    int chunk = 4096, device = 0, async_calls = 0, narrow_threads = 0;
    size_t device_batch = 0;
    bool preload = false;
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
            fun_amd::receiver_chain chain(device, async_calls, device_batch, narrow_threads);
            chain.process_samples(std::vector<std::complex<double> >(512));            // creates the handle outside the timed loop
            const auto t0 = std::chrono::steady_clock::now();
            for (auto &c : chunks) { on_packets(chain.process_samples(std::move(c))); }
            on_packets(chain.process_samples(std::vector<std::complex<double> >(512)));   // silence lets the pre-sync settle
            on_packets(chain.flush());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%.1f Msamples/s through process_samples (%zu samples in %.4f s, %zu calls of %d)\n", total / dt / 1e6, total, dt, chunks.size(), chunk);
        } else {
            fun_amd::receiver rx(on_packets, &src, device, chunk, async_calls, device_batch, narrow_threads);
            rx.wait_finished();
        }
        if (g_out) std::fclose(g_out);
        std::printf("%zu packets, %zu bytes, %zu calls of %d samples\n", g_packets, g_bytes, g_calls, chunk);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "foa_sim: %s\n", e.what());
        return 1;
    }
    return 0;
}
