// foa_sim -- offline receiver over a raw I/Q file (the shape of the reference's examples/test_sim.cpp
// with the simulated channel replaced by a capture).
//
//   foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--out FILE]
//
// Feeds the file through fun_amd::receiver (sample source -> receiver_chain::process_samples -> callback) in chunks of N
// samples (default 4096, the reference's NUM_RX_SAMPLES) and writes every received PSDU to FILE (default: stdout summary
// only) as a record: 4-byte little-endian length, then the bytes.  --async K: decode in asynchronous batches submitted
// every K calls (fun_amd::receiver_chain's streaming mode) instead of synchronously in every call.
//
// build:  g++ -O2 -std=c++17 examples/foa_sim.cpp -Iinclude -Lfun_ofdm_amd/csrc -lfun_ofdm_amd -lpthread -o foa_sim
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
    std::string path, format = "fc32", out;
    int chunk = 4096, device = 0, async_calls = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--format" && i + 1 < argc) format = argv[++i];
        else if (a == "--chunk" && i + 1 < argc) chunk = std::atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = std::atoi(argv[++i]);
        else if (a == "--async" && i + 1 < argc) async_calls = std::atoi(argv[++i]);
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else if (path.empty() && a[0] != '-') path = a;
        else { std::fprintf(stderr, "usage: foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--out FILE]\n"); return 2; }
    }
    if (path.empty() || chunk <= 0) { std::fprintf(stderr, "usage: foa_sim <iq file> [--format fc32|fc64] [--chunk N] [--device D] [--async K] [--out FILE]\n"); return 2; }
    try {
        if (!out.empty()) {
            g_out = std::fopen(out.c_str(), "wb");
            if (!g_out) { std::fprintf(stderr, "cannot open %s\n", out.c_str()); return 1; }
        }
        fun_amd::file_source src(path, format);
        {
            fun_amd::receiver rx(on_packets, &src, device, chunk, async_calls);
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
