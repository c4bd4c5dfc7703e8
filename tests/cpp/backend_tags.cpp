// backend_tags.cpp -- fun_amd::rx_backend over a GIVEN tagged stream, cut into work() calls of a given size: the payloads it hands on, one
// hex line each.  The caller (tests/test_backend_tags.py, tests/test_gpu_cpp_adaptors.py) makes the stream -- rotated complex<double>
// samples and one tag byte per sample, pile-ups of LTS1 tags included -- and compares with the oracle's block chain over the same tags.
//   backend_tags <samples.f64> <tags.u8> <chunk>
// Linked against tests/cpp/stub_abi.cpp (CPU: the C ABI answered by the oracle) or the real library (GPU).  TEST INFRASTRUCTURE.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fun_ofdm_amd/blocks.hpp"

static std::vector<unsigned char> slurp(const char *path)
{
    std::vector<unsigned char> v;
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    unsigned char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: backend_tags samples.f64 tags.u8 chunk\n"); return 2; }
    const std::vector<unsigned char> raw = slurp(argv[1]), tags = slurp(argv[2]);
    const size_t n = tags.size(), chunk = (size_t)atol(argv[3]);
    if (raw.size() != n * 16 || chunk == 0) { fprintf(stderr, "sizes do not match\n"); return 2; }
    const double *s = reinterpret_cast<const double *>(raw.data());
    try {
        fun_amd::rx_backend be;
        for (size_t x = 0; x < n; x += chunk) {
            const size_t m = x + chunk <= n ? chunk : n - x;
            be.input_buffer.resize(m);
            for (size_t i = 0; i < m; i++) {
                be.input_buffer[i].sample = std::complex<double>(s[2 * (x + i)], s[2 * (x + i) + 1]);
                be.input_buffer[i].tag = (fun::vector_tag)tags[x + i];
            }
            be.work();
            for (size_t p = 0; p < be.output_buffer.size(); p++) {
                for (size_t b = 0; b < be.output_buffer[p].size(); b++) printf("%02x", be.output_buffer[p][b]);
                printf("\n");
            }
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
