// probe_queues.hip -- how many HIP streams of one process really run side by side, and does a stream's PRIORITY give it a hardware
// queue of its own?  The library keeps six streams busy; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
// fixed when it starts, which only the host process can change.  If streams of different priority come from different queue pools, the
// library could spread its streams over priorities and would not depend on the host's environment.
// Each stream gets ONE single-wave kernel that spins for a fixed number of shader clocks; N of them take one spin if they overlap,
// ceil(N / queues) spins if they share queues.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_queues.bin tools/probe_queues.hip ;  build/probe_queues.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void k_spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *out = t - t0;
}

static double run(const std::vector<hipStream_t> &st, unsigned long long ticks, unsigned long long *d)
{
    for (auto s : st) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 1000ull, d);      // warm every stream's queue
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < st.size(); i++) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st[i], ticks, d + i);
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
}

int main()
{
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);                  // (lo = least, hi = greatest priority: numerically hi <= lo)
    printf("GPU_MAX_HW_QUEUES %s; stream priority range: least %d .. greatest %d\n", q ? q : "(unset: 4)", lo, hi);
    unsigned long long *d;
    (void)hipMalloc(&d, 64 * sizeof *d);
    const unsigned long long ticks = 100ull * 1000 * 1000 / 10;       // ~5 ms at 100 MHz s_memtime / ~2 GHz: long against launch costs
    std::vector<hipStream_t> one(1);
    (void)hipStreamCreateWithFlags(&one[0], hipStreamNonBlocking);
    const double t1 = run(one, ticks, d);
    printf("one stream: %.2f ms per spin\n", t1);
    for (int n : { 2, 4, 6, 8, 12 }) {
        std::vector<hipStream_t> st(n);
        for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const double t = run(st, ticks, d);
        printf("%2d streams, one priority:           %.2f ms = %.2f spins\n", n, t, t / t1);
        for (auto s : st) (void)hipStreamDestroy(s);
    }
    for (int n : { 6, 8, 12 }) {
        std::vector<hipStream_t> st(n);
        for (int i = 0; i < n; i++) {
            const int pr = hi + (i % (lo - hi + 1));                  // spread over every priority level there is
            (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, pr);
        }
        const double t = run(st, ticks, d);
        printf("%2d streams spread over %d priorities: %.2f ms = %.2f spins\n", n, lo - hi + 1, t, t / t1);
        for (auto s : st) (void)hipStreamDestroy(s);
    }
    return 0;
}
