// probe_pipes.hip -- which HIP streams of one process share a DISPATCHER?  Streams on different hardware queues run small kernels side by
// side (probe_queues.hip), but a grid that does not fit on the machine at once keeps its queue's dispatcher busy until its last workgroup
// has started, and a kernel on another queue served by the same dispatcher waits for that.  Two pipelined decode lanes placed like that run
// one after the other (profiles/r05_ab_stream_layout.txt: 27.7 against 37.2 Gsample/s, decided by the ORDER in which six streams were created).
// Here: S streams are created in order (priorities given on the command line, default all normal); for every ordered pair (i, j) a BIG grid
// (4 096 workgroups, two resident per CU through their LDS, ~4 ms) goes to stream i and one single-wave kernel (~0.2 ms) to stream j; the
// time until the small one is done says whether j had to wait for i's dispatch.
//   hipcc --offload-arch=gfx950 -O3 -o build/probe_pipes.bin tools/probe_pipes.hip ;  build/probe_pipes.bin [S] [priority pattern, e.g. 0,-1,-1,0,1,1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void k_big(unsigned long long ticks, unsigned long long *out)
{
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) { lds[0] = 1; *out = t - t0 + lds[0]; }
}
__global__ void k_small(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long t = t0;
    while (t - t0 < ticks) t = __builtin_readcyclecounter();
    if (threadIdx.x == 0) *out = t - t0;
}

int main(int argc, char **argv)
{
    const int S = argc > 1 ? atoi(argv[1]) : 12;
    std::vector<int> prio(S, 0);
    if (argc > 2) { char *p = strtok(argv[2], ","); for (int i = 0; i < S && p; i++, p = strtok(nullptr, ",")) prio[i] = atoi(p); }
    const char *q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES %s; %d streams, priorities:", q ? q : "(unset: 4)", S);
    for (int i = 0; i < S; i++) printf(" %d", prio[i]);
    printf("\n");
    (void)hipFuncSetAttribute((const void *)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    std::vector<hipStream_t> st(S);
    for (int i = 0; i < S; i++) (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, prio[i]);
    unsigned long long *d;
    (void)hipMalloc(&d, 64 * sizeof *d);
    const unsigned long long big_ticks = 1200000, small_ticks = 480000;  // the counter runs at ~2.4 GHz: 0.5 ms per workgroup (x 8 rounds), 0.2 ms
    for (int i = 0; i < S; i++) { hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st[i], 100ull, d); hipLaunchKernelGGL(k_big, dim3(8), dim3(256), 65536, st[i], 100ull, d + 1); }
    (void)hipDeviceSynchronize();
    printf("time until the small kernel on stream j is done, ms (rows: stream i with the big grid; '.' < 1 ms = side by side, '#' = waited)\n     ");
    for (int j = 0; j < S; j++) printf("%3d", j);
    printf("\n");
    for (int i = 0; i < S; i++) {
        printf("%3d  ", i);
        for (int j = 0; j < S; j++) {
            if (i == j) { printf("  -"); continue; }
            hipLaunchKernelGGL(k_big, dim3(4096), dim3(256), 65536, st[i], big_ticks, d + 1);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, st[j], small_ticks, d);
            (void)hipStreamSynchronize(st[j]);
            const double ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3;
            (void)hipDeviceSynchronize();
            printf("  %c", ms < 1.0 ? '.' : '#');
        }
        printf("\n");
    }
    return 0;
}
