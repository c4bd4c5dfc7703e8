// Probe: do scalar stores (s_store_dwordx2 + s_dcache_wb) work on gfx950, and are they visible to
// later vector loads of the same wave / a later kernel?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned long long *p, int n, unsigned long long *chk)
{
    unsigned long long *row = p + (size_t)blockIdx.x * n;
    for (int t = 0; t < n; t++) {
        unsigned long long d = __ballot(((threadIdx.x * 2654435761u + t * 40503u + blockIdx.x) >> 7) & 1);
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(d), "s"(row), "s"(t * 8) : "memory");
    }
    asm volatile("s_dcache_wb\n s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // read back with vector loads in the same wave
    unsigned long long acc = 0;
    for (int t = threadIdx.x; t < n; t += 64) acc ^= __builtin_nontemporal_load(row + t) * (t + 1);
    for (int o = 32; o; o >>= 1) acc ^= __shfl_xor(acc, o);
    if (threadIdx.x == 0) chk[blockIdx.x] = acc;
}
int main()
{
    const int nb = 2048, n = 8424;
    unsigned long long *d, *c;
    hipMalloc(&d, sizeof(*d) * nb * n); hipMalloc(&c, sizeof(*c) * nb);
    hipMemset(d, 0, sizeof(*d) * nb * n);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 2; it++) { hipEventRecord(e0); k<<<nb, 64>>>(d, n, c); hipEventRecord(e1); hipDeviceSynchronize(); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)nb * n), hc(nb);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, nb * 8, hipMemcpyDeviceToHost);
    size_t bad = 0, badchk = 0;
    for (int b = 0; b < nb; b++) {
        unsigned long long acc = 0;
        for (int t = 0; t < n; t++) {
            unsigned long long w = 0;
            for (unsigned l = 0; l < 64; l++) w |= (unsigned long long)(((l * 2654435761u + t * 40503u + b) >> 7) & 1) << l;
            if (h[(size_t)b * n + t] != w) bad++;
            acc ^= w * (t + 1);
        }
        if (acc != hc[b]) badchk++;
    }
    printf("sstore probe: %d blocks x %d steps, %.3f ms, host mismatches %zu, in-kernel readback mismatches %zu\n", nb, n, ms, bad, badchk);
    return bad || badchk;
}
