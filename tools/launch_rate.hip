// Micro-benchmark: how fast does the dispatcher place workgroups as a function of the
// dynamic LDS size and of the workgroup lifetime?  Each workgroup records the 100 MHz
// clock at entry and exit and spins for `spin` shader cycles in between.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void k(unsigned long long *t, unsigned spin, int touch) {
    extern __shared__ unsigned char smem[];
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c0 = __builtin_readcyclecounter();
    if (touch) smem[threadIdx.x] = (unsigned char)threadIdx.x;
    while (__builtin_readcyclecounter() - c0 < spin) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) {
        t[blockIdx.x * 2] = r0;
        t[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

int main() {
    const int grids[] = {1187, 7500};
    const size_t ldss[] = {0, 4096, 18432, 40000, 65536};
    const unsigned spins[] = {2000, 12000, 35000};
    unsigned long long *d;
    hipMalloc(&d, 2 * 8000 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int g : grids)
        for (size_t lds : ldss)
            for (unsigned spin : spins) {
                for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(g), dim3(256), lds, 0, d, spin, 1);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(2 * g);
                hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long lo = ~0ull, hi = 0, sum = 0, last_start = 0;
                for (int i = 0; i < g; ++i) {
                    lo = std::min(lo, h[2 * i]);
                    hi = std::max(hi, h[2 * i + 1]);
                    last_start = std::max(last_start, h[2 * i]);
                    sum += h[2 * i + 1] - h[2 * i];
                }
                printf("grid %5d lds %6zu spin %6u: span %7.2f us, last start at %7.2f us, mean life %6.2f us, mean resident %7.1f\n",
                       g, lds, spin, (hi - lo) / 100.0, (last_start - lo) / 100.0, sum / 100.0 / g,
                       (double)sum / (double)(hi - lo));
            }
    return 0;
}
