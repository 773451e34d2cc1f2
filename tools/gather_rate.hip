// Micro-benchmark: how many scattered (one per lane, random) reads per second the chip serves, as a function of
// the table size (L2 / Infinity Cache / HBM resident), the bytes per read (16 B, or a 32-B record as two 16-B
// loads) and whether a lane's reads are independent or each depends on the one before.  This is the ceiling the
// random-query interval kernels (locate --count, anno) are measured against (profiles/r02_gather_rate.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {   // a cheap hash: the next random record
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

// every lane: READS gathers of a WIDTH-byte record at a random 32-B aligned place in a table of `recs` records;
// DEP: the place of gather k+1 depends on the data of gather k
template <int READS, int WIDTH, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint4 *table, uint32_t recs_mask, uint32_t *sink) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    uint32_t h = mix(gid * 2654435761u + 12345u);
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < READS; ++k) {
        const uint4 *p = table + 2u * (size_t)(h & recs_mask);
        const uint4 a = p[0];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
        if (WIDTH == 32) {
            const uint4 b = p[1];
            acc ^= b.x ^ b.y ^ b.z ^ b.w;
        }
        h = mix(h + (DEP ? acc : 0u) + 0x9e3779b9u);
    }
    if (acc == 0xdeadbeefu) sink[gid & 1023u] = acc;
}

template <int READS, int WIDTH, bool DEP>
void run(const uint4 *d, uint32_t *sink, size_t bytes, uint32_t lanes) {
    const uint32_t recs = (uint32_t)(bytes / 32);
    uint32_t mask = 1;
    while (mask * 2u <= recs) mask *= 2u;
    mask -= 1u;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto launch = [&] { hipLaunchKernelGGL((gather<READS, WIDTH, DEP>), dim3(lanes / 256), dim3(256), 0, 0, d, mask, sink); };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("table %7.1f MB  %d x %2d B per lane, %s: %8.1f us  %6.2f G gathers/s  %6.2f G lanes/s\n",
           (mask + 1.0) * 32 / 1e6, READS, WIDTH, DEP ? "dependent  " : "independent", us,
           (double)lanes * READS / us / 1e3, (double)lanes / us / 1e3);
}

int main() {
    const size_t max_bytes = 2048ull << 20;
    uint4 *d;
    uint32_t *sink;
    CK(hipMalloc(&d, max_bytes));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(d, 0x5a, max_bytes));
    const uint32_t lanes = 12'500'000 / 256 * 256;
    for (size_t mb : {8, 64, 200, 512, 2048}) {
        const size_t bytes = mb << 20;
        run<1, 16, false>(d, sink, bytes, lanes);
        run<1, 32, false>(d, sink, bytes, lanes);
        run<2, 32, false>(d, sink, bytes, lanes);
        run<2, 32, true>(d, sink, bytes, lanes);
        run<3, 32, true>(d, sink, bytes, lanes);
        run<4, 16, false>(d, sink, bytes, lanes);
    }
    return 0;
}
