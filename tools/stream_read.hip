// Micro-benchmark: what a read-only stream over a genome-sized buffer can reach on this chip with the
// access pattern of the wave kernel (256-thread workgroups, each reading one contiguous tile as rows
// of 4 KiB, 16 B per lane), as a function of tile size, workgroups per CU and loop structure.
// This is the ceiling the wave kernel's load phase is measured against (profiles/r02_stream_read.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

// one tile per workgroup: ROWS back-to-back 16-B loads per thread, xor-reduced
template <int ROWS, int OCC, bool NT = false>
__global__ __launch_bounds__(256, OCC) void tile_read(const uint4 *src, unsigned *sink, unsigned lds_pad) {
    extern __shared__ unsigned pad[];
    const uint4 *p = src + (size_t)blockIdx.x * (ROWS * 256) + threadIdx.x;
    uint4 v[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        if (NT) {
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p + 256 * k));
            v[k] = make_uint4(t.x, t.y, t.z, t.w);
        } else {
            v[k] = p[256 * k];
        }
    }
    unsigned x = 0;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) x ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    if (lds_pad && x == 0x12345u) pad[threadIdx.x] = x;   // keeps the dynamic LDS (occupancy limiter) alive
    if (x == 0xdeadbeefu) sink[blockIdx.x] = x;
}

// persistent: each workgroup walks tiles b, b+G, ...; the next tile's loads are issued before the
// current tile is reduced (register double buffer)
template <int ROWS, int OCC>
__global__ __launch_bounds__(256, OCC) void persistent_read(const uint4 *src, unsigned *sink, unsigned n_tiles) {
    unsigned x = 0;
    uint4 cur[ROWS], nxt[ROWS];
    unsigned t = blockIdx.x;
    if (t >= n_tiles) return;
    {
        const uint4 *p = src + (size_t)t * (ROWS * 256) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) cur[k] = p[256 * k];
    }
    for (;;) {
        const unsigned tn = t + gridDim.x;
        const bool more = tn < n_tiles;
        const uint4 *p = src + (size_t)(more ? tn : t) * (ROWS * 256) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) nxt[k] = p[256 * k];
#pragma unroll
        for (int k = 0; k < ROWS; ++k) x ^= cur[k].x ^ cur[k].y ^ cur[k].z ^ cur[k].w;
        if (!more) break;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) cur[k] = nxt[k];
        t = tn;
    }
    if (x == 0xdeadbeefu) sink[blockIdx.x] = x;
}

template <typename F>
double time_us(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

template <int ROWS, int OCC, bool NT = false>
void run_tile(const uint4 *d, unsigned *sink, size_t bytes, unsigned lds) {
    const unsigned n_tiles = (unsigned)(bytes / (ROWS * 4096));
    if (NT) printf("(nontemporal) ");
    auto k = tile_read<ROWS, OCC, NT>;
    CK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const double us = time_us([&] { hipLaunchKernelGGL(k, dim3(n_tiles), dim3(256), lds, 0, d, sink, lds); }, 20);
    printf("tile_read  rows=%2d (%3d KiB/wg) occ<=%d lds=%6u  %8.1f us  %7.1f GB/s\n", ROWS, ROWS * 4, OCC, lds, us,
           bytes / us / 1e3);
}

template <int ROWS, int OCC>
void run_pers(const uint4 *d, unsigned *sink, size_t bytes, int wg_per_cu) {
    const unsigned n_tiles = (unsigned)(bytes / (ROWS * 4096));
    const double us = time_us(
        [&] { hipLaunchKernelGGL((persistent_read<ROWS, OCC>), dim3(256 * wg_per_cu), dim3(256), 0, 0, d, sink, n_tiles); },
        20);
    printf("persistent rows=%2d (%3d KiB/wg) wg/cu=%d           %8.1f us  %7.1f GB/s\n", ROWS, ROWS * 4, wg_per_cu, us,
           bytes / us / 1e3);
}

int main(int argc, char **argv) {
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 384;
    const size_t bytes = mb << 20;
    uint4 *d;
    unsigned *sink;
    CK(hipMalloc(&d, bytes + (1 << 20)));
    CK(hipMalloc(&sink, 1 << 22));
    CK(hipMemset(d, 0x41, bytes + (1 << 20)));
    printf("buffer %zu MiB\n", mb);
    run_tile<2, 8>(d, sink, bytes, 0);
    run_tile<4, 8>(d, sink, bytes, 0);
    run_tile<8, 8>(d, sink, bytes, 0);
    run_tile<8, 8, true>(d, sink, bytes, 0);
    run_tile<4, 8, true>(d, sink, bytes, 0);
    run_tile<8, 8>(d, sink, bytes, 40 * 1024);   // 4 workgroups per CU (LDS bound)
    run_tile<8, 8>(d, sink, bytes, 80 * 1024);   // 2 per CU
    run_tile<8, 4>(d, sink, bytes, 0);
    run_tile<16, 4>(d, sink, bytes, 0);
    run_tile<16, 4>(d, sink, bytes, 80 * 1024);
    run_pers<4, 8>(d, sink, bytes, 8);
    run_pers<8, 4>(d, sink, bytes, 4);
    run_pers<8, 4>(d, sink, bytes, 2);
    run_pers<4, 8>(d, sink, bytes, 4);
    run_pers<2, 8>(d, sink, bytes, 8);
    return 0;
}
