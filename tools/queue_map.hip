// Which of the streams a process creates can overlap kernels well?  S streams are created in order;
// four of them (a chosen subset) each get K back-to-back launches of a ~8 us kernel that fills the
// chip, queued by four host threads.  Prints us per launch for each subset.  Run under different
// GPU_MAX_HW_QUEUES values to see how streams map onto hardware queues.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ __launch_bounds__(256) void busy(float *p, int iters) {
    float a = threadIdx.x * 0.001f, b = a + 1.0f;
    for (int i = 0; i < iters; ++i) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
    if (a == 12345.678f) p[0] = a + b;
}
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 8;
    const int K = 1500;
    float *d;
    (void)hipMalloc(&d, 1024);
    std::vector<hipStream_t> st(S);
    for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    // calibrate iters for ~8 us with 1195 workgroups
    int iters = 600;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(busy, dim3(1195), dim3(256), 0, st[0], d, iters);
        (void)hipDeviceSynchronize();
    }
    {
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(busy, dim3(1195), dim3(256), 0, st[0], d, iters);
        (void)hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 200;
        printf("%d streams created; one stream alone: %.2f us per launch\n", S, us);
    }
    std::vector<std::vector<int>> subsets;
    for (int a = 0; a + 3 < S; ++a) subsets.push_back({a, a + 1, a + 2, a + 3});
    if (S >= 8) { subsets.push_back({0, 2, 4, 6}); subsets.push_back({0, 4, 1, 5}); subsets.push_back({0, 3, 4, 5}); subsets.push_back({0, 4, 5, 6}); }
    if (S >= 7) subsets.push_back({0, 3, 4, 5});
    for (auto &sub : subsets) {
        for (int s : sub) hipLaunchKernelGGL(busy, dim3(1195), dim3(256), 0, st[s], d, iters);
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (int k = 0; k < K; ++k) hipLaunchKernelGGL(busy, dim3(1195), dim3(256), 0, st[sub[t]], d, iters);
            });
        for (auto &x : th) x.join();
        (void)hipDeviceSynchronize();
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (4.0 * K);
        printf("  streams {%d,%d,%d,%d}: %.2f us per launch\n", sub[0], sub[1], sub[2], sub[3], us);
    }
    return 0;
}
