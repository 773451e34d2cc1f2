// How many kernel launches per second does the device take from S streams queued by T host threads?
// (an empty 1-workgroup kernel and a 1195-workgroup kernel that exits at once)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void nop(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    const int K = 4000;
    for (int grid : {1, 1195}) {
        for (int S : {1, 2, 4, 8}) {
            for (int T : {1, 2, 4}) {
                if (T > S) continue;
                std::vector<hipStream_t> st(S);
                for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
                for (auto &s : st) hipLaunchKernelGGL(nop, dim3(grid), dim3(256), 0, s, nullptr);
                (void)hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                std::vector<std::thread> th;
                for (int t = 0; t < T; ++t)
                    th.emplace_back([&, t] {
                        (void)hipSetDevice(0);
                        for (int k = 0; k < K; ++k)
                            for (int s = t; s < S; s += T) hipLaunchKernelGGL(nop, dim3(grid), dim3(256), 0, st[s], nullptr);
                    });
                for (auto &x : th) x.join();
                (void)hipDeviceSynchronize();
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf("grid %4d, %d streams, %d host threads: %.2f us per launch\n", grid, S, T, us / (K * S));
                for (auto &s : st) (void)hipStreamDestroy(s);
            }
        }
    }
    return 0;
}
