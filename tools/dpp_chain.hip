// Micro-benchmark: latency of the dependent row_shr:1 add chain used by the exact path,
// alone on a SIMD and next to busy waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define STEP "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define STEP0 "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define STEP2 "v_mov_b32 %1, %3\n\tv_mov_b32 %2, %3\n\tv_add_f32_dpp %0, %0, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define STEP3 "s_nop 0\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define STEP4 "v_mov_b32 %1, %2\n\tv_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define R15(S) S S S S S S S S S S S S S S S
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int mode, int busy_iters) {
    float x = threadIdx.x * 0.001f, s = x;
    const bool chain = (blockIdx.x % gridDim.y == 0) || true;
    if ((threadIdx.x >> 6) == 0 && blockIdx.y == 0) {
        float d1 = 0, d2 = 0;
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int r = 0; r < 14; ++r) {
            if (mode == 0) asm volatile(R15(STEP) : "+v"(s) : "v"(x));
            else if (mode == 1) asm volatile(R15(STEP0) "s_nop 1\n\t" : "+v"(s) : "v"(x));
            else if (mode == 2) asm volatile(R15(STEP2) : "+v"(s), "+v"(d1), "+v"(d2) : "v"(x));
            else if (mode == 3) asm volatile(R15(STEP3) : "+v"(s) : "v"(x));
            else if (mode == 4) asm volatile(R15(STEP4) : "+v"(s), "+v"(d1) : "v"(x));
            else if (mode == 5) {
#pragma unroll
                for (int q = 0; q < 15; ++q) asm volatile("v_add_f32 %0, %0, %1\n\t" : "+v"(s) : "v"(x));
            } else if (mode == 6) {
                // lane values -> SGPRs (independent v_readlane), chain of plain v_add_f32 with an SGPR operand
#pragma unroll
                for (int q = 0; q < 15; ++q)
                    s += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), (r * 15 + q) & 63));
            } else {
#pragma unroll
                for (int q = 0; q < 15; ++q) s = s + x;                      // compiler's own dependent chain
                asm volatile("" : "+v"(s));
            }
        }
        s += d1 + d2;
        unsigned long long t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    } else {
        float a = x, b = x + 1;
        for (int i = 0; i < busy_iters; ++i) { a = a * 1.0001f + b; b = b * 0.9999f + a; }
        s = a + b;
    }
    out[(blockIdx.x * gridDim.y + blockIdx.y) * 256 + threadIdx.x] = s;
    (void)chain;
}
int main() {
    float *d; unsigned long long *c;
    hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&c, 2048 * 8);
    for (int mode = 0; mode < 8; ++mode)
        for (int ny : {1, 5}) {
            hipLaunchKernelGGL(k, dim3(256, ny), dim3(256), 0, 0, d, c, mode, 20000);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
            double m = 0; for (auto v : h) m += v; m /= 256;
            printf("mode %d (%s), %d workgroups per CU: %.0f cycles for 210 steps = %.1f per step\n", mode,
                   mode == 0 ? "s_nop 1 + add_dpp" : mode == 1 ? "add_dpp back to back (hazard!)" : mode == 2 ? "2 v_mov fillers" : mode == 3 ? "s_nop 0" : mode == 4 ? "1 v_mov filler" : mode == 5 ? "plain dependent v_add_f32" : mode == 6 ? "v_readlane -> SGPR + v_add_f32" : "s = s + x", ny, m, m / 210);
        }
    return 0;
}
