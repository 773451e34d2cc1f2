// Micro-benchmark: dependent-issue latency of one wave for the instruction kinds the exact
// z-score path could be built from (plain VALU add, DPP add, LDS-fed add chains).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>
#define R4(S) S S S S
#define R16(S) R4(R4(S))
#define R256(S) R16(R16(S))

__device__ __forceinline__ float readlane_f32(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
#define GAMS_SEQ_STEP "s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ float seq_add_lanes(float carry, float x, uint32_t m) {
    for (uint32_t r = 0; r * 16u < m; ++r) {
        float s = x + carry;
        asm volatile(GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP
                         GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP
                             GAMS_SEQ_STEP GAMS_SEQ_STEP GAMS_SEQ_STEP
                     : "+v"(s)
                     : "v"(x));
        carry = readlane_f32(s, (int)(r * 16u + 15u));
    }
    return carry;
}
// the exact path as it was before (row-DPP recurrence)
__device__ __noinline__ float exact_dpp(const uint8_t *K, uint32_t tj, uint32_t n, float fsize) {
    const uint32_t lane = threadIdx.x & 63u;
    const float len = (float)n;
    float sum = 0.0f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const float x = c0 + lane < n ? (float)K[tj + c0 + lane] / fsize : 0.0f;
        sum = seq_add_lanes(sum, x, min(64u, n - c0));
    }
    const float mean = sum / len;
    float sq = 0.0f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const float x = (float)K[tj + min(c0 + lane, n - 1u)] / fsize;
        const float d = x - mean;
        const float dd = c0 + lane < n ? d * d : 0.0f;
        sq = seq_add_lanes(sq, dd, min(64u, n - c0));
    }
    return sqrtf(sq / (len - 1.0f)) + mean;
}
// candidate: lane values -> SGPRs (independent v_readlane), one plain dependent v_add_f32 per element
template <int N>
__device__ __forceinline__ float seq_add_sgpr(float carry, float x) {
#pragma unroll
    for (int j = 0; j < N; ++j) carry = carry + readlane_f32(x, j);
    return carry;
}
// same, 8 lanes fetched ahead into 8 SGPRs so that no add waits for its v_readlane
template <int N, int B = 0>
__device__ __forceinline__ float seq_add_sgpr8(float carry, float x) {
    if constexpr (N >= 8) {
        int t0 = __builtin_amdgcn_readlane(__float_as_int(x), B + 0), t1 = __builtin_amdgcn_readlane(__float_as_int(x), B + 1),
            t2 = __builtin_amdgcn_readlane(__float_as_int(x), B + 2), t3 = __builtin_amdgcn_readlane(__float_as_int(x), B + 3),
            t4 = __builtin_amdgcn_readlane(__float_as_int(x), B + 4), t5 = __builtin_amdgcn_readlane(__float_as_int(x), B + 5),
            t6 = __builtin_amdgcn_readlane(__float_as_int(x), B + 6), t7 = __builtin_amdgcn_readlane(__float_as_int(x), B + 7);
        asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(t3), "+s"(t4), "+s"(t5), "+s"(t6), "+s"(t7));
        carry = carry + __int_as_float(t0);
        carry = carry + __int_as_float(t1);
        carry = carry + __int_as_float(t2);
        carry = carry + __int_as_float(t3);
        carry = carry + __int_as_float(t4);
        carry = carry + __int_as_float(t5);
        carry = carry + __int_as_float(t6);
        carry = carry + __int_as_float(t7);
        asm volatile("" : "+v"(carry));
        return seq_add_sgpr8<N - 8, B + 8>(carry, x);
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) carry = carry + readlane_f32(x, B + j);
        return carry;
    }
}
template <int LAG>
__device__ __noinline__ float exact_sgpr8(const uint8_t *K, uint32_t tj, float fsize) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int N1 = LAG > 64 ? 64 : LAG, N2 = LAG > 64 ? LAG - 64 : 0;
    const float x0 = (int)lane < N1 ? (float)K[tj + lane] / fsize : 0.0f;
    const float x1 = (int)lane < N2 ? (float)K[tj + 64u + lane] / fsize : 0.0f;
    float sum = seq_add_sgpr8<N1>(0.0f, x0);
    if (N2) sum = seq_add_sgpr8<N2>(sum, x1);
    const float mean = sum / (float)LAG;
    const float d0 = x0 - mean, d1 = x1 - mean;
    float sq = seq_add_sgpr8<N1>(0.0f, d0 * d0);
    if (N2) sq = seq_add_sgpr8<N2>(sq, d1 * d1);
    return sqrtf(sq / ((float)LAG - 1.0f)) + mean;
}
template <int LAG>
__device__ __noinline__ float exact_sgpr(const uint8_t *K, uint32_t tj, float fsize) {
    const uint32_t lane = threadIdx.x & 63u;
    constexpr int N1 = LAG > 64 ? 64 : LAG, N2 = LAG > 64 ? LAG - 64 : 0;
    static_assert(LAG <= 128, "two chunks");
    const float x0 = (int)lane < N1 ? (float)K[tj + lane] / fsize : 0.0f;
    const float x1 = (int)lane < N2 ? (float)K[tj + 64u + lane] / fsize : 0.0f;
    float sum = seq_add_sgpr<N1>(0.0f, x0);
    if (N2) sum = seq_add_sgpr<N2>(sum, x1);
    const float mean = sum / (float)LAG;
    const float d0 = x0 - mean, d1 = x1 - mean;
    float sq = seq_add_sgpr<N1>(0.0f, d0 * d0);
    if (N2) sq = seq_add_sgpr<N2>(sq, d1 * d1);
    return sqrtf(sq / ((float)LAG - 1.0f)) + mean;
}

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, const unsigned char *kin, int lag) {
    __shared__ unsigned char K[4096];
    __shared__ float XT[256];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) K[i] = kin[i];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) XT[i] = (float)i / 100.0f;
    __syncthreads();
    float x = threadIdx.x * 0.001f + 1.0f, s = x;
    unsigned long long t0 = 0, t1 = 0;
    if ((threadIdx.x >> 6) == 0) {
        t0 = __builtin_readcyclecounter();
        if (MODE == 0) {
            asm volatile(R256("v_add_f32 %0, %0, %1\n\t") : "+v"(s) : "v"(x));
        } else if (MODE == 1) {
            asm volatile(R256("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(s) : "v"(x));
        } else if (MODE == 2) {
            asm volatile(R256("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(s) : "v"(x));
        } else if (MODE == 3) {
            // per-lane serial sum over `lag` counts starting at the lane's own offset (two-level LDS lookup)
            const int tj = threadIdx.x * 7;
            float acc = 0.0f;
#pragma unroll 16
            for (int j = 0; j < lag; ++j) acc = acc + XT[K[tj + j]];
            s = acc;
        } else if (MODE == 4) {
            // same, counts fetched as dwords (4 per load), table lookups per byte
            const int tj = threadIdx.x * 8;
            const unsigned *KW = reinterpret_cast<const unsigned *>(K);
            float acc = 0.0f;
#pragma unroll 8
            for (int j = 0; j < lag / 4; ++j) {
                const unsigned w = KW[tj / 4 + j];
                acc = acc + XT[w & 255u];
                acc = acc + XT[(w >> 8) & 255u];
                acc = acc + XT[(w >> 16) & 255u];
                acc = acc + XT[w >> 24];
            }
            s = acc;
        } else if (MODE == 5) {
            // two passes like the reference: mean, then squared deviations
            const int tj = threadIdx.x * 8;
            const unsigned *KW = reinterpret_cast<const unsigned *>(K);
            float acc = 0.0f;
#pragma unroll 8
            for (int j = 0; j < lag / 4; ++j) {
                const unsigned w = KW[tj / 4 + j];
                acc = acc + XT[w & 255u];
                acc = acc + XT[(w >> 8) & 255u];
                acc = acc + XT[(w >> 16) & 255u];
                acc = acc + XT[w >> 24];
            }
            const float mean = acc / (float)lag;
            float sq = 0.0f;
#pragma unroll 8
            for (int j = 0; j < lag / 4; ++j) {
                const unsigned w = KW[tj / 4 + j];
                float d;
                d = XT[w & 255u] - mean; sq = sq + d * d;
                d = XT[(w >> 8) & 255u] - mean; sq = sq + d * d;
                d = XT[(w >> 16) & 255u] - mean; sq = sq + d * d;
                d = XT[w >> 24] - mean; sq = sq + d * d;
            }
            s = sq;
        }
        else if (MODE == 6) s = exact_dpp(K, (blockIdx.x * 13) % 3000, (uint32_t)lag, 100.0f);
        else if (MODE == 7) s = exact_sgpr<100>(K, (blockIdx.x * 13) % 3000, 100.0f);
        else if (MODE == 8) s = exact_sgpr8<100>(K, (blockIdx.x * 13) % 3000, 100.0f);
        else if (MODE == 9) {   // cold: only one workgroup in 64 runs the evaluation, once
            if (blockIdx.x % 64 == 5) s = exact_sgpr8<100>(K, (blockIdx.x * 13) % 3000, 100.0f);
        } else if (MODE == 10) {
            if (blockIdx.x % 64 == 5) s = exact_dpp(K, (blockIdx.x * 13) % 3000, (uint32_t)lag, 100.0f);
        }
        asm volatile("" : "+v"(s));
        t1 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int steps, float *d, unsigned long long *c, unsigned char *kin, int threads) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, c, kin, 100);
        (void)hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= 256;
    printf("%-52s %3d threads: %7.0f cycles / %d steps = %5.1f per step\n", name, threads, m, steps, m / steps);
}

int main() {
    float *d;
    unsigned long long *c;
    unsigned char *kin;
    (void)hipMalloc(&d, 256 * 256 * 4);
    (void)hipMalloc(&c, 256 * 8);
    (void)hipMalloc(&kin, 4096);
    std::vector<unsigned char> hk(4096);
    for (int i = 0; i < 4096; ++i) hk[i] = (unsigned char)(30 + (i * 7) % 40);
    (void)hipMemcpy(kin, hk.data(), 4096, hipMemcpyHostToDevice);
    for (int threads : {64, 256}) {
        run<0>("v_add_f32 dependent chain", 256, d, c, kin, threads);
        run<1>("s_nop 1 + v_add_f32_dpp row_shr:1 chain", 256, d, c, kin, threads);
        run<2>("v_fma_f32 dependent chain", 256, d, c, kin, threads);
        run<3>("per-lane sum of XT[K[j]], byte loads", 100, d, c, kin, threads);
        run<4>("per-lane sum of XT[K[j]], dword loads", 100, d, c, kin, threads);
        run<5>("per-lane mean + squared deviations (2 passes)", 200, d, c, kin, threads);
        run<6>("exact path today (row DPP chain), lag 100", 200, d, c, kin, threads);
        run<7>("exact path via v_readlane -> SGPR + v_add, lag 100", 200, d, c, kin, threads);
        run<8>("exact path via 8 SGPRs ahead + v_add, lag 100", 200, d, c, kin, threads);
    }
    // cold instruction cache: first and only launch of these instantiations, 4 of 256 workgroups evaluate
    {
        std::vector<unsigned long long> h(256);
        hipLaunchKernelGGL(k<9>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        printf("cold, SGPR chain: %llu %llu %llu %llu cycles (others, skipping: %llu)\n", h[5], h[69], h[133], h[197], h[6]);
        hipLaunchKernelGGL(k<10>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        printf("cold, DPP chain:  %llu %llu %llu %llu cycles (others, skipping: %llu)\n", h[5], h[69], h[133], h[197], h[6]);
        hipLaunchKernelGGL(k<9>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), c, 256 * 8, hipMemcpyDeviceToHost);
        printf("second launch, SGPR chain: %llu %llu %llu %llu cycles\n", h[5], h[69], h[133], h[197]);
    }
    // both exact variants against the sequential f32 evaluation on the host
    for (int mode = 6; mode <= 8; ++mode) {
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        else if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        else hipLaunchKernelGGL(k<8>, dim3(256), dim3(256), 0, 0, d, c, kin, 100);
        (void)hipDeviceSynchronize();
        std::vector<float> ho(256 * 256);
        (void)hipMemcpy(ho.data(), d, ho.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < 256; ++b) {
            const int tj = (b * 13) % 3000;
            volatile float sum = 0.0f;
            for (int j = 0; j < 100; ++j) sum = sum + (float)hk[tj + j] / 100.0f;
            const float mean = sum / 100.0f;
            volatile float sq = 0.0f;
            for (int j = 0; j < 100; ++j) { const float dd = (float)hk[tj + j] / 100.0f - mean; sq = sq + dd * dd; }
            const float ref = sqrtf(sq / 99.0f) + mean;
            if (ref != ho[b * 256]) ++bad;
        }
        printf("mode %d: %d of 256 results differ from the host's sequential f32 evaluation\n", mode, bad);
    }
    return 0;
}
