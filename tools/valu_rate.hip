// Micro-benchmark: issue rate of the VALU instructions the wave kernel leans on, 8 waves per SIMD.
// Each thread runs a long chain of independent instructions of one kind; cycles per wave-instruction
// per SIMD = elapsed shader cycles * 4 SIMDs / (waves per CU * instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

template <int KIND>
__global__ __launch_bounds__(256, 8) void k(unsigned *out, unsigned seed, unsigned long long *cyc) {
    unsigned a = threadIdx.x + seed, b = a * 3u + 1u, c = a ^ 0x5b5b5b5bu, d = b + 7u;
    float fa = (float)a, fb = (float)b, fc = 1.0f, fd = 2.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        if (KIND == 0) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %3, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 2) { REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x6c\n v_bitop3_b32 %2, %2, %3, %0 bitop3:0x6c" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 3) { REP64(asm volatile("v_dot4_u32_u8 %0, %1, %1, %0\n v_dot4_u32_u8 %2, %3, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 4) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 5) { REP64(asm volatile("v_cvt_f32_u32 %0, %1\n v_cvt_f32_u32 %2, %3" : "+v"(fa), "+v"(b), "+v"(fc), "+v"(d));) }
        if (KIND == 6) { REP64(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 7) { REP64(asm volatile("v_sqrt_f32 %0, %1\n v_sqrt_f32 %2, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 8) { REP64(asm volatile("v_bfe_u32 %0, %1, 3, 10\n v_bcnt_u32_b32 %2, %0, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 9) { REP64(asm volatile("v_alignbit_b32 %0, %1, %0, 7\n v_alignbit_b32 %2, %3, %2, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 10) { REP64(asm volatile("v_cmp_gt_f32 vcc, %1, %3\n v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a), "+v"(fb), "+v"(c), "+v"(fd) : : "vcc");) }
        if (KIND == 11) { REP64(asm volatile("v_mul_f32 %0, %0, %1\n v_sub_f32 %2, %2, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 12) { REP64(asm volatile("v_mad_i32_i24 %0, %0, %1, %2\n v_mad_u32_u24 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 13) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_or_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)(fa + fb + fc + fd);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, unsigned *d_out, unsigned long long *d_cyc) {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1u, d_cyc);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 2u, d_cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double insts = 64.0 * 64 * 2;  // per wave
    // 8 waves share one SIMD: cycles per instruction per SIMD = mean cycles / (8 * insts)
    printf("%-28s %7.2f cycles per wave-instruction per SIMD (8 waves/SIMD)\n", name, mean / (8.0 * insts));
}

int main() {
    unsigned *d_out;
    unsigned long long *d_cyc;
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_cyc, 256 * 8 * 8);
    run<0>("v_add_u32", d_out, d_cyc);
    run<1>("v_fma_f32", d_out, d_cyc);
    run<2>("v_bitop3_b32", d_out, d_cyc);
    run<3>("v_dot4_u32_u8", d_out, d_cyc);
    run<4>("v_mul_u32_u24", d_out, d_cyc);
    run<5>("v_cvt_f32_u32", d_out, d_cyc);
    run<6>("v_add_u32_sdwa", d_out, d_cyc);
    run<7>("v_sqrt_f32", d_out, d_cyc);
    run<8>("v_bfe_u32 + v_bcnt", d_out, d_cyc);
    run<9>("v_alignbit_b32", d_out, d_cyc);
    run<10>("v_cmp_gt_f32 + v_addc", d_out, d_cyc);
    run<11>("v_mul_f32 + v_sub_f32", d_out, d_cyc);
    run<12>("v_mad_i32_i24 / u24", d_out, d_cyc);
    run<13>("v_cndmask + v_or", d_out, d_cyc);
    return 0;
}
