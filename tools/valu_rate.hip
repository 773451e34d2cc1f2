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
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pa = {fa, fb}, pb = {fb, fa}, pc = {1.0f, 2.0f}, pd = {2.0f, 1.0f};
    unsigned long long qa = a | ((unsigned long long)b << 32), qc = c | ((unsigned long long)d << 32);
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
        if (KIND == 14) { REP64(asm volatile("v_cvt_f32_ubyte1 %0, %1\n v_cvt_f32_ubyte2 %2, %3" : "+v"(fa), "+v"(b), "+v"(fc), "+v"(d));) }
        if (KIND == 15) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %2, %2, %3, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));) }
        if (KIND == 16) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %2, %2, %3" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd));) }
        if (KIND == 17) { REP64(asm volatile("v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 18) { REP64(asm volatile("v_bfe_u32 %0, %1, 3, 10\n v_bfe_u32 %2, %3, 5, 10" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 19) { REP64(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 20) { REP64(asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_and_or_b32 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 21) { REP64(asm volatile("v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 22) { REP64(asm volatile("v_mul_f32 %0, %0, %1 clamp\n v_mul_f32 %2, %2, %3 clamp" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));) }
        if (KIND == 23) { REP64(asm volatile("v_and_b32 %0, 0x3ff00, %1\n v_and_b32 %2, 0xffc00, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 24) { REP64(asm volatile("v_lshrrev_b32 %0, 7, %1\n v_lshlrev_b32 %2, 3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 25) { REP64(asm volatile("v_add3_u32 %0, %0, %1, %2\n v_or3_b32 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 26) { REP64(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_mad_u16 %2, %2, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 27) { REP64(asm volatile("v_mqsad_pk_u16_u8 %0, %0, %1, %0\n v_mqsad_pk_u16_u8 %2, %2, %3, %2" : "+v"(qa), "+v"(b), "+v"(qc), "+v"(d));) }
        if (KIND == 28) { REP64(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_gt_f32 s[4:5], %2, %3" : "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd) : : "vcc", "s4", "s5");) }
        if (KIND == 29) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 30) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %2, %2, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0x01010101u));) }
        if (KIND == 31) { REP64(asm volatile("v_mul_u32_u24 %0, 0x204081, %0\n v_mul_u32_u24 %2, 0x204081, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 32) { REP64(asm volatile("v_perm_b32 %0, %0, %0, %4\n v_perm_b32 %2, %2, %2, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0x03030303u));) }
        if (KIND == 33) { REP64(asm volatile("v_lshl_add_u32 %0, %0, 8, %0\n v_lshl_add_u32 %2, %2, 16, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 34) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 35) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 36) { REP64(asm volatile("v_add3_u32 %0, %0, %1, %4\n v_add3_u32 %2, %2, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0xfbfcfdffu));) }
        if (KIND == 40) { REP64(asm volatile("v_mul_lo_u32 %0, %1, %3\n v_mul_lo_u32 %2, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 41) { REP64(asm volatile("v_mul_u32_u24 %0, %1, %3\n v_mul_u32_u24 %2, %3, %1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 42) { REP64(asm volatile("v_perm_b32 %0, %1, %3, %4\n v_perm_b32 %2, %3, %1, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(0x03030303u));) }
        if (KIND == 43) { REP64(asm volatile("v_dot4_u32_u8 %0, %1, %3, %0\n v_dot4_u32_u8 %2, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 44) { REP64(asm volatile("v_mad_u32_u24 %0, %1, %3, %0\n v_mad_u32_u24 %2, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 45) { REP64(asm volatile("v_add3_u32 %0, %1, %3, %0\n v_add3_u32 %2, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 46) { REP64(asm volatile("v_bfe_u32 %0, %1, 7, 4\n v_bfe_u32 %2, %3, 11, 4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (KIND == 13) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_or_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + (unsigned)(fa + fb + fc + fd) + (unsigned)(pa.x + pa.y + pc.x + pc.y) + (unsigned)(qa + qc);
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
    run<14>("v_cvt_f32_ubyteN", d_out, d_cyc);
    run<15>("v_pk_fma_f32", d_out, d_cyc);
    run<16>("v_pk_mul_f32 + v_pk_add_f32", d_out, d_cyc);
    run<17>("v_bcnt_u32_b32", d_out, d_cyc);
    run<18>("v_bfe_u32", d_out, d_cyc);
    run<19>("v_perm_b32", d_out, d_cyc);
    run<20>("v_lshl_or + v_and_or", d_out, d_cyc);
    run<21>("v_sad_u8", d_out, d_cyc);
    run<22>("v_mul_f32 clamp", d_out, d_cyc);
    run<23>("v_and_b32 literal", d_out, d_cyc);
    run<24>("v_lshrrev + v_lshlrev", d_out, d_cyc);
    run<25>("v_add3 + v_or3", d_out, d_cyc);
    run<26>("v_pk_add_u16 + v_pk_mad_u16", d_out, d_cyc);
    run<27>("v_mqsad_pk_u16_u8", d_out, d_cyc);
    run<28>("v_cmp_f32 (vcc / sgpr)", d_out, d_cyc);
    run<29>("v_mul_lo_u32 + v_mul_hi_u32", d_out, d_cyc);
    run<30>("v_mul_lo_u32 v, v, s", d_out, d_cyc);
    run<34>("v_mul_lo_u32 v, v, v", d_out, d_cyc);
    run<35>("v_mul_hi_u32 v, v, v", d_out, d_cyc);
    run<31>("v_mul_u32_u24 literal", d_out, d_cyc);
    run<32>("v_perm_b32 v, v, v, s", d_out, d_cyc);
    run<33>("v_lshl_add_u32", d_out, d_cyc);
    run<36>("v_add3_u32 v, v, v, s", d_out, d_cyc);
    run<40>("v_mul_lo_u32 d, b, c (live operands)", d_out, d_cyc);
    run<41>("v_mul_u32_u24 d, b, c (live)", d_out, d_cyc);
    run<42>("v_perm_b32 d, b, c, s (live)", d_out, d_cyc);
    run<43>("v_dot4_u32_u8 d, b, c, d (live)", d_out, d_cyc);
    run<44>("v_mad_u32_u24 d, b, c, d (live)", d_out, d_cyc);
    run<45>("v_add3_u32 d, b, c, d (live)", d_out, d_cyc);
    run<46>("v_bfe_u32 d, b, 7, 4", d_out, d_cyc);
    return 0;
}
