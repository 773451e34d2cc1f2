// wave_kernels.hpp -- device side of the wave path (included by wave.hip only): the argument
// block, the two tile kernels, the serial / compaction / packing kernels.  See wave.hip for the
// scheme and the host side.
#pragma once

#include "common.hpp"

#include <type_traits>

namespace {

struct WaveCtgDev {
    uint64_t seq_off;   // byte offset of the ctg in the seqset buffer (256-B aligned)
    uint64_t win_base;  // index of the ctg's window 0 in the dense outputs
    uint32_t len;       // bases
    uint32_t n_win;     // windows
};

struct WaveTile {
    uint32_t ctg;
    uint32_t w0;        // first window of the tile
    uint32_t n_win;     // windows of the ctg
    uint32_t pad;
    uint64_t seq_off;   // copy of the ctg's geometry: one load per workgroup instead of two dependent ones
    uint64_t win_base;
};

// Peak records: every tile owns a fixed slot of `tile_cap` records (tile t writes
// peaks[t*tile_cap ...] and its count to tile_cnt[t]): no atomic, no ordering between
// workgroups; gams_wave_peaks() packs the slots on the device.  (A single appended-to
// counter serialises at ~88 atomics/us; 128 sharded counters still put a returning atomic
// on every workgroup's critical path.)  The diagnostics counters stay sharded: one word
// serialises at ~88 atomics/us, which a 1000-tile launch would feel.
constexpr uint32_t kShards = 128;
constexpr uint32_t kShardWords = 16;

struct WaveArgs {
    const uint8_t *seq;
    const WaveCtgDev *ctgs;
    const WaveTile *tiles;
    uint32_t size, step, lag, tw;
    uint32_t max_chunks;  // LDS carve: PM holds max_chunks+1 words
    uint32_t max_win;     // LDS carve: K holds max_win, Q1/Q2 hold max_win+1
    uint32_t flags;
    uint32_t no_signal;   // lag == 1: std is NaN (0/0), the reference never signals
    float thr, thr_abs, fsize, flag_f, cvar;
    float g0, g1, g2, g3;  // guard band  G = g0 + g1*S1 + g2*R + g3*D
    // the same band in the squared domain (wave_fast_kernel, see z_decide): decided signal <=>
    // A > 0 and A^2 > V, decided none <=> B^2 < V, with A = aA*D - (gA0 + gA1*S1),
    // B = aB*D + (gB0 + gB1*S1); every constant is already divided by thr*sqrt(n/(n-1))
    float aA, aB, gA0, gA1, gB0, gB1;
    // outputs
    gams_peak_t *peaks;
    uint32_t tile_cap;             // records per tile slot of `peaks`
    unsigned long long *counters;  // kShards x 16 words (one 128-B line per shard): [1] exact-path evaluations
    uint32_t *tile_cnt;
    uint32_t *dense_cnt;
    int8_t *dense_sig;
    const int8_t *const_sig;  // [size+1]: signal when the lag counts and the window's count all equal k
    unsigned long long *stamps;  // diagnostics: [tile][8] s_memtime at phase boundaries (NULL: off)
};

// Phase stamp of a diagnostic run (gams_wave_plan_set_stamps): thread 0 of the
// workgroup stores the shader clock.  Off (stamps == NULL) it is one scalar branch.
__device__ __forceinline__ void wave_stamp(const WaveArgs &a, int slot) {
    if (a.stamps != nullptr && threadIdx.x == 0) {
        // 16 words (one 128-B line) per workgroup: [0..6] shader clock at the phase
        // boundaries, [8] / [9] the constant 100 MHz clock at workgroup start / end
        a.stamps[(size_t)blockIdx.x * 16 + slot] = __builtin_readcyclecounter();
        if (slot == 0) {
            a.stamps[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memrealtime();
            // where it ran: HW_ID (wave/simd/cu/sh/se) and XCC_ID
            a.stamps[(size_t)blockIdx.x * 16 + 11] =
                ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) |
                (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        }
        if (slot == 6) a.stamps[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();
    }
}

// Sequence bytes are read once per pass (plus a 4 % halo): a non-temporal load (global_load ... nt)
// streams them past the caches' retention policy.  A pure read of this shape runs at 7.06 TB/s with
// it and 6.3 TB/s without (tools/stream_read.hip, profiles/r02_stream_read.txt).  A batch small enough
// to stay in L2 / the Infinity Cache between passes keeps plain loads (NT = false): the 12-Mb pass is
// 8 % slower with the hint, the 120-Mb and 384-Mb passes 4-5 % faster.
template <bool NT>
__device__ __forceinline__ uint4 load_stream16(const uint4 *p) {
    if constexpr (NT) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
        return make_uint4(t.x, t.y, t.z, t.w);
    } else {
        return *p;
    }
}

// ---- G/C/g/c classification of 16 packed bytes -> 16-bit mask ------------------
// 'C' 0x43, 'G' 0x47, 'c' 0x63, 'g' 0x67 are exactly the bytes with
// (b & 0xDB) == 0x43 (0xDB drops the case bit 0x20 and the C/G bit 0x04), i.e.
// bit 7 clear and (b & 0x5B) == 0x43.  Per dword: t = ((x & 0x5B..) ^ 0x43..) + 0x7F..
// sets bit 7 of every byte whose low part differs (no carry leaves a byte:
// <= 0x5B + 0x7F), so ~(t | x) has bit 7 set exactly on G/C/g/c bytes.  On gfx950
// this is v_bitop3 + v_add + v_bitop3.  The four flag bits of a dword are then
// gathered by v_dot4_u32_u8 against bit weights (0x80 * weight, undone by >> 7).
__device__ __forceinline__ uint32_t gc_flags(uint32_t x) {
    const uint32_t t = ((x & 0x5B5B5B5Bu) ^ 0x43434343u) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);  // 0x80 in every G/C/g/c byte, 0 elsewhere
}

__device__ __forceinline__ uint32_t gc_mask16(const uint4 v) {
    // lo = 128 * (bits 0..7 of the mask), hi = 128 * (bits 8..15)
    uint32_t lo = __builtin_amdgcn_udot4(gc_flags(v.x), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(gc_flags(v.y), 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(gc_flags(v.z), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(gc_flags(v.w), 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);
}

// #GC in tile bytes [0, x): chunk prefix + popcount of the chunk's low bits
__device__ __forceinline__ uint32_t gc_prefix_at(const uint32_t *PM, uint32_t x) {
    uint32_t e = PM[x >> 4];
    uint32_t low = e & ((1u << (x & 15u)) - 1u);  // mask lives in the low 16 bits
    return (e >> 16) + __popc(low);
}

// The reference's evaluation, bit for bit (stat.rs:1-14 and :36-38): K holds the
// gc counts of the tile's windows, tj = first averaged window, ti = this window.
template <typename KT>
__device__ __noinline__ int exact_signal(const KT *K, uint32_t tj, uint32_t ti, uint32_t n,
                                         float fsize, float thr) {
    const float len = (float)n;
    float sum = 0.0f;
    for (uint32_t q = 0; q < n; ++q) sum = sum + (float)K[tj + q] / fsize;   // mean: stat.rs:3
    const float mean = sum / len;                                             // stat.rs:5
    float sq = 0.0f;
    for (uint32_t q = 0; q < n; ++q) {                                        // stat.rs:12
        const float x = (float)K[tj + q] / fsize;
        const float d = x - mean;
        sq = sq + d * d;
    }
    const float sd = sqrtf(sq / (len - 1.0f));                                // stat.rs:13
    const float x = (float)K[ti] / fsize;
    if (fabsf(x - mean) > thr * sd) return x > mean ? 1 : -1;                 // stat.rs:36-38
    return 0;
}

template <typename KT, bool WIDE>
__global__ __launch_bounds__(256) void wave_tile_kernel(const WaveArgs a) {
    using Q2T = typename std::conditional<WIDE, uint64_t, uint32_t>::type;
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve: Q2 | Q1 | PM | scratch | PC | K
    const uint32_t mwp = (a.max_win + 3u) & ~1u;  // even element count keeps every array 8-B aligned
    Q2T *Q2 = reinterpret_cast<Q2T *>(smem);
    uint32_t *Q1 = reinterpret_cast<uint32_t *>(Q2 + mwp);
    uint32_t *PM = Q1 + mwp;
    uint64_t *scr = reinterpret_cast<uint64_t *>(PM + ((a.max_chunks + 4) & ~1u));
    uint32_t *PC = reinterpret_cast<uint32_t *>(scr + 8);  // 128 per-(iteration,wave) peak counts + base
    KT *K = reinterpret_cast<KT *>(PC + 132);

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const WaveTile tl = a.tiles[blockIdx.x];
    const struct { uint64_t seq_off, win_base; uint32_t n_win; } cg = {tl.seq_off, tl.win_base, tl.n_win};
    const uint32_t lag = a.lag, step = a.step, size = a.size;
    const uint32_t w0 = tl.w0;
    const uint32_t w1 = min(w0 + a.tw, cg.n_win);
    const uint32_t wh = w0 > lag ? w0 - lag - 1u : 0u;   // first window whose gc is needed
    const uint32_t nw = w1 - wh;
    const uint32_t b0 = wh * step;                        // ctg-relative byte of window wh
    const uint32_t a0 = b0 & ~15u;                        // tile byte 0 (16-B aligned)
    const uint32_t b1 = (w1 - 1u) * step + size;          // exclusive end, <= cg.len
    const uint32_t nchunk = (b1 - a0 + 15u) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.seq + cg.seq_off + a0);

    // ---- phase 1: load + classify ------------------------------------------
    for (uint32_t c = tid; c < nchunk; c += 1024u) {
        const uint32_t c1 = c + 256u, c2 = c + 512u, c3 = c + 768u;
        uint4 v0 = src[c];
        uint4 v1 = make_uint4(0, 0, 0, 0), v2 = v1, v3 = v1;
        if (c1 < nchunk) v1 = src[c1];
        if (c2 < nchunk) v2 = src[c2];
        if (c3 < nchunk) v3 = src[c3];
        PM[c] = gc_mask16(v0);
        if (c1 < nchunk) PM[c1] = gc_mask16(v1);
        if (c2 < nchunk) PM[c2] = gc_mask16(v2);
        if (c3 < nchunk) PM[c3] = gc_mask16(v3);
    }
    __syncthreads();

    // ---- phase 1b: exclusive prefix of chunk popcounts ----------------------
    {
        const uint32_t cpt = ((nchunk + 255u) >> 8) | 1u;  // odd stride: conflict-free LDS walks
        const uint32_t cb = min(tid * cpt, nchunk), ce = min(cb + cpt, nchunk);
        uint32_t s = 0;
        for (uint32_t c = cb; c < ce; ++c) s += __popc(PM[c]);
        uint32_t tot;
        uint32_t run = block_excl_scan_256<uint32_t>(s, reinterpret_cast<uint32_t *>(scr), tot);
        for (uint32_t c = cb; c < ce; ++c) {
            const uint32_t m = PM[c];
            PM[c] = (run << 16) | m;
            run += __popc(m);
        }
        if (cb < ce && ce == nchunk) PM[nchunk] = run << 16;  // sentinel for x on the tile end
    }
    __syncthreads();

    // ---- phase 2: window counts + prefixes of k and k^2 ---------------------
    {
        const uint32_t wpt = ((nw + 255u) >> 8) | 1u;
        const uint32_t tb = min(tid * wpt, nw), te = min(tb + wpt, nw);
        uint32_t s1 = 0;
        Q2T s2 = 0;
        uint32_t x = (wh + tb) * step - a0;
        for (uint32_t t = tb; t < te; ++t, x += step) {
            const uint32_t kk = gc_prefix_at(PM, x + size) - gc_prefix_at(PM, x);
            K[t] = (KT)kk;
            s1 += kk;
            s2 += (Q2T)kk * kk;
        }
        uint32_t tot1;
        Q2T tot2;
        uint32_t q1 = block_excl_scan_256<uint32_t>(s1, reinterpret_cast<uint32_t *>(scr), tot1);
        Q2T q2 = block_excl_scan_256<Q2T>(s2, reinterpret_cast<Q2T *>(scr), tot2);
        for (uint32_t t = tb; t < te; ++t) {
            Q1[t] = q1;
            Q2[t] = q2;
            const uint32_t kk = K[t];
            q1 += kk;
            q2 += (Q2T)kk * kk;
        }
        if (tb < te && te == nw) {
            Q1[nw] = q1;
            Q2[nw] = q2;
        }
    }
    __syncthreads();

    // ---- phase 3: z-score decision per window -------------------------------
    const bool want_peaks = (a.flags & GAMS_WAVE_PEAKS) != 0;
    const bool want_dense = (a.flags & GAMS_WAVE_DENSE) != 0;
    const uint32_t R = a.tw >> 8;  // iterations per thread, <= 32
    uint64_t sigbits = 0;          // 2 bits per iteration: 1 = crest, 3 = trough
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t i = w0 + (r << 8) + tid;
        int sg = 0;
        if (i < w1) {
            const uint32_t ti = i - wh;
            const uint32_t kk = K[ti];
            if (i >= lag && !a.no_signal) {
                const uint32_t tj = (i == lag ? 0u : i - 1u - lag) - wh;
                const uint32_t S1 = Q1[tj + lag] - Q1[tj];
                const Q2T S2 = Q2[tj + lag] - Q2[tj];
                const int64_t di = (int64_t)lag * kk - (int64_t)S1;   // n*k - S1, sign = side of the mean
                float Df, Vf;
                if (WIDE) {
                    const uint64_t D = (uint64_t)(di < 0 ? -di : di);
                    const uint64_t V = (uint64_t)lag * (uint64_t)S2 - (uint64_t)S1 * S1;
                    Df = (float)D;
                    Vf = (float)V;
                } else {
                    const uint32_t D = (uint32_t)(di < 0 ? -di : di);
                    const uint32_t V = lag * (uint32_t)S2 - S1 * S1;
                    Df = (float)D;
                    Vf = (float)V;
                }
                const float Rf = a.thr_abs * __builtin_amdgcn_sqrtf(a.cvar * Vf);
                const float diff = Df - Rf;
                const float G = a.g0 + a.g1 * (float)S1 + a.g2 * Rf + a.g3 * Df;
                if (fabsf(diff) > G) {
                    sg = diff > 0.0f ? (di > 0 ? 1 : -1) : 0;
                } else {
                    sg = exact_signal<KT>(K, tj, ti, lag, a.fsize, a.thr);
                    atomicAdd(&a.counters[(blockIdx.x & (kShards - 1u)) * kShardWords + 1u], 1ull);
                }
            }
            if (want_dense) {
                a.dense_cnt[cg.win_base + i] = kk;
                a.dense_sig[cg.win_base + i] = (int8_t)sg;
            }
        }
        if (want_peaks) {
            const unsigned long long bal = __ballot(sg != 0);
            if (lane == 0) PC[(r << 2) + wv] = (uint32_t)__popcll(bal);
            sigbits |= (uint64_t)(sg & 3) << (2u * r);
        }
    }

    // ---- phase 4: ordered compaction of the tile's peaks ---------------------
    if (want_peaks) {
        __syncthreads();
        const uint32_t ncell = R << 2;
        const uint32_t mine = tid < ncell ? PC[tid] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256<uint32_t>(mine, reinterpret_cast<uint32_t *>(scr), tot);
        if (tid < ncell) PC[tid] = ex;
        if (tid == 0) a.tile_cnt[blockIdx.x] = tot;
        const uint32_t base = 0;
        gams_peak_t *const region = a.peaks + (size_t)blockIdx.x * a.tile_cap;
        if (tot) {
            for (uint32_t r = 0; r < R; ++r) {
                const uint32_t code = (uint32_t)(sigbits >> (2u * r)) & 3u;
                const unsigned long long bal = __ballot(code != 0);
                if (code) {
                    const uint32_t i = w0 + (r << 8) + tid;
                    const uint32_t pos = base + PC[(r << 2) + wv] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    if (pos < a.tile_cap) {
                        gams_peak_t pk;
                        pk.ctg = tl.ctg;
                        pk.window = i;
                        pk.gc_count = K[i - wh];
                        pk.signal = code == 1u ? 1 : -1;
                        region[pos] = pk;
                    }
                }
            }
        }
    }
}

// Same evaluation, executed by a whole wavefront for ONE window: the lanes fetch and
// convert the lag counts in parallel; only the two f32 sums stay sequential, in the
// reference's order.  Every lane returns the same signal.  Control flow must be wave-uniform
// at the call.
// carry + x[B] + x[B+1] + ... + x[B+7] (lane values of x), added strictly left to right.
// The eight lane values go to eight SGPRs first (v_readlane, independent of the sum), then one
// plain dependent v_add_f32 per element with an SGPR operand: 8 + 8 issue slots per 8 elements.
// The empty asm pins all eight in SGPRs before the first add, so that no add waits for its own
// v_readlane (the compiler's single-SGPR version needs an s_nop 1 per element).
// Cycles per lag-100 evaluation, alone on a SIMD (tools/dep_latency.hip): 2349, against 4952
// for the row-DPP chain (s_nop 1 + v_add_f32_dpp row_shr:1 per element) this replaced and
// 3733 for the single-SGPR form; a cold first execution on a CU adds ~180 cycles.  The phase
// stamps of the S288c launch show ~4.9k cycles (5.7k before), the stamping wave's wait for its
// own stamp stores included.
// Tried and dropped: terms through LDS with broadcast ds_read_b128 + VGPR adds (8.4k cycles in
// the kernel: the loop control eats the shorter chain), and inlining everything (scratch
// spills under the 64-VGPR cap, 8.2 -> 11.7 us per launch).
template <int B>
__device__ __forceinline__ float seq_add8(float carry, float x) {
    const int xi = __float_as_int(x);
    int t0 = __builtin_amdgcn_readlane(xi, B + 0), t1 = __builtin_amdgcn_readlane(xi, B + 1),
        t2 = __builtin_amdgcn_readlane(xi, B + 2), t3 = __builtin_amdgcn_readlane(xi, B + 3),
        t4 = __builtin_amdgcn_readlane(xi, B + 4), t5 = __builtin_amdgcn_readlane(xi, B + 5),
        t6 = __builtin_amdgcn_readlane(xi, B + 6), t7 = __builtin_amdgcn_readlane(xi, B + 7);
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
    return carry;
}

// Sequential f32 sum of the first `m` lane values of `x` (m <= 64, wave-uniform) on top of
// `carry`.  Lanes past m must hold +0.0f: the sums here are of non-negative terms and start at
// +0.0f, so a padding term leaves them unchanged; whole groups of eight past m are skipped.
// Out of line: one 0.8-KB body shared by the four calls of an evaluation.
__device__ __noinline__ float seq_add_lanes(float carry, float x, uint32_t m) {
    carry = seq_add8<0>(carry, x);
    if (m > 8u) carry = seq_add8<8>(carry, x);
    if (m > 16u) carry = seq_add8<16>(carry, x);
    if (m > 24u) carry = seq_add8<24>(carry, x);
    if (m > 32u) carry = seq_add8<32>(carry, x);
    if (m > 40u) carry = seq_add8<40>(carry, x);
    if (m > 48u) carry = seq_add8<48>(carry, x);
    if (m > 56u) carry = seq_add8<56>(carry, x);
    return carry;
}

// Inlined into the kernel, where K is known to be LDS (ds_read); behind a call boundary it is a
// generic pointer and every count costs a flat load (~1k cycles per evaluation).  The
// sequential part stays out of line (seq_add_lanes) and takes values, not pointers.
__device__ __forceinline__ int exact_signal_wave(const uint8_t *K, uint32_t tj, uint32_t ti, uint32_t n,
                                                 float fsize, float thr) {
    const uint32_t lane = threadIdx.x & 63u;
    const float len = (float)n;
    // lanes past the end contribute +0.0f, which leaves an f32 sum of non-negative terms unchanged
    float sum = 0.0f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const float x = c0 + lane < n ? (float)K[tj + c0 + lane] / fsize : 0.0f;
        sum = seq_add_lanes(sum, x, min(64u, n - c0));                           // stat.rs:3
    }
    const float mean = sum / len;                                                // stat.rs:5
    float sq = 0.0f;
    for (uint32_t c0 = 0; c0 < n; c0 += 64u) {
        const float x = (float)K[tj + min(c0 + lane, n - 1u)] / fsize;
        const float d = x - mean;
        const float dd = c0 + lane < n ? d * d : 0.0f;
        sq = seq_add_lanes(sq, dd, min(64u, n - c0));                            // stat.rs:12
    }
    const float sd = sqrtf(sq / (len - 1.0f));                                   // stat.rs:13
    const float x = (float)K[ti] / fsize;
    if (fabsf(x - mean) > thr * sd) return x > mean ? 1 : -1;                    // stat.rs:36-38
    return 0;
}

// const_sig[k]: the reference's verdict when the lag averaged windows and the
// tested window all have gc count k (homopolymer / N runs: V == 0 and D == 0,
// which the integer decision cannot settle).  One thread per k, exact order.
__global__ void wave_const_table_kernel(int8_t *const_sig, uint32_t size, uint32_t lag, float thr) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > size) return;
    const float fsize = (float)size, len = (float)lag;
    const float x = (float)k / fsize;
    float sum = 0.0f;
    for (uint32_t q = 0; q < lag; ++q) sum = sum + x;
    const float mean = sum / len;
    float sq = 0.0f;
    const float d = x - mean;
    for (uint32_t q = 0; q < lag; ++q) sq = sq + d * d;
    const float sd = sqrtf(sq / (len - 1.0f));
    int sg = 0;
    if (fabsf(x - mean) > thr * sd) sg = x > mean ? 1 : -1;
    const_sig[k] = (int8_t)sg;
}

// The integer decision of wave_fast_kernel without a square root or an int<->float conversion.
// Exact quantities: D = |n*k - S1|, V = n*S2 - S1^2 (integers).  The reference signals iff
// D > R = thr*sqrt(n*V/(n-1)) up to its f32 rounding, bounded by the guard band
// G = c + g*(D + R), c = g0 + g1*S1, g = g2 + g3 (wave_guard_band):
//   D - R >  G  <=>  A := (D*(1-g) - c)/(1+g) > R  <=>  A > 0 and A^2 > R^2
//   R - D >  G  <=>  B := (D*(1+g) + c)/(1-g) < R  <=>  B^2 < R^2
// and with both sides divided by thr^2*n/(n-1) the right-hand side is V itself.  S1, S2, k are
// carried as f32 (integers < 2^24, exact); V is formed with an error-free square
// (P + Plo = S1^2 exactly), so its relative error is <= 2u; the host rounds aA down / aB, gA*,
// gB* up by a few u (wave_squared_band), which covers the roundings of A, B and V.  The two
// comparisons themselves are single fma's: the sign of fl(V - A*|A|) is the sign of the exact
// value.  Returns raw words whose SIGN BITS say: xs = decided signal, xn = decided none,
// dn = k below the mean.
__device__ __forceinline__ void z_decide(float nf, float kkf, float S1f, float S2f, float aA, float aB,
                                         float cA, float cB, uint32_t &xs, uint32_t &xn, uint32_t &dn) {
    const float di = __builtin_fmaf(nf, kkf, -S1f);              // n*k - S1, exact
    const float P = S1f * S1f;
    const float V0 = __builtin_fmaf(nf, S2f, -P);                // n*S2 - P, one rounding
    const float Plo = __builtin_fmaf(S1f, S1f, -P);              // S1^2 - P, exact
    const float V = V0 - Plo;
    const float A = __builtin_fmaf(__builtin_fabsf(di), aA, -cA);
    const float B = __builtin_fmaf(__builtin_fabsf(di), aB, cB);
    xs = __float_as_uint(__builtin_fmaf(-A, __builtin_fabsf(A), V));   // < 0  <=>  A > 0 and A^2 > V
    xn = __float_as_uint(__builtin_fmaf(B, B, -V));                     // < 0  <=>  B^2 < V
    dn = __float_as_uint(di);
}

// =============================================================================
// wave_fast_tile<W,SIZE,STEP,LAG,NT,NTH>: the tile algorithm for 8-bit counts (size <= 255),
// step <= 32 and 24-bit sums (lag*size <= 65535, lag*size^2 < 2^24), i.e. every BASELINE
// configuration.  NTH threads (256; 64 for the step-1 kernels and 128 for the step-5 one of peaks-only
// plans: a tile per one or two waves, see NTH below), W windows per thread; a tile holds NTH*W - lag - 1
// windows when the parameters are baked in, 256*W otherwise.  wave_fast_kernel runs one tile size per launch,
// wave_fast_taper_kernel W = 12 tiles followed by W = 8 and W = 4 tiles (the launch's tail).
//
//   phase 1  tile bytes HBM -> registers (16 B/lane, coalesced; NT: the first and last row with
//            plain loads -- they hold / leave the halo -- the rows between with the streaming hint)
//            -> G/C/g/c flags (v_bitop3, v_add, v_bitop3 per dword; v_dot4 gathers 4 flags) ->
//            16-bit mask per 16-B chunk -> LDS bit stream BM (1 bit per base).
//   phase 2  window counts, rolling over the bit stream: k(w+1) = k(w) + popc(step bits
//            entering) - popc(step bits leaving); stored as bytes K[].  Slot idx holds
//            window vb+idx with vb = w0-lag-1; windows before the ctg start read as 0.
//            Baked: every thread owns W slots and also leaves their (sum k, sum k^2) in PS[].
//   phase 3  thread t owns windows w0 + t*W + [0,W).  S1 = sum k, S2 = sum k^2 over the lag
//            counts in front of its first window: lag / W blocks of PS + lag % W counts of K
//            (baked), or v_dot4 over the packed bytes; then rolled in f32 (exact): S(q+1) =
//            S(q) - K[tW+q] + K[tW+q+lag].  K traffic is whole dwords at an odd dword stride
//            between lanes (W/4 in {1,3,5}): conflict-free.  Branch-free decision in the squared
//            domain (z_decide: no sqrt, no int<->float conversion); three sign bits per window are
//            shifted into accumulators with v_alignbit.  Window i == lag (averages [0,lag):
//            stat.rs:30-31) is redone by its owner.  Windows inside the guard band: const_sig
//            table (homopolymer runs) or exact_signal_wave.
//   phase 4  dense rows through LDS (coalesced stores) and/or per-thread peak counts ->
//            one workgroup scan -> records in window order into the tile's fixed slot.
// =============================================================================
// __launch_bounds__(256, 8): 8 waves per SIMD = 8 workgroups per CU, i.e. at most 64 VGPRs.
// Interleaved A/B (tools/ab.py) on 384 Mb: W = 12 runs 115 -> 105 us with the cap.
// SIZE/STEP/LAG != 0 bake the parameters into the instruction stream (constant bit-field offsets in phase 2);
// LAG == 0 with SIZE/STEP != 0: the lag stays an argument; all 0 = taken from the arguments at run time.
template <int W, int SIZE, int STEP, int LAG, bool NT, int NTH = 256>
// `tiles` and `seq` are kernel parameters of their own, in front of the argument block: with
// -mllvm -amdgpu-kernarg-preload-count the command processor delivers the first kernel-argument
// dwords in SGPRs at wave start, so the tile descriptor's load does not wait for a scalar load of
// the arguments first (two dependent round trips before a workgroup's first sequence load -> one).
__device__ __forceinline__ void wave_fast_tile(const WaveTile tl, const uint8_t *const seq_p, const WaveArgs &a) {
    static_assert(W % 4 == 0 && W <= 28 && (((W / 4) & 1) == 1 || W == 8), "W/4 odd (LDS bank stride), or W = 8 (64-bit reads); one mask bit per window");
    static_assert((SIZE == 0) == (STEP == 0) && (LAG == 0 || STEP != 0), "bake size and step together; lag only with them");
    // NTH threads per workgroup.  256 everywhere except the step-1 kernels, which may run a tile per one or two
    // waves (NTH = 64 / 128, W > 12, baked): every barrier of the tile then waits for fewer waves (see
    // profiles/r03_barrier_skew.txt: the four waves of a workgroup reach a barrier 3,000-4,500 cycles apart).
    static_assert(NTH == 256 || ((NTH == 64 || NTH == 128) && STEP != 0), "narrow workgroups: baked kernels");
    constexpr uint32_t NT_ = (uint32_t)NTH;
    // Windows per tile.  Baked parameters: 256*W - LAG - 1, so that the tile's K slots (its windows
    // plus the lag+1 in front) are exactly 256*W: every thread of phase 2 owns W slots, which are
    // also the block of outgoing counts of the phase-3 thread with the same index.
    // (SIZE and STEP baked, LAG == 0: the lag is an argument -- `gams wave --lag N` keeps the baked kernel; the
    // reference's own benchmark runs 100 / 5 / 200 and 100 / 20 / 50, doc/benchmark/Atha.md:55,276-280)
    const uint32_t TW = STEP != 0 ? NT_ * W - (LAG ? (uint32_t)LAG : a.lag) - 1u : NT_ * W;
    constexpr int WD = W / 4;
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS carve: BM (1 bit per base, 16 per chunk) | scratch (16 words) | K | PS | SG (dense only)
    uint16_t *BM = reinterpret_cast<uint16_t *>(smem);
    const uint32_t *BW = reinterpret_cast<const uint32_t *>(smem);
    uint32_t *scr = reinterpret_cast<uint32_t *>(smem) + ((((a.max_chunks + 8u) >> 1) + 16u + 3u) & ~3u);
    uint8_t *K = reinterpret_cast<uint8_t *>(scr + 16);
    const uint32_t *KW = reinterpret_cast<const uint32_t *>(K);
    // PS[t] = (sum k, sum k^2) over K slots [t*W, t*W + W)   (baked kernels; 256 + 16 entries)
    uint2 *PS = reinterpret_cast<uint2 *>(K + ((NT_ * W + (LAG ? (uint32_t)LAG : a.lag) + 1u + 31u) & ~15u));
    uint16_t *RK = reinterpret_cast<uint16_t *>(PS + NT_ + 16u);   // phase 4b (W > 12): a thread's rank inside its wave
    uint8_t *SG = reinterpret_cast<uint8_t *>(RK + NT_);

    const uint32_t tid = threadIdx.x;
    if (a.stamps != nullptr && threadIdx.x == 0)
        a.stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memrealtime();  // workgroup entry
    const struct { uint64_t seq_off, win_base; uint32_t n_win; } cg = {tl.seq_off, tl.win_base, tl.n_win};
    const uint32_t lag = LAG ? (uint32_t)LAG : a.lag, step = STEP ? (uint32_t)STEP : a.step,
                   size = SIZE ? (uint32_t)SIZE : a.size;
    const uint32_t w0 = tl.w0;
    const uint32_t w1 = min(w0 + TW, cg.n_win);
    const uint32_t nvalid = w1 - w0;
    const int32_t vb = (int32_t)w0 - (int32_t)lag - 1;       // window held by K slot 0 (may be < 0)
    const uint32_t wh = vb > 0 ? (uint32_t)vb : 0u;
    const uint32_t b0 = wh * step;
    const uint32_t a0 = b0 & ~15u;
    const uint32_t b1 = (w1 - 1u) * step + size;
    const uint32_t nchunk = (b1 - a0 + 15u) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(seq_p + cg.seq_off + a0);

    wave_stamp(a, 0);
    // ---- phase 1: load + classify ------------------------------------------
    if constexpr (STEP != 0) {
        // Baked parameters: the tile never has more than NCH chunks, so every thread issues
        // all of its NLD loads back to back with no bounds logic (the bytes past the tile
        // are the next tile's or the seqset's tail slack), classifies, and stores.
        constexpr uint32_t NCH = (NT_ * W * STEP + SIZE + 30u) / 16u + 1u;   // TW + lag + 1 = 256 * W windows' worth
        constexpr uint32_t NLD = (NCH + NT_ - 1u) / NT_;
        uint4 v[NLD];
        // The tile's first row holds the halo (the previous tile's last 1.1 KB) and its last row the bytes
        // the next tile will want as ITS halo: those two rows are read with plain loads -- the last one leaves
        // its lines in the caches, the first one finds them there -- and the rows in between with the streaming
        // hint.  384-Mb launch: 70.4 us all streaming, 69.4 first row plain, 70.9 last row plain, 67.1 both
        // (gpurun_out/r2_ab12.log); the two-round 120-Mb launch does not care (27.2-27.4 us).
        v[0] = load_stream16<false>(src + tid);
#pragma unroll
        for (uint32_t k = 1; k + 1u < NLD; ++k) v[k] = load_stream16<NT>(src + tid + NT_ * k);
        // the last row is only partly inside the largest tile: lanes past it re-read the tile's last
        // chunk (one more request for the same line; their masks land behind the tile's last chunk in
        // BM, where nothing reads).  Unconditional: a predicated load sits in its own basic block and
        // makes hipcc wait vmcnt(0) right behind it -- which the tapered kernel's inlined bodies did.
        v[NLD - 1u] = load_stream16<false>(src + min(tid + NT_ * (NLD - 1u), NCH - 1u));
#pragma unroll
        for (uint32_t k = 0; k < NLD; ++k) BM[tid + NT_ * k] = (uint16_t)gc_mask16(v[k]);
    } else
    // Two batches of four 16-B loads stay in flight per thread: the next batch is issued
    // before the current one is classified.  Loads and LDS stores are unconditional (index
    // clamped to the last chunk / parked on the slot behind the last chunk, never read):
    // a predicated load sits in its own basic block and makes hipcc wait vmcnt(0).
    {
        const uint32_t last = nchunk - 1u;
        uint4 cur[4], nxt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = load_stream16<NT>(src + min(tid + 256u * k, last));
        for (uint32_t c0 = tid; c0 < nchunk; c0 += 1024u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) nxt[k] = load_stream16<NT>(src + min(c0 + 1024u + 256u * k, last));
#pragma unroll
            for (int k = 0; k < 4; ++k) BM[min(c0 + 256u * k, nchunk)] = (uint16_t)gc_mask16(cur[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
        }
    }
    __syncthreads();
    wave_stamp(a, 1);
    wave_stamp(a, 2);

    // ---- phase 2: k of every slot, rolling over the bit stream ------------------
    // Thread t owns a run of consecutive slots.  Its first window is counted directly
    // (popcount of `size` bits); every further window adds the popcount of the `step`
    // bits entering on the right and subtracts the `step` bits leaving on the left:
    // two 64-bit LDS reads, two shifts, two masks, two popcounts per window, for any
    // size and any step <= 32.  No byte prefix is ever built.
    bool counted = false;
    if constexpr (STEP != 0) {
        // Baked parameters, tiles behind the ctg start (vb >= 0): a thread takes RUN
        // consecutive slots, pulls the RUN*STEP + SIZE bits they span into registers once,
        // realigns them to bit 0 (v_alignbit), and every field is then a v_bfe at a constant
        // offset: ~6 VALU per window, no LDS access inside the run, counts stored 4 per dword.
        if (vb >= 0) {
            counted = true;
            constexpr uint32_t NK_MAX = NT_ * W;                    // = lag + 1 + TW
            constexpr uint32_t RUN = (uint32_t)W;
            static_assert(NK_MAX == NT_ * RUN, "phase 2: W slots per thread");
            constexpr uint32_t NBITS = RUN * (uint32_t)STEP + (uint32_t)SIZE;
            constexpr uint32_t NDW = (NBITS + 31u) / 32u + 1u;           // + 1 for the realignment
            const uint32_t nK = lag + 1u + nvalid;
            const uint32_t idx0 = tid * RUN;
            if (idx0 < nK) {
                const uint32_t x0 = ((uint32_t)vb + idx0) * (uint32_t)STEP - a0;
                const uint32_t d0 = x0 >> 5, sh = x0 & 31u;
                uint32_t r[NDW];
#pragma unroll
                for (uint32_t i = 0; i < NDW; ++i) r[i] = BW[d0 + i];
#pragma unroll
                for (uint32_t i = 0; i + 1u < NDW; ++i) r[i] = __builtin_amdgcn_alignbit(r[i + 1u], r[i], sh);
                auto field = [&](uint32_t off) -> uint32_t {           // STEP bits at constant bit `off`
                    const uint32_t d = off >> 5, s = off & 31u;
                    if (s + (uint32_t)STEP <= 32u) return __builtin_amdgcn_ubfe(r[d], s, (uint32_t)STEP);
                    return __builtin_amdgcn_alignbit(r[d + 1u], r[d], s) & ((1u << STEP) - 1u);
                };
                uint32_t k = 0;
#pragma unroll
                for (uint32_t d = 0; d < (uint32_t)SIZE / 32u; ++d) k += __popc(r[d]);
                if constexpr (SIZE % 32 != 0) k += __popc(r[SIZE / 32] & ((1u << (SIZE % 32)) - 1u));
                uint32_t packed = k;
                uint32_t s1 = 0, s2 = 0;                                  // block sums for phase 3
#pragma unroll
                for (uint32_t j = 1; j < RUN; ++j) {
                    k += __popc(field((uint32_t)SIZE + (j - 1u) * (uint32_t)STEP));
                    k -= __popc(field((j - 1u) * (uint32_t)STEP));
                    if ((j & 3u) == 0u) {
                        // (The test is not needed for correctness -- K has room for NTH * W slots and nothing reads
                        // the ones past nK -- but the step-1 kernel is 15 % SLOWER without it, 284 -> 327 us on 384 Mb,
                        // although it is three instructions per store: the load phase of the OTHER tiles on the CU
                        // grows from 6,250 to 9,770 cycles, everything else stays; gpurun_out/r3_stamps_ab3.txt.
                        // Yield points put into the z-score loop instead do nothing, profiles/r03_step1_valu_attempts.txt.)
                        if (idx0 + j - 4u < nK) reinterpret_cast<uint32_t *>(K)[(idx0 + j - 4u) >> 2] = packed;
                        s1 = __builtin_amdgcn_udot4(packed, 0x01010101u, s1, false);
                        s2 = __builtin_amdgcn_udot4(packed, packed, s2, false);
                        packed = k;
                    } else {
                        packed |= k << (8u * (j & 3u));
                    }
                }
                if (idx0 + RUN - 4u < nK) reinterpret_cast<uint32_t *>(K)[(idx0 + RUN - 4u) >> 2] = packed;
                s1 = __builtin_amdgcn_udot4(packed, 0x01010101u, s1, false);
                s2 = __builtin_amdgcn_udot4(packed, packed, s2, false);
                PS[tid] = make_uint2(s1, s2);   // slots past nK only reach windows past the tile's end
            }
        }
    }
    if (!counted) {
        const uint32_t nK = lag + 1u + nvalid;
        const uint32_t run = (nK + NT_ - 1u) / NT_;
        uint32_t idx = tid * run;
        const uint32_t iend = min(idx + run, nK);
        for (; idx < iend && vb + (int32_t)idx < 0; ++idx) K[idx] = 0;   // before the ctg start
        if (idx < iend) {
            uint32_t x = (uint32_t)(vb + (int32_t)idx) * step - a0;      // bit = base position
            // direct count of bits [x, x + size)
            uint32_t k = 0;
            {
                const uint32_t e = x + size;
                for (uint32_t d = x >> 5; d <= (e - 1u) >> 5; ++d) {
                    uint32_t w = BW[d];
                    const uint32_t lo = d << 5;
                    if (lo < x) w &= ~0u << (x - lo);
                    if (lo + 32u > e) w &= ~0u >> (lo + 32u - e);
                    k += __popc(w);
                }
            }
            K[idx++] = (uint8_t)k;
            const uint32_t fmask = step >= 32u ? ~0u : (1u << step) - 1u;
            // four windows per trip: all eight 64-bit LDS reads are issued before any store
            // (K and the bit stream share the LDS array, so the compiler will not reorder them)
            for (; idx + 4u <= iend; idx += 4u, x += 4u * step) {
                uint32_t nl[4], nh[4], ol[4], oh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t po = x + (uint32_t)j * step, pn = po + size;
                    nl[j] = BW[pn >> 5];
                    nh[j] = BW[(pn >> 5) + 1u];
                    ol[j] = BW[po >> 5];
                    oh[j] = BW[(po >> 5) + 1u];
                }
                uint32_t kq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t po = x + (uint32_t)j * step, pn = po + size;
                    const uint64_t wn = ((uint64_t)nh[j] << 32) | nl[j];
                    const uint64_t wo = ((uint64_t)oh[j] << 32) | ol[j];
                    k += __popc((uint32_t)(wn >> (pn & 31u)) & fmask);
                    k -= __popc((uint32_t)(wo >> (po & 31u)) & fmask);
                    kq[j] = k;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) K[idx + j] = (uint8_t)kq[j];
            }
            for (; idx < iend; ++idx, x += step) {
                const uint32_t pn = x + size, po = x;
                const uint64_t wn = ((uint64_t)BW[(pn >> 5) + 1u] << 32) | BW[pn >> 5];
                const uint64_t wo = ((uint64_t)BW[(po >> 5) + 1u] << 32) | BW[po >> 5];
                k += __popc((uint32_t)(wn >> (pn & 31u)) & fmask);
                k -= __popc((uint32_t)(wo >> (po & 31u)) & fmask);
                K[idx] = (uint8_t)k;
            }
        }
    }
    __syncthreads();
    wave_stamp(a, 3);

    // ---- phase 3: rolling sums + decision, W consecutive windows per thread ---
    const bool want_peaks = (a.flags & GAMS_WAVE_PEAKS) != 0;
    const bool want_dense = (a.flags & GAMS_WAVE_DENSE) != 0;
    const uint32_t base = tid * (uint32_t)W;        // K slot of the outgoing count of window q = 0
    uint32_t crest = 0, trough = 0;                  // one bit per owned window
    uint32_t pend = 0;                               // windows whose decision sits inside the guard band
    uint32_t in[WD + 1];                             // bytes K[base+lag .. base+lag+W]: incoming + own counts
    {
        // Every thread runs this block (wave-uniform control flow, no branches in the
        // window loop): threads past the tile's last window read LDS inside the
        // allocation and their decisions are masked off.
        const uint32_t bw = base >> 2;               // dword index (W % 4 == 0)
        uint32_t og[WD];                             // outgoing counts K[base + q]
#pragma unroll
        for (int d = 0; d < WD; ++d) og[d] = KW[bw + d];
        // S1, S2 over K[base, base+lag)
        uint32_t S1 = 0, S2 = 0;
        const uint32_t nfull = lag >> 2;
        const uint32_t sh = lag & 3u;                // wave-uniform
        bool summed = false;
        if constexpr (STEP != 0) {
            // Baked parameters: phase 2 left the sums of every thread's W slots in PS, so the lag
            // counts in front of this thread's windows are LAG / W whole blocks (one 8-byte LDS read
            // each, consecutive lanes on consecutive entries) plus LAG % W counts straight from K.
            if (counted) {
                summed = true;
                if constexpr (LAG != 0) {
                    constexpr uint32_t FULL = (uint32_t)LAG / (uint32_t)W, REM = (uint32_t)LAG % (uint32_t)W;
#pragma unroll
                    for (uint32_t j = 0; j < FULL; ++j) {
                        const uint2 ps = PS[tid + j];
                        S1 += ps.x;
                        S2 += ps.y;
                    }
#pragma unroll
                    for (uint32_t d = 0; d < (REM + 3u) / 4u; ++d) {
                        uint32_t x = KW[bw + FULL * (uint32_t)WD + d];
                        if (d == REM / 4u && (REM & 3u)) x &= (1u << (8u * (REM & 3u))) - 1u;
                        S1 = __builtin_amdgcn_udot4(x, 0x01010101u, S1, false);
                        S2 = __builtin_amdgcn_udot4(x, x, S2, false);
                    }
                } else {
                    // the lag is an argument: the same sums with run-time trip counts.  A thread that owns windows
                    // of the tile never reaches past block 255 (t + lag / W <= 255); threads behind the tile's
                    // last window are masked off and stop at the array's end.
                    const uint32_t full = lag / (uint32_t)W, rem = lag % (uint32_t)W;
                    const uint32_t trips = min(full, NT_ + 16u - tid);     // (masked threads: stay inside PS)
#pragma unroll 4
                    for (uint32_t j = 0; j < trips; ++j) {
                        const uint2 ps = PS[tid + j];
                        S1 += ps.x;
                        S2 += ps.y;
                    }
                    const uint32_t rd = (rem + 3u) >> 2;
                    for (uint32_t d = 0; d < rd; ++d) {
                        uint32_t x = KW[bw + full * (uint32_t)WD + d];
                        if (d == (rem >> 2) && (rem & 3u)) x &= (1u << (8u * (rem & 3u))) - 1u;
                        S1 = __builtin_amdgcn_udot4(x, 0x01010101u, S1, false);
                        S2 = __builtin_amdgcn_udot4(x, x, S2, false);
                    }
                }
            }
        }
        if (!summed) {
#pragma unroll 5
            for (uint32_t d = 0; d < nfull; ++d) {
                const uint32_t x = KW[bw + d];
                S1 = __builtin_amdgcn_udot4(x, 0x01010101u, S1, false);
                S2 = __builtin_amdgcn_udot4(x, x, S2, false);
            }
            if (sh) {
                const uint32_t x = KW[bw + nfull] & ((1u << (8u * sh)) - 1u);
                S1 = __builtin_amdgcn_udot4(x, 0x01010101u, S1, false);
                S2 = __builtin_amdgcn_udot4(x, x, S2, false);
            }
        }
        {
            const uint32_t ib = bw + nfull;
            if constexpr (LAG != 0 && (LAG & 3) == 0) {          // whole dwords (v_alignbyte by 0 is not folded away)
#pragma unroll
                for (int d = 0; d <= WD; ++d) in[d] = KW[ib + d];
            } else {
                uint32_t lo = KW[ib];
#pragma unroll
                for (int d = 0; d <= WD; ++d) {
                    const uint32_t hi = KW[ib + d + 1];
                    in[d] = __builtin_amdgcn_alignbyte(hi, lo, sh);
                    lo = hi;
                }
            }
        }
        // windows this thread may decide: inside the tile, i >= lag, signalling enabled
        uint32_t can = 0;
        if (!a.no_signal && base < nvalid) {
            const uint32_t hi_q = min((uint32_t)W, nvalid - base);            // q < hi_q
            const uint32_t lo_q = w0 + base >= lag ? 0u : min((uint32_t)W, lag - (w0 + base));  // q >= lo_q
            can = ((1u << hi_q) - 1u) & ~((1u << lo_q) - 1u);
        }
        // per window three sign bits (z_decide): sig (decided signal), nos (decided none), dn
        // (n*k < S1).  The S1 term of the guard band uses the thread's upper bound S1 + W*size (S1
        // grows by at most `size` per window): two fma's per thread instead of two per window.
        // S1, S2 and the counts are carried as f32 from here on: integers below 2^24, every
        // update exact (S2 drops the outgoing square before it takes the incoming one).
        uint32_t sigm = 0, nosm = 0, dnm = 0;
        const float nf = (float)lag, aA = a.aA, aB = a.aB;
        const float s1ub = (float)(S1 + (uint32_t)W * size);
        const float cA = __builtin_fmaf(a.gA1, s1ub, a.gA0), cB = __builtin_fmaf(a.gB1, s1ub, a.gB0);
        float S1f = (float)S1, S2f = (float)S2;
        float kinf = (float)(in[0] & 0xFFu);
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const float koutf = (float)((og[q >> 2] >> (8 * (q & 3))) & 0xFFu);              // v_cvt_f32_ubyteN
            const float kkf = (float)((in[(q + 1) >> 2] >> (8 * ((q + 1) & 3))) & 0xFFu);
            uint32_t xs, xn, dn;
            z_decide(nf, kkf, S1f, S2f, aA, aB, cA, cB, xs, xn, dn);
            // one bit per test, shifted in MSB first: (acc << 1) | sign(x) is a single v_alignbit
            sigm = __builtin_amdgcn_alignbit(sigm, xs, 31);
            nosm = __builtin_amdgcn_alignbit(nosm, xn, 31);
            dnm = __builtin_amdgcn_alignbit(dnm, dn, 31);
            S1f = (S1f - koutf) + kinf;
            S2f = __builtin_fmaf(-koutf, koutf, S2f);
            S2f = __builtin_fmaf(kinf, kinf, S2f);
            kinf = kkf;
        }
        // window q sits at bit W-1-q of the accumulators: flip to bit q
        sigm = __brev(sigm) >> (32 - W);
        nosm = __brev(nosm) >> (32 - W);
        dnm = __brev(dnm) >> (32 - W);
        if (!(a.g0 < INFINITY)) sigm = nosm = 0;   // thresholds outside the decidable range (wave_guard_band): every window is exact
        uint32_t decided = sigm | nosm;
        crest = sigm & ~dnm;                     // D > R + G > 0, so di != 0 here
        trough = sigm & dnm;
        // Window i == lag averages windows [0,lag) (stat.rs:30-31) like i == lag+1, not
        // [i-1-lag, i-1): its owner (one thread per ctg) redoes the integer decision with the
        // sums over K[base+qlag+1, base+qlag+1+lag).
        const uint32_t qlag = lag - (w0 + base);                             // wraps when i == lag is not here
        if (qlag < (uint32_t)W) {
            const uint32_t bit = 1u << qlag;
            uint32_t s1 = 0, s2 = 0;
            for (uint32_t j = 0; j < lag; ++j) {
                const uint32_t kv = K[base + qlag + 1u + j];
                s1 += kv;
                s2 += kv * kv;
            }
            const uint32_t kk = K[base + qlag + lag + 1u];
            uint32_t xs, xn, dn;
            z_decide(nf, (float)kk, (float)s1, (float)s2, aA, aB, __builtin_fmaf(a.gA1, (float)s1, a.gA0),
                     __builtin_fmaf(a.gB1, (float)s1, a.gB0), xs, xn, dn);
            const bool sg = (xs >> 31) != 0u, no = (xn >> 31) != 0u, below = (dn >> 31) != 0u;
            const bool exact_all = !(a.g0 < INFINITY);
            decided = ((sg || no) && !exact_all) ? (decided | bit) : (decided & ~bit);
            crest = (sg && !below) ? (crest | bit) : (crest & ~bit);
            trough = (sg && below) ? (trough | bit) : (trough & ~bit);
        }
        crest &= can & decided;
        trough &= can & decided;
        pend = can & ~decided;
        // homopolymer / N runs: all lag averaged counts and the window's own count equal
        // (V == 0 and D == 0, undecidable in integers): settled by the precomputed table.
        // Tested on demand for pending windows only (rare), straight from K.
        if (__ballot(pend != 0u)) {
            uint32_t todo = pend;
            while (todo) {
                const int q = __ffs((int)todo) - 1;
                todo &= todo - 1u;
                const uint32_t i = w0 + base + (uint32_t)q;
                const uint32_t tj = base + (uint32_t)q + (i == lag ? 1u : 0u);
                const uint32_t kk = K[base + (uint32_t)q + lag + 1u];
                bool same = true;
                for (uint32_t j = 0; j < lag && same; ++j) same = K[tj + j] == kk;
                if (same) {
                    const int sg = a.const_sig[kk];
                    pend &= ~(1u << q);
                    crest |= sg > 0 ? 1u << q : 0u;
                    trough |= sg < 0 ? 1u << q : 0u;
                }
            }
        }
    }
    wave_stamp(a, 4);
    // Guard-band windows: exact f32 order, one window at a time by the whole wave.
    {
        const uint32_t lane = tid & 63u;
        unsigned long long bal = __ballot(pend != 0u);
        uint32_t n_exact = 0;
        while (bal) {
            const int L = __ffsll(bal) - 1;                                   // wave-uniform
            const uint32_t pL = (uint32_t)__builtin_amdgcn_readlane((int)pend, L);
            const uint32_t q = (uint32_t)__ffs((int)pL) - 1u;
            const uint32_t baseL = ((tid & ~63u) + (uint32_t)L) * (uint32_t)W;
            const uint32_t first = (w0 + baseL + q == lag) ? 1u : 0u;
            const int sg = exact_signal_wave(K, baseL + q + first, baseL + q + lag + 1u, lag, a.fsize, a.thr);
            if (lane == (uint32_t)L) {
                pend &= ~(1u << q);
                crest |= sg > 0 ? 1u << q : 0u;
                trough |= sg < 0 ? 1u << q : 0u;
            }
            ++n_exact;
            bal = __ballot(pend != 0u);
        }
        if (n_exact && lane == 0)
            atomicAdd(&a.counters[(blockIdx.x & (kShards - 1u)) * kShardWords + 1u], (unsigned long long)n_exact);
    }

    wave_stamp(a, 5);
    // ---- phase 4a: dense rows, coalesced through LDS -------------------------
    if (want_dense) {
        if (base < nvalid) {
#pragma unroll
            for (int q = 0; q < W; ++q)
                SG[base + q] = (uint8_t)(((crest >> q) & 1u) | (((trough >> q) & 1u) ? 0xFFu : 0u));
        }
        __syncthreads();
        // Rows leave four windows per thread: a 16-B store of four counts and a 4-B store of four signals, on
        // groups aligned in the GLOBAL row index (the tile's first row sits anywhere); the bytes come out of LDS
        // as two dwords + v_alignbyte.  The up to three rows in front of the first aligned group and behind the
        // last one are stored one by one.  (One row per thread and trip -- a 4-B and a 1-B store per window --
        // cost 155 us of a 476-us pass over 3.8e8 windows at step 1.)
        const uint64_t g0 = cg.win_base + w0;
        const uint32_t head = min((uint32_t)(0u - (uint32_t)g0) & 3u, nvalid);
        const uint32_t ngrp = (nvalid - head) >> 2;
        const uint32_t *SW = reinterpret_cast<const uint32_t *>(SG);
        for (uint32_t j = tid; j < ngrp; j += NT_) {
            const uint32_t idx = head + 4u * j;
            const uint32_t ka = idx + lag + 1u;
            const uint32_t kc = __builtin_amdgcn_alignbyte(KW[(ka >> 2) + 1u], KW[ka >> 2], ka & 3u);
            const uint32_t sc = __builtin_amdgcn_alignbyte(SW[(idx >> 2) + 1u], SW[idx >> 2], idx & 3u);
            *reinterpret_cast<uint4 *>(a.dense_cnt + g0 + idx) =
                make_uint4(kc & 0xFFu, (kc >> 8) & 0xFFu, (kc >> 16) & 0xFFu, kc >> 24);
            *reinterpret_cast<uint32_t *>(a.dense_sig + g0 + idx) = sc;
        }
        const uint32_t tail0 = head + 4u * ngrp;
        constexpr uint32_t TL = NTH > 64 ? 64u : 8u;               // first lane of the tail rows: another wave than the head's, if there is one
        if (tid < head) {
            a.dense_cnt[g0 + tid] = K[tid + lag + 1u];
            a.dense_sig[g0 + tid] = (int8_t)SG[tid];
        } else if (tid >= TL && tail0 + (tid - TL) < nvalid) {
            const uint32_t idx = tail0 + (tid - TL);
            a.dense_cnt[g0 + idx] = K[idx + lag + 1u];
            a.dense_sig[g0 + idx] = (int8_t)SG[idx];
        }
    }

    // ---- phase 4b: ordered compaction: record order == window order -----------------------
    // W <= 12 (step 10: a thread holds at most a few peaks): every thread writes its own records at
    // its rank in the tile.  W >= 20 (step 1, where a GC crest spans 10-30 adjacent windows, i.e. one
    // or two threads hold all of a wave's peaks and that loop runs W times with 62 lanes idle):
    // every thread publishes its window masks and its rank inside its wave in LDS and the tile's records are
    // dealt out one per thread -- record r finds its wave (three compares with the wave totals), its
    // owner (six-step search over the wave's ranks), its window (the n-th set bit of the owner's
    // mask) and stores 16 B next to its neighbours': a fixed ~80 instructions per 256 records.
    // 384 Mb at step 1: 361 -> 336 us; at step 10 the dealt-out form is no faster (27.4 -> 27.1 us
    // on 120 Mb, 64.8 -> 65.7 on 384 Mb: the LDS search lengthens every workgroup's life), hence W.
    if (want_peaks) {
        const uint32_t both = crest | trough;
        const uint32_t mine = (uint32_t)__popc(both);
        gams_peak_t *const region = a.peaks + (size_t)blockIdx.x * a.tile_cap;
        if constexpr (W <= 12) {
            uint32_t tot;
            const uint32_t ex = block_excl_scan_256<uint32_t, NTH>(mine, scr, tot);
            if (tid == 0) a.tile_cnt[blockIdx.x] = tot;
            if (mine) {
                uint32_t pos = ex;
                uint32_t bits = both;
                while (bits) {
                    const int q = __ffs((int)bits) - 1;
                    bits &= bits - 1u;
                    const uint32_t code = (crest >> q) & 1u;
                    if (pos < a.tile_cap) {
                        gams_peak_t pk;
                        pk.ctg = tl.ctg;
                        pk.window = w0 + base + (uint32_t)q;
                        pk.gc_count = K[base + (uint32_t)q + lag + 1u];
                        pk.signal = code == 1u ? 1 : -1;
                        region[pos] = pk;
                    }
                    ++pos;
                }
            }
        } else {
            const uint32_t inc = wave_incl_scan(mine);
            if ((tid & 63u) == 63u) scr[tid >> 6] = inc;
            __syncthreads();                               // every wave is past phase 3: PS can be reused
            uint2 *const MB = PS;
            MB[tid] = make_uint2(both, crest);
            RK[tid] = (uint16_t)(inc - mine);              // < 64 * W
            const uint32_t c1 = scr[0], c2 = NTH > 64 ? c1 + scr[1] : c1, c3 = NTH > 128 ? c2 + scr[2] : c2,
                           tot = NTH > 128 ? c3 + scr[3] : c3;   // (one or two waves: the missing wave totals count as empty)
            __syncthreads();
            if (tid == 0) a.tile_cnt[blockIdx.x] = tot;
            const uint32_t lim = min(tot, a.tile_cap);
            for (uint32_t r = tid; r < lim; r += NT_) {
                const uint32_t wr = NTH == 64 ? 0u : NTH == 128 ? (r >= c1 ? 1u : 0u)
                                                          : (r >= c1 ? 1u : 0u) + (r >= c2 ? 1u : 0u) + (r >= c3 ? 1u : 0u);
                const uint32_t rl = r - (wr == 3u ? c3 : wr == 2u ? c2 : wr == 1u ? c1 : 0u);
                // the last thread of wave wr whose rank is <= rl (ranks do not decrease; a thread without
                // peaks shares its rank with its successor, so the last one is the owner)
                uint32_t t = wr * 64u;
#pragma unroll
                for (uint32_t s = 32u; s != 0u; s >>= 1)
                    if (RK[t + s] <= rl) t += s;
                const uint2 m = MB[t];
                uint32_t n = rl - RK[t], msk = m.x, q = 0;
#pragma unroll
                for (uint32_t s = 16u; s != 0u; s >>= 1) {  // n-th set bit: skip the low s bits while they hold <= n
                    const uint32_t c = (uint32_t)__popc(msk & ((1u << s) - 1u));
                    const bool skip = n >= c;
                    n -= skip ? c : 0u;
                    msk = skip ? msk >> s : msk;
                    q += skip ? s : 0u;
                }
                gams_peak_t pk;
                pk.ctg = tl.ctg;
                pk.window = w0 + t * (uint32_t)W + q;
                pk.gc_count = K[t * (uint32_t)W + q + lag + 1u];
                pk.signal = ((m.y >> q) & 1u) ? 1 : -1;
                region[r] = pk;
            }
        }
    }
    wave_stamp(a, 6);
}

template <int W, int SIZE, int STEP, int LAG, bool NT, int NTH = 256>
__global__ __launch_bounds__(NTH, 8) void wave_fast_kernel(const WaveTile *const tiles_p, const uint8_t *const seq_p,
                                                           const WaveArgs a) {
    const WaveTile tl = tiles_p[blockIdx.x];         // issued before anything waits for the argument block
    wave_fast_tile<W, SIZE, STEP, LAG, NT, NTH>(tl, seq_p, a);
}

// Tapered launch (baked parameters): the tile table ends in tiles of 8 and then 4 windows per thread
// (WaveTile::pad = W of the tile; all tiles of a ctg have one size).  Workgroups are dispatched in
// blockIdx order, so the launch ends in short-lived workgroups: its tail -- the time the last
// workgroups load and compute alone, ~10 us of a 28-us launch over 120 Mb -- shrinks to the life of
// a W = 4 tile, for 3-4 % more work (the small tiles carry relatively more halo).
template <int SIZE, int STEP, int LAG, bool NT>
__global__ __launch_bounds__(256, 8) void wave_fast_taper_kernel(const WaveTile *const tiles_p,
                                                                 const uint8_t *const seq_p, const WaveArgs a) {
    const WaveTile tl = tiles_p[blockIdx.x];
    if (tl.pad == 12u)
        wave_fast_tile<12, SIZE, STEP, LAG, NT>(tl, seq_p, a);
    else if (tl.pad == 8u)
        wave_fast_tile<8, SIZE, STEP, LAG, NT>(tl, seq_p, a);
    else
        wave_fast_tile<4, SIZE, STEP, LAG, NT>(tl, seq_p, a);
}

// ---- parameters whose halo does not fit a tile (large step or lag): no tiling ---------------
// (lag+1)*step + size + 256*step > 64 KB, or the LDS prefix arrays > 160 KB: one lane per window
// counts its own bases straight from HBM (windows barely overlap at such steps), a second kernel
// evaluates every window in the reference's exact f32 order over the dense counts (no guard band
// to speak of: this IS the exact path), wave_compact_kernel collects the peaks.  Few windows per
// base, so the simple form costs little; bit-exact like everything else.
__device__ __forceinline__ uint32_t wave_ctg_of(const WaveCtgDev *ctgs, uint32_t n_ctg, uint64_t g) {
    uint32_t lo = 0, hi = n_ctg;               // last ctg whose win_base <= g
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ctgs[mid].win_base <= g)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void wave_direct_count_kernel(const uint8_t *seq, const WaveCtgDev *ctgs,
                                                                 uint32_t n_ctg, uint64_t total, uint32_t size,
                                                                 uint32_t step, uint32_t *dense_cnt) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= total) return;
    const WaveCtgDev cg = ctgs[wave_ctg_of(ctgs, n_ctg, g)];
    const uint64_t b0 = cg.seq_off + (g - cg.win_base) * (uint64_t)step, b1 = b0 + size;   // window.rs:78-94
    uint32_t cnt = 0;
    // whole aligned dwords around [b0, b1): bytes outside are masked (the buffer is 256-B aligned
    // in front and has tail slack behind)
    for (uint64_t a = b0 & ~(uint64_t)3; a < b1; a += 4) {
        uint32_t f = gc_flags(*reinterpret_cast<const uint32_t *>(seq + a));
        if (a < b0) f &= ~0u << (8u * (uint32_t)(b0 - a));
        if (a + 4 > b1) f &= ~0u >> (8u * (uint32_t)(a + 4 - b1));
        cnt += (uint32_t)__popc(f);
    }
    dense_cnt[g] = cnt;
}

__global__ __launch_bounds__(256) void wave_direct_signal_kernel(const WaveCtgDev *ctgs, uint32_t n_ctg,
                                                                  uint64_t total, const uint32_t *dense_cnt,
                                                                  int8_t *dense_sig, uint32_t lag, float thr,
                                                                  float fsize, uint32_t no_signal) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= total) return;
    const WaveCtgDev cg = ctgs[wave_ctg_of(ctgs, n_ctg, g)];
    const uint32_t i = (uint32_t)(g - cg.win_base);
    int sg = 0;
    if (!no_signal && i >= lag) {
        const uint32_t tj = i == lag ? 0u : i - 1u - lag;              // stat.rs:30-31 / :51-52
        sg = exact_signal<uint32_t>(dense_cnt + cg.win_base, tj, i, lag, fsize, thr);
    }
    dense_sig[g] = (int8_t)sg;
}

// ---- influence != 1: the filtered[] recurrence is serial per ctg -------------
// One lane per ctg, the reference's loop verbatim (stat.rs:16-56) over the dense
// gc counts a counts-only pass of wave_tile_kernel left in HBM.  `ring` holds
// filtered[] for the ctg (n_win floats, written once, read lag times).
__global__ void wave_serial_kernel(const WaveCtgDev *ctgs, uint32_t n_ctg, const uint32_t *dense_cnt,
                                   int8_t *dense_sig, float *filtered, uint32_t lag, float thr,
                                   float influence, float fsize) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ctg) return;
    const WaveCtgDev cg = ctgs[c];
    const uint32_t n = cg.n_win;
    const uint32_t *k = dense_cnt + cg.win_base;
    int8_t *sig = dense_sig + cg.win_base;
    float *f = filtered + cg.win_base;
    const float len = (float)lag;
    for (uint32_t i = 0; i < n; ++i) {
        f[i] = (float)k[i] / fsize;   // filtered_data = data.to_owned()  (stat.rs:21)
        sig[i] = 0;
    }
    if (n < lag || lag == 0) return;
    float avg, sd;
    {
        float sum = 0.0f;
        for (uint32_t q = 0; q < lag; ++q) sum = sum + f[q];
        avg = sum / len;
        float sq = 0.0f;
        for (uint32_t q = 0; q < lag; ++q) {
            const float d = f[q] - avg;
            sq = sq + d * d;
        }
        sd = sqrtf(sq / (len - 1.0f));
    }
    for (uint32_t i = lag; i < n; ++i) {
        const float x = (float)k[i] / fsize;
        if (fabsf(x - avg) > thr * sd) {                               // stat.rs:36
            sig[i] = x > avg ? 1 : -1;
            const float a = influence * x;
            const float b = (1.0f - influence) * f[i - 1];
            f[i] = a + b;                                              // stat.rs:42
        } else {
            f[i] = x;
        }
        float sum = 0.0f;
        for (uint32_t q = i - lag; q < i; ++q) sum = sum + f[q];       // stat.rs:51
        avg = sum / len;
        float sq = 0.0f;
        for (uint32_t q = i - lag; q < i; ++q) {                       // stat.rs:52
            const float d = f[q] - avg;
            sq = sq + d * d;
        }
        sd = sqrtf(sq / (len - 1.0f));
    }
}

// The same recurrence with one WAVEFRONT per ctg (lag + 1 <= kSerialRing): the last lag + 1 values of
// filtered[] live in an LDS ring, the lanes fetch the lag values of a window in parallel and only the
// two f32 sums stay sequential (seq_add_lanes: the reference's left-to-right order), so a window costs
// ~2.4k cycles at lag 100 instead of 2 * lag dependent global loads on a single lane (a 500-kb ctg at
// step 10 took minutes).  Every lane computes the same decision; lane 0 stores.
constexpr uint32_t kSerialRing = 16384;   // floats of LDS ring (64 KiB)
__global__ __launch_bounds__(64) void wave_serial_wave_kernel(const WaveCtgDev *ctgs, uint32_t n_ctg,
                                                              const uint32_t *dense_cnt, int8_t *dense_sig,
                                                              uint32_t lag, float thr, float influence,
                                                              float fsize) {
    extern __shared__ float ring[];       // ring[q % R] = filtered[q], R = lag + 1 rounded up to 64
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    if (c >= n_ctg) return;
    const WaveCtgDev cg = ctgs[c];
    const uint32_t n = cg.n_win;
    const uint32_t *k = dense_cnt + cg.win_base;
    int8_t *sig = dense_sig + cg.win_base;
    const uint32_t R = (lag + 1u + 63u) & ~63u;
    for (uint32_t i = lane; i < n; i += 64u) sig[i] = 0;
    if (n < lag || lag == 0) return;
    for (uint32_t q = lane; q < lag; q += 64u) ring[q] = (float)k[q] / fsize;   // filtered = data (stat.rs:21)
    __builtin_amdgcn_s_waitcnt(0xc07f);                                          // lgkmcnt(0): one wave, no barrier needed
    const float len = (float)lag;
    // mean and sample sd of filtered[first, first + lag), in the reference's order (stat.rs:1-14)
    auto stats = [&](uint32_t first, float &avg, float &sd) {
        float sum = 0.0f;
        for (uint32_t c0 = 0; c0 < lag; c0 += 64u) {
            const float x = c0 + lane < lag ? ring[(first + c0 + lane) % R] : 0.0f;
            sum = seq_add_lanes(sum, x, min(64u, lag - c0));
        }
        avg = sum / len;
        float sq = 0.0f;
        for (uint32_t c0 = 0; c0 < lag; c0 += 64u) {
            const float x = ring[(first + min(c0 + lane, lag - 1u)) % R];
            const float d = x - avg;
            const float dd = c0 + lane < lag ? d * d : 0.0f;
            sq = seq_add_lanes(sq, dd, min(64u, lag - c0));
        }
        sd = sqrtf(sq / (len - 1.0f));
    };
    float avg, sd;
    stats(0u, avg, sd);                                                          // stat.rs:30-31
    for (uint32_t i = lag; i < n; ++i) {
        const float x = (float)k[i] / fsize;
        float f = x;
        int sg = 0;
        if (fabsf(x - avg) > thr * sd) {                                         // stat.rs:36
            sg = x > avg ? 1 : -1;
            const float a = influence * x;
            const float b = (1.0f - influence) * ring[(i - 1u) % R];
            f = a + b;                                                           // stat.rs:42
        }
        if (lane == 0) {
            if (sg) sig[i] = (int8_t)sg;
            ring[i % R] = f;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        stats(i - lag, avg, sd);                                                 // stat.rs:51-52: [i - lag, i)
    }
}

// Ordered compaction of dense signals (serial path): same tile/offset scheme as
// phase 4 of wave_tile_kernel.
__global__ __launch_bounds__(256) void wave_compact_kernel(const WaveCtgDev *ctgs, const WaveTile *tiles,
                                                            uint32_t tw, const uint32_t *dense_cnt,
                                                            const int8_t *dense_sig, gams_peak_t *peaks,
                                                            uint32_t tile_cap, uint32_t *tile_cnt) {
    __shared__ uint32_t PC[132];
    __shared__ uint64_t scr[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const WaveTile tl = tiles[blockIdx.x];
    const struct { uint64_t win_base; uint32_t n_win; } cg = {tl.win_base, tl.n_win};
    const uint32_t w0 = tl.w0, w1 = min(w0 + tw, cg.n_win);
    const uint32_t R = (tw + 255u) >> 8;       // the baked fast tiles hold 256*W - lag - 1 windows: not a multiple of 256
    uint64_t sigbits = 0;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t i = w0 + (r << 8) + tid;
        int sg = 0;
        if (i < w1) sg = dense_sig[cg.win_base + i];
        const unsigned long long bal = __ballot(sg != 0);
        if (lane == 0) PC[(r << 2) + wv] = (uint32_t)__popcll(bal);
        sigbits |= (uint64_t)(sg & 3) << (2u * r);
    }
    __syncthreads();
    const uint32_t ncell = R << 2;
    const uint32_t mine = tid < ncell ? PC[tid] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_scan_256<uint32_t>(mine, reinterpret_cast<uint32_t *>(scr), tot);
    if (tid < ncell) PC[tid] = ex;
    if (tid == 0) tile_cnt[blockIdx.x] = tot;
    const uint32_t base = 0;
    if (!tot) return;
    peaks += (size_t)blockIdx.x * tile_cap;
    for (uint32_t r = 0; r < R; ++r) {
        const uint32_t code = (uint32_t)(sigbits >> (2u * r)) & 3u;
        const unsigned long long bal = __ballot(code != 0);
        if (code) {
            const uint32_t i = w0 + (r << 8) + tid;
            const uint32_t pos = base + PC[(r << 2) + wv] + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (pos < tile_cap) {
                gams_peak_t pk;
                pk.ctg = tl.ctg;
                pk.window = i;
                pk.gc_count = dense_cnt[cg.win_base + i];
                pk.signal = code == 1u ? 1 : -1;
                peaks[pos] = pk;
            }
        }
    }
}

// Exclusive prefix of the per-tile peak counts -> tile_off; totals[0] = all peaks, totals[1] = the
// fullest tile.  One workgroup of 1024 lanes, each summing a contiguous run of tiles.
__global__ __launch_bounds__(1024) void wave_offsets_kernel(const uint32_t *tile_cnt, uint32_t nt,
                                                            unsigned long long *tile_off,
                                                            unsigned long long *totals) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t wmax[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (nt + 1023u) / 1024u;
    const uint32_t t0 = min(nt, tid * per), t1 = min(nt, t0 + per);
    unsigned long long mine = 0;
    uint32_t mx = 0;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = tile_cnt[t];
        mine += c;
        mx = max(mx, c);
    }
    unsigned long long inc = wave_incl_scan_u64(mine);
    for (int d = 32; d; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if (lane == 63) wsum[wv] = inc;
    if (lane == 0) wmax[wv] = mx;
    __syncthreads();
    unsigned long long base = 0, all = 0;
    uint32_t worst = 0;
    for (uint32_t w = 0; w < 16; ++w) {
        if (w < wv) base += wsum[w];
        all += wsum[w];
        worst = max(worst, wmax[w]);
    }
    unsigned long long off = base + inc - mine;
    for (uint32_t t = t0; t < t1; ++t) {
        tile_off[t] = off;
        off += tile_cnt[t];
    }
    if (tid == 0) {
        totals[0] = all;
        totals[1] = worst;
    }
}

// The same for tables too long for one workgroup (a tile per wave at step 1: 1.8 million tiles for a human genome, where
// the kernel above -- every lane walking its own 1,800 counts, a cache line apart from its neighbour's -- took 5.8 ms,
// more than twice the pass itself): spans of kOffSpan counts, three launches.  (1) a workgroup per span: its sum and
// its maximum, packed into the span's first offset word (sum below bit 40, maximum above: a span holds at most
// 8192 x 65535 records); (2) one workgroup: exclusive prefix over the span words, in place, and the two totals;
// (3) a workgroup per span: the offsets, from the span's base.  Every load and store is coalesced.
constexpr uint32_t kOffSpan = 8192;
constexpr unsigned long long kOffSumMask = (1ull << 40) - 1ull;

// exclusive prefix of v over the 1024 threads of the workgroup; total = the sum over all of them
__device__ __forceinline__ unsigned long long block_excl_scan_1024(unsigned long long v, unsigned long long *wsum,
                                                                   unsigned long long &total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long inc = wave_incl_scan_u64(v);
    __syncthreads();                                    // (wsum may still be read from the previous call)
    if (lane == 63u) wsum[wv] = inc;
    __syncthreads();
    unsigned long long base = 0, all = 0;
    for (uint32_t w = 0; w < 16u; ++w) {
        const unsigned long long x = wsum[w];
        if (w < wv) base += x;
        all += x;
    }
    total = all;
    return base + inc - v;
}

__global__ __launch_bounds__(1024) void wave_offsets_sum_kernel(const uint32_t *cnt, uint32_t n, unsigned long long *off) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t wmax[16];
    const uint32_t tid = threadIdx.x, b0 = blockIdx.x * kOffSpan;
    unsigned long long mine = 0;
    uint32_t mx = 0;
    for (uint32_t i = 0; i < kOffSpan / 1024u; ++i) {
        const uint32_t t = b0 + i * 1024u + tid;
        const uint32_t c = t < n ? cnt[t] : 0u;
        mine += c;
        mx = max(mx, c);
    }
    for (int d = 32; d; d >>= 1) {
        mine += (unsigned long long)__shfl_xor((long long)mine, d, 64);
        mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    }
    if ((tid & 63u) == 0u) {
        wsum[tid >> 6] = mine;
        wmax[tid >> 6] = mx;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long all = 0;
        uint32_t worst = 0;
        for (uint32_t w = 0; w < 16u; ++w) {
            all += wsum[w];
            worst = max(worst, wmax[w]);
        }
        off[b0] = all | ((unsigned long long)worst << 40);
    }
}

__global__ __launch_bounds__(1024) void wave_offsets_base_kernel(uint32_t n, unsigned long long *off,
                                                                 unsigned long long *totals) {
    __shared__ unsigned long long wsum[16];
    __shared__ uint32_t wmax[16];
    const uint32_t tid = threadIdx.x, nb = (n + kOffSpan - 1u) / kOffSpan;
    unsigned long long running = 0;
    uint32_t mx = 0;
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024u) {
        const uint32_t b = c0 + tid;
        const unsigned long long word = b < nb ? off[(size_t)b * kOffSpan] : 0ull;
        mx = max(mx, (uint32_t)(word >> 40));
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan_1024(word & kOffSumMask, wsum, tot);
        if (b < nb) off[(size_t)b * kOffSpan] = running + ex;
        running += tot;
    }
    for (int d = 32; d; d >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, d, 64));
    if ((tid & 63u) == 0u) wmax[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        uint32_t worst = 0;
        for (uint32_t w = 0; w < 16u; ++w) worst = max(worst, wmax[w]);
        totals[0] = running;
        totals[1] = worst;
    }
}

__global__ __launch_bounds__(1024) void wave_offsets_scan_kernel(const uint32_t *cnt, uint32_t n, unsigned long long *off) {
    __shared__ unsigned long long wsum[16];
    const uint32_t tid = threadIdx.x, b0 = blockIdx.x * kOffSpan;
    unsigned long long running = off[b0];               // every thread, before anybody overwrites it (the scan's first barrier)
    for (uint32_t i = 0; i < kOffSpan / 1024u; ++i) {
        const uint32_t t = b0 + i * 1024u + tid;
        const uint32_t c = t < n ? cnt[t] : 0u;
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan_1024(c, wsum, tot);
        if (t < n) off[t] = running + ex;
        running += tot;
    }
}

// Pack the per-tile slots into one dense, (ctg, window)-ordered array: one wave per tile.  A tile
// that would not fit `cap` records (or overflowed its slot) is skipped: the host regrows and repeats.
__global__ __launch_bounds__(64) void wave_gather_kernel(const gams_peak_t *slots, uint32_t tile_cap,
                                                         const uint32_t *tile_cnt, const unsigned long long *tile_off,
                                                         gams_peak_t *dense, unsigned long long cap) {
    const uint32_t t = blockIdx.x;
    const uint32_t n = tile_cnt[t];
    const unsigned long long o = tile_off[t];
    if (n > tile_cap || o + n > cap) return;
    const gams_peak_t *src = slots + (size_t)t * tile_cap;
    gams_peak_t *dst = dense + o;
    for (uint32_t i = threadIdx.x; i < n; i += 64u) dst[i] = src[i];
}

}  // namespace
