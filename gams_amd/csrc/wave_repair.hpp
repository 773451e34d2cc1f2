// wave_repair.hpp -- influence != 1 by guess-and-iterate (included by wave.hip only).
//
// stat.rs:16-56 with influence != 1: filtered[i] = influence*x[i] + (1-influence)*filtered[i-1] at a
// signalled window, x[i] otherwise, and window i is tested against mean / sd of filtered[i-1-lag, i-1)
// -- a serial recurrence per ctg.  Two observations make it parallel:
//
//   * GIVEN the signals, filtered[] is cheap: it is the data except inside runs of consecutive
//     signalled windows, where it is a first-order recurrence from the run's start (jac_filter_kernel:
//     one lane per run), and GIVEN filtered[], every window's decision is independent of every other
//     one's (jac_eval_kernel: one lane per window, the lag values of its history out of an LDS tile, the
//     two sums strictly left to right in the reference's f32 order);
//   * the map "signals -> filtered -> signals" has the reference's answer as its ONLY fixed point, and
//     iterating it from any guess reaches it: if the guess is right on all windows before p, filtered is
//     right before p, so window p comes out right and the windows before it stay right (induction from
//     the left; at least one more window per sweep, in practice nearly all of them).
//
// The guess is what the influence == 1 kernels produce (pass "S1": the plan's ordinary kernels write counts
// and S1 signals to the dense rows); filtered differs from the data only at signalled windows, so S1 is
// right wherever no signal lies within lag + 1 windows in front.  On an A. thaliana-shaped genome the
// sweep settles in 3-5 rounds for every threshold from 1 to 3 at influence 0.5 (CPU model,
// tools/experiments/jacobi_model.py; round 3's first form -- one lane per "zone" behind a signal, kept as
// tools/experiments/r03_zone_walk_repair.hpp, 2.8 ms per 30-Mb chromosome -- needed the zones to be
// separable and fell back to a serial walk at threshold 2).  What stays slow is influence 0 with signals
// nearly everywhere (a window that signalled keeps the old baseline forever, so runs of thousands of
// windows grow by ~lag/3 windows per sweep): after kJacMaxSweeps sweeps, or at the first run longer than
// kJacRunCap, the host hands the batch to the one-wavefront-per-ctg recurrence of round 2 -- the exact
// answer either way.
//
// Only sweeps after the first look at tiles whose history changed (per block of 256 rows: the sweep in
// which filtered last changed there).  Termination is exact: a sweep that flips no signal has reproduced
// its own input.
#pragma once

#include "wave_kernels.hpp"

namespace {

constexpr uint32_t kJacTile = 256;          // windows per workgroup = rows per dirty block
constexpr uint32_t kJacMaxSweeps = 48;      // then the serial recurrence takes over
constexpr uint32_t kJacRunCap = 8192;       // a run of signalled windows longer than this: the same
constexpr uint32_t kJacWords = 64;          // control words per way: flips[0 .. kJacMaxSweeps), [kJacAbandon]
constexpr uint32_t kJacAbandon = 63;

struct JacTile {
    uint32_t ctg, w0;      // first window of the tile
    uint32_t n_win;        // windows of the ctg
    uint32_t pad;
    uint64_t win_base;     // the ctg's first row in the dense arrays
};

struct JacArgs {
    const JacTile *tiles;
    uint32_t n_tiles;
    const uint32_t *cnt;        // dense gc counts
    int8_t *sig;                // dense signals: the S1 guess on entry, the reference's answer at the fixed point
    const float *xtab;          // xtab[k] = k as f32 / size as f32 (the data value of a window with count k)
    float *f;                   // filtered[], one per row
    uint32_t *fblk;             // per block of kJacTile rows: 1 + the sweep in which filtered last changed there
    unsigned long long *ctl;    // flips per sweep, abandon flag
    uint32_t lag, sweep;
    float thr, influence;
};

// the sweeps queued behind a finished pass (or behind a hopeless one) have nothing to do
__device__ __forceinline__ bool jac_done(const JacArgs &a) {
    if (a.ctl[kJacAbandon] != 0ull) return true;
    return a.sweep > 0u && a.ctl[a.sweep - 1u] == 0ull;
}

// filtered := data, control words cleared (one launch per pass, in front of the sweeps)
__global__ __launch_bounds__(256) void jac_init_kernel(const JacArgs a) {
    if (blockIdx.x == 0 && threadIdx.x < kJacWords) a.ctl[threadIdx.x] = 0ull;
    if (blockIdx.x >= a.n_tiles) return;
    const JacTile t = a.tiles[blockIdx.x];
    const uint32_t i = t.w0 + threadIdx.x;
    if (i < t.n_win) a.f[t.win_base + i] = a.xtab[a.cnt[t.win_base + i]];
    if (threadIdx.x == 0) a.fblk[(t.win_base + t.w0) / kJacTile] = 0u;
    if (threadIdx.x == 1) a.fblk[(t.win_base + min(t.w0 + kJacTile, t.n_win) - 1u) / kJacTile] = 0u;
}

// filtered[] from the signals: a lane per window; the lane of a run's first window walks the run
__global__ __launch_bounds__(256) void jac_filter_kernel(const JacArgs a) {
    if (jac_done(a)) return;
    const JacTile t = a.tiles[blockIdx.x];
    const uint32_t i = t.w0 + threadIdx.x;
    if (i >= t.n_win) return;
    const int8_t *sg = a.sig + t.win_base;
    const uint32_t *k = a.cnt + t.win_base;
    float *f = a.f + t.win_base;
    const uint32_t mark = a.sweep + 1u;
    auto put = [&](uint32_t j, float v) {
        if (__float_as_uint(f[j]) != __float_as_uint(v)) {
            f[j] = v;
            a.fblk[(t.win_base + j) / kJacTile] = mark;       // (every writer of a block stores the same value)
        }
    };
    const int s = sg[i];
    if (s == 0) {
        put(i, a.xtab[k[i]]);                                  // stat.rs:45 (and the untested windows, :21)
        return;
    }
    if (i > 0u && sg[i - 1u] != 0) return;                     // inside a run: its first window's lane writes this one
    // signals only exist from window lag on, so i >= lag >= 2 and filtered[i-1] is the data
    float prev = a.xtab[k[i - 1u]];
    uint32_t j = i;
    for (; j < t.n_win && sg[j] != 0; ++j) {
        if (j - i >= kJacRunCap) {
            a.ctl[kJacAbandon] = 1ull;                         // a run this long: hand the batch to the serial kernel
            return;
        }
        const float t1 = a.influence * a.xtab[k[j]];
        const float t2 = (1.0f - a.influence) * prev;
        prev = t1 + t2;                                        // stat.rs:42
        put(j, prev);
    }
}

// every window of a tile whose history changed: the reference's decision from filtered[] (stat.rs:30-38, :51-52)
__global__ __launch_bounds__(256) void jac_eval_kernel(const JacArgs a) {
    extern __shared__ float L[];                               // filtered[w0 - 1 - lag .. w0 + 254]
    __shared__ uint32_t flips_wg;
    if (jac_done(a)) return;
    const JacTile t = a.tiles[blockIdx.x];
    const uint32_t tid = threadIdx.x, lag = a.lag;
    const int64_t hbase = (int64_t)t.w0 - 1 - (int64_t)lag;    // window held by L[0] (may be < 0)
    const uint32_t nL = kJacTile + lag + 1u;
    if (a.sweep > 0u) {
        // did filtered change, in the sweep just made, anywhere in the rows this tile reads?
        const uint64_t r0 = t.win_base + (uint64_t)(hbase > 0 ? hbase : 0), r1 = t.win_base + min(t.w0 + kJacTile, t.n_win) - 1u;
        bool dirty = false;
        for (uint64_t b = r0 / kJacTile; b <= r1 / kJacTile; ++b) dirty |= a.fblk[b] == a.sweep + 1u;
        if (!dirty) return;                                    // (uniform: every thread reads the same words)
    }
    const float *f = a.f + t.win_base;
    for (uint32_t q = tid; q < nL; q += 256u) {
        const int64_t w = hbase + (int64_t)q;
        L[q] = (w >= 0 && w < (int64_t)t.n_win) ? f[w] : 0.0f;
    }
    if (tid == 0) flips_wg = 0u;
    __syncthreads();
    const uint32_t i = t.w0 + tid;
    uint32_t flipped = 0u;
    if (i < t.n_win && i >= lag) {
        const float *h = L + tid + (i == lag ? 1u : 0u);        // window i == lag averages [0, lag) like window lag + 1
        const float len = (float)lag;
        float sum = 0.0f;
#pragma unroll 8
        for (uint32_t c = 0; c < lag; ++c) sum = sum + h[c];                                   // stat.rs:3
        const float mean = sum / len;                                                          // stat.rs:5
        float sq = 0.0f;
#pragma unroll 8
        for (uint32_t c = 0; c < lag; ++c) {
            const float d = h[c] - mean;
            sq = sq + d * d;                                                                   // stat.rs:12
        }
        const float sd = sqrtf(sq / (len - 1.0f));                                             // stat.rs:13
        const float x = a.xtab[a.cnt[t.win_base + i]];
        int s = 0;
        if (fabsf(x - mean) > a.thr * sd) s = x > mean ? 1 : -1;                               // stat.rs:36-38
        int8_t *sp = a.sig + t.win_base + i;
        if (*sp != (int8_t)s) {
            *sp = (int8_t)s;
            flipped = 1u;
        }
    }
    const unsigned long long bal = __ballot(flipped != 0u);
    if ((tid & 63u) == 0u && bal) atomicAdd(&flips_wg, (uint32_t)__popcll(bal));
    __syncthreads();
    if (tid == 0 && flips_wg) atomicAdd(&a.ctl[a.sweep], (unsigned long long)flips_wg);
}

}  // namespace
