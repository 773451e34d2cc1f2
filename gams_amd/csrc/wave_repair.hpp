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
constexpr uint32_t kJacMaxSweeps0 = 240;    // the same for influence == 0 (jac0_* kernels: a sweep costs no run walk there)
constexpr uint32_t kJacRunCap = 8192;       // a run of signalled windows longer than this: the same
constexpr uint32_t kJacWords = 256;         // control words per way: flips[0 .. kJacMaxSweeps0), [kJacAbandon]
constexpr uint32_t kJacAbandon = 255;
constexpr unsigned long long kJacNoFreeze = ~0ull;
constexpr uint32_t kJac0Group = 8;          // tiles per workgroup of jac0_scan / jac0_fill (a tile is 256 windows of one ctg)

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
    // influence == 0 (jac0_* kernels)
    int32_t *lastu;                 // per row: the last window <= i of the tile (jac0_scan) that does not signal, -1: none
    int32_t *tile_last;             // per tile: its last window that does not signal, -1: every window of the tile signals
    unsigned long long *freeze;     // [2][n_ctg], by sweep parity: (first window whose run of signals has reached `lag`) << 16
                                    // | the count at the run's foot; kJacNoFreeze: no such run
    const int8_t *ftab;             // (size + 1)^2: the decision for count k after `lag` copies of count kc, [kc * size1 + k]
    const uint8_t *frow;            // per kc: 1 while ftab[kc][k] == 0 only at k == kc
    uint32_t size1, n_ctg;
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
    if (a.freeze != nullptr && t.w0 == 0u && threadIdx.x < 2u) a.freeze[threadIdx.x * a.n_ctg + t.ctg] = kJacNoFreeze;
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

// inclusive maximum over the 256 threads of the workgroup up to each thread; total = the workgroup's maximum
__device__ __forceinline__ int jac0_block_max_scan(int v, int *ws, int &total) {
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if ((int)lane >= d) v = max(v, o);
    }
    if (lane == 63u) ws[wv] = v;
    __syncthreads();
    const int w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];
    __syncthreads();
    const int before = wv == 0 ? -1 : wv == 1 ? w0 : wv == 2 ? max(w0, w1) : max(max(w0, w1), w2);
    total = max(max(w0, w1), max(w2, w3));
    return max(v, before);
}

constexpr uint32_t kJac0InnerFrom = 0;     // influence 0: from this sweep on a dirty tile iterates on itself ...
constexpr uint32_t kJac0Inner = 32;         // ... up to this many times per sweep

// every window of a tile whose history changed: the reference's decision from filtered[] (stat.rs:30-38, :51-52)
__global__ __launch_bounds__(256) void jac_eval_kernel(const JacArgs a) {
    extern __shared__ float L[];                               // filtered[w0 - 1 - lag .. w0 + 254]
    __shared__ uint32_t flips_wg;
    __shared__ float X[kJacTile];                              // influence 0, inner iterations: the tile's data values
    __shared__ int ws[4];
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
    const bool tested = i < t.n_win && i >= lag;
    const float x = i < t.n_win ? a.xtab[a.cnt[t.win_base + i]] : 0.0f;
    auto decide = [&]() -> int {
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
        int s = 0;
        if (fabsf(x - mean) > a.thr * sd) s = x > mean ? 1 : -1;                               // stat.rs:36-38
        return s;
    };
    int s = tested ? decide() : 0;
    // Influence 0, late sweeps: what is left then are the runs in front of the freeze points, which a sweep extends by
    // the few windows whose history has just become right.  The tile repeats the two steps on itself -- filtered[] of its
    // own windows from the decisions just made (a fill-forward inside the tile, in front of it L[lag] as it stands),
    // then the decisions again -- until nothing changes or kJac0Inner times: another guess, like every state of the
    // iteration; the flips are counted against the signals the sweep started from.
    if (a.lastu != nullptr && a.sweep >= kJac0InnerFrom) {
        X[tid] = x;
        for (uint32_t it = 1; it < kJac0Inner; ++it) {
            int tot;
            const int last = jac0_block_max_scan(i < t.n_win && s == 0 ? (int)tid : -1, ws, tot);
            const float fnew = last >= 0 ? X[last] : L[lag];
            __syncthreads();
            L[lag + 1u + tid] = fnew;
            __syncthreads();
            const int s2 = tested ? decide() : 0;
            const int changed = s2 != s;
            s = s2;
            if (!__syncthreads_or(changed)) break;
        }
    }
    if (tested) {
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

// ---- influence == 0 ---------------------------------------------------------------------------------------------
// stat.rs:42 with influence 0 is filtered[i] = filtered[i-1]: a signalled window repeats the value in front of it, so
// filtered[i] is the data of the LAST WINDOW <= i THAT DID NOT SIGNAL -- a fill-forward, i.e. a max-scan of indices,
// however long the runs are (jac_filter_kernel walks them, and gives up at kJacRunCap).  And once `lag` windows in a
// row have signalled, the averaged history of the next window is `lag` copies of one value c, it signals unless its
// own value is c (|x - mean| against thr * sd with sd ~ 1e-8: the rounding of a sum of equal terms), filtered stays c
// either way, and so on to the end of the ctg: behind such a point the reference's answer is a function of two counts,
// the one at the foot of the run and the window's own (ftab, filled by jac0_table_kernel with the reference's own
// arithmetic on `lag` equal values).  That is the dense regime -- low thresholds, signals nearly everywhere -- in which
// plain sweeps crawl (a run grows by one decision per sweep once its history is all c: 400 sweeps did not settle a
// 50,000-window ctg in tools/experiments/freeze_model.py) and round 2's one-wavefront-per-ctg kernel took 230 ms per
// 30-Mb chromosome.  Here every sweep first REPLACES the signals behind the first such point of a ctg (as the previous
// sweep's filtered[] located it) by the table's -- a guess like any other: the sweep that follows evaluates every window
// whose signal or history changed, and only a sweep that neither guesses nor flips anything ends the iteration, so the
// fixed point reached is the reference's (the correct prefix of a ctg still grows by at least one window per sweep: a
// guess only touches windows behind a run that lies behind it, or is the true freeze).  The model settles in 7-45 sweeps.
__global__ __launch_bounds__(256) void jac0_table_kernel(int8_t *ftab, uint8_t *frow, const float *xtab, uint32_t size1,
                                                         uint32_t lag, float thr) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= size1 * size1) return;
    const uint32_t kc = q / size1, k = q - kc * size1;
    const float c = xtab[kc], x = xtab[k], len = (float)lag;
    float sum = 0.0f;
    for (uint32_t j = 0; j < lag; ++j) sum = sum + c;                                          // stat.rs:3
    const float mean = sum / len;                                                              // stat.rs:5
    float sq = 0.0f;
    for (uint32_t j = 0; j < lag; ++j) {
        const float d = c - mean;
        sq = sq + d * d;                                                                       // stat.rs:12
    }
    const float sd = sqrtf(sq / (len - 1.0f));                                                 // stat.rs:13
    int sg = 0;
    if (fabsf(x - mean) > thr * sd) sg = x > mean ? 1 : -1;                                    // stat.rs:36-38
    ftab[q] = (int8_t)sg;
    if (sg == 0 && k != kc) frow[kc] = 0;          // (frow starts as all ones; every writer stores the same value)
}

// (1) the guess behind the ctg's freeze point, then the last window of the tile, up to each lane, that does not signal
__global__ __launch_bounds__(256) void jac0_scan_kernel(const JacArgs a) {
    __shared__ int ws[4];
    __shared__ uint32_t changed_wg;
    if (jac_done(a)) return;
    const uint32_t tid = threadIdx.x;
    if (tid == 0u) changed_wg = 0u;                     // (the scans below hold the barriers that order this)
    for (uint32_t tile = blockIdx.x * kJac0Group; tile < min(a.n_tiles, (blockIdx.x + 1u) * kJac0Group); ++tile) {
    const JacTile t = a.tiles[tile];
    const uint32_t i = t.w0 + tid;
    const bool in = i < t.n_win;
    const unsigned long long fz = a.freeze[((a.sweep + 1u) & 1u) * a.n_ctg + t.ctg];       // written by the sweep before
    if (t.w0 == 0u && tid == 0u) a.freeze[(a.sweep & 1u) * a.n_ctg + t.ctg] = kJacNoFreeze;    // this sweep's, for jac0_fill
    int s = 0;
    uint32_t changed = 0u;
    if (in) {
        int8_t *sp = a.sig + t.win_base + i;
        s = *sp;
        if (fz != kJacNoFreeze) {
            const uint32_t p = (uint32_t)(fz >> 16), kc = (uint32_t)(fz & 0xFFFFull);
            if (i > p && a.frow[kc]) {
                const int g = a.ftab[kc * a.size1 + a.cnt[t.win_base + i]];
                if (g != s) {
                    *sp = (int8_t)g;
                    s = g;
                    changed = 1u;
                    a.fblk[(t.win_base + i) / kJacTile] = a.sweep + 1u;    // its signal changed: evaluate it again
                }
            }
        }
    }
    int tot;
    const int last = jac0_block_max_scan(in && s == 0 ? (int)i : -1, ws, tot);
    if (in) a.lastu[t.win_base + i] = last;
    if (tid == 0u) a.tile_last[tile] = tot;
    const unsigned long long bal = __ballot(changed != 0u);
    if ((tid & 63u) == 0u && bal) atomicAdd(&changed_wg, (uint32_t)__popcll(bal));
    }
    __syncthreads();
    // one word takes ~88 atomics per microsecond: one per workgroup, not one per wave and tile
    if (tid == 0u && changed_wg) atomicAdd(&a.ctl[a.sweep], (unsigned long long)changed_wg);
}

// (2) filtered[] = the data of the last window that did not signal (in the tile, else in the tiles in front of it);
//     where a run of signals reaches `lag` windows: a candidate for the ctg's freeze point
__global__ __launch_bounds__(256) void jac0_fill_kernel(const JacArgs a) {
    __shared__ int carry_s;
    __shared__ unsigned long long wmin[4];
    if (jac_done(a)) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t tile = blockIdx.x * kJac0Group; tile < min(a.n_tiles, (blockIdx.x + 1u) * kJac0Group); ++tile) {
    const JacTile t = a.tiles[tile];
    const uint32_t i = t.w0 + tid;
    __syncthreads();                                   // (carry_s / wmin of the tile before)
    if (tid < 64u) {
        // the ctg's tiles are consecutive in the table: look back 64 at a time (all but a few tiles hold a window that does
        // not signal -- behind a freeze point those whose count is the frozen one)
        const int nb = (int)(t.w0 / kJacTile);
        int carry = -1;
        for (int base = 1; base <= nb && carry < 0; base += 64) {
            const int j = base + (int)lane;
            const int v = j <= nb ? a.tile_last[tile - (uint32_t)j] : -1;
            const unsigned long long m = __ballot(v >= 0);
            if (m) carry = __shfl(v, __ffsll((long long)m) - 1, 64);
        }
        if (lane == 0u) carry_s = carry;
    }
    __syncthreads();
    unsigned long long cand = kJacNoFreeze;
    if (i < t.n_win) {
        const uint32_t *k = a.cnt + t.win_base;
        const int own = a.lastu[t.win_base + i];
        const int L = own >= 0 ? own : carry_s;                    // (>= 0: the first `lag` windows of a ctg never signal)
        const uint32_t kl = k[L >= 0 ? (uint32_t)L : i];
        const float v = a.xtab[kl];
        float *fp = a.f + t.win_base + i;
        if (__float_as_uint(*fp) != __float_as_uint(v)) {
            *fp = v;
            a.fblk[(t.win_base + i) / kJacTile] = a.sweep + 1u;
        }
        if (L >= 0 && i - (uint32_t)L >= a.lag) cand = ((unsigned long long)i << 16) | kl;
    }
    for (int d = 32; d; d >>= 1) {
        const unsigned long long o = (unsigned long long)__shfl_xor((long long)cand, d, 64);
        cand = o < cand ? o : cand;
    }
    if (lane == 0u) wmin[tid >> 6] = cand;
    __syncthreads();
    if (tid == 0u) {
        unsigned long long m = wmin[0];
        for (int w = 1; w < 4; ++w) m = wmin[w] < m ? wmin[w] : m;
        if (m != kJacNoFreeze) atomicMin(&a.freeze[(a.sweep & 1u) * a.n_ctg + t.ctg], m);
    }
    }
}

}  // namespace
