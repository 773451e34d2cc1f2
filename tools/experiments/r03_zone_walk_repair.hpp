// wave_repair.hpp -- influence != 1 by speculate-and-repair (included by wave.hip only).
//
// stat.rs:16-56 with influence != 1: filtered[i] = influence*x[i] + (1-influence)*filtered[i-1] at a
// signalled window, x[i] otherwise, and window i is tested against mean / sd of filtered[i-1-lag, i-1)
// -- a serial recurrence per ctg.  But filtered differs from the data ONLY at signalled windows, so
//
//   (1) a window whose history [i-1-lag, i-1) holds no signalled window is decided exactly as with
//       influence == 1, which the fast kernels already do (pass "S1": dense counts + S1 signals);
//   (2) behind a true signal at j the windows j+1 .. j+lag+1 see a filtered value that is not the
//       data (j+1 only through filtered[j] if it signals itself) and have to be walked in order, in
//       the reference's f32 order; every further true signal in that stretch extends it.  Such a
//       "zone" ends at e = (last true signal) + lag + 2, where the history is clean again;
//   (3) by induction the next true signal at or after e is the next S1 signal at or after e, and it
//       starts the next zone from a clean state.  Every S1 signal is therefore either a zone start
//       or lies inside a zone, and between zones all signals are 0.
//
// A zone depends on nothing but the counts, so zones are evaluated in PARALLEL -- one LANE per zone,
// 64 zones per wavefront in lockstep, each lane running its own sequential f32 chain (the
// wave-cooperative form of round 2 spent a whole wavefront on one chain: 2.4k cycles per window; this
// is ~45 cycles per window and lane).  Where a zone ends is only known once it has been walked, so:
//
//   zone_spec_kernel     a lane for EVERY S1 signal walks the zone that would start there (at most
//                        kZoneCap steps) and records where it ends and which S1 signal follows;
//   zone_resolve_kernel  one wavefront per ctg follows that chain from the ctg's first S1 signal:
//                        the zones on the chain are the real ones (a zone that hit the cap is walked
//                        to its end right here, by the whole wavefront, and written);
//   zone_commit_kernel   a lane per real zone walks it again and writes the true signals over the
//                        S1 signals of dense_sig[j, e).
//
// The lanes of zone_spec that started inside somebody else's zone did wasted work (about 2/3 of them
// at the default parameters); that is the price of having every possible restart point at hand in ONE
// round, with no host in the loop.  Worst case (a threshold so low that everything signals): one
// zone per ctg, walked by the resolver at the old kernel's speed.
#pragma once

#include "wave_kernels.hpp"

namespace {

constexpr uint32_t kZoneBlock = 1024;     // windows per block of the S1-signal compaction
constexpr uint32_t kZoneUnfinished = 0xFFFFFFFFu;
constexpr uint32_t kZoneNone = 0xFFFFFFFFu;   // nxt: no further S1 signal in this ctg

struct ZoneArgs {
    const WaveCtgDev *ctgs;
    uint32_t n_ctg;
    const uint32_t *cnt;       // dense gc counts, [total_windows]
    int8_t *sig;               // dense signals: S1 on entry, the true ones behind zone_commit
    const float *xtab;         // xtab[k] = k as f32 / size as f32 (the data value of a window with count k)
    const uint32_t *zlist;     // global row index of every S1 signal, ascending
    const unsigned long long *totals;   // [0] = number of S1 signals
    uint2 *zinfo;              // per S1 signal: x = end e of the zone starting there (window index in its ctg;
                               // kZoneUnfinished: hit the cap), y = index of the first S1 signal of the ctg at or behind e
    uint32_t lag, cap;
    uint32_t n_xtab;           // size + 1
    float thr, influence;
};

// nonzero bytes of four packed signals (0x00, 0x01, 0xFF)
__device__ __forceinline__ uint32_t sig_nz4(uint32_t v) { return (uint32_t)__popc(v & 0x01010101u); }

// S1 signals per block of kZoneBlock rows of the dense arrays
__global__ __launch_bounds__(256) void zone_count_kernel(const int8_t *sig, uint64_t total, uint32_t *blk_cnt) {
    __shared__ uint32_t ws[4];
    const uint64_t i = (uint64_t)blockIdx.x * kZoneBlock + threadIdx.x * 4u;
    uint32_t c = 0;
    if (i + 4u <= total) {
        c = sig_nz4(*reinterpret_cast<const uint32_t *>(sig + i));
    } else {
        for (uint64_t q = i; q < total; ++q) c += sig[q] != 0;
    }
    for (int d = 32; d; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
    if ((threadIdx.x & 63u) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// ... and their row indices, in order, at the block's offset
__global__ __launch_bounds__(256) void zone_scatter_kernel(const int8_t *sig, uint64_t total,
                                                           const unsigned long long *blk_off, uint32_t *zlist) {
    __shared__ uint32_t scr[4];
    const uint64_t i = (uint64_t)blockIdx.x * kZoneBlock + threadIdx.x * 4u;
    uint32_t v = 0;
    if (i + 4u <= total) {
        v = *reinterpret_cast<const uint32_t *>(sig + i);
    } else {
        for (uint64_t q = i; q < total; ++q) v |= (uint32_t)(uint8_t)sig[q] << (8u * (uint32_t)(q - i));
    }
    const uint32_t mine = sig_nz4(v);
    uint32_t tot;
    uint32_t pos = block_excl_scan_256<uint32_t>(mine, scr, tot);
    if (mine) {
        uint32_t *dst = zlist + blk_off[blockIdx.x];
        for (uint32_t b = 0; b < 4u; ++b)
            if ((v >> (8u * b)) & 1u) dst[pos++] = (uint32_t)(i + b);
    }
}

// first index in [lo, hi) of the ascending zlist whose value is >= key
__device__ __forceinline__ uint32_t zlist_lower_bound(const uint32_t *zlist, uint32_t lo, uint32_t hi, uint64_t key) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)zlist[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// One lane walks one zone.  All lanes of the wavefront advance one window per trip, so the ring index
// of "the value lag+1 windows back" is the same in every lane: ring[slot][lane], conflict-free, and the
// slot arithmetic stays scalar.  Slot of window q of a zone starting at j: (q - (j-1-lag)) mod R, R = lag+2.
// COMMIT: write the true signals of [j+1, e) to sig (the start keeps its S1 signal, which is true).
// Returns the zone's end e (window index in the ctg) or kZoneUnfinished after `cap` windows (cap even).
// XT_LDS: the data-value table xtab sits behind the ring in LDS (staged by zone_stage_xtab), else in global memory.
//
// Two windows per trip.  The statistics of window i+1 cover filtered[i-lag, i): they do not contain filtered[i],
// so they do not wait for the decision at i (only filtered[i+1] itself does, through one multiply-add at the
// end).  A trip therefore streams the lag + 1 ring values of windows i-1-lag .. i-1 once and feeds two
// INDEPENDENT pairs of sequential sums with them -- a wavefront is alone on its SIMD here, and two chains keep
// the VALU busy where one leaves every other issue slot to the dependent add's latency.
template <bool COMMIT, bool XT_LDS>
__device__ __forceinline__ uint32_t zone_walk(const ZoneArgs &a, float *ring, bool active,
                                              const WaveCtgDev cg, uint32_t j, uint32_t cap) {
    const float *const xt_lds = ring + (size_t)(a.lag + 2u) * 64u;
    auto xval = [&](uint32_t kk) -> float { return XT_LDS ? xt_lds[kk] : a.xtab[kk]; };
    const uint32_t lane = threadIdx.x & 63u, lag = a.lag, R = lag + 2u;
    const uint32_t n = cg.n_win;
    const uint32_t *k = a.cnt + cg.win_base;
    int8_t *sg = a.sig + cg.win_base;
    const float len = (float)lag, thr = a.thr, infl = a.influence;
    float *my = ring + lane;                                   // my[slot * 64]
    // windows j-1-lag .. j-1 are data (nothing before the zone signalled within reach).  Eight counts are in
    // flight per lane before the first one is used.
    {
        const int64_t q0 = (int64_t)j - 1 - (int64_t)lag;       // window of slot 0 (-1 only for j == lag)
        uint32_t s = 0;
        for (; s + 8u <= lag + 1u; s += 8u) {
            uint32_t kk[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) kk[u] = (active && q0 + (int64_t)(s + u) >= 0) ? k[q0 + (int64_t)(s + u)] : 0u;
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) my[(s + u) * 64u] = xval(kk[u]);
        }
        for (; s <= lag; ++s) my[s * 64u] = (active && q0 + (int64_t)s >= 0) ? xval(k[q0 + (int64_t)s]) : 0.0f;
        if (q0 < 0) my[0] = 0.0f;                              // no window -1; the slot is never read
    }
    {
        const float xj = active ? xval(k[j]) : 0.0f;
        const float t1 = infl * xj;                            // stat.rs:42, the zone's start is a true signal
        const float t2 = (1.0f - infl) * my[lag * 64u];
        my[(lag + 1u) * 64u] = t1 + t2;
    }
    uint32_t last = j;
    uint32_t first = 1u;                                       // slot of window i-1-lag: (i - j) mod R
    bool capped = false;
    uint32_t kn0 = (active && j + 1u < n) ? k[j + 1u] : 0u;    // the counts of the trip's two windows, fetched a trip ahead
    uint32_t kn1 = (active && j + 2u < n) ? k[j + 2u] : 0u;
    for (uint32_t t = 1;; t += 2u) {
        const uint32_t i = j + t;
        const bool liveA = active && i < n && i <= last + lag + 1u;
        if (!__any(liveA)) break;
        if (t > cap) {
            capped = liveA;                                    // lanes still inside their zone did not finish
            break;
        }
        const uint32_t kA = kn0, kB = kn1;
        kn0 = (active && i + 2u < n) ? k[i + 2u] : 0u;          // land while this trip's sums are added up
        kn1 = (active && i + 3u < n) ? k[i + 3u] : 0u;
        // value c of the stream = slot (first + c) mod R, c = 0 .. lag: window A sums c in [0, lag), window B c in [1, lag]
        const uint32_t run1 = min(lag + 1u, R - first);        // contiguous slots before the wrap
        float sumA, sumB;
        {
            const float v0 = my[first * 64u];
            sumA = 0.0f + v0;                                                                  // stat.rs:3
            sumB = 0.0f;
        }
#pragma unroll 8
        for (uint32_t c = 1; c < run1; ++c) {
            const float v = my[(first + c) * 64u];
            if (c < lag) sumA = sumA + v;
            sumB = sumB + v;
        }
#pragma unroll 8
        for (uint32_t c = run1 > 1u ? run1 : 1u; c <= lag; ++c) {
            const float v = my[(c - run1) * 64u];
            if (c < lag) sumA = sumA + v;
            sumB = sumB + v;
        }
        const float meanA = sumA / len, meanB = sumB / len;                                    // stat.rs:5
        float sqA, sqB = 0.0f;
        {
            const float d = my[first * 64u] - meanA;
            sqA = 0.0f + d * d;                                                                // stat.rs:12
        }
#pragma unroll 8
        for (uint32_t c = 1; c < run1; ++c) {
            const float v = my[(first + c) * 64u];
            const float dA = v - meanA, dB = v - meanB;
            if (c < lag) sqA = sqA + dA * dA;
            sqB = sqB + dB * dB;
        }
#pragma unroll 8
        for (uint32_t c = run1 > 1u ? run1 : 1u; c <= lag; ++c) {
            const float v = my[(c - run1) * 64u];
            const float dA = v - meanA, dB = v - meanB;
            if (c < lag) sqA = sqA + dA * dA;
            sqB = sqB + dB * dB;
        }
        const float sdA = sqrtf(sqA / (len - 1.0f)), sdB = sqrtf(sqB / (len - 1.0f));          // stat.rs:13
        // window i
        const float xA = xval(kA);
        const bool hitA = fabsf(xA - meanA) > thr * sdA;                                       // stat.rs:36
        const uint32_t prevA = first + lag >= R ? first + lag - R : first + lag;               // slot of window i-1
        const uint32_t curA = prevA + 1u == R ? 0u : prevA + 1u;                               // slot of window i
        float fA = xA;
        if (hitA) {
            const float t1 = infl * xA;
            const float t2 = (1.0f - infl) * my[prevA * 64u];
            fA = t1 + t2;                                                                      // stat.rs:42
        }
        if (liveA) {
            my[curA * 64u] = fA;
            if (hitA) last = i;
            if (COMMIT) sg[i] = (int8_t)(hitA ? (xA > meanA ? 1 : -1) : 0);                    // stat.rs:37-38
        }
        // window i+1: its statistics were independent of window i; its filtered value is not
        const bool liveB = liveA && i + 1u < n && i + 1u <= last + lag + 1u;   // (cap is even: B is inside it when A is)
        const float xB = xval(kB);
        const bool hitB = fabsf(xB - meanB) > thr * sdB;
        float fB = xB;
        if (hitB) {
            const float t1 = infl * xB;
            const float t2 = (1.0f - infl) * fA;
            fB = t1 + t2;
        }
        if (liveB) {
            my[first * 64u] = fB;                                                              // slot of window i+1 = (first + lag + 2) mod R
            if (hitB) last = i + 1u;
            if (COMMIT) sg[i + 1u] = (int8_t)(hitB ? (xB > meanB ? 1 : -1) : 0);
        }
        first = first + 2u >= R ? first + 2u - R : first + 2u;
    }
    if (capped) return kZoneUnfinished;
    return min(last + lag + 2u, n);
}

// xtab in LDS when it is small (size <= kXtabLds - 1, i.e. always for the fast kernels' sizes).  One wave per
// workgroup: no barrier, a wait on the LDS counter.
constexpr uint32_t kXtabLds = 1024;
__device__ __forceinline__ void zone_stage_xtab(const ZoneArgs &a, float *lds_xt) {
    for (uint32_t q = threadIdx.x; q < a.n_xtab; q += 64u) lds_xt[q] = a.xtab[q];
    __builtin_amdgcn_s_waitcnt(0xc07f);
}

// the ctg a row of the dense arrays belongs to
__device__ __forceinline__ uint32_t zone_ctg_of(const ZoneArgs &a, uint64_t g) { return wave_ctg_of(a.ctgs, a.n_ctg, g); }

// A lane for every S1 signal: the zone that would start there.  Grid-stride over chunks of 64 signals (the
// number of signals lives on the device; no host wait sizes the grid).
template <bool XT_LDS>
__device__ __forceinline__ void zone_spec_body(const ZoneArgs &a, float *ring) {
    const uint32_t lane = threadIdx.x;
    const uint64_t total = a.totals[0];
    for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < total; base += (uint64_t)gridDim.x * 64u) {
        const uint64_t p = base + lane;
        const bool active = p < total;
        const uint32_t g = a.zlist[active ? p : total - 1u];
        const uint32_t c = zone_ctg_of(a, g);
        const WaveCtgDev cg = a.ctgs[c];
        const uint32_t j = (uint32_t)(g - cg.win_base);
        const uint32_t e = zone_walk<false, XT_LDS>(a, ring, active, cg, j, a.cap);
        if (active) {
            uint32_t nxt = kZoneNone;
            if (e != kZoneUnfinished && e < cg.n_win) {
                // the first S1 signal at or behind e: usually one of the next few entries
                const uint64_t key = cg.win_base + e;
                uint32_t lo = (uint32_t)p + 1u, step = 1u, hi = lo;
                while (hi < (uint32_t)total && (uint64_t)a.zlist[hi] < key) {
                    lo = hi + 1u;
                    hi = (uint32_t)min((uint64_t)hi + step, total);
                    step <<= 1;
                }
                const uint32_t at = zlist_lower_bound(a.zlist, lo, min(hi, (uint32_t)total), key);
                if (at < (uint32_t)total && (uint64_t)a.zlist[at] < cg.win_base + cg.n_win) nxt = at;
            }
            a.zinfo[p] = make_uint2(e, nxt);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);              // the ring is reused by the next chunk (one wave: no barrier)
    }
}

__global__ __launch_bounds__(64) void zone_spec_kernel(const ZoneArgs a) {
    extern __shared__ float ring[];                      // [lag + 2][64] | xtab (when it fits)
    if (a.n_xtab <= kXtabLds) {
        zone_stage_xtab(a, ring + (size_t)(a.lag + 2u) * 64u);
        zone_spec_body<true>(a, ring);
    } else {
        zone_spec_body<false>(a, ring);
    }
}

// A zone beyond the cap, walked to its end by the whole wavefront (the round-2 evaluator: the lanes fetch
// the lag values of a window in parallel, the two sums stay sequential), signals written as it goes.
// `ring` holds R = lag + 2 floats.  Returns e.
__device__ __noinline__ uint32_t zone_walk_long(const ZoneArgs &a, float *ring, const WaveCtgDev cg, uint32_t j) {
    const uint32_t lane = threadIdx.x & 63u, lag = a.lag, R = lag + 2u, n = cg.n_win;
    const uint32_t *k = a.cnt + cg.win_base;
    int8_t *sg = a.sig + cg.win_base;
    const float len = (float)lag;
    // slot of window q: (q - (j-1-lag)) mod R
    for (uint32_t s = lane; s <= lag; s += 64u) {
        const int64_t q = (int64_t)j - 1 - (int64_t)lag + (int64_t)s;
        ring[s] = q >= 0 ? a.xtab[k[q]] : 0.0f;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) {
        const float t1 = a.influence * a.xtab[k[j]];
        const float t2 = (1.0f - a.influence) * ring[lag];
        ring[lag + 1u] = t1 + t2;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    uint32_t last = j;
    for (uint32_t i = j + 1u; i < n && i <= last + lag + 1u; ++i) {
        const uint32_t first = (i - j) % R;
        float sum = 0.0f;
        for (uint32_t c0 = 0; c0 < lag; c0 += 64u) {
            const float x = c0 + lane < lag ? ring[(first + c0 + lane) % R] : 0.0f;
            sum = seq_add_lanes(sum, x, min(64u, lag - c0));
        }
        const float mean = sum / len;
        float sq = 0.0f;
        for (uint32_t c0 = 0; c0 < lag; c0 += 64u) {
            const float x = ring[(first + min(c0 + lane, lag - 1u)) % R];
            const float d = x - mean;
            const float dd = c0 + lane < lag ? d * d : 0.0f;
            sq = seq_add_lanes(sq, dd, min(64u, lag - c0));
        }
        const float sd = sqrtf(sq / (len - 1.0f));
        const float x = a.xtab[k[i]];
        const bool hit = fabsf(x - mean) > a.thr * sd;
        float f = x;
        if (hit) {
            const float t1 = a.influence * x;
            const float t2 = (1.0f - a.influence) * ring[(first + lag) % R];
            f = t1 + t2;
            last = i;
        }
        if (lane == 0) {
            ring[(first + lag + 1u) % R] = f;
            sg[i] = (int8_t)(hit ? (x > mean ? 1 : -1) : 0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    return min(last + lag + 2u, n);
}

// One wavefront per ctg follows the chain of zones from the ctg's first S1 signal.  vlist[pb + m] = index (into
// zlist) of the m-th real zone of the ctg that zone_commit has to write, vcount[c] how many, pbase[c] = pb.
constexpr uint32_t kResolveStage = 8192;   // zinfo entries of a ctg staged in LDS (64 KiB)
__global__ __launch_bounds__(64) void zone_resolve_kernel(const ZoneArgs a, uint32_t *vlist, uint32_t *vcount,
                                                          uint32_t *pbase) {
    extern __shared__ float lds[];                        // ring (lag + 2 floats, rounded to 64) | staged zinfo
    const uint32_t lane = threadIdx.x;
    const uint32_t ring_words = (a.lag + 2u + 63u) & ~63u;
    float *ring = lds;
    uint2 *stage = reinterpret_cast<uint2 *>(lds + ring_words);
    const uint32_t total = (uint32_t)a.totals[0];
    for (uint32_t c = blockIdx.x; c < a.n_ctg; c += gridDim.x) {
        const WaveCtgDev cg = a.ctgs[c];
        const uint32_t pb = zlist_lower_bound(a.zlist, 0u, total, cg.win_base);
        const uint32_t pe = zlist_lower_bound(a.zlist, pb, total, cg.win_base + cg.n_win);
        const uint32_t np = pe - pb;
        const bool staged = np <= kResolveStage;
        if (staged)
            for (uint32_t q = lane; q < np; q += 64u) stage[q] = a.zinfo[pb + q];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        uint32_t m = 0;
        uint32_t p = pb;
        while (p < pe) {                                   // wave-uniform
            uint2 zi = make_uint2(0u, 0u);
            if (lane == 0) zi = staged ? stage[p - pb] : a.zinfo[p];
            zi.x = (uint32_t)__builtin_amdgcn_readfirstlane((int)zi.x);
            zi.y = (uint32_t)__builtin_amdgcn_readfirstlane((int)zi.y);
            if (zi.x == kZoneUnfinished) {
                const uint32_t j = (uint32_t)((uint64_t)a.zlist[p] - cg.win_base);
                const uint32_t e = zone_walk_long(a, ring, cg, j);
                p = e < cg.n_win ? zlist_lower_bound(a.zlist, p + 1u, pe, cg.win_base + e) : pe;
            } else {
                if (lane == 0) vlist[pb + m] = p;
                ++m;
                p = zi.y == kZoneNone ? pe : zi.y;
            }
        }
        if (lane == 0) {
            vcount[c] = m;
            pbase[c] = pb;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// A lane per real zone: walk it again, writing the true signals.  vstart = exclusive prefix of vcount
// (wave_offsets_kernel), vtotals[0] = number of real zones.
template <bool XT_LDS>
__device__ __forceinline__ void zone_commit_body(const ZoneArgs &a, float *ring, const uint32_t *vlist, const uint32_t *pbase,
                                                 const unsigned long long *vstart, const unsigned long long *vtotals) {
    const uint32_t lane = threadIdx.x;
    const uint64_t total = vtotals[0];
    for (uint64_t base = (uint64_t)blockIdx.x * 64u; base < total; base += (uint64_t)gridDim.x * 64u) {
        const uint64_t v = base + lane;
        const bool active = v < total;
        const uint64_t vv = active ? v : total - 1u;
        uint32_t lo = 0, hi = a.n_ctg;                     // last ctg whose vstart <= vv
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (vstart[mid] <= vv)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t c = lo;
        const uint32_t p = vlist[pbase[c] + (uint32_t)(vv - vstart[c])];
        const WaveCtgDev cg = a.ctgs[c];
        const uint32_t j = (uint32_t)((uint64_t)a.zlist[p] - cg.win_base);
        (void)zone_walk<true, XT_LDS>(a, ring, active, cg, j, a.cap);
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

__global__ __launch_bounds__(64) void zone_commit_kernel(const ZoneArgs a, const uint32_t *vlist, const uint32_t *pbase,
                                                         const unsigned long long *vstart,
                                                         const unsigned long long *vtotals) {
    extern __shared__ float ring[];
    if (a.n_xtab <= kXtabLds) {
        zone_stage_xtab(a, ring + (size_t)(a.lag + 2u) * 64u);
        zone_commit_body<true>(a, ring, vlist, pbase, vstart, vtotals);
    } else {
        zone_commit_body<false>(a, ring, vlist, pbase, vstart, vtotals);
    }
}

}  // namespace
