// interval.hip -- sorted-interval binary-search kernels replacing the rust_lapper
// `idx:` index and the IntSpan intersections of `anno`.
//
//   interval_count_kernel   gams::count_rg  (src/libs/utils.rs:24-36) = Lapper::count:
//       #{start < qe} - #{stop <= qs} by two lower bounds over the group's
//       independently sorted starts[] and stops[] (rust-lapper 1.1.0, BITS).
//   interval_locate_kernel  gams::find_one_idx (src/libs/utils.rs:7-22) =
//       Lapper::find(qs,qe).next(): first interval in (start,stop) order with
//       start < qe && stop > qs; the scan begins at lower_bound(qs - max_len)
//       exactly like the crate's iterator and stops at start >= qe.
//   span_cover_kernel       anno (src/cmd_gams/anno.rs:128-139):
//       |set[chr] & [ctg] & [range]| / |range| from a prefix of covered bases
//       over the chr's sorted disjoint spans (two upper bounds per query).
//
// Intervals are half-open [start, stop) with stop = end+1 as the reference
// stores them (src/libs/redis.rs:245-248, 291-294); queries are passed the way
// the reference passes them: (rg.start, rg.end), i.e. the end is exclusive.
// One lane per query; 16 B/query of traffic plus the probed table lines.
//
// Random queries make a plain binary search fetch ~log2(n) scattered cache lines per sorted
// array (measured: 2.8 G queries/s, the L2 line rate).  Every sorted array therefore carries a
// bucket directory: dir[b] = rank of the first key >= key0 + (b << shift), with about one
// bucket per key, so a lower bound is one directory line plus a search inside dir[b]..dir[b+1]
// (typically 0-2 keys, one line).  Skewed keys only lengthen that inner search; the answer is
// the same lower bound either way.

#include "common.hpp"

#include <rocprim/rocprim.hpp>   // the segmented LSD radix sort, for groups beyond a workgroup (index_build_kernel does the others)

#include <algorithm>
#include <atomic>
#include <numeric>
#include <thread>

// bucket directory of one sorted key array of one group (32-bit keys, order-preserving bias applied)
struct KeyDir {
    uint32_t key0;    // smallest key of the group
    uint32_t shift;   // bucket b holds keys in [key0 + (b << shift), key0 + ((b+1) << shift))
    uint32_t nb;      // buckets; nb <= n, dir has nb+1 entries at dir[off + g]
    uint32_t pad;
};

// everything a query needs to know about its group, one 48-B record (one or two lines)
struct IvRec {
    uint32_t start, stop;
    uint64_t orig;
};

struct IndexGroup {
    uint64_t off;     // first interval of the group
    uint32_t n;       // intervals in the group
    uint32_t maxlen;  // max(stop - start), Lapper::max_len
    KeyDir start, stop;       // one bucket per key, rank only (locate's lower bound)
    KeyDir bk_start, bk_stop; // ~2^kCellShift keys per cell, rank + the next seven keys inline (count's two lower bounds)
};

// What a count query needs of its group, one 32-B record (the 80-B IndexGroup costs a second line and
// three more loads per query; random queries are bound by lines moved, not by bytes)
struct CountGroup {
    uint32_t off, n;                 // first interval (m < 2^32), intervals
    uint32_t s_key0, s_nb;           // bucket records over the starts
    uint32_t t_key0, t_nb;           // ... over the stops
    uint32_t s_shift, t_shift;
};

// Bucket record of the count path: everything a lower bound needs in ONE 32-B access -- the rank of
// the first key at or after the bucket's edge and the seven keys that follow it (0xffffffff past the
// group's end).  A key that falls in bucket b is compared with those seven; only when all seven are
// smaller (a cell of more than seven keys: 0.1 % of uniform cells at 2 keys per cell) the search goes on in the key
// array.  Random queries are bound by scattered lines out of the Infinity Cache: the separate
// directory + key array of the locate path cost two lines per bound, this one.
struct BkRec {
    uint32_t rank;
    uint32_t k[7];
};
#ifndef GAMS_CELL_SHIFT
#define GAMS_CELL_SHIFT 1
#endif
constexpr uint32_t kCellShift = GAMS_CELL_SHIFT;   // about 2^kCellShift keys per cell of the count path's grid

struct SpanRec {
    int32_t lo, hi;   // inclusive
    uint64_t cum;     // covered bases in the group's spans before this one
};

struct SpanGroup {
    uint64_t off;
    uint32_t n;
    uint32_t pad;
    KeyDir lo;
};

struct gams_index {
    uint32_t n_groups = 0;
    uint64_t m = 0;
    IndexGroup *d_groups = nullptr;
    uint32_t *d_stops = nullptr;     // per group, ascending, sorted independently (for count)
    uint32_t *d_lstart = nullptr;    // per group, starts of the (start,stop)-sorted pairs: ascending (both searches)
    IvRec *d_lrec = nullptr;         // the sorted pairs + the caller's index of each, 16 B (locate's scan)
    uint32_t *d_dir_start = nullptr; // m + n_groups entries: group g's directory begins at off[g] + g (locate)
    BkRec *d_bk_start = nullptr;     // 2 * ((m >> kCellShift) + 2*n_groups + 2) records, the starts' and the stops' record of a cell side by side;
    BkRec *d_bk_stop = nullptr;      // = d_bk_start + 1; group g's cells begin at (off[g] >> kCellShift) + 2g
    CountGroup *d_cgroups = nullptr;
    // everything above lives in one pooled HBM block
    uint8_t *arena = nullptr;
    size_t arena_bytes = 0;
};

// Cell record of the anno path, 64 B: everything a "covered bases up to x" lookup needs when x falls in cell b of the
// group's grid (the grid of the `lo` directory, about one span per cell) -- the rank at the cell's edge, the span in
// front of it with the covered bases before that one, and the next five spans inline.  One line per position instead
// of a directory line, a search and a record line; a cell that starts more than five spans falls back to those.
struct SpanCell {
    uint64_t base;        // covered bases in the group's spans before span rank-1 (0 when rank == 0)
    uint32_t rank;        // spans with lo < the cell's edge
    int32_t plo, phi;     // span rank-1; plo > phi when there is none
    int32_t lo[5], hi[5]; // spans rank .. rank+4 (unused entries past the group's end)
    uint32_t pad;
};
static_assert(sizeof(SpanCell) == 64, "SpanCell is one 64-B record");

struct gams_spans {
    uint32_t n_groups = 0;
    uint64_t m = 0;
    SpanGroup *d_groups = nullptr;
    SpanRec *d_rec = nullptr;        // one 16-B record per span: the search and its two follow-up reads share a line
    uint32_t *d_dir_lo = nullptr;
    SpanCell *d_cells = nullptr;     // m + n_groups + 1 records: group g's cell b at off[g] + g + b (like its directory)
};

namespace {

// Number of keys < key among the group's n ascending keys a[0..n) (rank of the lower bound).
// BIAS = 0x80000000 compares int32 keys stored as they are (x ^ BIAS is order preserving).
template <uint32_t BIAS>
__device__ __forceinline__ uint32_t dir_lower_bound(const uint32_t *a, const uint32_t *dir, uint32_t n,
                                                    const KeyDir d, uint64_t key) {
    if (n == 0 || key <= (uint64_t)d.key0) return 0;
    const uint64_t b = (key - d.key0) >> d.shift;
    if (b >= d.nb) return n;                       // beyond the last bucket: beyond the largest key
    uint32_t lo = dir[b], hi = dir[b + 1];
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)(a[mid] ^ BIAS) < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// lower bound through the bucket records (see BkRec)
__device__ __forceinline__ uint32_t bk_lower_bound(const uint32_t *keys, const BkRec *bk, uint32_t n, const KeyDir d,
                                                   uint64_t key) {
    if (n == 0 || key <= (uint64_t)d.key0) return 0;
    const uint64_t b = (key - d.key0) >> d.shift;
    if (b >= d.nb) return n;
    const uint4 *rp = reinterpret_cast<const uint4 *>(bk + 2u * b);   // the cell's pair of records: starts', stops'
    const uint4 r0 = rp[0], r1 = rp[1];
    const uint32_t rank = r0.x;
    uint32_t c = (uint32_t)((uint64_t)r0.y < key) + (uint32_t)((uint64_t)r0.z < key) + (uint32_t)((uint64_t)r0.w < key) +
                 (uint32_t)((uint64_t)r1.x < key) + (uint32_t)((uint64_t)r1.y < key) + (uint32_t)((uint64_t)r1.z < key) +
                 (uint32_t)((uint64_t)r1.w < key);
    c = min(c, n - rank);                            // padding past the group's end does not count
    if (c < 7u || rank + 7u >= n) return rank + c;
    // A crowded cell (more than seven keys; 5 % of uniform cells, i.e. some lane of nearly every wave): the
    // answer lies between rank + 7 and the next cell's rank.  One load for that rank (the neighbouring
    // record), then the next eight keys in one batch of independent loads -- two round trips instead of the
    // ~12 dependent ones of a binary search over the group, which made every wave of random queries live 26 us
    // (33 dependent loads, profiles/r02_count_latency_chain.txt).  A cell of more than 15 keys goes on with
    // the binary search, inside the cell.
    uint32_t lo = rank + 7u;
    uint32_t hi = b + 1u < d.nb ? reinterpret_cast<const uint32_t *>(bk + 2u * (b + 1u))[0] : n;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) {
        const uint32_t k = keys[min(lo + i, n - 1u)];
        cnt += (lo + i < hi && (uint64_t)k < key) ? 1u : 0u;
    }
    if (hi - lo <= 8u) return lo + cnt;
    if (cnt < 8u) return lo + cnt;                   // sorted keys: the first one >= key ends the count
    lo += 8u;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void interval_count_kernel(const CountGroup *groups, const uint32_t *starts,
                                                             const uint32_t *stops, const BkRec *bk_start,
                                                             const BkRec *bk_stop, uint32_t n_groups,
                                                             const uint32_t *group, const uint32_t *qs,
                                                             const uint32_t *qe, uint64_t nq, int32_t *out) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t g = group[q];
    if (g >= n_groups) {  // utils.rs:29-32: ctg not in the index -> 0
        out[q] = 0;
        return;
    }
    const uint4 *gp = reinterpret_cast<const uint4 *>(groups + g);
    const uint4 g0 = gp[0], g1 = gp[1];
    const uint32_t off = g0.x, n = g0.y;
    const KeyDir ds{g0.z, g1.z, g0.w, 0u}, dt{g1.x, g1.w, g1.y, 0u};
    const uint64_t boff = (uint64_t)(off >> kCellShift) + 2ull * g;
    // Lapper::count: first = bsearch_seq(start + 1, stops); last = bsearch_seq(stop, starts)
    // (bk_start = the interleaved array, bk_stop = bk_start + 1: a range shorter than a cell finds both of its
    // records in one 64-B line or in two neighbouring ones)
    const uint32_t first = bk_lower_bound(stops + off, bk_stop + 2u * boff, n, dt, (uint64_t)qs[q] + 1u);
    const uint32_t last = bk_lower_bound(starts + off, bk_start + 2u * boff, n, ds, (uint64_t)qe[q]);
    out[q] = (int32_t)((int64_t)last - (int64_t)first);
}

__global__ __launch_bounds__(256) void interval_locate_kernel(const IndexGroup *groups, const uint32_t *lstart,
                                                              const IvRec *lrec, const uint32_t *dir_start,
                                                              uint32_t n_groups, const uint32_t *group,
                                                              const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                                                              int64_t *out) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t g = group[q];
    int64_t hit = -1;
    if (g < n_groups) {
        const IndexGroup G = groups[g];
        const uint32_t s = qs[q], e = qe[q];
        const uint32_t from = s > G.maxlen ? s - G.maxlen : 0u;  // checked_sub(max_len).unwrap_or(0)
        // Lapper::lower_bound: first interval whose start >= from
        const uint64_t hi = G.off + G.n;
        for (uint64_t i = G.off + dir_lower_bound<0u>(lstart + G.off, dir_start + G.off + g, G.n, G.start, from);
             i < hi; ++i) {
            const IvRec r = lrec[i];
            if (r.start < e && r.stop > s) {  // Interval::overlap
                hit = (int64_t)r.orig;
                break;
            }
            if (r.start >= e) break;
        }
    }
    out[q] = hit;
}

// covered positions <= x inside the group's spans
__device__ __forceinline__ uint64_t covered_upto(const SpanRec *rec, const uint32_t *dir, const SpanGroup &G,
                                                 uint32_t g, int32_t x, uint32_t *rank) {
    // spans with lo <= x = keys < x+1 in biased order; directory lookup, then a search over rec[].lo
    const SpanRec *r = rec + G.off;
    uint32_t i = 0;
    const uint64_t key = (uint64_t)((uint32_t)x ^ 0x80000000u) + 1u;
    if (G.n != 0 && key > (uint64_t)G.lo.key0) {
        const uint64_t b = (key - G.lo.key0) >> G.lo.shift;
        if (b >= G.lo.nb) {
            i = G.n;
        } else {
            const uint32_t *d = dir + G.off + g;
            uint32_t lo = d[b], hi = d[b + 1];
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if ((uint64_t)((uint32_t)r[mid].lo ^ 0x80000000u) < key)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            i = lo;
        }
    }
    if (rank) *rank = i;
    if (i == 0) return 0;
    const SpanRec s = r[i - 1];
    const int32_t top = s.hi < x ? s.hi : x;
    return s.cum + (uint64_t)((int64_t)top - s.lo + 1);
}

// The same for a second position x2 <= x whose predecessor `rank` (spans with lo <= x) is known: a range is
// short next to the spans' spacing, so the spans with lo <= x2 end zero to two records further down -- a walk
// over neighbouring 16-B records instead of a second directory line + search (two scattered requests less per
// line).  A long walk gives up and searches.
__device__ __forceinline__ uint64_t covered_upto_below(const SpanRec *rec, const uint32_t *dir, const SpanGroup &G,
                                                       uint32_t g, int32_t x2, uint32_t rank) {
    const SpanRec *r = rec + G.off;
    uint32_t i = rank;
    for (int step = 0; step < 6; ++step) {
        if (i == 0) return 0;
        const SpanRec s = r[i - 1];
        if (s.lo <= x2) {
            const int32_t top = s.hi < x2 ? s.hi : x2;
            return s.cum + (uint64_t)((int64_t)top - s.lo + 1);
        }
        --i;
    }
    return covered_upto(rec, dir, G, g, x2, nullptr);
}

// covered positions <= x through the cell records; *cell_no receives the cell (so that a second position in the
// same cell reuses the record), `ok` = false: the cell starts more than five spans, use covered_upto
__device__ __forceinline__ uint64_t covered_cell(const SpanCell &c, uint32_t n, int32_t x, bool &ok) {
    const uint32_t valid = min(5u, n - c.rank);
    uint32_t t = 0;
#pragma unroll
    for (uint32_t u = 0; u < 5u; ++u) t += (u < valid && c.lo[u] <= x) ? 1u : 0u;
    ok = !(t == 5u && c.rank + 5u < n);
    const bool has_pred = c.plo <= c.phi;
    if (t == 0u) {
        if (!has_pred) return 0;
        const int32_t top = c.phi < x ? c.phi : x;
        return c.base + (uint64_t)((int64_t)top - c.plo + 1);
    }
    uint64_t cum = c.base + (has_pred ? (uint64_t)((int64_t)c.phi - c.plo + 1) : 0ull);
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u)
        if (u + 1u < t) cum += (uint64_t)((int64_t)c.hi[u] - c.lo[u] + 1);
    const int32_t lo = c.lo[t - 1u], hi = c.hi[t - 1u];
    const int32_t top = hi < x ? hi : x;
    return cum + (uint64_t)((int64_t)top - lo + 1);
}

// cell of position x in the group's grid; false: no span of the group has lo <= x (the answer is 0)
__device__ __forceinline__ bool span_cell_of(const SpanGroup &G, int32_t x, uint32_t &b) {
    const uint64_t key = (uint64_t)((uint32_t)x ^ 0x80000000u) + 1u;   // spans with lo <= x = biased keys < key
    if (G.n == 0 || key <= (uint64_t)G.lo.key0) return false;
    const uint64_t bb = (key - 1u - G.lo.key0) >> G.lo.shift;          // the cell x itself falls in
    b = (uint32_t)(bb < G.lo.nb ? bb : G.lo.nb - 1u);                   // past the last cell: the last cell's spans
    return true;
}

__device__ __forceinline__ SpanCell load_cell(const SpanCell *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    union {
        uint4 v[4];
        SpanCell c;
    } u;
    u.v[0] = q[0];
    u.v[1] = q[1];
    u.v[2] = q[2];
    u.v[3] = q[3];
    return u.c;
}

__global__ __launch_bounds__(256) void span_cell_kernel(const SpanGroup *groups, uint32_t n_groups, const SpanRec *rec,
                                                        const uint32_t *dir, uint64_t slots, SpanCell *cells) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= slots) return;
    uint32_t lo = 0, hi = n_groups;                 // last group with off[g] + g <= j
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (groups[mid].off + mid <= j)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t g = lo;
    const SpanGroup G = groups[g];
    const uint64_t b = j - (G.off + g);
    if (G.n == 0 || b >= G.lo.nb) return;
    const uint32_t rank = dir[j];                   // spans with biased lo < key0 + (b << shift)
    const SpanRec *r = rec + G.off;
    SpanCell c;
    c.rank = rank;
    c.pad = 0;
    if (rank) {
        const SpanRec p = r[rank - 1u];
        c.base = p.cum;
        c.plo = p.lo;
        c.phi = p.hi;
    } else {
        c.base = 0;
        c.plo = 1;
        c.phi = 0;
    }
#pragma unroll
    for (uint32_t u = 0; u < 5u; ++u) {
        const bool in = rank + u < G.n;
        const SpanRec q = r[in ? rank + u : 0u];
        c.lo[u] = in ? q.lo : 0;
        c.hi[u] = in ? q.hi : 0;
    }
    cells[j] = c;
}

__global__ __launch_bounds__(256) void span_cover_kernel(const SpanGroup *groups, const SpanRec *rec,
                                                         const uint32_t *dir, const SpanCell *cells, uint32_t n_groups,
                                                         const uint32_t *group, const int32_t *clip_lo,
                                                         const int32_t *clip_hi, const int32_t *qs,
                                                         const int32_t *qe, uint64_t nq, float *out) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t g = group[q];
    const int32_t s = qs[q], e = qe[q];
    float prop = 0.0f;  // anno.rs:128: chr absent from the set
    if (g < n_groups && e >= s) {
        const int32_t L = s > clip_lo[q] ? s : clip_lo[q];
        const int32_t H = e < clip_hi[q] ? e : clip_hi[q];
        uint64_t card = 0;
        if (H >= L) {
            const SpanGroup G = groups[g];
            // one 64-B cell record per end of the range (the same record when both ends fall in one cell)
            uint32_t bh = 0, bl = 0;
            uint64_t upto_h = 0, upto_l = 0;
            bool ok_h = true, ok_l = true;
            const bool any_h = span_cell_of(G, H, bh);
            const bool any_l = L > INT32_MIN && span_cell_of(G, L - 1, bl);
            if (any_h) {
                const SpanCell ch = load_cell(cells + G.off + g + bh);
                upto_h = covered_cell(ch, G.n, H, ok_h);
                if (any_l) {
                    if (bl == bh)
                        upto_l = covered_cell(ch, G.n, L - 1, ok_l);
                    else
                        upto_l = covered_cell(load_cell(cells + G.off + g + bl), G.n, L - 1, ok_l);
                }
            }
            if (!(ok_h && ok_l)) {                     // a crowded cell: directory + search + walk
                uint32_t rank_h;
                upto_h = covered_upto(rec, dir, G, g, H, &rank_h);
                upto_l = L > INT32_MIN ? covered_upto_below(rec, dir, G, g, L - 1, rank_h) : 0;
            }
            card = upto_h - upto_l;
        }
        const int32_t total = (int32_t)((int64_t)e - s + 1);
        prop = (float)(int32_t)card / (float)total;  // cardinality() as f32 / cardinality() as f32
    }
    out[q] = prop;
}

// ---- index build on the device (Lapper::new, src/libs/redis.rs:236-324) -----------------------
// pack (start, stop) into one 64-bit key and number the intervals: a stable sort of the keys inside
// every group is intervals.sort() of rust-lapper (ties keep the caller's order)
__global__ __launch_bounds__(256) void index_pack_kernel(const uint32_t *starts, const uint32_t *stops, uint64_t m,
                                                         uint64_t *key, uint32_t *val) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    key[i] = ((uint64_t)starts[i] << 32) | stops[i];
    val[i] = (uint32_t)i;
}

// sorted keys + permutation -> the ascending starts and the 16-B scan records
__global__ __launch_bounds__(256) void index_unpack_kernel(const uint64_t *key, const uint32_t *perm, uint64_t m,
                                                           uint32_t *lstart, IvRec *lrec) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint64_t k = key[i];
    lstart[i] = (uint32_t)(k >> 32);
    lrec[i] = IvRec{(uint32_t)(k >> 32), (uint32_t)k, (uint64_t)perm[i]};
}

__device__ __forceinline__ KeyDir dir_params(uint32_t first, uint32_t last, uint32_t n) {
    KeyDir d{0u, 0u, 0u, 0u};
    if (n == 0) return d;
    d.key0 = first;
    // 64-bit range: a one-key grid (n == 1, e.g. the cell grid of a one-interval group [0, 2^31 + 1)) needs
    // range >> shift == 0, i.e. shift == 32 when the range has bit 31 set -- a 32-bit shift by 32 is masked
    // to 0 on this hardware and the loop would never end.  Every consumer shifts in 64 bits.
    const uint64_t range = (uint64_t)last - (uint64_t)first;
    while ((range >> d.shift) >= n) ++d.shift;     // (range >> shift) + 1 <= n buckets; ends at shift <= 32
    d.nb = (uint32_t)(range >> d.shift) + 1u;
    return d;
}

// one wavefront per group: max(stop - start) (Lapper::max_len) and the directory headers
__global__ __launch_bounds__(64) void index_group_kernel(const uint32_t *off32, uint32_t n_groups,
                                                         const IvRec *lrec, const uint32_t *lstart,
                                                         const uint32_t *stops_sorted, IndexGroup *groups,
                                                         CountGroup *cgroups) {
    const uint32_t g = blockIdx.x;
    if (g >= n_groups) return;
    const uint32_t lo = off32[g], hi = off32[g + 1], n = hi - lo;
    uint32_t ml = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 64u) {
        const IvRec r = lrec[i];
        if (r.stop > r.start) ml = max(ml, r.stop - r.start);
    }
    for (int d = 32; d; d >>= 1) ml = max(ml, (uint32_t)__shfl_xor((int)ml, d, 64));
    if (threadIdx.x == 0) {
        IndexGroup G;
        G.off = lo;
        G.n = n;
        G.maxlen = ml;
        G.start = n ? dir_params(lstart[lo], lstart[hi - 1], n) : KeyDir{0u, 0u, 0u, 0u};
        G.stop = n ? dir_params(stops_sorted[lo], stops_sorted[hi - 1], n) : KeyDir{0u, 0u, 0u, 0u};
        // the count path's bucket records: ONE grid of cells over the group's coordinates for both key
        // arrays, so that the record of the starts and the record of the stops of a cell sit side by side
        G.bk_start = n ? dir_params(min(lstart[lo], stops_sorted[lo]), max(lstart[hi - 1], stops_sorted[hi - 1]), (n >> kCellShift) + 1u)
                       : KeyDir{0u, 0u, 0u, 0u};
        G.bk_stop = G.bk_start;
        groups[g] = G;
        cgroups[g] = CountGroup{lo, n, G.bk_start.key0, G.bk_start.nb, G.bk_stop.key0, G.bk_stop.nb, G.bk_start.shift,
                                G.bk_stop.shift};
    }
}

// bucket directories: slot j of the directory array belongs to the group g with off[g] + g <= j
// (its bucket b = j - off[g] - g); dir[b] = rank of the first key >= key0 + (b << shift), dir[nb] = n
__global__ __launch_bounds__(256) void index_dir_kernel(const uint32_t *off32, uint32_t n_groups, uint64_t slots,
                                                        const IndexGroup *groups, const uint32_t *lstart,
                                                        uint32_t *dir_start) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= slots) return;
    uint32_t lo = 0, hi = n_groups;                 // last group with off[g] + g <= j
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)off32[mid] + mid <= j)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t g = lo;
    const IndexGroup G = groups[g];
    const uint32_t b = (uint32_t)(j - ((uint64_t)off32[g] + g));
    auto fill = [&](const uint32_t *keys, const KeyDir d, uint32_t *dir) {
        if (b > d.nb || (G.n == 0 && b > 0)) return;
        uint32_t r = G.n;
        if (b < d.nb) {
            const uint64_t edge = (uint64_t)d.key0 + ((uint64_t)b << d.shift);
            uint32_t a = 0, z = G.n;
            while (a < z) {
                const uint32_t mid = a + ((z - a) >> 1);
                if ((uint64_t)keys[G.off + mid] < edge)
                    a = mid + 1;
                else
                    z = mid;
            }
            r = a;
        }
        dir[j] = G.n == 0 ? 0u : r;
    };
    fill(lstart, G.start, dir_start);
}

// bucket records of the count path: slot j belongs to the group g with off[g]/4 + 2g <= j, bucket b = j - that
__global__ __launch_bounds__(256) void index_bk_kernel(const uint32_t *off32, uint32_t n_groups, uint64_t slots,
                                                       const IndexGroup *groups, const uint32_t *lstart,
                                                       const uint32_t *stops_sorted, BkRec *bk_start, BkRec *bk_stop) {
    const uint64_t j = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (j >= slots) return;
    uint32_t lo = 0, hi = n_groups;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((uint64_t)(off32[mid] >> kCellShift) + 2ull * mid <= j)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t g = lo;
    const IndexGroup G = groups[g];
    const uint64_t b = j - ((uint64_t)(off32[g] >> kCellShift) + 2ull * g);
    auto fill = [&](const uint32_t *keys, const KeyDir d, BkRec *bk) {
        if (G.n == 0 || b >= d.nb) return;
        const uint64_t edge = (uint64_t)d.key0 + (b << d.shift);
        uint32_t a = 0, z = G.n;
        while (a < z) {
            const uint32_t mid = a + ((z - a) >> 1);
            if ((uint64_t)keys[G.off + mid] < edge)
                a = mid + 1;
            else
                z = mid;
        }
        BkRec r;
        r.rank = a;
#pragma unroll
        for (uint32_t i = 0; i < 7u; ++i) r.k[i] = a + i < G.n ? keys[G.off + a + i] : 0xffffffffu;
        bk[2u * j] = r;                                // cell j: starts' record, stops' record (64 B together)
    };
    fill(lstart, G.bk_start, bk_start);
    fill(stops_sorted, G.bk_stop, bk_stop);
}

// ---- the whole build of ONE group in one workgroup (groups of at most kBuildCap intervals) --------------
// A group of idx:rg:{ctg} / idx:ctg:{chr} is a few thousand intervals: it fits a workgroup's LDS several times
// over, so Lapper::new (redis.rs:253,299: intervals.sort(), the stops sorted on their own, max_len) and every
// table derived from the sorted arrays are made where the data already is, with one pass over HBM in and one
// out -- against eight radix passes of a generic 64-bit segmented sort plus five kernels that each search the
// sorted arrays in global memory again (2.4 ms per 1.25e7 intervals in 4,000 groups).
//   sort      bitonic network over N = 2^ceil(log2 n) slots on the triple (start, stop, position in the
//             caller's order): the position breaks ties, which makes the network's result the STABLE order of
//             intervals.sort(); the stops ride along as a second, keys-only network in the same stages
//             (one barrier per stage for both);
//   outputs   ascending starts, 16-B scan records, sorted stops, max_len, the group's directory headers,
//             the bucket directory (one lower bound per slot, in LDS) and the count path's cell records.
constexpr uint32_t kBuildCap = 8192;
template <uint32_t CAP, uint32_t T>
__global__ __launch_bounds__(T) void index_build_kernel(const uint32_t *off32, uint32_t n_groups, const uint32_t *starts,
                                                        const uint32_t *stops, uint32_t *lstart, IvRec *lrec,
                                                        uint32_t *stops_sorted, IndexGroup *groups, CountGroup *cgroups,
                                                        uint32_t *dir_start, BkRec *bk_start, BkRec *bk_stop) {
    extern __shared__ __align__(16) unsigned char smem_b[];
    uint64_t *key = reinterpret_cast<uint64_t *>(smem_b);            // (start << 32) | stop
    uint32_t *skey = reinterpret_cast<uint32_t *>(key + CAP);        // stops, sorted on their own
    uint16_t *pos = reinterpret_cast<uint16_t *>(skey + CAP);        // position in the caller's order
    uint32_t *scr = reinterpret_cast<uint32_t *>(pos + CAP);         // 96 words: reductions [0, 64), the group record's parameters [64, 96)
    const uint32_t tid = threadIdx.x;
    const uint32_t g = blockIdx.x;
    if (g >= n_groups) return;
    const uint32_t lo = off32[g], n = off32[g + 1] - lo;
    uint32_t N = 4;
    while (N < n) N <<= 1;
    // The ranges of a real group are narrow (a ctg's coordinates: 20-odd bits): when (start - min start),
    // (stop - min stop) and the position fit 64 bits together, the triple is ONE word -- unique, so no tie
    // logic -- and a compare-exchange is two 8-B reads, one compare, two 8-B writes.  Otherwise the triple
    // stays (64-bit key, 16-bit position).
    uint32_t smin = ~0u, smax = 0u, tmin = ~0u, tmax = 0u;
    for (uint32_t i = tid; i < n; i += T) {
        const uint32_t a = starts[lo + i], b = stops[lo + i];
        smin = min(smin, a);
        smax = max(smax, a);
        tmin = min(tmin, b);
        tmax = max(tmax, b);
    }
    for (int d = 32; d; d >>= 1) {
        smin = min(smin, (uint32_t)__shfl_xor((int)smin, d, 64));
        smax = max(smax, (uint32_t)__shfl_xor((int)smax, d, 64));
        tmin = min(tmin, (uint32_t)__shfl_xor((int)tmin, d, 64));
        tmax = max(tmax, (uint32_t)__shfl_xor((int)tmax, d, 64));
    }
    if ((tid & 63u) == 0) {
        scr[(tid >> 6) * 4u + 0u] = smin;     // T / 64 <= 16 waves: scr[0..63] (the parameters below live behind them)
        scr[(tid >> 6) * 4u + 1u] = smax;
        scr[(tid >> 6) * 4u + 2u] = tmin;
        scr[(tid >> 6) * 4u + 3u] = tmax;
    }
    __syncthreads();
    for (uint32_t w = 0; w < T / 64u; ++w) {
        smin = min(smin, scr[w * 4u + 0u]);
        smax = max(smax, scr[w * 4u + 1u]);
        tmin = min(tmin, scr[w * 4u + 2u]);
        tmax = max(tmax, scr[w * 4u + 3u]);
    }
    __syncthreads();
    const uint32_t bs = n ? 32u - (uint32_t)__clz((int)((smax - smin) | 1u)) : 1u;
    const uint32_t bt = n ? 32u - (uint32_t)__clz((int)((tmax - tmin) | 1u)) : 1u;
    const uint32_t bp = 32u - (uint32_t)__clz((int)((N - 1u) | 1u));
    const bool one_word = bs + bt + bp <= 64u;
    for (uint32_t i = tid; i < N; i += T) {
        if (i < n) {
            const uint32_t a = starts[lo + i], b = stops[lo + i];
            key[i] = one_word ? ((uint64_t)(a - smin) << (bt + bp)) | ((uint64_t)(b - tmin) << bp) | i
                              : ((uint64_t)a << 32) | b;
            skey[i] = b;
            pos[i] = (uint16_t)i;
        } else {
            key[i] = ~0ull;                 // pads sort behind every interval: equal keys are ordered by position
            skey[i] = ~0u;
            pos[i] = 0xFFFFu;
        }
    }
    __syncthreads();
    if (one_word) {
        // Slots come in blocks of four (block o = slots 4o .. 4o+3, one thread each): the stages with j = 2 and j = 1 of every level (and
        // the levels k = 2 and k = 4 entirely) never leave them, so they run in registers between one 16-B/32-B
        // read and one write per array -- two barriers and four LDS round trips less per level.
        auto cex = [](uint64_t &x, uint64_t &y, bool up) {
            if ((x > y) == up && x != y) {
                const uint64_t t_ = x;
                x = y;
                y = t_;
            }
        };
        auto cex32 = [](uint32_t &x, uint32_t &y, bool up) {
            if ((x > y) == up && x != y) {
                const uint32_t t_ = x;
                x = y;
                y = t_;
            }
        };
        auto local = [&](uint32_t k) {            // k == 0: levels 2 and 4 from scratch; else stages j = 2, 1 of level k >= 8
            for (uint32_t o = tid; o < (N >> 2); o += T) {
                uint64_t *kp = key + 4u * o;
                uint32_t *sp = skey + 4u * o;
                uint64_t a0 = kp[0], a1 = kp[1], a2 = kp[2], a3 = kp[3];
                uint4 sv = *reinterpret_cast<uint4 *>(sp);
                const bool up = k ? ((4u * o) & k) == 0u : ((4u * o) & 4u) == 0u;
                if (k == 0u) {                     // level 2: pairs (0,1) ascending, (2,3) descending
                    cex(a0, a1, true);
                    cex(a2, a3, false);
                    cex32(sv.x, sv.y, true);
                    cex32(sv.z, sv.w, false);
                }
                cex(a0, a2, up);                   // j = 2
                cex(a1, a3, up);
                cex(a0, a1, up);                   // j = 1
                cex(a2, a3, up);
                cex32(sv.x, sv.z, up);
                cex32(sv.y, sv.w, up);
                cex32(sv.x, sv.y, up);
                cex32(sv.z, sv.w, up);
                kp[0] = a0;
                kp[1] = a1;
                kp[2] = a2;
                kp[3] = a3;
                *reinterpret_cast<uint4 *>(sp) = sv;
            }
            __syncthreads();
        };
        local(0u);
        for (uint32_t k = 8; k <= N; k <<= 1) {
            for (uint32_t j = k >> 1; j >= 4u; j >>= 1) {
                for (uint32_t t = tid; t < (N >> 1); t += T) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));   // bit j clear
                    const uint32_t q = i | j;
                    const bool up = (i & k) == 0u;
                    const uint64_t ka = key[i], kb = key[q];
                    if ((ka > kb) == up && ka != kb) {
                        key[i] = kb;
                        key[q] = ka;
                    }
                    const uint32_t sa = skey[i], sb = skey[q];
                    if ((sa > sb) == up && sa != sb) {
                        skey[i] = sb;
                        skey[q] = sa;
                    }
                }
                __syncthreads();
            }
            local(k);
        }
        // back to (start << 32 | stop) and the position, in place
        const uint64_t mt = bt >= 32u ? 0xFFFFFFFFull : (1ull << bt) - 1ull, mp = (1ull << bp) - 1ull;
        for (uint32_t i = tid; i < n; i += T) {
            const uint64_t c = key[i];
            const uint32_t a = (uint32_t)(c >> (bt + bp)) + smin, b = (uint32_t)((c >> bp) & mt) + tmin;
            key[i] = ((uint64_t)a << 32) | b;
            pos[i] = (uint16_t)(c & mp);
        }
        __syncthreads();
    } else {
        for (uint32_t k = 2; k <= N; k <<= 1) {
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                for (uint32_t t = tid; t < (N >> 1); t += T) {
                    const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));   // bit j clear
                    const uint32_t q = i | j;
                    const bool up = (i & k) == 0u;
                    const uint64_t ka = key[i], kb = key[q];
                    const uint32_t pa = pos[i], pb = pos[q];
                    const bool gt = ka > kb || (ka == kb && pa > pb);
                    if (gt == up) {
                        key[i] = kb;
                        key[q] = ka;
                        pos[i] = (uint16_t)pb;
                        pos[q] = (uint16_t)pa;
                    }
                    const uint32_t sa = skey[i], sb = skey[q];
                    if ((sa > sb) == up && sa != sb) {
                        skey[i] = sb;
                        skey[q] = sa;
                    }
                }
                __syncthreads();
            }
        }
    }
    // sorted arrays out; max(stop - start)
    uint32_t ml = 0;
    for (uint32_t i = tid; i < n; i += T) {
        const uint64_t kk = key[i];
        const uint32_t a = (uint32_t)(kk >> 32), b = (uint32_t)kk;
        lstart[lo + i] = a;
        lrec[lo + i] = IvRec{a, b, (uint64_t)lo + pos[i]};
        stops_sorted[lo + i] = skey[i];
        if (b > a) ml = max(ml, b - a);
    }
    for (int d = 32; d; d >>= 1) ml = max(ml, (uint32_t)__shfl_xor((int)ml, d, 64));
    if ((tid & 63u) == 0) scr[tid >> 6] = ml;
    __syncthreads();
    if (tid == 0) {
        for (uint32_t w = 1; w < T / 64u; ++w) ml = max(ml, scr[w]);
        IndexGroup G;
        G.off = lo;
        G.n = n;
        G.maxlen = ml;
        const KeyDir none{0u, 0u, 0u, 0u};
        const uint32_t s0 = n ? (uint32_t)(key[0] >> 32) : 0u, s1 = n ? (uint32_t)(key[n - 1] >> 32) : 0u;
        const uint32_t t0 = n ? skey[0] : 0u, t1 = n ? skey[n - 1] : 0u;
        G.start = n ? dir_params(s0, s1, n) : none;
        G.stop = n ? dir_params(t0, t1, n) : none;
        G.bk_start = n ? dir_params(min(s0, t0), max(s1, t1), (n >> kCellShift) + 1u) : none;
        G.bk_stop = G.bk_start;
        groups[g] = G;
        cgroups[g] = CountGroup{lo, n, G.bk_start.key0, G.bk_start.nb, G.bk_stop.key0, G.bk_stop.nb, G.bk_start.shift,
                                G.bk_stop.shift};
        scr[64] = G.start.key0;
        scr[65] = G.start.shift;
        scr[66] = G.start.nb;
        scr[67] = G.bk_start.key0;
        scr[68] = G.bk_start.shift;
        scr[69] = G.bk_start.nb;
    }
    __syncthreads();
    if (n == 0) {
        if (tid == 0) dir_start[(uint64_t)lo + g] = 0u;
        return;
    }
    // bucket directory over the starts (locate): slot b of the group's n + 1 at dir_start[lo + g + b]
    {
        const uint32_t key0 = scr[64], shift = scr[65], nb = scr[66];
        for (uint32_t b = tid; b <= nb; b += T) {
            uint32_t r = n;
            if (b < nb) {
                const uint64_t edge = (uint64_t)key0 + ((uint64_t)b << shift);
                uint32_t a = 0, z = n;
                while (a < z) {
                    const uint32_t mid = a + ((z - a) >> 1);
                    if ((key[mid] >> 32) < edge)
                        a = mid + 1;
                    else
                        z = mid;
                }
                r = a;
            }
            dir_start[(uint64_t)lo + g + b] = r;
        }
    }
    // cell records of the count path: cell b of the group at (lo >> kCellShift) + 2g + b, the starts' and the
    // stops' record side by side
    {
        const uint32_t key0 = scr[67], shift = scr[68], nb = scr[69];
        const uint64_t cell0 = (uint64_t)(lo >> kCellShift) + 2ull * g;
        for (uint32_t b = tid; b < nb; b += T) {
            const uint64_t edge = (uint64_t)key0 + ((uint64_t)b << shift);
            BkRec rs, rt;
            {
                uint32_t a = 0, z = n;
                while (a < z) {
                    const uint32_t mid = a + ((z - a) >> 1);
                    if ((key[mid] >> 32) < edge)
                        a = mid + 1;
                    else
                        z = mid;
                }
                rs.rank = a;
#pragma unroll
                for (uint32_t i = 0; i < 7u; ++i) rs.k[i] = a + i < n ? (uint32_t)(key[a + i] >> 32) : 0xffffffffu;
            }
            {
                uint32_t a = 0, z = n;
                while (a < z) {
                    const uint32_t mid = a + ((z - a) >> 1);
                    if ((uint64_t)skey[mid] < edge)
                        a = mid + 1;
                    else
                        z = mid;
                }
                rt.rank = a;
#pragma unroll
                for (uint32_t i = 0; i < 7u; ++i) rt.k[i] = a + i < n ? skey[a + i] : 0xffffffffu;
            }
            bk_start[2u * (cell0 + b)] = rs;
            bk_stop[2u * (cell0 + b)] = rt;
        }
    }
}

template <uint32_t CAP, uint32_t T>
hipError_t index_build_launch(gams_gpu_t *h, hipStream_t st, const uint32_t *off32, uint32_t n_groups, const uint32_t *starts,
                              const uint32_t *stops, gams_index_t *ix) {
    const size_t lds = (size_t)CAP * (8 + 4 + 2) + 96 * 4;
    auto kern = index_build_kernel<CAP, T>;
    hipError_t e = gams_lds_attr(h, reinterpret_cast<const void *>(kern), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_groups), dim3(T), lds, st, off32, n_groups, starts, stops, ix->d_lstart, ix->d_lrec,
                       ix->d_stops, ix->d_groups, ix->d_cgroups, ix->d_dir_start, ix->d_bk_start, ix->d_bk_stop);
    return hipGetLastError();
}

// Host side of the directory: keys[0..n) ascending (already biased); dir gets nb+1 <= n+1 entries.
KeyDir build_dir(const uint32_t *keys, uint32_t n, uint32_t *dir) {
    KeyDir d{0u, 0u, 0u, 0u};
    if (n == 0) {
        dir[0] = 0;
        return d;
    }
    d.key0 = keys[0];
    const uint64_t range = (uint64_t)keys[n - 1] - (uint64_t)keys[0];
    while ((range >> d.shift) >= n) ++d.shift;     // (range >> shift) + 1 <= n buckets; ends at shift <= 32
    d.nb = (uint32_t)(range >> d.shift) + 1u;
    uint32_t i = 0;
    for (uint32_t b = 0; b < d.nb; ++b) {
        const uint64_t edge = (uint64_t)d.key0 + ((uint64_t)b << d.shift);
        while (i < n && (uint64_t)keys[i] < edge) ++i;
        dir[b] = i;
    }
    dir[d.nb] = n;
    return d;
}

// run fn(g) for every group on up to 16 host threads (the per-group sorts dominate index creation)
template <typename F>
void for_groups(uint32_t n_groups, F fn) {
    const unsigned T = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), n_groups}));
    if (T <= 1) {
        for (uint32_t g = 0; g < n_groups; ++g) fn(g);
        return;
    }
    std::atomic<uint32_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t)
        pool.emplace_back([&] {
            for (uint32_t g = next.fetch_add(1); g < n_groups; g = next.fetch_add(1)) fn(g);
        });
    for (auto &th : pool) th.join();
}

template <typename T>
hipError_t to_device(T **d, const T *hsrc, size_t n) {
    hipError_t e = hipMalloc(d, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (n) e = hipMemcpy(*d, hsrc, n * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

#define Q_HIP(call)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return gams_fail(h, GAMS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Host columns in, one host column out, through the three query kernels.  The queries go in chunks of
// 2^21: chunk c+1 is copied in (copy stream) while chunk c is searched (compute stream) and chunk c-1 is
// copied out (readback stream), two device slots.  With page-locked host arrays (gams_gpu_host_alloc) the
// three overlap and the call runs at the rate of the PCIe link; pageable arrays work, staged by the runtime.
struct QCol {
    const void *host;
    size_t elem;
};

template <typename F>
int query_pipeline(gams_gpu_t *h, uint64_t nq, const QCol *cols, int ncol, void *out_host, size_t out_elem, F launch) {
    const uint64_t CH = std::min<uint64_t>(nq, 1ull << 21);
    const uint64_t nchunk = (nq + CH - 1) / CH;
    const int nslot = nchunk > 1 ? 2 : 1;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t slot_bytes = al(CH * out_elem);
    for (int k = 0; k < ncol; ++k) slot_bytes += al(CH * cols[k].elem);
    uint8_t *block = nullptr;
    size_t block_cap = 0;
    hipError_t e = gams_pool_alloc(h, false, slot_bytes * nslot, reinterpret_cast<void **>(&block), &block_cap);
    if (e != hipSuccess)
        return gams_fail(h, e == hipErrorOutOfMemory ? GAMS_ENOMEM : GAMS_EHIP,
                         std::string("query buffers: ") + hipGetErrorString(e));
    struct Guard {
        gams_gpu_t *h;
        uint8_t *p;
        size_t cap;
        ~Guard() {
            (void)hipStreamSynchronize(h->copy);
            (void)hipStreamSynchronize(h->compute);
            (void)hipStreamSynchronize(h->readback);
            gams_pool_free(h, false, p, cap);
        }
    } guard{h, block, block_cap};
    for (int s = 0; s < 2; ++s)
        for (int k = 0; k < 3; ++k)
            if (!h->q_ev[s][k]) Q_HIP(hipEventCreateWithFlags(&h->q_ev[s][k], hipEventDisableTiming));
    while (h->kq.size() < nchunk) {
        hipEvent_t a = nullptr, b = nullptr;
        Q_HIP(hipEventCreate(&a));
        Q_HIP(hipEventCreate(&b));
        h->kq.emplace_back(a, b);
    }
    h->kq_used = 0;
    // whatever the caller queued on the compute stream before (index build, uploads) comes first
    Q_HIP(hipEventRecord(h->k0, h->compute));
    Q_HIP(hipStreamWaitEvent(h->copy, h->k0, 0));
    std::vector<void *> d_in((size_t)ncol);
    for (uint64_t c = 0; c < nchunk; ++c) {
        const int s = (int)(c & 1);
        const uint64_t lo = c * CH, n = std::min<uint64_t>(CH, nq - lo);
        uint8_t *p = block + (size_t)s * slot_bytes;
        if (c >= 2) Q_HIP(hipStreamWaitEvent(h->copy, h->q_ev[s][1], 0));       // chunk c-2's kernel has read its inputs
        for (int k = 0; k < ncol; ++k) {
            d_in[(size_t)k] = p;
            Q_HIP(hipMemcpyAsync(p, static_cast<const uint8_t *>(cols[k].host) + lo * cols[k].elem, n * cols[k].elem,
                                 hipMemcpyHostToDevice, h->copy));
            p += al(CH * cols[k].elem);
        }
        void *d_out = p;
        Q_HIP(hipEventRecord(h->q_ev[s][0], h->copy));
        Q_HIP(hipStreamWaitEvent(h->compute, h->q_ev[s][0], 0));
        if (c >= 2) Q_HIP(hipStreamWaitEvent(h->compute, h->q_ev[s][2], 0));    // chunk c-2's results are out
        Q_HIP(hipEventRecord(h->kq[(size_t)c].first, h->compute));
        launch(d_in.data(), d_out, n, h->compute);
        Q_HIP(hipGetLastError());
        Q_HIP(hipEventRecord(h->kq[(size_t)c].second, h->compute));
        Q_HIP(hipEventRecord(h->q_ev[s][1], h->compute));
        Q_HIP(hipStreamWaitEvent(h->readback, h->q_ev[s][1], 0));
        Q_HIP(hipMemcpyAsync(static_cast<uint8_t *>(out_host) + lo * out_elem, d_out, n * out_elem,
                             hipMemcpyDeviceToHost, h->readback));
        Q_HIP(hipEventRecord(h->q_ev[s][2], h->readback));
    }
    h->kq_used = (int)nchunk;
    h->k_valid = true;
    Q_HIP(hipStreamSynchronize(h->readback));
    return GAMS_OK;
}

}  // namespace

extern "C" {

int gams_index_create(gams_gpu_t *h, uint32_t n_groups, const uint64_t *group_off, const uint32_t *starts,
                      const uint32_t *stops, gams_index_t **out) {
    if (!h || !out || !group_off) return gams_fail(h, GAMS_EINVAL, "index_create: null argument");
    const uint64_t m = group_off[n_groups];
    if (m && (!starts || !stops)) return gams_fail(h, GAMS_EINVAL, "index_create: null interval arrays");
    for (uint32_t g = 0; g < n_groups; ++g)
        if (group_off[g] > group_off[g + 1]) return gams_fail(h, GAMS_EINVAL, "index_create: group_off not ascending");
    if (m > 0xfffffff0ull)
        return gams_fail(h, GAMS_EUNSUPPORTED, "index_create: more than 2^32-16 intervals in one index");
    GAMS_HIP(h, hipSetDevice(h->device));
    // Lapper::new on the device: intervals.sort() by (start, stop) inside every group = a stable
    // segmented radix sort of the packed 64-bit keys; the stops are also sorted on their own.  The
    // bucket directories and Lapper::max_len come from three small kernels over the sorted arrays.
    const uint32_t ng1 = std::max<uint32_t>(n_groups, 1);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_cg = al((size_t)ng1 * sizeof(CountGroup));
    const size_t b_groups = al((size_t)ng1 * sizeof(IndexGroup)), b_u32 = al(std::max<uint64_t>(m, 1) * 4),
                 b_rec = al(std::max<uint64_t>(m, 1) * sizeof(IvRec)), b_dir = al((m + n_groups + 1) * 4);
    const uint64_t bk_slots = (m >> kCellShift) + 2ull * n_groups + 2;
    const size_t b_bk = al(bk_slots * sizeof(BkRec));
    gams_index_t *ix = new gams_index_t();
    ix->n_groups = n_groups;
    ix->m = m;
    hipError_t e = gams_pool_alloc(h, false, b_groups + b_cg + 2 * b_u32 + b_rec + b_dir + 2 * b_bk,
                                   reinterpret_cast<void **>(&ix->arena), &ix->arena_bytes);
    // scratch: raw columns, packed keys in/out, permutation in/out, 32-bit offsets, radix-sort storage
    const size_t b_key = al(std::max<uint64_t>(m, 1) * 8), b_off = al(((size_t)n_groups + 1) * 4);
    uint8_t *scratch = nullptr;
    size_t scratch_bytes = 0, tmp_bytes_pairs = 0, tmp_bytes_keys = 0;
    uint8_t *d_tmp = nullptr;
    size_t d_tmp_bytes = 0;
    auto fail = [&](hipError_t err, const char *what) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->compute);
        gams_pool_free(h, false, scratch, scratch_bytes);
        gams_pool_free(h, false, d_tmp, d_tmp_bytes);
        gams_index_destroy(h, ix);
        return gams_fail(h, err == hipErrorOutOfMemory ? GAMS_ENOMEM : GAMS_EHIP,
                         std::string("index_create: ") + what + ": " + hipGetErrorString(err));
    };
    if (e != hipSuccess) return fail(e, "hipMalloc(index)");
    {
        uint8_t *p = ix->arena;
        ix->d_groups = reinterpret_cast<IndexGroup *>(p);
        p += b_groups;
        ix->d_cgroups = reinterpret_cast<CountGroup *>(p);
        p += b_cg;
        ix->d_stops = reinterpret_cast<uint32_t *>(p);
        p += b_u32;
        ix->d_lstart = reinterpret_cast<uint32_t *>(p);
        p += b_u32;
        ix->d_lrec = reinterpret_cast<IvRec *>(p);
        p += b_rec;
        ix->d_dir_start = reinterpret_cast<uint32_t *>(p);
        p += b_dir;
        ix->d_bk_start = reinterpret_cast<BkRec *>(p);     // 2 * bk_slots records, interleaved: cell j = [2j] starts, [2j+1] stops
        ix->d_bk_stop = ix->d_bk_start + 1;
    }
    e = gams_pool_alloc(h, false, 2 * b_u32 + 2 * b_key + 2 * b_u32 + b_off, reinterpret_cast<void **>(&scratch),
                        &scratch_bytes);
    if (e != hipSuccess) return fail(e, "hipMalloc(scratch)");
    uint32_t *d_starts_in = reinterpret_cast<uint32_t *>(scratch);
    uint32_t *d_stops_in = reinterpret_cast<uint32_t *>(scratch + b_u32);
    uint64_t *d_key_in = reinterpret_cast<uint64_t *>(scratch + 2 * b_u32);
    uint64_t *d_key_out = reinterpret_cast<uint64_t *>(scratch + 2 * b_u32 + b_key);
    uint32_t *d_val_in = reinterpret_cast<uint32_t *>(scratch + 2 * b_u32 + 2 * b_key);
    uint32_t *d_val_out = reinterpret_cast<uint32_t *>(scratch + 3 * b_u32 + 2 * b_key);
    uint32_t *d_off32 = reinterpret_cast<uint32_t *>(scratch + 4 * b_u32 + 2 * b_key);
    hipStream_t st = h->compute;
    std::vector<uint32_t> off32((size_t)n_groups + 1);
    for (uint32_t g = 0; g <= n_groups; ++g) off32[g] = (uint32_t)group_off[g];
    if ((e = hipMemcpyAsync(d_off32, off32.data(), off32.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess)
        return fail(e, "copy offsets");
    uint32_t max_n = 0;
    for (uint32_t g = 0; g < n_groups; ++g) max_n = std::max(max_n, off32[g + 1] - off32[g]);
    if (m) {
        if ((e = hipMemcpyAsync(d_starts_in, starts, m * 4, hipMemcpyHostToDevice, st)) != hipSuccess)
            return fail(e, "copy starts");
        if ((e = hipMemcpyAsync(d_stops_in, stops, m * 4, hipMemcpyHostToDevice, st)) != hipSuccess)
            return fail(e, "copy stops");
    }
    if (n_groups && max_n <= kBuildCap) {
        // every group fits a workgroup: sort and every derived table in one kernel, one workgroup per group
        if (max_n <= 256)
            e = index_build_launch<256, 128>(h, st, d_off32, n_groups, d_starts_in, d_stops_in, ix);
        else if (max_n <= 1024)
            e = index_build_launch<1024, 512>(h, st, d_off32, n_groups, d_starts_in, d_stops_in, ix);
        else if (max_n <= 2048)
            e = index_build_launch<2048, 1024>(h, st, d_off32, n_groups, d_starts_in, d_stops_in, ix);
        else if (max_n <= 4096)
            e = index_build_launch<4096, 1024>(h, st, d_off32, n_groups, d_starts_in, d_stops_in, ix);
        else
            e = index_build_launch<8192, 1024>(h, st, d_off32, n_groups, d_starts_in, d_stops_in, ix);
        if (e != hipSuccess) return fail(e, "group build");
    } else if (m) {
        // a group beyond the workgroup's LDS: the library's segmented radix sort + the table kernels
        const unsigned blocks = (unsigned)((m + 255) / 256);
        hipLaunchKernelGGL(index_pack_kernel, dim3(blocks), dim3(256), 0, st, d_starts_in, d_stops_in, m, d_key_in,
                           d_val_in);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "pack");
        e = rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes_pairs, d_key_in, d_key_out, d_val_in, d_val_out,
                                                (unsigned)m, n_groups, d_off32, d_off32 + 1, 0, 64, st);
        if (e != hipSuccess) return fail(e, "radix sort (size query)");
        e = rocprim::segmented_radix_sort_keys(nullptr, tmp_bytes_keys, d_stops_in, ix->d_stops, (unsigned)m, n_groups,
                                               d_off32, d_off32 + 1, 0, 32, st);
        if (e != hipSuccess) return fail(e, "radix sort (size query)");
        e = gams_pool_alloc(h, false, std::max<size_t>(std::max(tmp_bytes_pairs, tmp_bytes_keys), 256),
                            reinterpret_cast<void **>(&d_tmp), &d_tmp_bytes);
        if (e != hipSuccess) return fail(e, "hipMalloc(sort storage)");
        e = rocprim::segmented_radix_sort_pairs(d_tmp, tmp_bytes_pairs, d_key_in, d_key_out, d_val_in, d_val_out,
                                                (unsigned)m, n_groups, d_off32, d_off32 + 1, 0, 64, st);
        if (e != hipSuccess) return fail(e, "radix sort of (start, stop)");
        e = rocprim::segmented_radix_sort_keys(d_tmp, tmp_bytes_keys, d_stops_in, ix->d_stops, (unsigned)m, n_groups,
                                               d_off32, d_off32 + 1, 0, 32, st);
        if (e != hipSuccess) return fail(e, "radix sort of stops");
        hipLaunchKernelGGL(index_unpack_kernel, dim3(blocks), dim3(256), 0, st, d_key_out, d_val_out, m, ix->d_lstart,
                           ix->d_lrec);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "unpack");
    }
    if (n_groups && max_n > kBuildCap) {
        hipLaunchKernelGGL(index_group_kernel, dim3(n_groups), dim3(64), 0, st, d_off32, n_groups, ix->d_lrec,
                           ix->d_lstart, ix->d_stops, ix->d_groups, ix->d_cgroups);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "group records");
        const uint64_t slots = m + n_groups;        // the last group's directory ends at m + n_groups (inclusive slot)
        hipLaunchKernelGGL(index_dir_kernel, dim3((unsigned)((slots + 1 + 255) / 256)), dim3(256), 0, st, d_off32,
                           n_groups, slots + 1, ix->d_groups, ix->d_lstart, ix->d_dir_start);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "directories");
        hipLaunchKernelGGL(index_bk_kernel, dim3((unsigned)((bk_slots + 255) / 256)), dim3(256), 0, st, d_off32,
                           n_groups, bk_slots, ix->d_groups, ix->d_lstart, ix->d_stops, ix->d_bk_start, ix->d_bk_stop);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e, "bucket records");
    }
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return fail(e, "index build");
    gams_pool_free(h, false, scratch, scratch_bytes);
    gams_pool_free(h, false, d_tmp, d_tmp_bytes);
    *out = ix;
    return GAMS_OK;
}

void gams_index_destroy(gams_gpu_t *h, gams_index_t *ix) {
    if (!ix) return;
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->compute);
    }
    gams_pool_free(h, false, ix->arena, ix->arena_bytes);
    delete ix;
}


int gams_gpu_count(gams_gpu_t *h, gams_index_t *ix, const uint32_t *group, const uint32_t *qs,
                   const uint32_t *qe, uint64_t nq, int32_t *count) {
    if (!h || !ix || (nq && (!group || !qs || !qe || !count)))
        return gams_fail(h, GAMS_EINVAL, "gpu_count: null argument");
    if (nq == 0) return GAMS_OK;
    if ((nq + 255) / 256 > 0x7fffffffull) return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_count: too many queries");
    GAMS_HIP(h, hipSetDevice(h->device));
    const QCol cols[3] = {{group, 4}, {qs, 4}, {qe, 4}};
    return query_pipeline(h, nq, cols, 3, count, sizeof(int32_t), [&](void *const *d, void *d_out, uint64_t n, hipStream_t st) {
        hipLaunchKernelGGL(interval_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ix->d_cgroups,
                           ix->d_lstart, ix->d_stops, ix->d_bk_start, ix->d_bk_stop, ix->n_groups,
                           static_cast<const uint32_t *>(d[0]), static_cast<const uint32_t *>(d[1]),
                           static_cast<const uint32_t *>(d[2]), n, static_cast<int32_t *>(d_out));
    });
}

int gams_gpu_locate(gams_gpu_t *h, gams_index_t *ix, const uint32_t *group, const uint32_t *qs,
                    const uint32_t *qe, uint64_t nq, int64_t *hit) {
    if (!h || !ix || (nq && (!group || !qs || !qe || !hit)))
        return gams_fail(h, GAMS_EINVAL, "gpu_locate: null argument");
    if (nq == 0) return GAMS_OK;
    if ((nq + 255) / 256 > 0x7fffffffull) return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_locate: too many queries");
    GAMS_HIP(h, hipSetDevice(h->device));
    const QCol cols[3] = {{group, 4}, {qs, 4}, {qe, 4}};
    return query_pipeline(h, nq, cols, 3, hit, sizeof(int64_t), [&](void *const *d, void *d_out, uint64_t n, hipStream_t st) {
        hipLaunchKernelGGL(interval_locate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ix->d_groups,
                           ix->d_lstart, ix->d_lrec, ix->d_dir_start, ix->n_groups, static_cast<const uint32_t *>(d[0]),
                           static_cast<const uint32_t *>(d[1]), static_cast<const uint32_t *>(d[2]), n,
                           static_cast<int64_t *>(d_out));
    });
}

int gams_spans_create(gams_gpu_t *h, uint32_t n_groups, const uint64_t *group_off, const int32_t *lo,
                      const int32_t *hi, gams_spans_t **out) {
    if (!h || !out || !group_off) return gams_fail(h, GAMS_EINVAL, "spans_create: null argument");
    const uint64_t m = group_off[n_groups];
    if (m && (!lo || !hi)) return gams_fail(h, GAMS_EINVAL, "spans_create: null span arrays");
    std::vector<SpanRec> rec(m);
    for (uint32_t g = 0; g < n_groups; ++g) {
        if (group_off[g] > group_off[g + 1]) return gams_fail(h, GAMS_EINVAL, "spans_create: group_off not ascending");
        if (group_off[g + 1] - group_off[g] > 0xfffffff0ull)
            return gams_fail(h, GAMS_EUNSUPPORTED, "spans_create: a group holds more than 2^32-16 spans");
        uint64_t c = 0;
        for (uint64_t i = group_off[g]; i < group_off[g + 1]; ++i) {
            if (hi[i] < lo[i] || (i > group_off[g] && lo[i] <= hi[i - 1]))
                return gams_fail(h, GAMS_EINVAL, "spans_create: spans must be sorted and disjoint");
            rec[i] = SpanRec{lo[i], hi[i], c};
            c += (uint64_t)((int64_t)hi[i] - lo[i] + 1);
        }
    }
    std::vector<SpanGroup> groups(std::max<uint32_t>(n_groups, 1));
    std::vector<uint32_t> dir(m + n_groups + 1);
    for_groups(n_groups, [&](uint32_t g) {
        const uint64_t o = group_off[g];
        const uint32_t n = (uint32_t)(group_off[g + 1] - o);
        std::vector<uint32_t> biased(n);
        for (uint32_t i = 0; i < n; ++i) biased[i] = (uint32_t)lo[o + i] ^ 0x80000000u;
        groups[g].off = o;
        groups[g].n = n;
        groups[g].lo = build_dir(biased.data(), n, dir.data() + o + g);
    });
    GAMS_HIP(h, hipSetDevice(h->device));
    gams_spans_t *sp = new gams_spans_t();
    sp->n_groups = n_groups;
    sp->m = m;
    hipError_t e = to_device(&sp->d_groups, groups.data(), groups.size());
    if (e == hipSuccess) e = to_device(&sp->d_rec, rec.data(), m);
    if (e == hipSuccess) e = to_device(&sp->d_dir_lo, dir.data(), dir.size());
    const uint64_t slots = m + n_groups + 1;
    if (e == hipSuccess) e = hipMalloc(&sp->d_cells, slots * sizeof(SpanCell));
    if (e == hipSuccess && n_groups) {
        hipLaunchKernelGGL(span_cell_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, h->compute, sp->d_groups,
                           n_groups, sp->d_rec, sp->d_dir_lo, slots, sp->d_cells);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(h->compute);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        gams_spans_destroy(h, sp);
        return gams_fail(h, e == hipErrorOutOfMemory ? GAMS_ENOMEM : GAMS_EHIP,
                         std::string("spans_create: ") + hipGetErrorString(e));
    }
    *out = sp;
    return GAMS_OK;
}

void gams_spans_destroy(gams_gpu_t *h, gams_spans_t *sp) {
    if (!sp) return;
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->compute);
    }
    (void)hipFree(sp->d_groups);
    (void)hipFree(sp->d_dir_lo);
    (void)hipFree(sp->d_rec);
    (void)hipFree(sp->d_cells);
    delete sp;
}

int gams_gpu_cover(gams_gpu_t *h, gams_spans_t *sp, const uint32_t *group, const int32_t *clip_lo,
                   const int32_t *clip_hi, const int32_t *qs, const int32_t *qe, uint64_t nq, float *prop) {
    if (!h || !sp || (nq && (!group || !clip_lo || !clip_hi || !qs || !qe || !prop)))
        return gams_fail(h, GAMS_EINVAL, "gpu_cover: null argument");
    if (nq == 0) return GAMS_OK;
    if ((nq + 255) / 256 > 0x7fffffffull) return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_cover: too many queries");
    GAMS_HIP(h, hipSetDevice(h->device));
    const QCol cols[5] = {{group, 4}, {clip_lo, 4}, {clip_hi, 4}, {qs, 4}, {qe, 4}};
    return query_pipeline(h, nq, cols, 5, prop, sizeof(float), [&](void *const *d, void *d_out, uint64_t n, hipStream_t st) {
        hipLaunchKernelGGL(span_cover_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sp->d_groups, sp->d_rec,
                           sp->d_dir_lo, sp->d_cells, sp->n_groups, static_cast<const uint32_t *>(d[0]),
                           static_cast<const int32_t *>(d[1]), static_cast<const int32_t *>(d[2]),
                           static_cast<const int32_t *>(d[3]), static_cast<const int32_t *>(d[4]), n,
                           static_cast<float *>(d_out));
    });
}

}  // extern "C"
