// wave_rows.hpp -- the TSV rows of `gams wave` made on the device (included by wave.hip only).
//
// wave.rs:157-252 turns the signalled windows of a ctg into rows: crests and troughs separately, windows whose
// ranges overlap are merged (merge_ints: a graph over the pairwise intersections, connected components), a
// component prints once -- at its first window -- as "{chr}(+):{min}-{max}", a window without a partner as
// "{chr}:{start}-{end}", both followed by "\t{gc_content}\t{signal}\n" with the FIRST window's gc_content.
// All windows have one size and sit on a grid of `step`, so whether two of them are linked depends on their
// distance d alone, and for every --coverage up to 1 (the default is 0.2) EVERY overlap links: d <= dmax =
// ceil(size / step) - 1.  Components are then chains of consecutive same-sign windows at most dmax apart:
//
//   rows_link_kernel   per packed peak record (ordered by ctg, window): has it a same-sign neighbour within dmax
//                      in front / behind (a bounded look at the adjacent records)?  -> head / tail / merged
//                      flags; block-wise prefix maximum of "position of the last head", per sign;
//   rows_heads_kernel  prefix maximum over the blocks' aggregates (one workgroup);
//   rows_tail_kernel   every tail tells its head where the component ends (between a head and its tail there
//                      is no other head of that sign, so the tail's head is the last head at or before it);
//   rows_len_kernel    byte length of every head's row (others: 0) + block sums;   wave_offsets_kernel: block offsets;
//   rows_write_kernel  block-wise prefix of the lengths, then every head writes its row where it belongs.
//
// The text is what the host layer's merge + formatting produced before (tests: the reference's I.peaks.tsv, byte
// for byte); gc_content is printed through a table of the size + 1 possible values, formatted on the host the way
// Rust's `{}` prints an f32 (shortest digits that round-trip).  Other coverages (a distance window [dmin, dmax]
// with dmin > 1: components are no longer chains) stay with the host's union-find.
#pragma once

#include "wave_kernels.hpp"

namespace {

constexpr uint32_t kRowsPer = 2;             // records per thread (a pass over 184 k records is ~360 workgroups: enough to fill the chip)
constexpr uint32_t kRowsBlock = 256 * kRowsPer;   // records per workgroup
constexpr uint32_t kRowsStage = 16384;       // bytes of a block's text staged in LDS by rows_write_kernel
constexpr uint32_t kGcStride = 24;           // bytes per entry of the gc_content text table: [len][chars]
constexpr uint8_t kRowHead = 1, kRowMerged = 2, kRowTail = 4;

struct RowCtg {                              // per ctg, 16 B
    uint32_t name_off, name_len;             // its chromosome's name in the blob
    int32_t chr_start;                       // chromosome coordinate of its first base
    uint32_t pad;
};

struct RowArgs {
    const gams_peak_t *rec;                  // packed peaks of the pass, ordered by (ctg, window)
    const unsigned long long *n_rec;         // [0] = how many, [1] = the fullest tile (device side: nobody waited for the pass)
    uint64_t cap;                            // records the packed array and the tables below have room for
    uint32_t tile_cap;                       // records a tile's slot holds
    const RowCtg *ctgs;
    const char *names;
    const uint8_t *gctab;                    // [size + 1][kGcStride]
    uint32_t size, step, dmax;
    uint8_t *flags;                          // per record
    int2 *headpos;                           // per record: position of the last crest head / trough head at or before it
    int2 *blk_head;                          // per block: the block's last heads; then (rows_heads_kernel) the heads before the block
    uint32_t *tailwin;                       // per record (read at heads): window of the component's last member
    uint32_t *len;                           // per record: bytes of its row
    uint32_t *blk_len;                       // per block
    const unsigned long long *blk_off;       // per block: exclusive prefix of blk_len; [nb_cap] = all bytes
    uint32_t nb_cap;                         // blocks the tables have room for = cap / kRowsBlock rounded up
    char *text;
    uint64_t text_cap;
    uint32_t n_ctg;
    unsigned long long *words;               // what the host reads, one block: [0] records, [1] text bytes, [2] peaks, [3] fullest
                                             // tile, [4 + c] where ctg c's rows begin (~0: it has none; the host fills those)
};

// Records to work on.  The packed records are only complete when no tile overflowed its slot and the packed
// array held them all (wave_gather_kernel skips what does not fit, leaving whatever the pooled block held before);
// the host finds that out when it reads the totals and runs the pass again with more room -- until then the
// rows kernels must not look at a single record.
__device__ __forceinline__ uint64_t rows_n(const RowArgs &a) {
    const unsigned long long total = a.n_rec[0], worst = a.n_rec[1];
    return (worst > a.tile_cap || total > a.cap) ? 0ull : (uint64_t)total;
}

__device__ __forceinline__ uint32_t dec_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
         : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
// decimal digits of v ending just before `end`; returns where they start
__device__ __forceinline__ char *put_dec_back(char *end, uint32_t v) {
    do {
        *--end = (char)('0' + v % 10u);
        v /= 10u;
    } while (v);
    return end;
}

__global__ __launch_bounds__(256) void rows_link_kernel(const RowArgs a) {
    __shared__ int2 wtot[4];
    const uint64_t n = rows_n(a);
    const uint64_t base = (uint64_t)blockIdx.x * kRowsBlock;
    if (base >= n) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    // thread t owns kRowsPer consecutive records (the prefix maximum runs through them in order)
    int2 run = make_int2(-1, -1);
    int2 mine[kRowsPer];
#pragma unroll
    for (uint32_t u = 0; u < kRowsPer; ++u) {
        const uint64_t i = base + kRowsPer * tid + u;
        mine[u] = make_int2(-1, -1);
        if (i >= n) continue;
        const gams_peak_t r = a.rec[i];
        bool prev = false, next = false;
        for (uint64_t j = i; j-- > 0;) {                       // same-sign neighbour within dmax in front?
            const gams_peak_t q = a.rec[j];
            if (q.ctg != r.ctg || r.window - q.window > a.dmax) break;
            if (q.signal == r.signal) {
                prev = true;
                break;
            }
        }
        for (uint64_t j = i + 1; j < n; ++j) {                 // ... behind?
            const gams_peak_t q = a.rec[j];
            if (q.ctg != r.ctg || q.window - r.window > a.dmax) break;
            if (q.signal == r.signal) {
                next = true;
                break;
            }
        }
        const uint8_t f = (uint8_t)((prev ? 0 : kRowHead) | ((prev || next) ? kRowMerged : 0) | (next ? 0 : kRowTail));
        a.flags[i] = f;
        if (!prev) {
            if (r.signal > 0)
                run.x = (int)(i - base);
            else
                run.y = (int)(i - base);
        }
        mine[u] = run;                                         // inclusive, inside this thread's own
    }
    // prefix maximum across the threads of the block (positions relative to the block, -1 = none)
    int2 inc = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int ox = __shfl_up(inc.x, d, 64), oy = __shfl_up(inc.y, d, 64);
        if ((int)lane >= d) {
            inc.x = max(inc.x, ox);
            inc.y = max(inc.y, oy);
        }
    }
    if (lane == 63u) wtot[wv] = inc;
    __syncthreads();
    int2 before = make_int2(-1, -1);                           // heads in the waves before this one
    for (uint32_t w = 0; w < wv; ++w) {
        before.x = max(before.x, wtot[w].x);
        before.y = max(before.y, wtot[w].y);
    }
    int2 excl;                                                  // heads in the threads before this one
    excl.x = max(before.x, __shfl_up(inc.x, 1, 64));
    excl.y = max(before.y, __shfl_up(inc.y, 1, 64));
    if (lane == 0u) excl = before;
#pragma unroll
    for (uint32_t u = 0; u < kRowsPer; ++u) {
        const uint64_t i = base + kRowsPer * tid + u;
        if (i >= n) continue;
        const int hx = max(mine[u].x, excl.x), hy = max(mine[u].y, excl.y);
        // block-relative -> absolute (fits 32 bits: the rows path is limited to 2^31 records), -1 stays -1
        a.headpos[i] = make_int2(hx < 0 ? -1 : (int)(base + (uint32_t)hx), hy < 0 ? -1 : (int)(base + (uint32_t)hy));
    }
    if (tid == 255u) {
        const int2 tot = make_int2(max(before.x, inc.x), max(before.y, inc.y));
        a.blk_head[blockIdx.x] = make_int2(tot.x < 0 ? -1 : (int)(base + (uint32_t)tot.x), tot.y < 0 ? -1 : (int)(base + (uint32_t)tot.y));
    }
}

// exclusive prefix maximum over the blocks' last heads, in place (one workgroup)
__global__ __launch_bounds__(1024) void rows_heads_kernel(const RowArgs a) {
    __shared__ int2 wtot[16];
    for (uint32_t c = threadIdx.x; c <= a.n_ctg; c += 1024u) a.words[4u + c] = ~0ull;   // (rows_write_kernel sets the ctgs that have rows)
    const uint64_t n = rows_n(a);
    const uint32_t nb = (uint32_t)((n + kRowsBlock - 1) / kRowsBlock);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
    int2 mine = make_int2(-1, -1);
    for (uint32_t b = b0; b < b1; ++b) {
        const int2 v = a.blk_head[b];
        mine.x = max(mine.x, v.x);
        mine.y = max(mine.y, v.y);
    }
    int2 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int ox = __shfl_up(inc.x, d, 64), oy = __shfl_up(inc.y, d, 64);
        if ((int)lane >= d) {
            inc.x = max(inc.x, ox);
            inc.y = max(inc.y, oy);
        }
    }
    if (lane == 63u) wtot[wv] = inc;
    __syncthreads();
    int2 run = make_int2(-1, -1);
    for (uint32_t w = 0; w < wv; ++w) {
        run.x = max(run.x, wtot[w].x);
        run.y = max(run.y, wtot[w].y);
    }
    const int px = __shfl_up(inc.x, 1, 64), py = __shfl_up(inc.y, 1, 64);
    if (lane != 0u) {
        run.x = max(run.x, px);
        run.y = max(run.y, py);
    }
    for (uint32_t b = b0; b < b1; ++b) {
        const int2 v = a.blk_head[b];
        a.blk_head[b] = run;
        run.x = max(run.x, v.x);
        run.y = max(run.y, v.y);
    }
}

__global__ __launch_bounds__(256) void rows_tail_kernel(const RowArgs a) {
    const uint64_t n = rows_n(a);
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (!(a.flags[i] & kRowTail)) return;
    const gams_peak_t r = a.rec[i];
    const int2 loc = a.headpos[i], pre = a.blk_head[i / kRowsBlock];
    const int head = r.signal > 0 ? max(loc.x, pre.x) : max(loc.y, pre.y);
    a.tailwin[head] = r.window;             // a tail's component has a head at or before it, in the same ctg
}

// bytes of the row record i prints (0: it is printed with its component's head)
__device__ __forceinline__ uint32_t row_len(const RowArgs &a, uint64_t i, const gams_peak_t r, uint8_t f) {
    if (!(f & kRowHead)) return 0u;
    const RowCtg cg = a.ctgs[r.ctg];
    const uint32_t s = (uint32_t)cg.chr_start + r.window * a.step;
    const uint32_t lastw = (f & kRowMerged) ? a.tailwin[i] : r.window;
    const uint32_t e = (uint32_t)cg.chr_start + lastw * a.step + a.size - 1u;
    uint32_t len = cg.name_len + ((f & kRowMerged) ? 3u : 0u) + 1u + dec_digits(s);
    if (e != s) len += 1u + dec_digits(e);                      // IntSpan runlist: "s" for a single position
    len += 1u + a.gctab[(size_t)r.gc_count * kGcStride] + 1u + (r.signal > 0 ? 1u : 2u) + 1u;
    return len;
}

__global__ __launch_bounds__(256) void rows_len_kernel(const RowArgs a) {
    __shared__ uint32_t ws[4];
    const uint64_t n = rows_n(a);
    const uint64_t base = (uint64_t)blockIdx.x * kRowsBlock;
    if (base >= n) {
        if (threadIdx.x == 0) a.blk_len[blockIdx.x] = 0u;      // the prefix over the blocks runs to the table's end
        return;
    }
    const uint32_t tid = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t u = 0; u < kRowsPer; ++u) {
        const uint64_t i = base + kRowsPer * tid + u;
        if (i >= n) continue;
        const uint32_t l = row_len(a, i, a.rec[i], a.flags[i]);
        a.len[i] = l;
        sum += l;
    }
    for (int d = 32; d; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    if ((tid & 63u) == 0u) ws[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0u) a.blk_len[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one row's bytes at p (global or LDS)
__device__ __forceinline__ void row_put(const RowArgs &a, char *p, uint64_t i, const gams_peak_t r) {
    const uint8_t f = a.flags[i];
    const RowCtg cg = a.ctgs[r.ctg];
    const uint32_t s = (uint32_t)cg.chr_start + r.window * a.step;
    const uint32_t lastw = (f & kRowMerged) ? a.tailwin[i] : r.window;
    const uint32_t e = (uint32_t)cg.chr_start + lastw * a.step + a.size - 1u;
    for (uint32_t q = 0; q < cg.name_len; ++q) *p++ = a.names[cg.name_off + q];
    if (f & kRowMerged) {
        *p++ = '(';
        *p++ = '+';
        *p++ = ')';
    }
    *p++ = ':';
    p += dec_digits(s);
    put_dec_back(p, s);
    if (e != s) {
        *p++ = '-';
        p += dec_digits(e);
        put_dec_back(p, e);
    }
    *p++ = '\t';
    const uint8_t *g = a.gctab + (size_t)r.gc_count * kGcStride;
    for (uint32_t q = 0; q < g[0]; ++q) *p++ = (char)g[1u + q];
    *p++ = '\t';
    if (r.signal < 0) *p++ = '-';
    *p++ = '1';
    *p++ = '\n';
}

// The rows of a block are one contiguous stretch of the text: they are put together in LDS (where a row's ~28
// single-byte stores cost nothing) and leave as 16-B stores on 16-B boundaries of the text, the ragged ends byte by
// byte.  A block whose text does not fit the stage (long names) writes its rows straight to the text.
__global__ __launch_bounds__(256) void rows_write_kernel(const RowArgs a) {
    __shared__ uint32_t scr[4];
    __shared__ __align__(16) char stage[kRowsStage + 16];
    const uint64_t n = rows_n(a);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.words[0] = a.n_rec[0];
        a.words[1] = a.blk_off[a.nb_cap];                      // all bytes (blocks past the last record count 0)
        a.words[2] = a.n_rec[0];
        a.words[3] = a.n_rec[1];
    }
    const uint64_t base = (uint64_t)blockIdx.x * kRowsBlock;
    if (base >= n) return;
    const uint32_t tid = threadIdx.x;
    uint32_t l[kRowsPer], mine = 0;
#pragma unroll
    for (uint32_t u = 0; u < kRowsPer; ++u) {
        const uint64_t i = base + kRowsPer * tid + u;
        l[u] = i < n ? a.len[i] : 0u;
        mine += l[u];
    }
    uint32_t tot;
    const uint32_t rel = block_excl_scan_256<uint32_t>(mine, scr, tot);
    const uint64_t blk0 = a.blk_off[blockIdx.x];               // the block's first byte in the text
    const uint32_t mis = (uint32_t)(blk0 & 15u);               // the stage starts at the same offset inside a 16-B unit
    const bool staged = tot <= kRowsStage && blk0 + tot <= a.text_cap;
    uint32_t at = rel;
#pragma unroll
    for (uint32_t u = 0; u < kRowsPer; ++u) {
        const uint64_t i = base + kRowsPer * tid + u;
        if (i >= n) break;
        const gams_peak_t r = a.rec[i];
        if (i == 0 || a.rec[i - 1].ctg != r.ctg) a.words[4u + r.ctg] = blk0 + at;   // the ctg's rows begin here (its first record is a head)
        if (l[u]) {
            if (staged)
                row_put(a, stage + mis + at, i, r);
            else if (blk0 + at + l[u] <= a.text_cap)
                row_put(a, a.text + blk0 + at, i, r);
        }
        at += l[u];
    }
    if (!staged) return;
    __syncthreads();
    // stage[mis, mis + tot) -> text[blk0, blk0 + tot): whole 16-B units in the middle
    const uint32_t head = min(tot, (16u - mis) & 15u);          // bytes in front of the first 16-B boundary
    char *const dst = a.text + blk0;
    if (tid < head) dst[tid] = stage[mis + tid];
    const uint32_t units = (tot - head) >> 4;
    const uint4 *const su = reinterpret_cast<const uint4 *>(stage + mis + head);   // 16-B aligned: mis + head is 0 mod 16
    uint4 *const du = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t q = tid; q < units; q += 256u) du[q] = su[q];
    const uint32_t done = head + (units << 4);
    if (tid < tot - done) dst[done + tid] = stage[mis + done + tid];
}

// ---- `wave --signal`: a row for EVERY window (wave.rs:158-168) ----------------------------------------------------
// "{chr}:{start}-{end}\t{gc_content}\t{signal}\n" from the dense rows of the pass (count, signal per window): a
// workgroup takes 256 consecutive windows of one ctg (SigTile), sig_len_kernel sums the row lengths per tile,
// wave_offsets_* turns them into the tiles' offsets in the text, sig_write_kernel builds a tile's text in LDS and stores
// it in 16-B units like rows_write_kernel.  120 Mb at step 10: 1.2e7 rows, 310 MB of text.
constexpr uint32_t kSigRows = 256;

struct SigTile {
    uint32_t ctg, w0;          // first window of the tile
    uint32_t n_win;            // windows of the ctg
    uint32_t pad;
    uint64_t win_base;         // the ctg's first row in the dense arrays
};

struct SigArgs {
    const SigTile *tiles;
    uint32_t n_tiles;
    const uint32_t *cnt;
    const int8_t *sig;
    const RowCtg *ctgs;
    const char *names;
    const uint8_t *gctab;                    // [size + 1][kGcStride]
    uint32_t size, step;
    uint32_t *blk_len;                       // per tile: bytes of its rows
    const unsigned long long *blk_off;       // per tile: where they begin
    char *text;
    uint64_t text_cap;
    unsigned long long *words;               // per ctg: where its rows begin (ctgs without a window keep ~0)
};

__device__ __forceinline__ uint32_t sig_row_len(const SigArgs &a, const RowCtg cg, uint32_t i, uint32_t k, int sg) {
    const uint32_t s = (uint32_t)cg.chr_start + i * a.step, e = s + a.size - 1u;
    return cg.name_len + 1u + dec_digits(s) + (e != s ? 1u + dec_digits(e) : 0u) + 1u + a.gctab[(size_t)k * kGcStride] + 1u +
           (sg < 0 ? 2u : 1u) + 1u;
}

__global__ __launch_bounds__(256) void sig_len_kernel(const SigArgs a) {
    __shared__ uint32_t ws[4];
    const SigTile t = a.tiles[blockIdx.x];
    const uint32_t tid = threadIdx.x, i = t.w0 + tid;
    uint32_t len = 0;
    if (i < t.n_win) len = sig_row_len(a, a.ctgs[t.ctg], i, a.cnt[t.win_base + i], a.sig[t.win_base + i]);
    for (int d = 32; d; d >>= 1) len += (uint32_t)__shfl_xor((int)len, d, 64);
    if ((tid & 63u) == 0u) ws[tid >> 6] = len;
    __syncthreads();
    if (tid == 0u) a.blk_len[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

__device__ __forceinline__ void sig_row_put(const SigArgs &a, char *p, const RowCtg cg, uint32_t i, uint32_t k, int sg) {
    const uint32_t s = (uint32_t)cg.chr_start + i * a.step, e = s + a.size - 1u;
    for (uint32_t q = 0; q < cg.name_len; ++q) *p++ = a.names[cg.name_off + q];
    *p++ = ':';
    p += dec_digits(s);
    put_dec_back(p, s);
    if (e != s) {                                   // IntSpan::runlist of a single position is that position
        *p++ = '-';
        p += dec_digits(e);
        put_dec_back(p, e);
    }
    *p++ = '\t';
    const uint8_t *g = a.gctab + (size_t)k * kGcStride;
    for (uint32_t q = 0; q < g[0]; ++q) *p++ = (char)g[1u + q];
    *p++ = '\t';
    if (sg < 0) *p++ = '-';
    *p++ = sg == 0 ? '0' : '1';
    *p++ = '\n';
}

__global__ __launch_bounds__(256) void sig_write_kernel(const SigArgs a) {
    __shared__ uint32_t scr[4];
    __shared__ __align__(16) char stage[kRowsStage + 16];
    const SigTile t = a.tiles[blockIdx.x];
    const uint32_t tid = threadIdx.x, i = t.w0 + tid;
    const RowCtg cg = a.ctgs[t.ctg];
    uint32_t k = 0, len = 0;
    int sg = 0;
    if (i < t.n_win) {
        k = a.cnt[t.win_base + i];
        sg = a.sig[t.win_base + i];
        len = sig_row_len(a, cg, i, k, sg);
    }
    uint32_t tot;
    const uint32_t rel = block_excl_scan_256<uint32_t>(len, scr, tot);
    const uint64_t blk0 = a.blk_off[blockIdx.x];
    if (t.w0 == 0u && tid == 0u) a.words[t.ctg] = blk0;
    const uint32_t mis = (uint32_t)(blk0 & 15u);
    const bool staged = tot <= kRowsStage && blk0 + tot <= a.text_cap;
    if (len) {
        if (staged)
            sig_row_put(a, stage + mis + rel, cg, i, k, sg);
        else if (blk0 + rel + len <= a.text_cap)
            sig_row_put(a, a.text + blk0 + rel, cg, i, k, sg);
    }
    if (!staged) return;
    __syncthreads();
    const uint32_t head = min(tot, (16u - mis) & 15u);
    char *const dst = a.text + blk0;
    if (tid < head) dst[tid] = stage[mis + tid];
    const uint32_t units = (tot - head) >> 4;
    const uint4 *const su = reinterpret_cast<const uint4 *>(stage + mis + head);
    uint4 *const du = reinterpret_cast<uint4 *>(dst + head);
    for (uint32_t q = tid; q < units; q += 256u) du[q] = su[q];
    const uint32_t done = head + (units << 4);
    if (tid < tot - done) dst[done + tid] = stage[mis + done + tid];
}

}  // namespace
