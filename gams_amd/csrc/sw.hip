// sw.hip -- windows around features: center_sw geometry + range GC + flank statistics.
//
// Replaces, per feature of one ctg (src/cmd_gams/sw.rs:141-184):
//   gams::center_sw         src/libs/window.rs:3-56
//   gams::cache_gc_content  src/libs/utils.rs:141-162   (round4 on return)
//   gams::center_resize     src/libs/window.rs:96-124
//   gams::cache_gc_stat     src/libs/utils.rs:189-213 -> gc_stat utils.rs:164-187
//
// GC of an arbitrary range comes from a prefix index over the whole seqset
// buffer, built once per seqset by two kernels:
//   gc_index_build  one workgroup per 64 KiB segment: 16-bit G/C mask per 16-B
//                   chunk + segment-local prefix (same layout as the wave tile)
//   gc_index_scan   exclusive scan of the segment totals (u64)
// P(y) = seg_base[y >> 16] + (PM[y >> 4] >> 16) + popcount(PM[y >> 4] & low(y & 15))
// and gc_count(range) = P(end+1) - P(start): 6 loads per range (4 B + 8 B each side).
// The parent is a single span, so IntSpan index/slice/at are plain arithmetic.
// Statistics are evaluated in the reference's f32 order (-ffp-contract=off).

#include "common.hpp"

#include <string>

#include <algorithm>

struct gams_gcindex {
    uint32_t *d_pm = nullptr;        // per 16-B chunk: local prefix << 16 | mask
    uint64_t *d_seg = nullptr;       // per 64 KiB segment: GC count before it
    uint64_t n_chunks = 0, n_segs = 0;
};

namespace {

__device__ __forceinline__ uint32_t gc_nibble(uint32_t x) {
    uint32_t y = (x & 0xDBDBDBDBu) ^ 0x43434343u;
    uint32_t t = (y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    uint32_t f = ~(t | y) & 0x80808080u;
    uint32_t g = f | (f >> 7);
    g |= g >> 14;
    return (g >> 7) & 0xFu;
}

// one workgroup (256 threads) per segment of 4096 chunks = 64 KiB
__global__ __launch_bounds__(256) void gc_index_build(const uint8_t *seq, uint64_t n_chunks, uint32_t *pm,
                                                       uint64_t *seg_tot) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t msk[4096 + 16];
    const uint32_t tid = threadIdx.x;
    const uint64_t c0 = (uint64_t)blockIdx.x * 4096u;
    const uint4 *src = reinterpret_cast<const uint4 *>(seq);
    // coalesced: in trip q the workgroup reads 256 consecutive chunks (4 KiB)
#pragma unroll 4
    for (uint32_t q = 0; q < 16; ++q) {
        const uint32_t lc = q * 256u + tid;
        const uint64_t c = c0 + lc;
        uint32_t m = 0;
        if (c < n_chunks) {
            const uint4 v = load_once16(src + c);   // the index is built in one pass over the sequence
            m = gc_nibble(v.x) | (gc_nibble(v.y) << 4) | (gc_nibble(v.z) << 8) | (gc_nibble(v.w) << 12);
        }
        msk[lc + (lc >> 8)] = m;  // +1 word per 256: thread t's 16-word run starts on bank 17t
    }
    __syncthreads();
    // thread t owns chunks [16t, 16t+16) of the segment
    uint32_t s = 0;
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        const uint32_t lc = tid * 16u + q;
        s += __popc(msk[lc + (lc >> 8)]);
    }
    uint32_t tot;
    uint32_t run = block_excl_scan_256<uint32_t>(s, ws, tot);
#pragma unroll
    for (uint32_t q = 0; q < 16; ++q) {
        const uint32_t lc = tid * 16u + q;
        const uint32_t m = msk[lc + (lc >> 8)];
        if (c0 + lc < n_chunks) pm[c0 + lc] = (run << 16) | m;
        run += __popc(m);
    }
    if (tid == 0) seg_tot[blockIdx.x] = tot;  // <= 65536; local prefixes (exclusive) stay below 2^16
}

// single workgroup: in-place exclusive scan of the segment totals
__global__ __launch_bounds__(256) void gc_index_scan(uint64_t *seg, uint64_t n_segs) {
    __shared__ uint64_t ws[4];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (n_segs + 255u) / 256u;
    const uint64_t b = min((uint64_t)tid * per, n_segs), e = min(b + per, n_segs);
    uint64_t s = 0;
    for (uint64_t i = b; i < e; ++i) s += seg[i];
    uint64_t tot;
    uint64_t run = block_excl_scan_256<uint64_t>(s, ws, tot);
    for (uint64_t i = b; i < e; ++i) {
        const uint64_t v = seg[i];
        seg[i] = run;
        run += v;
    }
}

__device__ __forceinline__ uint64_t gc_before(const uint32_t *pm, const uint64_t *seg, uint64_t y) {
    const uint32_t e = pm[y >> 4];
    return seg[y >> 16] + (e >> 16) + __popc(e & ((1u << (y & 15u)) - 1u));
}

// ---- geometry shared by host (row offsets) and device ------------------------
struct SwGeom {
    int32_t m_s, m_e;  // the M window, chromosome coordinates
    int32_t n_l, n_r;  // number of L and R windows
};

// window.rs:96-124 for a single-span parent [ps,pe] and span [is,ie]
__host__ __device__ inline void center_resize_1(int32_t ps, int32_t pe, int32_t is, int32_t ie, int32_t resize,
                                                int32_t &os, int32_t &oe) {
    const int32_t psize = pe - ps + 1;
    const int32_t half_size = (ie - is + 1) / 2;
    const int32_t mid_left = half_size == 0 ? is : is + half_size - 1;
    const int32_t mid_right = half_size == 0 ? is : is + half_size;
    const int32_t half_resize = resize / 2;
    int32_t left_idx = (mid_left - ps + 1) - half_resize + 1;
    if (left_idx < 1) left_idx = 1;
    int32_t right_idx = (mid_right - ps + 1) + half_resize - 1;
    if (right_idx > psize) right_idx = psize;
    os = ps + left_idx - 1;
    oe = ps + right_idx - 1;
}

// window.rs:3-56: M, then L windows while sw_start >= 1, then R windows while
// sw_end <= parent.size(), at most `max` each
__host__ __device__ inline SwGeom sw_geometry(int32_t ps, int32_t pe, int32_t fs, int32_t fe, int32_t size,
                                              int32_t max) {
    SwGeom g;
    center_resize_1(ps, pe, fs, fe, size, g.m_s, g.m_e);
    const int64_t psize = (int64_t)pe - ps + 1;
    const int64_t m_min_idx = (int64_t)g.m_s - ps + 1, m_max_idx = (int64_t)g.m_e - ps + 1;
    // L: sw_end = m_min_idx - 1 - (d-1)*size, sw_start = sw_end - size + 1 >= 1, sw_end <= psize
    int64_t nl = 0, nr = 0;
    if (max > 0 && size > 0) {
        const int64_t room_l = m_min_idx - 1;  // indices strictly left of M
        nl = room_l >= 0 ? room_l / size : 0;
        if (m_min_idx - 1 > psize) nl = 0;     // first L window already past the end (:33)
        const int64_t room_r = psize - m_max_idx;
        nr = room_r >= 0 ? room_r / size : 0;
        if (m_max_idx + 1 < 1) nr = 0;         // first R window before the start (:30)
        if (nl > max) nl = max;
        if (nr > max) nr = max;
    }
    g.n_l = (int32_t)nl;
    g.n_r = (int32_t)nr;
    return g;
}

__device__ __forceinline__ float round4(float x) {  // utils.rs:135-138 with decimals = 4
    const float y = 10000.0f;
    return roundf(x * y) / y;
}

// one selected ctg of a batched sw call
struct SwCtg {
    uint64_t seq_off;     // buffer offset of ctg base 0
    uint32_t len;
    int32_t chr_start, chr_end;
    uint32_t feat_first;  // index of the ctg's first feature in the call's feature arrays
    uint32_t pad[2];
};

struct SwArgs {
    const uint32_t *pm;
    const uint64_t *seg;
    uint64_t seq_off;   // buffer offset of ctg base 0   (sw_kernel: filled per thread from ctgs[])
    uint32_t len;
    int32_t chr_start, chr_end;
    const SwCtg *ctgs;        // sw_kernel: the selected ctgs
    const uint32_t *fctg;     // sw_kernel: selected-ctg number of every feature
    const int32_t *fs, *fe;
    const uint64_t *row_off;  // exclusive prefix of rows per feature
    uint32_t nf;
    int32_t size, max, resize;
    gams_sw_row_t *rows;
    uint64_t cap;
};

// gc_content of chromosome range [s,e] (inclusive) inside the ctg: count / len, f32
__device__ __forceinline__ float range_gc(const SwArgs &a, int32_t s, int32_t e) {
    // parent.index(): from = s - chr_start + 1 (1-based), bytes [from-1, to)
    int64_t b0 = (int64_t)s - a.chr_start, b1 = (int64_t)e - a.chr_start + 1;
    const float flen = (float)(int32_t)(b1 - b0);
    b0 = b0 < 0 ? 0 : (b0 > a.len ? a.len : b0);  // never read outside the ctg
    b1 = b1 < 0 ? 0 : (b1 > a.len ? a.len : b1);
    const uint64_t c = gc_before(a.pm, a.seg, a.seq_off + (uint64_t)b1) -
                       gc_before(a.pm, a.seg, a.seq_off + (uint64_t)b0);
    return (float)(uint32_t)c / flen;
}

// G/C bases in front of ctg byte offset b (clamped into the ctg like range_gc)
__device__ __forceinline__ uint64_t gc_prefix_at(const SwArgs &a, int64_t b) {
    b = b < 0 ? 0 : (b > a.len ? a.len : b);
    return gc_before(a.pm, a.seg, a.seq_off + (uint64_t)b);
}

// one thread per (feature, slot); slot 0 = M, 1..max = L, max+1..2max = R
// All selected ctgs of a call in ONE launch (a ctg's ~400 features are a handful of workgroups: launched
// per ctg the kernel is all launch latency).
__global__ __launch_bounds__(256) void sw_kernel(const SwArgs a0) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slots = 1u + 2u * (uint32_t)a0.max;
    const uint64_t f = gid / slots;
    const uint32_t slot = (uint32_t)(gid % slots);
    if (f >= a0.nf) return;
    const SwCtg cg = a0.ctgs[a0.fctg[f]];
    SwArgs a = a0;
    a.seq_off = cg.seq_off;
    a.len = cg.len;
    a.chr_start = cg.chr_start;
    a.chr_end = cg.chr_end;
    const SwGeom g = sw_geometry(a.chr_start, a.chr_end, a.fs[f], a.fe[f], a.size, a.max);
    int32_t type, dist, ws, we;
    uint64_t row = a.row_off[f];
    if (slot == 0) {
        type = 0;
        dist = 0;
        ws = g.m_s;
        we = g.m_e;
    } else if (slot <= (uint32_t)a.max) {
        dist = (int32_t)slot;
        if (dist > g.n_l) return;
        type = 1;
        we = g.m_s - 1 - (dist - 1) * a.size;   // window.rs:24,48-49
        ws = we - a.size + 1;
        row += (uint64_t)dist;
    } else {
        dist = (int32_t)slot - a.max;
        if (dist > g.n_r) return;
        type = 2;
        ws = g.m_e + 1 + (dist - 1) * a.size;   // window.rs:21,45-46
        we = ws + a.size - 1;
        row += (uint64_t)g.n_l + (uint64_t)dist;
    }
    if (row >= a.cap) return;
    gams_sw_row_t r;
    r.feature = (uint32_t)f - cg.feat_first;
    r.type = type;
    r.distance = dist;
    r.start = ws;
    r.end = we;
    r.gc_content = round4(range_gc(a, ws, we));                         // utils.rs:161
    // flank: center_resize(parent, window, resize) cut into size-bp tiles (sw.rs:175-178)
    int32_t rs, re;
    center_resize_1(a.chr_start, a.chr_end, ws, we, a.resize, rs, re);
    const int32_t flen = re - rs + 1;
    const int32_t nt = flen >= a.size ? (flen - a.size) / a.size + 1 : 0;  // sliding(range,size,size)
    const float len = (float)nt;
    // Consecutive tiles share a boundary: one prefix lookup per boundary (nt + 1) instead of two
    // per tile and pass; the tile values are kept for the second pass when they fit.
    constexpr int kKeep = 8;
    float xs[kKeep];
    float sum = 0.0f;
    {
        uint64_t before = gc_prefix_at(a, (int64_t)rs - a.chr_start);
#pragma unroll   // constant indices keep xs[] in registers
        for (int32_t t = 0; t < kKeep; ++t) {
            if (t < nt) {
                const uint64_t upto = gc_prefix_at(a, (int64_t)rs - a.chr_start + (int64_t)(t + 1) * a.size);
                xs[t] = round4((float)(uint32_t)(upto - before) / (float)a.size);
                before = upto;
                sum = sum + xs[t];                                          // stat.rs:3
            }
        }
        for (int32_t t = kKeep; t < nt; ++t) {
            const uint64_t upto = gc_prefix_at(a, (int64_t)rs - a.chr_start + (int64_t)(t + 1) * a.size);
            sum = sum + round4((float)(uint32_t)(upto - before) / (float)a.size);
            before = upto;
        }
    }
    const float mean = sum / len;                                       // stat.rs:5
    float sq = 0.0f;
#pragma unroll
    for (int32_t t = 0; t < kKeep; ++t) {
        if (t < nt) {
            const float d = xs[t] - mean;
            sq = sq + d * d;                                                // stat.rs:12
        }
    }
    for (int32_t t = kKeep; t < nt; ++t) {
        const int32_t ts = rs + t * a.size;
        const float x = round4(range_gc(a, ts, ts + a.size - 1));
        const float d = x - mean;
        sq = sq + d * d;
    }
    const float sd = sqrtf(sq / (len - 1.0f));                          // stat.rs:13
    float cv;                                                           // utils.rs:169-175
    if (mean == 0.0f || mean == 1.0f)
        cv = 0.0f;
    else if (mean <= 0.5f)
        cv = sd / mean;
    else
        cv = sd / (1.0f - mean);
    r.gc_mean = round4(mean);
    r.gc_stddev = round4(sd);
    r.gc_cv = round4(cv);
    a.rows[row] = r;
}

// peak (src/cmd_gams/peak.rs:79) / any caller of cache_gc_content: one lane per range, the ranges of every
// selected ctg in one launch (a.ctgs / a.fctg as in sw_kernel)
__global__ __launch_bounds__(256) void range_gc_kernel(const SwArgs a0, const int32_t *rs, const int32_t *re,
                                                       uint32_t n, float *gc) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const SwCtg cg = a0.ctgs[a0.fctg[q]];
    SwArgs a = a0;
    a.seq_off = cg.seq_off;
    a.len = cg.len;
    a.chr_start = cg.chr_start;
    a.chr_end = cg.chr_end;
    gc[q] = round4(range_gc(a, rs[q], re[q]));                          // utils.rs:157-161
}

}  // namespace

// lazily built per seqset; owned by the seqset (freed in gams_seqset_destroy)
int gams_seqset_gcindex(gams_gpu_t *h, gams_seqset_t *s) {
    if (s->gcindex) return GAMS_OK;
    h->reader_epoch.fetch_add(1, std::memory_order_relaxed);   // the build reads the sequence bytes
    int wrc = gams_seqset_wait_uploads(h, s);
    if (wrc != GAMS_OK) return wrc;
    gams_gcindex *ix = new gams_gcindex();
    ix->n_chunks = s->bytes / 16;
    ix->n_segs = (ix->n_chunks + 4095) / 4096;
    hipError_t e = hipMalloc(&ix->d_pm, ix->n_chunks * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&ix->d_seg, (ix->n_segs + 1) * sizeof(uint64_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();   // reported below, not left sticky
        (void)hipFree(ix->d_pm);
        (void)hipFree(ix->d_seg);
        delete ix;
        return gams_fail(h, GAMS_ENOMEM, std::string("gcindex: hipMalloc: ") + hipGetErrorString(e));
    }
    hipLaunchKernelGGL(gc_index_build, dim3((unsigned)ix->n_segs), dim3(256), 0, h->compute, s->d_seq,
                       ix->n_chunks, ix->d_pm, ix->d_seg);
    hipLaunchKernelGGL(gc_index_scan, dim3(1), dim3(256), 0, h->compute, ix->d_seg, ix->n_segs);
    e = hipGetLastError();
    if (e != hipSuccess) {
        (void)hipGetLastError();   // reported below, not left sticky
        (void)hipFree(ix->d_pm);
        (void)hipFree(ix->d_seg);
        delete ix;
        return gams_fail(h, GAMS_EHIP, std::string("gcindex: launch: ") + hipGetErrorString(e));
    }
    s->gcindex = ix;
    return GAMS_OK;
}

void gams_seqset_gcindex_free(gams_seqset_t *s) {
    if (!s->gcindex) return;
    (void)hipFree(s->gcindex->d_pm);
    (void)hipFree(s->gcindex->d_seg);
    delete s->gcindex;
    s->gcindex = nullptr;
}

// ---- the rows as TSV text (sw.rs:152-190, the Display of Sw: data.rs:58-83) -------------------------------------
// "sw:{feature id}:{serial}\t{chr}:{start}-{end}\t{M|L|R}\t{distance}\t{gc_content}\t{gc_mean}\t{gc_stddev}\t{gc_cv}\t\n"
// (the last field, rg_count, is empty here as in `gams sw` without --rg).  The four floats are round(x, 4) values
// (utils.rs:135-138, :161, :186): the f32 nearest to m / 10^4 for an integer m, and Rust's `{}` prints the shortest
// digits that round-trip -- for m below 10^7 (values below 1000: gc values are below 1, cv a few units) that is m / 10^4
// with its trailing zeros dropped, since no shorter decimal lies within half an ulp of it.  Anything else (a value of
// 1000 or more, a negative coordinate) raises a flag and the host formats the batch as before.
namespace {
constexpr uint32_t kSwTextBlock = 512;     // rows per workgroup of the text kernels (256 threads x 2)
constexpr uint32_t kSwTextStage = 49152;   // bytes of a block's text staged in LDS

struct SwTextArgs {
    const gams_sw_row_t *rows;
    uint64_t n_rows;
    const SwCtg *ctgs;
    const uint64_t *ctg_row_off;       // [n_sel + 1]: first row of every selected ctg
    uint32_t n_sel;
    const uint64_t *feat_row_off;      // [nf + 1]: first row of every feature
    const uint32_t *name_off;          // [n_sel + 1] into names
    const char *names;
    const uint32_t *id_off;            // [nf + 1] into ids
    const char *ids;
    uint32_t *len;                     // per row
    uint32_t *blk_len;                 // per block
    const unsigned long long *blk_off; // exclusive prefix, [nb] = all bytes
    uint32_t nb;
    char *text;
    uint64_t text_cap;
    unsigned long long *words;         // [0] = all bytes, [1] = flag (a value this formatter does not cover), [2 + k] = first byte of ctg k
};

__device__ __forceinline__ uint32_t sw_digits(uint32_t v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u
         : v < 10000000u ? 7u : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ char *sw_put_dec(char *p, uint32_t v) {
    const uint32_t n = sw_digits(v);
    char *e = p + n;
    do {
        *--e = (char)('0' + v % 10u);
        v /= 10u;
    } while (v);
    return p + n;
}
// a round4 value as Rust prints it; returns the length (p == nullptr: length only); *bad set for values not covered
__device__ __forceinline__ uint32_t sw_put_f4(char *p, float v, bool *bad) {
    if (v != v) {
        if (p) { p[0] = 'N'; p[1] = 'a'; p[2] = 'N'; }
        return 3u;
    }
    if (!(v >= 0.0f) || !(v < 1000.0f)) {
        *bad = true;
        return 1u;
    }
    if (v == 0.0f && (__float_as_uint(v) >> 31)) {        // round(x, 4) of a tiny negative: Rust prints "-0"
        if (p) { p[0] = '-'; p[1] = '0'; }
        return 2u;
    }
    const uint32_t m = (uint32_t)((double)v * 10000.0 + 0.5);     // v is the f32 nearest to m / 10^4: exact in double
    const uint32_t ip = m / 10000u;
    uint32_t fr = m % 10000u, nd = 4u;
    while (nd && fr % 10u == 0u) {
        fr /= 10u;
        --nd;
    }
    const uint32_t n = sw_digits(ip) + (nd ? 1u + nd : 0u);
    if (p) {
        p = sw_put_dec(p, ip);
        if (nd) {
            *p++ = '.';
            char *e = p + nd;
            for (uint32_t q = 0; q < nd; ++q) {
                *--e = (char)('0' + fr % 10u);
                fr /= 10u;
            }
        }
    }
    return n;
}

struct SwRowCtx {
    uint32_t k, f, serial;    // selected ctg, feature (index into the call's arrays), 1-based row of the feature
};
__device__ __forceinline__ SwRowCtx sw_row_ctx(const SwTextArgs &a, uint64_t r, const gams_sw_row_t &w) {
    uint32_t lo = 0, hi = a.n_sel;                       // last ctg whose first row <= r
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.ctg_row_off[mid] <= r)
            lo = mid;
        else
            hi = mid;
    }
    SwRowCtx c;
    c.k = lo;
    c.f = a.ctgs[lo].feat_first + w.feature;
    c.serial = (uint32_t)(r - a.feat_row_off[c.f]) + 1u;    // sw.rs:148: the serial restarts with every feature
    return c;
}
// the row's text at p (nullptr: its length only)
__device__ __forceinline__ uint32_t sw_row_text(const SwTextArgs &a, uint64_t r, char *p, bool *bad) {
    const gams_sw_row_t w = a.rows[r];
    const SwRowCtx c = sw_row_ctx(a, r, w);
    const uint32_t id0 = a.id_off[c.f], idn = a.id_off[c.f + 1u] - id0;
    const uint32_t nm0 = a.name_off[c.k], nmn = a.name_off[c.k + 1u] - nm0;
    if (w.start < 0 || w.end < 0 || w.distance < 0 || (uint32_t)w.type > 2u) *bad = true;
    const uint32_t st = (uint32_t)w.start, en = (uint32_t)w.end, di = (uint32_t)w.distance;
    if (!p) {
        uint32_t n = 3u + idn + 1u + sw_digits(c.serial) + 1u + nmn + 1u + sw_digits(st);
        if (en != st) n += 1u + sw_digits(en);
        n += 1u + 1u + 1u + sw_digits(di) + 1u;
        n += sw_put_f4(nullptr, w.gc_content, bad) + 1u + sw_put_f4(nullptr, w.gc_mean, bad) + 1u +
             sw_put_f4(nullptr, w.gc_stddev, bad) + 1u + sw_put_f4(nullptr, w.gc_cv, bad) + 2u;
        return n;
    }
    char *q = p;
    *q++ = 's';
    *q++ = 'w';
    *q++ = ':';
    for (uint32_t i = 0; i < idn; ++i) *q++ = a.ids[id0 + i];
    *q++ = ':';
    q = sw_put_dec(q, c.serial);
    *q++ = '\t';
    for (uint32_t i = 0; i < nmn; ++i) *q++ = a.names[nm0 + i];
    *q++ = ':';
    q = sw_put_dec(q, st);
    if (en != st) {
        *q++ = '-';
        q = sw_put_dec(q, en);
    }
    *q++ = '\t';
    *q++ = "MLR"[(uint32_t)w.type > 2u ? 0u : (uint32_t)w.type];
    *q++ = '\t';
    q = sw_put_dec(q, di);
    *q++ = '\t';
    q += sw_put_f4(q, w.gc_content, bad);
    *q++ = '\t';
    q += sw_put_f4(q, w.gc_mean, bad);
    *q++ = '\t';
    q += sw_put_f4(q, w.gc_stddev, bad);
    *q++ = '\t';
    q += sw_put_f4(q, w.gc_cv, bad);
    *q++ = '\t';
    *q++ = '\n';
    return (uint32_t)(q - p);
}

__global__ __launch_bounds__(256) void sw_text_len_kernel(const SwTextArgs a) {
    __shared__ uint32_t ws[4];
    const uint64_t base = (uint64_t)blockIdx.x * kSwTextBlock;
    const uint32_t tid = threadIdx.x;
    uint32_t sum = 0;
    bool bad = false;
#pragma unroll
    for (uint32_t u = 0; u < 2u; ++u) {
        const uint64_t r = base + 2u * tid + u;
        if (r >= a.n_rows) continue;
        const uint32_t l = sw_row_text(a, r, nullptr, &bad);
        a.len[r] = l;
        sum += l;
    }
    if (bad) a.words[1] = 1ull;
    for (int d = 32; d; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    if ((tid & 63u) == 0u) ws[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0u) a.blk_len[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive prefix of the blocks' byte counts (one workgroup), the total behind them
__global__ __launch_bounds__(1024) void sw_text_scan_kernel(const uint32_t *blk_len, uint32_t nb, unsigned long long *blk_off,
                                                            unsigned long long *words) {
    __shared__ unsigned long long wsum[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t per = (nb + 1023u) / 1024u;
    const uint32_t b0 = min(nb, tid * per), b1 = min(nb, b0 + per);
    unsigned long long mine = 0;
    for (uint32_t b = b0; b < b1; ++b) mine += blk_len[b];
    const unsigned long long inc = wave_incl_scan_u64(mine);
    if (lane == 63u) wsum[wv] = inc;
    __syncthreads();
    unsigned long long base = 0, all = 0;
    for (uint32_t w = 0; w < 16u; ++w) {
        if (w < wv) base += wsum[w];
        all += wsum[w];
    }
    unsigned long long off = base + inc - mine;
    for (uint32_t b = b0; b < b1; ++b) {
        blk_off[b] = off;
        off += blk_len[b];
    }
    if (tid == 0u) {
        blk_off[nb] = all;
        words[0] = all;
    }
}

__global__ __launch_bounds__(256) void sw_text_write_kernel(const SwTextArgs a) {
    __shared__ uint32_t scr[4];
    __shared__ __align__(16) char stage[kSwTextStage + 16];
    const uint64_t base = (uint64_t)blockIdx.x * kSwTextBlock;
    const uint32_t tid = threadIdx.x;
    uint32_t l[2], mine = 0;
#pragma unroll
    for (uint32_t u = 0; u < 2u; ++u) {
        const uint64_t r = base + 2u * tid + u;
        l[u] = r < a.n_rows ? a.len[r] : 0u;
        mine += l[u];
    }
    uint32_t tot;
    const uint32_t rel = block_excl_scan_256<uint32_t>(mine, scr, tot);
    const uint64_t blk0 = a.blk_off[blockIdx.x];
    const uint32_t mis = (uint32_t)(blk0 & 15u);
    const bool fits = blk0 + tot <= a.text_cap;
    const bool staged = tot <= kSwTextStage && fits;
    uint32_t at = rel;
    bool bad = false;
#pragma unroll
    for (uint32_t u = 0; u < 2u; ++u) {
        const uint64_t r = base + 2u * tid + u;
        if (r >= a.n_rows) break;
        if (staged)
            (void)sw_row_text(a, r, stage + mis + at, &bad);
        else if (fits)
            (void)sw_row_text(a, r, a.text + blk0 + at, &bad);
        at += l[u];
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

// where every selected ctg's text begins: the bytes of the rows in front of its first row
__global__ __launch_bounds__(64) void sw_text_ctg_kernel(const SwTextArgs a) {
    const uint32_t k = blockIdx.x, lane = threadIdx.x;
    if (k > a.n_sel) return;
    const uint64_t r = a.ctg_row_off[k];                 // (entry n_sel = all rows)
    const uint64_t b = r / kSwTextBlock;
    unsigned long long off = 0;
    for (uint64_t q = b * kSwTextBlock + lane; q < r; q += 64u) off += a.len[q];
    for (int d = 32; d; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)off, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(off >> 32), d, 64);
        off += ((unsigned long long)hi << 32) | lo;
    }
    if (lane == 0) a.words[2u + k] = (b < a.nb ? a.blk_off[b] : a.blk_off[a.nb]) + off;
}

struct SwTextReq {                       // what gams_gpu_sw_text adds to a batch call
    const char *const *chr;              // per selected ctg
    const char *const *feat_id;          // per feature
    const char **text;
    uint64_t *text_bytes;
    const uint64_t **ctg_off;
};
}  // namespace

static int sw_batch_impl(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                         const int32_t *chr_start, const uint64_t *feat_off, const int32_t *feat_start,
                         const int32_t *feat_end, int32_t size, int32_t max, int32_t resize, gams_sw_row_t *rows,
                         uint64_t cap, uint64_t *row_off, uint64_t *n_rows, const SwTextReq *tx);

extern "C" int gams_gpu_sw_batch(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                                 const int32_t *chr_start, const uint64_t *feat_off, const int32_t *feat_start,
                                 const int32_t *feat_end, int32_t size, int32_t max, int32_t resize,
                                 gams_sw_row_t *rows, uint64_t cap, uint64_t *row_off, uint64_t *n_rows) {
    return sw_batch_impl(h, s, n_sel, ctg_index, chr_start, feat_off, feat_start, feat_end, size, max, resize, rows, cap,
                         row_off, n_rows, nullptr);
}

extern "C" int gams_gpu_sw_text(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                                const char *const *chr, const int32_t *chr_start, const uint64_t *feat_off,
                                const int32_t *feat_start, const int32_t *feat_end, const char *const *feat_id,
                                int32_t size, int32_t max, int32_t resize, const char **text, uint64_t *text_bytes,
                                const uint64_t **ctg_off, uint64_t *n_rows) {
    if (!h || !text || !text_bytes || !n_rows || (n_sel && (!chr || !feat_off)))
        return gams_fail(h, GAMS_EINVAL, "gpu_sw_text: null argument");
    if (n_sel && feat_off[n_sel] && !feat_id) return gams_fail(h, GAMS_EINVAL, "gpu_sw_text: null feature ids");
    const SwTextReq tx{chr, feat_id, text, text_bytes, ctg_off};
    *text = nullptr;
    *text_bytes = 0;
    return sw_batch_impl(h, s, n_sel, ctg_index, chr_start, feat_off, feat_start, feat_end, size, max, resize, nullptr, 0,
                         nullptr, n_rows, &tx);
}

static int sw_batch_impl(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                         const int32_t *chr_start, const uint64_t *feat_off, const int32_t *feat_start,
                         const int32_t *feat_end, int32_t size, int32_t max, int32_t resize, gams_sw_row_t *rows,
                         uint64_t cap, uint64_t *row_off, uint64_t *n_rows, const SwTextReq *tx) {
    if (!h || !s || !n_rows || (n_sel && (!ctg_index || !chr_start || !feat_off)))
        return gams_fail(h, GAMS_EINVAL, "gpu_sw: null argument");
    // size or resize 1: half_resize = 0 makes center_resize slice [mid+1, mid-1] (window.rs:113-123),
    // an empty span whose min()/max() the reference then asks for -- no defined answer to mirror
    if (size < 2 || max < 0 || resize < 2)
        return gams_fail(h, GAMS_EINVAL, "gpu_sw: size >= 2, max >= 0, resize >= 2 (center_resize of 1 bp is an empty span)");
    *n_rows = 0;
    if (row_off)
        for (uint32_t k = 0; k <= n_sel; ++k) row_off[k] = 0;
    if (n_sel == 0) return GAMS_OK;
    if (feat_off[0] != 0) return gams_fail(h, GAMS_EINVAL, "gpu_sw: feat_off[0] must be 0");
    for (uint32_t k = 0; k < n_sel; ++k) {
        if (ctg_index[k] >= s->n_ctg) return gams_fail(h, GAMS_EINVAL, "gpu_sw: ctg index out of range");
        if (feat_off[k + 1] < feat_off[k]) return gams_fail(h, GAMS_EINVAL, "gpu_sw: feat_off must not decrease");
        const uint32_t len = s->len[ctg_index[k]];
        if (len == 0 || len > 0x7fffffffu) return gams_fail(h, GAMS_EINVAL, "gpu_sw: ctg length out of range");
    }
    const uint64_t nf64 = feat_off[n_sel];
    if (nf64 == 0) return GAMS_OK;
    if (!feat_start || !feat_end) return gams_fail(h, GAMS_EINVAL, "gpu_sw: null argument");
    // one thread per (feature, slot); max beyond 2^24 windows a side cannot exist in a ctg of < 2^31 bases
    // and would overflow the product
    if (max > (1 << 24)) return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_sw: max beyond 2^24 windows a side");
    const uint64_t threads = nf64 * (1u + 2u * (uint64_t)max);
    if (nf64 > 0xffffffffull || (threads + 255) / 256 > 0x7fffffffull)
        return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_sw: too many feature slots for one launch");
    const uint32_t nf = (uint32_t)nf64;
    GAMS_HIP(h, hipSetDevice(h->device));

    // Inputs are assembled in one page-locked block (one DMA): ctgs | fs | fe | fctg | off; the rows per
    // feature come from the closed form (window.rs:29-41) -> exclusive offsets.
    const size_t b_ctg = ((size_t)n_sel * sizeof(SwCtg) + 255) & ~(size_t)255;
    const size_t b_i32 = ((size_t)nf * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t b_off = (((size_t)nf + 1) * sizeof(uint64_t) + 255) & ~(size_t)255;
    const size_t in_bytes = b_ctg + 3 * b_i32 + b_off;
    const bool size_query = !tx && (!rows || cap == 0);
    std::vector<uint64_t> crow((size_t)n_sel + 1, 0);    // first row of every selected ctg (text mode)
    uint8_t *pin = nullptr, *dev = nullptr;
    size_t pin_cap = 0, dev_cap = 0;
    std::vector<uint64_t> off_host;                 // size query: no device, no pinned memory
    uint64_t *off = nullptr;
    SwCtg *cg = nullptr;
    int32_t *fs = nullptr, *fe = nullptr;
    uint32_t *fctg = nullptr;
    if (size_query) {
        off_host.resize((size_t)nf + 1);
        off = off_host.data();
    } else {
        const hipError_t e = gams_pool_alloc(h, true, in_bytes, reinterpret_cast<void **>(&pin), &pin_cap);
        if (e != hipSuccess) return gams_fail(h, GAMS_ENOMEM, std::string("gpu_sw: pinned staging: ") + hipGetErrorString(e));
        cg = reinterpret_cast<SwCtg *>(pin);
        fs = reinterpret_cast<int32_t *>(pin + b_ctg);
        fe = reinterpret_cast<int32_t *>(pin + b_ctg + b_i32);
        fctg = reinterpret_cast<uint32_t *>(pin + b_ctg + 2 * b_i32);
        off = reinterpret_cast<uint64_t *>(pin + b_ctg + 3 * b_i32);
    }
    auto release = [&]() {
        if (pin) gams_pool_free(h, true, pin, pin_cap);
        if (dev) gams_pool_free(h, false, dev, dev_cap);
    };
    uint64_t tot = 0;
    for (uint32_t k = 0; k < n_sel; ++k) {
        const uint32_t i = ctg_index[k];
        const int32_t cs = chr_start[k], ce = cs + (int32_t)s->len[i] - 1;
        if (row_off) row_off[k] = tot;
        crow[k] = tot;
        if (cg) cg[k] = SwCtg{s->off[i], s->len[i], cs, ce, (uint32_t)feat_off[k], {0u, 0u}};
        for (uint64_t f = feat_off[k]; f < feat_off[k + 1]; ++f) {
            // window.rs:98-110: the middle pair of the feature must be members of the ctg span --
            // IntSpan::index of a non-member has no defined answer in the reference to mirror
            const int64_t flen = (int64_t)feat_end[f] - feat_start[f] + 1, half = flen / 2;
            const int64_t mid_l = half == 0 ? feat_start[f] : (int64_t)feat_start[f] + half - 1;
            const int64_t mid_r = half == 0 ? feat_start[f] : (int64_t)feat_start[f] + half;
            if (flen < 1 || mid_l < cs || mid_r > ce) {
                release();
                return gams_fail(h, GAMS_EINVAL, "gpu_sw: feature " + std::to_string(f - feat_off[k]) +
                                                     (n_sel > 1 ? " of selected ctg " + std::to_string(k) : std::string()) +
                                                     " is empty or has its middle outside the ctg");
            }
            off[f] = tot;
            const SwGeom g = sw_geometry(cs, ce, feat_start[f], feat_end[f], size, max);
            tot += 1u + (uint64_t)g.n_l + (uint64_t)g.n_r;
            if (fs) {
                fs[f] = feat_start[f];
                fe[f] = feat_end[f];
                fctg[f] = k;
            }
        }
    }
    off[nf] = tot;
    if (row_off) row_off[n_sel] = tot;
    crow[n_sel] = tot;
    *n_rows = tot;
    if (size_query) return GAMS_OK;
    if (tx) cap = tot;                                   // text mode: every row, kept on the device

    int rc = gams_seqset_gcindex(h, s);
    if (rc != GAMS_OK) {
        release();
        return rc;
    }
    const uint64_t n_out = std::min<uint64_t>(tot, cap);
    const size_t b_rows = (size_t)std::max<uint64_t>(n_out, 1) * sizeof(gams_sw_row_t);
    hipError_t e = gams_pool_alloc(h, false, in_bytes + b_rows, reinterpret_cast<void **>(&dev), &dev_cap);
    if (e != hipSuccess) {
        release();
        return gams_fail(h, GAMS_ENOMEM, std::string("gpu_sw: device buffers: ") + hipGetErrorString(e));
    }
#define SW_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (void)hipStreamSynchronize(h->compute);                                    \
            release();                                                                 \
            return gams_fail(h, GAMS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                              \
    } while (0)
    SW_HIP(hipMemcpyAsync(dev, pin, in_bytes, hipMemcpyHostToDevice, h->compute));
    gams_sw_row_t *d_rows = reinterpret_cast<gams_sw_row_t *>(dev + in_bytes);
    SwArgs a{};
    a.pm = s->gcindex->d_pm;
    a.seg = s->gcindex->d_seg;
    a.ctgs = reinterpret_cast<const SwCtg *>(dev);
    a.fs = reinterpret_cast<const int32_t *>(dev + b_ctg);
    a.fe = reinterpret_cast<const int32_t *>(dev + b_ctg + b_i32);
    a.fctg = reinterpret_cast<const uint32_t *>(dev + b_ctg + 2 * b_i32);
    a.row_off = reinterpret_cast<const uint64_t *>(dev + b_ctg + 3 * b_i32);
    a.nf = nf;
    a.size = size;
    a.max = max;
    a.resize = resize;
    a.rows = d_rows;
    a.cap = n_out;
    SW_HIP(hipEventRecord(h->k0, h->compute));
    hipLaunchKernelGGL(sw_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->compute, a);
    SW_HIP(hipGetLastError());
    SW_HIP(hipEventRecord(h->k1, h->compute));
    h->k_valid = true;
    h->kq_used = 0;
    if (tx) {
        // ---- rows -> TSV text on the device ----------------------------------------------------------------------
        std::vector<uint32_t> name_off((size_t)n_sel + 1), id_off((size_t)nf + 1);
        std::string names, ids;
        size_t max_name = 0, max_id = 0;
        for (uint32_t k = 0; k < n_sel; ++k) {
            name_off[k] = (uint32_t)names.size();
            if (!tx->chr[k]) {
                (void)hipStreamSynchronize(h->compute);
                release();
                return gams_fail(h, GAMS_EINVAL, "gpu_sw_text: null chromosome name");
            }
            const size_t before = names.size();
            names += tx->chr[k];
            max_name = std::max(max_name, names.size() - before);
        }
        name_off[n_sel] = (uint32_t)names.size();
        for (uint32_t f = 0; f < nf; ++f) {
            id_off[f] = (uint32_t)ids.size();
            if (!tx->feat_id[f] || ids.size() > 0xF0000000ull) {
                (void)hipStreamSynchronize(h->compute);
                release();
                return gams_fail(h, GAMS_EINVAL, "gpu_sw_text: null feature id, or more than 4 GB of ids");
            }
            const size_t before = ids.size();
            ids += tx->feat_id[f];
            max_id = std::max(max_id, ids.size() - before);
        }
        id_off[nf] = (uint32_t)ids.size();
        auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const uint32_t nb = (uint32_t)((n_out + kSwTextBlock - 1) / kSwTextBlock);
        const size_t t_crow = al(((size_t)n_sel + 1) * 8), t_noff = al(((size_t)n_sel + 1) * 4), t_names = al(names.size() + 1),
                     t_ioff = al(((size_t)nf + 1) * 4), t_ids = al(ids.size() + 1);
        const size_t tab_bytes = t_crow + t_noff + t_names + t_ioff + t_ids;
        const size_t b_len = al((size_t)std::max<uint64_t>(n_out, 1) * 4), b_blen = al((size_t)std::max(nb, 1u) * 4),
                     b_boff = al(((size_t)nb + 1) * 8), b_words = al(((size_t)n_sel + 3) * 8);
        const uint64_t text_cap = std::max<uint64_t>(n_out * (uint64_t)(max_id + max_name + 96), 4096);
        uint8_t *tpin = nullptr, *tdev = nullptr;
        size_t tpin_cap = 0, tdev_cap = 0;
        auto release_text = [&]() {
            if (tpin) gams_pool_free(h, true, tpin, tpin_cap);
            if (tdev) gams_pool_free(h, false, tdev, tdev_cap);
        };
#define SWT_HIP(call)                                                                  \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (void)hipStreamSynchronize(h->compute);                                    \
            release_text();                                                            \
            release();                                                                 \
            return gams_fail(h, e_ == hipErrorOutOfMemory ? GAMS_ENOMEM : GAMS_EHIP,   \
                             std::string(#call) + ": " + hipGetErrorString(e_));       \
        }                                                                              \
    } while (0)
        SWT_HIP(gams_pool_alloc(h, true, tab_bytes, reinterpret_cast<void **>(&tpin), &tpin_cap));
        SWT_HIP(gams_pool_alloc(h, false, tab_bytes + b_len + b_blen + b_boff + b_words + al(text_cap),
                                reinterpret_cast<void **>(&tdev), &tdev_cap));
        std::memcpy(tpin, crow.data(), crow.size() * 8);
        std::memcpy(tpin + t_crow, name_off.data(), name_off.size() * 4);
        std::memcpy(tpin + t_crow + t_noff, names.data(), names.size());
        std::memcpy(tpin + t_crow + t_noff + t_names, id_off.data(), id_off.size() * 4);
        std::memcpy(tpin + t_crow + t_noff + t_names + t_ioff, ids.data(), ids.size());
        SWT_HIP(hipMemcpyAsync(tdev, tpin, tab_bytes, hipMemcpyHostToDevice, h->compute));
        SwTextArgs ta{};
        ta.rows = d_rows;
        ta.n_rows = n_out;
        ta.ctgs = a.ctgs;
        ta.ctg_row_off = reinterpret_cast<const uint64_t *>(tdev);
        ta.n_sel = n_sel;
        ta.feat_row_off = a.row_off;
        ta.name_off = reinterpret_cast<const uint32_t *>(tdev + t_crow);
        ta.names = reinterpret_cast<const char *>(tdev + t_crow + t_noff);
        ta.id_off = reinterpret_cast<const uint32_t *>(tdev + t_crow + t_noff + t_names);
        ta.ids = reinterpret_cast<const char *>(tdev + t_crow + t_noff + t_names + t_ioff);
        uint8_t *q = tdev + tab_bytes;
        ta.len = reinterpret_cast<uint32_t *>(q);
        ta.blk_len = reinterpret_cast<uint32_t *>(q + b_len);
        unsigned long long *blk_off = reinterpret_cast<unsigned long long *>(q + b_len + b_blen);
        ta.blk_off = blk_off;
        ta.nb = nb;
        ta.words = reinterpret_cast<unsigned long long *>(q + b_len + b_blen + b_boff);
        ta.text = reinterpret_cast<char *>(q + b_len + b_blen + b_boff + b_words);
        ta.text_cap = text_cap;
        SWT_HIP(hipMemsetAsync(ta.words, 0, b_words, h->compute));
        if (nb) {
            hipLaunchKernelGGL(sw_text_len_kernel, dim3(nb), dim3(256), 0, h->compute, ta);
            hipLaunchKernelGGL(sw_text_scan_kernel, dim3(1), dim3(1024), 0, h->compute, ta.blk_len, nb, blk_off, ta.words);
            hipLaunchKernelGGL(sw_text_write_kernel, dim3(nb), dim3(256), 0, h->compute, ta);
            hipLaunchKernelGGL(sw_text_ctg_kernel, dim3(n_sel + 1), dim3(64), 0, h->compute, ta);
            SWT_HIP(hipGetLastError());
        }
        // the words first (total, flag, per-ctg offsets), then the text -- whose size they say -- into the handle's
        // page-locked text buffer, valid until the next call
        const size_t n_words = (size_t)n_sel + 3;
        if (h->sw_words_bytes < n_words * 8) {
            gams_pool_free(h, true, h->sw_words, h->sw_words_bytes);
            h->sw_words = nullptr;
            h->sw_words_bytes = 0;
            SWT_HIP(gams_pool_alloc(h, true, n_words * 8, reinterpret_cast<void **>(&h->sw_words), &h->sw_words_bytes));
        }
        SWT_HIP(hipMemcpyAsync(h->sw_words, ta.words, n_words * 8, hipMemcpyDeviceToHost, h->compute));
        SWT_HIP(hipStreamSynchronize(h->compute));
        const uint64_t bytes = h->sw_words[0];
        if (h->sw_words[1] != 0 || bytes > text_cap) {
            release_text();
            release();
            return gams_fail(h, GAMS_EUNSUPPORTED,
                             "gpu_sw_text: a value this formatter does not cover (a statistic of 1000 or more, a negative "
                             "coordinate): format gams_gpu_sw_batch's rows on the host");
        }
        if (h->sw_text_bytes < bytes) {
            gams_pool_free(h, true, h->sw_text, h->sw_text_bytes);
            h->sw_text = nullptr;
            h->sw_text_bytes = 0;
            SWT_HIP(gams_pool_alloc(h, true, bytes + bytes / 8 + 4096, reinterpret_cast<void **>(&h->sw_text), &h->sw_text_bytes));
        }
        if (bytes) SWT_HIP(hipMemcpyAsync(h->sw_text, ta.text, bytes, hipMemcpyDeviceToHost, h->compute));
        SWT_HIP(hipStreamSynchronize(h->compute));
#undef SWT_HIP
        *tx->text = bytes ? h->sw_text : nullptr;
        *tx->text_bytes = bytes;
        if (tx->ctg_off) *tx->ctg_off = reinterpret_cast<const uint64_t *>(h->sw_words + 2);
        release_text();
        release();
        return GAMS_OK;
    }
    SW_HIP(hipMemcpyAsync(rows, d_rows, n_out * sizeof(gams_sw_row_t), hipMemcpyDeviceToHost, h->compute));
    SW_HIP(hipStreamSynchronize(h->compute));
#undef SW_HIP
    release();
    return GAMS_OK;
}

extern "C" int gams_gpu_sw(gams_gpu_t *h, gams_seqset_t *s, uint32_t i, int32_t chr_start,
                           const int32_t *feat_start, const int32_t *feat_end, uint32_t nf, int32_t size,
                           int32_t max, int32_t resize, gams_sw_row_t *rows, uint64_t cap, uint64_t *n_rows) {
    if (!h || !s || !n_rows || (nf && (!feat_start || !feat_end)))
        return gams_fail(h, GAMS_EINVAL, "gpu_sw: null argument");
    const uint64_t feat_off[2] = {0, nf};
    return gams_gpu_sw_batch(h, s, 1, &i, &chr_start, feat_off, feat_start, feat_end, size, max, resize, rows, cap,
                             nullptr, n_rows);
}

// gc_content (round4) of arbitrary chromosome ranges inside ctg i: gams::cache_gc_content
// (src/libs/utils.rs:141-162) as `gams peak` uses it (src/cmd_gams/peak.rs:79).
extern "C" int gams_gpu_range_gc_batch(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                                       const int32_t *chr_start, const uint64_t *range_off, const int32_t *range_start,
                                       const int32_t *range_end, float *gc) {
    if (!h || !s || (n_sel && (!ctg_index || !chr_start || !range_off)))
        return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: null argument");
    if (n_sel == 0) return GAMS_OK;
    if (range_off[0] != 0) return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: range_off[0] must be 0");
    for (uint32_t k = 0; k < n_sel; ++k) {
        if (ctg_index[k] >= s->n_ctg) return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: ctg index out of range");
        if (range_off[k + 1] < range_off[k]) return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: range_off must not decrease");
        const uint32_t len = s->len[ctg_index[k]];
        if (len == 0 || len > 0x7fffffffu) return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: ctg length out of range");
    }
    const uint64_t n64 = range_off[n_sel];
    if (n64 == 0) return GAMS_OK;
    if (!range_start || !range_end || !gc) return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: null argument");
    if (n64 > 0x7fffff00ull) return gams_fail(h, GAMS_EUNSUPPORTED, "gpu_range_gc: too many ranges for one launch");
    const uint32_t n = (uint32_t)n64;
    GAMS_HIP(h, hipSetDevice(h->device));
    // inputs in one page-locked block (one DMA): ctgs | rs | re | rctg; results in the same device block
    const size_t b_ctg = ((size_t)n_sel * sizeof(SwCtg) + 255) & ~(size_t)255;
    const size_t b_i32 = ((size_t)n * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t in_bytes = b_ctg + 3 * b_i32;
    uint8_t *pin = nullptr, *dev = nullptr;
    size_t pin_cap = 0, dev_cap = 0;
    hipError_t e = gams_pool_alloc(h, true, in_bytes, reinterpret_cast<void **>(&pin), &pin_cap);
    if (e != hipSuccess) return gams_fail(h, GAMS_ENOMEM, std::string("gpu_range_gc: pinned staging: ") + hipGetErrorString(e));
    auto release = [&]() {
        if (pin) gams_pool_free(h, true, pin, pin_cap);
        if (dev) gams_pool_free(h, false, dev, dev_cap);
    };
    SwCtg *cg = reinterpret_cast<SwCtg *>(pin);
    int32_t *rs = reinterpret_cast<int32_t *>(pin + b_ctg), *re = reinterpret_cast<int32_t *>(pin + b_ctg + b_i32);
    uint32_t *rctg = reinterpret_cast<uint32_t *>(pin + b_ctg + 2 * b_i32);
    for (uint32_t k = 0; k < n_sel; ++k) {
        const uint32_t i = ctg_index[k];
        const int32_t cs = chr_start[k];
        const int64_t ce = (int64_t)cs + s->len[i] - 1;
        cg[k] = SwCtg{s->off[i], s->len[i], cs, (int32_t)ce, (uint32_t)range_off[k], {0u, 0u}};
        // utils.rs:151-156 slices seq[from-1..to): a range outside the ctg (or inverted) panics there
        for (uint64_t q = range_off[k]; q < range_off[k + 1]; ++q) {
            if (range_start[q] < cs || range_end[q] > ce || range_end[q] < range_start[q]) {
                release();
                return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: range " + std::to_string(q - range_off[k]) +
                                                     (n_sel > 1 ? " of selected ctg " + std::to_string(k) : std::string()) +
                                                     " is not inside the ctg");
            }
            rs[q] = range_start[q];
            re[q] = range_end[q];
            rctg[q] = k;
        }
    }
    int rc = gams_seqset_gcindex(h, s);
    if (rc != GAMS_OK) {
        release();
        return rc;
    }
    e = gams_pool_alloc(h, false, in_bytes + b_i32, reinterpret_cast<void **>(&dev), &dev_cap);
    if (e != hipSuccess) {
        release();
        return gams_fail(h, GAMS_ENOMEM, std::string("gpu_range_gc: device buffers: ") + hipGetErrorString(e));
    }
#define R_HIP(call)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipStreamSynchronize(h->compute);                                            \
            release();                                                                         \
            return gams_fail(h, GAMS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)
    R_HIP(hipMemcpyAsync(dev, pin, in_bytes, hipMemcpyHostToDevice, h->compute));
    float *d_gc = reinterpret_cast<float *>(dev + in_bytes);
    SwArgs a{};
    a.pm = s->gcindex->d_pm;
    a.seg = s->gcindex->d_seg;
    a.ctgs = reinterpret_cast<const SwCtg *>(dev);
    a.fctg = reinterpret_cast<const uint32_t *>(dev + b_ctg + 2 * b_i32);
    R_HIP(hipEventRecord(h->k0, h->compute));
    hipLaunchKernelGGL(range_gc_kernel, dim3((n + 255) / 256), dim3(256), 0, h->compute, a,
                       reinterpret_cast<const int32_t *>(dev + b_ctg), reinterpret_cast<const int32_t *>(dev + b_ctg + b_i32), n,
                       d_gc);
    R_HIP(hipGetLastError());
    R_HIP(hipEventRecord(h->k1, h->compute));
    h->k_valid = true;
    h->kq_used = 0;
    R_HIP(hipMemcpyAsync(gc, d_gc, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, h->compute));
    R_HIP(hipStreamSynchronize(h->compute));
#undef R_HIP
    release();
    return GAMS_OK;
}

extern "C" int gams_gpu_range_gc(gams_gpu_t *h, gams_seqset_t *s, uint32_t i, int32_t chr_start,
                                 const int32_t *range_start, const int32_t *range_end, uint32_t n, float *gc) {
    if (!h || !s || (n && (!range_start || !range_end || !gc)))
        return gams_fail(h, GAMS_EINVAL, "gpu_range_gc: null argument");
    const uint64_t range_off[2] = {0, n};
    return gams_gpu_range_gc_batch(h, s, 1, &i, &chr_start, range_off, range_start, range_end, gc);
}
