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

extern "C" int gams_gpu_sw_batch(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                                 const int32_t *chr_start, const uint64_t *feat_off, const int32_t *feat_start,
                                 const int32_t *feat_end, int32_t size, int32_t max, int32_t resize,
                                 gams_sw_row_t *rows, uint64_t cap, uint64_t *row_off, uint64_t *n_rows) {
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
    const bool size_query = !rows || cap == 0;
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
    *n_rows = tot;
    if (size_query) return GAMS_OK;

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
