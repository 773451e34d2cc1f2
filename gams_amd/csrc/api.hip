// api.hip -- handle, timers and HBM-resident ctg sequences of libgams_gpu.

#include "common.hpp"

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <thread>

extern "C" {

int gams_gpu_create(int device, gams_gpu_t **out) {
    if (!out) return GAMS_EINVAL;
    *out = nullptr;
    // (Kernel arguments in device memory -- HIP_FORCE_DEV_KERNARG=1 -- are the HOST's to ask for, before
    // its first HIP call: the library does not touch the process environment.  INTEGRATION.md.)
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GAMS_ENODEV;
    if (device < 0 || device >= n) return GAMS_ENODEV;
    gams_gpu_t *h = new gams_gpu_t();
    h->device = device;
    auto bail = [&](hipError_t e, const char *what) {
        fprintf(stderr, "gams_gpu_create: %s: %s\n", what, hipGetErrorString(e));
        gams_gpu_destroy(h);
        return GAMS_EHIP;
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail(e, "hipGetDeviceProperties");
    h->cus = prop.multiProcessorCount;
    h->hbm = prop.totalGlobalMem;
    snprintf(h->arch, sizeof h->arch, "%s", prop.gcnArchName);
    if ((e = hipStreamCreateWithFlags(&h->compute, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate(compute)");
    // The streams a plan of depth D runs on are created first and together: HIP hands streams to
    // its hardware queues (GPU_MAX_HW_QUEUES, 4 by default, one per CP pipe) in creation order, and
    // streams that end up on queues of the same pipe overlap their kernels badly (tools/queue_map.hip:
    // 9.8 us per launch for four consecutive streams, 14-17 us for {0,2,4,6} on 8 queues).
    for (int k = 0; k < gams_gpu::kMaxWays - 1; ++k)
        if ((e = hipStreamCreateWithFlags(&h->aux[k], hipStreamNonBlocking)) != hipSuccess)
            return bail(e, "hipStreamCreate(aux)");
    if ((e = hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate(copy)");
    if ((e = hipEventCreateWithFlags(&h->copy2_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipStreamCreateWithFlags(&h->readback, hipStreamNonBlocking)) != hipSuccess)
        return bail(e, "hipStreamCreate(readback)");
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->k0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->k1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipHostMalloc(reinterpret_cast<void **>(&h->pin_scratch), 4096, hipHostMallocDefault)) != hipSuccess)
        return bail(e, "hipHostMalloc(scratch)");
    *out = h;
    return GAMS_OK;
}

void gams_gpu_destroy(gams_gpu_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->compute) {
        (void)hipStreamSynchronize(h->compute);
        (void)hipStreamDestroy(h->compute);
    }
    if (h->copy) {
        (void)hipStreamSynchronize(h->copy);
        (void)hipStreamDestroy(h->copy);
    }
    if (h->copy2_ev) (void)hipEventDestroy(h->copy2_ev);
    if (h->readback) {
        (void)hipStreamSynchronize(h->readback);
        (void)hipStreamDestroy(h->readback);
    }
    for (int k = 0; k < gams_gpu::kMaxWays - 1; ++k) {
        if (h->aux[k]) {
            (void)hipStreamSynchronize(h->aux[k]);
            (void)hipStreamDestroy(h->aux[k]);
        }
        if (h->aux_ev[k]) (void)hipEventDestroy(h->aux_ev[k]);
    }
    for (auto &e : h->rd_ev)
        if (e) (void)hipEventDestroy(e);
    gams_pool_free(h, true, h->sw_text, h->sw_text_bytes);      // back to the pool, which is released below
    gams_pool_free(h, true, h->sw_words, h->sw_words_bytes);
    h->sw_text = nullptr;
    h->sw_words = nullptr;
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    for (auto &slot : h->q_ev)
        for (auto &e : slot)
            if (e) (void)hipEventDestroy(e);
    for (auto &pr : h->kq) {
        if (pr.first) (void)hipEventDestroy(pr.first);
        if (pr.second) (void)hipEventDestroy(pr.second);
    }
    if (h->k0) (void)hipEventDestroy(h->k0);
    if (h->k1) (void)hipEventDestroy(h->k1);
    for (int k = 0; k < gams_gpu::kStageSlots; ++k) {
        if (h->stage[k]) (void)hipHostFree(h->stage[k]);
        if (h->stage_free[k]) (void)hipEventDestroy(h->stage_free[k]);
    }
    if (h->pin_scratch) (void)hipHostFree(h->pin_scratch);
    for (auto &b : h->dev_pool) (void)hipFree(b.p);
    for (auto &b : h->pin_pool) (void)hipHostFree(b.p);
    delete h;
}

const char *gams_gpu_last_error(gams_gpu_t *h) { return h ? h->err.c_str() : "null handle"; }

int gams_gpu_device_info(gams_gpu_t *h, char *arch, size_t arch_len, int32_t *compute_units,
                         uint64_t *hbm_bytes) {
    if (!h) return GAMS_EINVAL;
    if (arch && arch_len) snprintf(arch, arch_len, "%s", h->arch);
    if (compute_units) *compute_units = h->cus;
    if (hbm_bytes) *hbm_bytes = h->hbm;
    return GAMS_OK;
}

int gams_gpu_sync(gams_gpu_t *h) {
    if (!h) return GAMS_EINVAL;
    GAMS_HIP(h, hipSetDevice(h->device));
    GAMS_HIP(h, hipStreamSynchronize(h->copy));
    GAMS_HIP(h, hipStreamSynchronize(h->compute));
    for (int k = 0; k < gams_gpu::kMaxWays - 1; ++k)
        if (h->aux[k]) GAMS_HIP(h, hipStreamSynchronize(h->aux[k]));
    GAMS_HIP(h, hipStreamSynchronize(h->readback));
    return GAMS_OK;
}

int gams_gpu_release_cached(gams_gpu_t *h, uint64_t *cached_bytes) {
    if (!h) return GAMS_EINVAL;
    GAMS_HIP(h, hipSetDevice(h->device));
    uint64_t held = 0;
    for (auto &b : h->dev_pool) {
        held += b.bytes;
        (void)hipFree(b.p);
    }
    for (auto &b : h->pin_pool) {
        held += b.bytes;
        (void)hipHostFree(b.p);
    }
    h->dev_pool.clear();
    h->pin_pool.clear();
    if (cached_bytes) *cached_bytes = held;
    return GAMS_OK;
}

int gams_gpu_timer_start(gams_gpu_t *h) {
    if (!h) return GAMS_EINVAL;
    GAMS_HIP(h, hipSetDevice(h->device));
    GAMS_HIP(h, hipEventRecord(h->ev0, h->compute));
    return GAMS_OK;
}

int gams_gpu_timer_stop(gams_gpu_t *h, float *ms) {
    if (!h || !ms) return gams_fail(h, GAMS_EINVAL, "timer_stop: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    // the timed region ends when every stream the plans run on has drained: the compute stream
    // queues behind the auxiliary ones before the closing event
    for (int k = 0; k < gams_gpu::kMaxWays - 1; ++k) {
        if (!h->aux[k]) continue;
        if (!h->aux_ev[k]) GAMS_HIP(h, hipEventCreateWithFlags(&h->aux_ev[k], hipEventDisableTiming));
        GAMS_HIP(h, hipEventRecord(h->aux_ev[k], h->aux[k]));
        GAMS_HIP(h, hipStreamWaitEvent(h->compute, h->aux_ev[k], 0));
    }
    GAMS_HIP(h, hipEventRecord(h->ev1, h->compute));
    GAMS_HIP(h, hipEventSynchronize(h->ev1));
    GAMS_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return GAMS_OK;
}

int gams_gpu_last_kernel_ms(gams_gpu_t *h, float *ms) {
    if (!h || !ms) return gams_fail(h, GAMS_EINVAL, "last_kernel_ms: null argument");
    if (!h->k_valid) return gams_fail(h, GAMS_ESTATE, "last_kernel_ms: no timed call yet");
    GAMS_HIP(h, hipSetDevice(h->device));
    if (h->kq_used > 0) {          // a chunked query call: the sum of its kernels
        float total = 0.0f;
        for (int c = 0; c < h->kq_used; ++c) {
            float part = 0.0f;
            GAMS_HIP(h, hipEventSynchronize(h->kq[c].second));
            GAMS_HIP(h, hipEventElapsedTime(&part, h->kq[c].first, h->kq[c].second));
            total += part;
        }
        *ms = total;
        return GAMS_OK;
    }
    GAMS_HIP(h, hipEventSynchronize(h->k1));
    GAMS_HIP(h, hipEventElapsedTime(ms, h->k0, h->k1));
    return GAMS_OK;
}

int gams_gpu_host_alloc(gams_gpu_t *h, uint64_t bytes, void **p) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "host_alloc: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    // from the handle's pool of page-locked blocks: pinning 150 MB takes tens of milliseconds, a host that asks
    // for the same columns batch after batch gets the block of the batch before
    size_t cap = 0;
    hipError_t e = gams_pool_alloc(h, true, (size_t)std::max<uint64_t>(bytes, 1), p, &cap);
    if (e != hipSuccess) return gams_fail(h, GAMS_ENOMEM, std::string("host_alloc: ") + hipGetErrorString(e));
    h->host_blocks.push_back({*p, cap});
    return GAMS_OK;
}

void gams_gpu_host_free(gams_gpu_t *h, void *p) {
    if (!p) return;
    if (!h) {
        (void)hipHostFree(p);
        return;
    }
    (void)hipSetDevice(h->device);
    for (size_t i = 0; i < h->host_blocks.size(); ++i)
        if (h->host_blocks[i].p == p) {
            const size_t cap = h->host_blocks[i].bytes;
            h->host_blocks.erase(h->host_blocks.begin() + (long)i);
            gams_pool_free(h, true, p, cap);
            return;
        }
    (void)hipHostFree(p);   // not one of this handle's blocks
}

// ---------------------------------------------------------------------------
// seqset: the batch of gunzipped `seq:{ctg}` values, 1 B/base, in one HBM buffer
// ---------------------------------------------------------------------------
int gams_seqset_create(gams_gpu_t *h, uint32_t n_ctg, const uint32_t *lengths, gams_seqset_t **out) {
    if (!h || !out || (n_ctg && !lengths)) return gams_fail(h, GAMS_EINVAL, "seqset_create: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    gams_seqset_t *s = new gams_seqset_t();
    s->n_ctg = n_ctg;
    s->len.assign(lengths, lengths + n_ctg);
    s->off.resize(n_ctg);
    uint64_t o = 0;
    for (uint32_t i = 0; i < n_ctg; ++i) {
        s->off[i] = o;
        o += ((uint64_t)lengths[i] + 255u) & ~(uint64_t)255u;  // next ctg on a 256-B boundary
    }
    s->bytes = o + 65536;  // tail slack: kernels read whole 16-B chunks, the baked wave kernels whole 4-KiB rows
    hipError_t e = gams_pool_alloc(h, false, s->bytes, reinterpret_cast<void **>(&s->d_seq), &s->cap);
    if (e != hipSuccess) {
        delete s;
        return gams_fail(h, e == hipErrorOutOfMemory ? GAMS_ENOMEM : GAMS_EHIP,
                         std::string("seqset_create: hipMalloc: ") + hipGetErrorString(e));
    }
    // Padding bytes are never counted, but keep them defined (the block may be a recycled one).
    // Queued on the copy stream in front of the uploads; nothing waits for it on the host.
    e = hipMemsetAsync(s->d_seq, 0, s->bytes, h->copy);
    if (e != hipSuccess) {
        gams_pool_free(h, false, s->d_seq, s->cap);
        delete s;
        return gams_fail(h, GAMS_EHIP, std::string("seqset_create: memset: ") + hipGetErrorString(e));
    }
    // readers on the compute stream order themselves behind the memset like behind an upload
    if (hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming) == hipSuccess &&
        hipEventRecord(s->uploaded, h->copy) == hipSuccess)
        s->dirty = true;
    ++s->upload_gen;
    *out = s;
    return GAMS_OK;
}

int gams_seqset_upload(gams_gpu_t *h, gams_seqset_t *s, uint32_t i, const uint8_t *seq) {
    if (!h || !s || !seq) return gams_fail(h, GAMS_EINVAL, "seqset_upload: null argument");
    if (i >= s->n_ctg) return gams_fail(h, GAMS_EINVAL, "seqset_upload: ctg index out of range");
    GAMS_HIP(h, hipSetDevice(h->device));
    if (s->len[i] == 0) return GAMS_OK;
    if (s->gcindex) {
        GAMS_HIP(h, hipStreamSynchronize(h->compute));
        gams_seqset_gcindex_free(s);  // the bytes it indexed are about to change
    }
    // Pageable host memory -> pinned ring -> HBM: the call returns once the last piece is queued;
    // the host memcpy of piece k+1 overlaps the DMA of piece k.  Kernels that read the seqset
    // wait on s->uploaded (gams_seqset_wait_uploads), not the host.
    {
        int rc = gams_stage_ring(h);
        if (rc != GAMS_OK) return rc;
    }
    if (!s->uploaded) GAMS_HIP(h, hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming));
    // passes queued earlier on any stream of the handle may still be reading these bytes
    {
        int rc = gams_order_after_readers(h, h->copy);
        if (rc != GAMS_OK) return rc;
    }
    const uint64_t len = s->len[i];
    for (uint64_t o = 0; o < len; o += gams_gpu::kStageBytes) {
        const size_t n = (size_t)std::min<uint64_t>(gams_gpu::kStageBytes, len - o);
        const int k = h->stage_next;
        h->stage_next = (k + 1) % gams_gpu::kStageSlots;
        GAMS_HIP(h, hipEventSynchronize(h->stage_free[k]));          // the DMA that used this slot is done
        std::memcpy(h->stage[k], seq + o, n);
        GAMS_HIP(h, hipMemcpyAsync(s->d_seq + s->off[i] + o, h->stage[k], n, hipMemcpyHostToDevice, h->copy));
        GAMS_HIP(h, hipEventRecord(h->stage_free[k], h->copy));
    }
    GAMS_HIP(h, hipEventRecord(s->uploaded, h->copy));
    s->dirty = true;
    ++s->upload_gen;
    return GAMS_OK;
}

int gams_seqset_upload_all(gams_gpu_t *h, gams_seqset_t *s, const uint8_t *const *seqs) {
    if (!h || !s || (s->n_ctg && !seqs)) return gams_fail(h, GAMS_EINVAL, "seqset_upload_all: null argument");
    for (uint32_t i = 0; i < s->n_ctg; ++i)
        if (s->len[i] && !seqs[i]) return gams_fail(h, GAMS_EINVAL, "seqset_upload_all: null sequence");
    GAMS_HIP(h, hipSetDevice(h->device));
    if (s->n_ctg == 0) return GAMS_OK;
    if (s->gcindex) {
        GAMS_HIP(h, hipStreamSynchronize(h->compute));
        gams_seqset_gcindex_free(s);
    }
    {
        int rc = gams_stage_ring(h);
        if (rc != GAMS_OK) return rc;
    }
    if (!s->uploaded) GAMS_HIP(h, hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming));
    // The staging slots mirror the device layout (ctgs on 256-B boundaries, gaps zeroed), so one
    // window of kStageBytes is one DMA.  Worker t owns slot t and takes windows t, t+T, ...: its
    // memcpy of the next window overlaps the DMAs the other workers queued.
    const uint64_t end = s->off[s->n_ctg - 1] + s->len[s->n_ctg - 1];
    // a small batch is cut finer so that all workers and the DMA engine still overlap
    const uint64_t W = std::min<uint64_t>(gams_gpu::kStageBytes,
                                          std::max<uint64_t>(1u << 20, (end / (2 * gams_gpu::kStageSlots) + 4095) & ~4095ull));
    const uint64_t n_win = (end + W - 1) / W;
    const unsigned T = (unsigned)std::min<uint64_t>(gams_gpu::kStageSlots, std::max<uint64_t>(n_win, 1));
    std::atomic<int> failed{0};
    std::string err[gams_gpu::kStageSlots];
    // odd slots go through a second stream (a second DMA engine) -- the readback stream, idle during
    // an upload; a stream of its own would cost one more of the few hardware queues HIP maps
    // streams onto (GPU_MAX_HW_QUEUES, 4 by default).  It starts behind whatever `copy` already
    // holds for this seqset (the memset of gams_seqset_create, earlier uploads).
    hipStream_t second = h->readback;
    // passes queued earlier on any stream of the handle may still be reading these bytes
    {
        int rc = gams_order_after_readers(h, h->copy);
        if (rc != GAMS_OK) return rc;
    }
    GAMS_HIP(h, hipEventRecord(s->uploaded, h->copy));
    GAMS_HIP(h, hipStreamWaitEvent(second, s->uploaded, 0));
    auto work = [&](unsigned t) {
        hipStream_t cs = (t & 1u) ? second : h->copy;
        hipError_t e = hipSetDevice(h->device);
        for (uint64_t w = t; w < n_win && e == hipSuccess && !failed.load(); w += T) {
            const uint64_t lo = w * W, hi = std::min(end, lo + W);
            e = hipEventSynchronize(h->stage_free[t]);              // the DMA that used this slot is done
            if (e != hipSuccess) break;
            uint8_t *dst = h->stage[t];
            // first ctg whose bytes reach into the window
            uint32_t i = (uint32_t)(std::upper_bound(s->off.begin(), s->off.end(), lo) - s->off.begin());
            i = i ? i - 1 : 0;
            uint64_t filled = lo;                                       // stage holds [lo, filled)
            for (; i < s->n_ctg && s->off[i] < hi; ++i) {
                const uint64_t a = std::max(lo, s->off[i]), b = std::min(hi, s->off[i] + s->len[i]);
                if (a > filled) std::memset(dst + (filled - lo), 0, a - filled);   // alignment gap
                if (b > a) {
                    std::memcpy(dst + (a - lo), seqs[i] + (a - s->off[i]), b - a);
                    filled = b;
                } else if (a > filled) {
                    filled = a;
                }
            }
            if (hi > filled) std::memset(dst + (filled - lo), 0, hi - filled);
            e = hipMemcpyAsync(s->d_seq + lo, dst, hi - lo, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipEventRecord(h->stage_free[t], cs);
        }
        if (e != hipSuccess) {
            err[t] = hipGetErrorString(e);
            failed.store(1);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    if (failed.load()) {
        for (auto &m : err)
            if (!m.empty()) return gams_fail(h, GAMS_EHIP, "seqset_upload_all: " + m);
    }
    GAMS_HIP(h, hipEventRecord(h->copy2_ev, second));
    GAMS_HIP(h, hipStreamWaitEvent(h->copy, h->copy2_ev, 0));
    GAMS_HIP(h, hipEventRecord(s->uploaded, h->copy));
    s->dirty = true;
    ++s->upload_gen;
    return GAMS_OK;
}

int gams_seqset_layout(gams_gpu_t *h, const gams_seqset_t *s, uint64_t *offsets, uint64_t *bytes) {
    if (!s) return gams_fail(h, GAMS_EINVAL, "seqset_layout: null seqset");
    if (offsets)
        for (uint32_t i = 0; i < s->n_ctg; ++i) offsets[i] = s->off[i];
    if (bytes) *bytes = s->bytes;
    return GAMS_OK;
}

int gams_seqset_upload_image(gams_gpu_t *h, gams_seqset_t *s, const uint8_t *image, uint64_t lo, uint64_t hi) {
    if (!h || !s || !image) return gams_fail(h, GAMS_EINVAL, "seqset_upload_image: null argument");
    if (lo > hi || hi > s->bytes) return gams_fail(h, GAMS_EINVAL, "seqset_upload_image: range outside the seqset");
    GAMS_HIP(h, hipSetDevice(h->device));
    if (lo == hi) return GAMS_OK;
    if (s->gcindex) {
        GAMS_HIP(h, hipStreamSynchronize(h->compute));
        gams_seqset_gcindex_free(s);  // the bytes it indexed are about to change
    }
    if (!s->uploaded) GAMS_HIP(h, hipEventCreateWithFlags(&s->uploaded, hipEventDisableTiming));
    // passes queued earlier on any stream of the handle may still be reading these bytes; once per
    // generation of readers is enough, so only the first range of a burst pays the four event pairs
    const uint64_t epoch = h->reader_epoch.load(std::memory_order_relaxed);
    if (s->ordered_epoch != epoch) {
        int rc = gams_order_after_readers(h, h->copy);
        if (rc != GAMS_OK) return rc;
        s->ordered_epoch = epoch;
    }
    GAMS_HIP(h, hipMemcpyAsync(s->d_seq + lo, image + lo, hi - lo, hipMemcpyHostToDevice, h->copy));
    GAMS_HIP(h, hipEventRecord(s->uploaded, h->copy));
    s->dirty = true;
    ++s->upload_gen;
    return GAMS_OK;
}

void gams_seqset_destroy(gams_gpu_t *h, gams_seqset_t *s) {
    if (!s) return;
    if (h) {
        (void)hipSetDevice(h->device);
        (void)hipStreamSynchronize(h->compute);
    }
    if (h) (void)hipStreamSynchronize(h->copy);
    if (h) (void)hipStreamSynchronize(h->readback);
    gams_seqset_gcindex_free(s);
    if (s->uploaded) (void)hipEventDestroy(s->uploaded);
    gams_pool_free(h, false, s->d_seq, s->cap);
    delete s;
}

}  // extern "C"

hipError_t gams_pool_alloc(gams_gpu_t *h, bool pinned, size_t bytes, void **out, size_t *cap) {
    auto &pool = pinned ? h->pin_pool : h->dev_pool;
    bytes = std::max<size_t>(bytes, 1);
    // best fit among the kept blocks, but never hand a huge block to a small request
    int best = -1;
    for (int i = 0; i < (int)pool.size(); ++i)
        if (pool[i].bytes >= bytes && pool[i].bytes <= std::max<size_t>(2 * bytes, 4u << 20) &&
            (best < 0 || pool[i].bytes < pool[best].bytes))
            best = i;
    if (best >= 0) {
        *out = pool[best].p;
        *cap = pool[best].bytes;
        pool.erase(pool.begin() + best);
        return hipSuccess;
    }
    const size_t want = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);   // 2-MiB granules
    hipError_t e = pinned ? hipHostMalloc(out, want, hipHostMallocDefault) : hipMalloc(out, want);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        for (auto &b : pool) (void)(pinned ? hipHostFree(b.p) : hipFree(b.p));
        pool.clear();
        e = pinned ? hipHostMalloc(out, want, hipHostMallocDefault) : hipMalloc(out, want);
    }
    if (e == hipSuccess)
        *cap = want;
    else
        (void)hipGetLastError();   // the failure is reported through `e`; do not leave it sticky for the next launch check
    return e;
}

void gams_pool_free(gams_gpu_t *h, bool pinned, void *p, size_t cap) {
    if (!p) return;
    if (!h) {
        (void)(pinned ? hipHostFree(p) : hipFree(p));
        return;
    }
    auto &pool = pinned ? h->pin_pool : h->dev_pool;
    pool.push_back({p, cap});
    // keep at most 16 blocks and a quarter of the HBM (1 GiB of pinned memory): drop the oldest
    const size_t limit = pinned ? ((size_t)1 << 30) : (size_t)(h->hbm / 4);
    size_t total = 0;
    for (auto &b : pool) total += b.bytes;
    while (!pool.empty() && (pool.size() > 16 || total > limit)) {
        total -= pool.front().bytes;
        (void)(pinned ? hipHostFree(pool.front().p) : hipFree(pool.front().p));
        pool.erase(pool.begin());
    }
}

// the pinned staging ring, allocated on first use
int gams_stage_ring(gams_gpu_t *h) {
    for (int k = 0; k < gams_gpu::kStageSlots; ++k) {
        if (!h->stage[k]) {
            GAMS_HIP(h, hipHostMalloc(reinterpret_cast<void **>(&h->stage[k]), gams_gpu::kStageBytes, hipHostMallocDefault));
            GAMS_HIP(h, hipEventCreateWithFlags(&h->stage_free[k], hipEventDisableTiming));
            GAMS_HIP(h, hipEventRecord(h->stage_free[k], h->copy));
        }
    }
    return GAMS_OK;
}

int gams_order_after_readers(gams_gpu_t *h, hipStream_t st) {
    for (int k = 0; k < gams_gpu::kMaxWays; ++k) {
        hipStream_t src = k == 0 ? h->compute : h->aux[k - 1];
        if (!src) continue;
        if (!h->rd_ev[k]) GAMS_HIP(h, hipEventCreateWithFlags(&h->rd_ev[k], hipEventDisableTiming));
        GAMS_HIP(h, hipEventRecord(h->rd_ev[k], src));
        GAMS_HIP(h, hipStreamWaitEvent(st, h->rd_ev[k], 0));
    }
    return GAMS_OK;
}

int gams_seqset_wait_uploads(gams_gpu_t *h, gams_seqset_t *s) {
    if (s->dirty) {
        GAMS_HIP(h, hipStreamWaitEvent(h->compute, s->uploaded, 0));
        s->dirty = false;
    }
    return GAMS_OK;
}

