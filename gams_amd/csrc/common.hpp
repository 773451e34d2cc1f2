// common.hpp -- shared host/device plumbing of libgams_gpu (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/gams_gpu.h"
#include "../../include/gams_gpu_diag.h"

struct gams_gpu {
    int device = 0;
    hipStream_t compute = nullptr;  // every kernel of the library runs here
    hipStream_t copy = nullptr;     // H2D staging of seq: bytes
    hipEvent_t copy2_ev = nullptr;  // gams_seqset_upload_all: joins the second DMA queue (the readback stream) into `copy`
    hipStream_t readback = nullptr; // packing + D2H of a finished run's results (waits on that run only)
    // a wave plan of depth D rotates its runs over `compute` and aux[0..D-2] (gams_wave_plan_set_depth)
    static constexpr int kMaxWays = 4;
    hipStream_t aux[kMaxWays - 1] = {};
    hipEvent_t aux_ev[kMaxWays - 1] = {};
    hipEvent_t rd_ev[kMaxWays] = {};   // uploads queue behind every stream that may still read a seqset
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;  // around the kernel of the last query-style call
    bool k_valid = false;
    // chunked query calls (gams_gpu_count / locate / cover): per-slot ordering events and one timed pair
    // per chunk; gams_gpu_last_kernel_ms sums the pairs of the last call when kq_used > 0
    hipEvent_t q_ev[2][3] = {};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> kq;
    int kq_used = 0;
    // pinned staging ring of the copy stream: the CPU fills slot k+1 while the DMA drains slot k
    static constexpr int kStageSlots = 4;
    static constexpr size_t kStageBytes = 16u << 20;
    uint8_t *stage[kStageSlots] = {};
    hipEvent_t stage_free[kStageSlots] = {};
    int stage_next = 0;
    unsigned long long *pin_scratch = nullptr;   // 4 KiB of pinned host memory for few-word readbacks
    int cus = 0;
    uint64_t hbm = 0;
    char arch[64] = {0};
    std::string err;
    std::mutex err_mu;
    // Freed HBM / pinned-host blocks kept for the next batch: hipMalloc of a few hundred MB and
    // hipHostMalloc of the peak buffer cost milliseconds each (13 ms + 5 ms per 384-Mb batch).
    struct Block {
        void *p;
        size_t bytes;
    };
    std::vector<Block> dev_pool, pin_pool;
    std::vector<Block> host_blocks;   // page-locked blocks handed out by gams_gpu_host_alloc (their pooled sizes)
    // gams_gpu_sw_text: the last call's text and words (total, flag, per-ctg offsets), page-locked, valid until the next call
    char *sw_text = nullptr;
    size_t sw_text_bytes = 0;
    unsigned long long *sw_words = nullptr;
    size_t sw_words_bytes = 0;
    // hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel FUNCTION (per device), not to a plan:
    // the largest value any launch on this handle has asked for, per function; only ever raised (gams_lds_attr)
    // counts the launches that read a seqset (any stream of the handle): gams_seqset_upload_image queues the
    // copy stream behind the handle's streams only when a reader was launched since it last did so
    std::atomic<uint64_t> reader_epoch{1};
    std::unordered_map<const void *, size_t> lds_attr;
    std::mutex lds_mu;
};

// Make sure launches of `func` on the handle's device may use `bytes` of dynamic LDS.  Plans with different
// tile sizes / lags share one kernel function; lowering the attribute behind another plan's back would fail
// that plan's next launch, so the recorded maximum only grows.
inline hipError_t gams_lds_attr(gams_gpu_t *h, const void *func, size_t bytes) {
    std::lock_guard<std::mutex> lk(h->lds_mu);
    size_t &have = h->lds_attr[func];
    if (bytes <= have) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// Pooled allocation on the handle's device (pinned = page-locked host memory).  *cap is the size
// of the block handed out (>= bytes); pass it back to gams_pool_free.
hipError_t gams_pool_alloc(gams_gpu_t *h, bool pinned, size_t bytes, void **out, size_t *cap);
void gams_pool_free(gams_gpu_t *h, bool pinned, void *p, size_t cap);

inline int gams_fail(gams_gpu_t *h, int code, const std::string &msg) {
    (void)hipGetLastError();   // a failed HIP call is reported through the code; it must not stay sticky
    if (h) {
        std::lock_guard<std::mutex> lk(h->err_mu);   // gams_wave_run_n queues from two host threads
        h->err = msg;
    }
    return code;
}

#define GAMS_HIP(h, call)                                                           \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            (void)hipGetLastError(); /* reported here; not left sticky */           \
            return gams_fail((h), GAMS_EHIP,                                        \
                             std::string(#call) + ": " + hipGetErrorString(e_));    \
        }                                                                           \
    } while (0)

struct gams_gcindex;  // sw.hip: prefix index over the seqset buffer

struct gams_seqset {
    uint32_t n_ctg = 0;
    std::vector<uint32_t> len;      // bases per ctg
    std::vector<uint64_t> off;      // byte offset of ctg i inside d_seq (256-B aligned)
    uint64_t bytes = 0;             // bytes in use (with tail slack for 16-B over-reads)
    size_t cap = 0;                 // size of the pooled block behind d_seq
    uint8_t *d_seq = nullptr;
    hipEvent_t uploaded = nullptr;    // recorded on the copy stream after the last upload; kernels wait on it
    bool dirty = false;               // an upload happened since the last wait was queued (compute stream)
    uint64_t upload_gen = 0;          // counts uploads; streams other than `compute` compare it with what they saw
    uint64_t ordered_epoch = 0;       // gams_gpu::reader_epoch when an upload last queued itself behind the readers
    gams_gcindex *gcindex = nullptr;  // built lazily by gams_gpu_sw, dropped by every upload
};

// make the compute stream wait for every upload queued so far (no host blocking)
int gams_seqset_wait_uploads(gams_gpu_t *h, gams_seqset_t *s);
int gams_stage_ring(gams_gpu_t *h);   // allocate the pinned staging slots on first use
// make `st` (a copy stream) wait for everything queued so far on the compute and auxiliary streams
int gams_order_after_readers(gams_gpu_t *h, hipStream_t st);
int gams_seqset_gcindex(gams_gpu_t *h, gams_seqset_t *s);
void gams_seqset_gcindex_free(gams_seqset_t *s);

// 16 bytes of sequence that this kernel will not read again: non-temporal load (global_load ... nt).
// A pure streaming read runs at 7.06 TB/s with the hint, 6.3 TB/s without (profiles/r02_stream_read.txt).
__device__ __forceinline__ uint4 load_once16(const uint4 *p) {
    typedef unsigned v4u_ __attribute__((ext_vector_type(4)));
    const v4u_ t = __builtin_nontemporal_load(reinterpret_cast<const v4u_ *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

// ---------------------------------------------------------------------------
// wave64 / workgroup scan primitives (device)
// ---------------------------------------------------------------------------
// Inclusive add-scan across the 64 lanes of a wavefront with DPP row shifts and
// the GFX9 row broadcasts: 6 VALU adds, no LDS traffic.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return (uint32_t)v;
}

// 64-bit variant (only the wide z-score path needs it): shuffle based.
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t x) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = __shfl_up((uint32_t)x, d, 64);
        uint32_t hi = __shfl_up((uint32_t)(x >> 32), d, 64);
        uint64_t y = ((uint64_t)hi << 32) | lo;
        if (lane >= d) x += y;
    }
    return x;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) { return wave_incl_scan_u32(x); }
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t x) { return wave_incl_scan_u64(x); }

// Exclusive scan over a 256-thread workgroup (4 waves).  `ws` is LDS scratch of
// 4 elements; the call contains two barriers and may be used back to back.
template <typename T, int NTH = 256>
__device__ __forceinline__ T block_excl_scan_256(T v, T *ws, T &total) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    T inc = wave_incl_scan(v);
    if (lane == 63) ws[wv] = inc;
    __syncthreads();
    // (NTH = 64 / 128: workgroups of one or two waves, the missing waves count as empty)
    T w0 = ws[0], w1 = NTH > 64 ? ws[1] : (T)0, w2 = NTH > 128 ? ws[2] : (T)0, w3 = NTH > 128 ? ws[3] : (T)0;
    __syncthreads();
    T base = (wv > 0 ? w0 : (T)0) + (wv > 1 ? w1 : (T)0) + (wv > 2 ? w2 : (T)0);
    total = w0 + w1 + w2 + w3;
    return base + inc - v;
}

