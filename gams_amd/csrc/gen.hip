// gen.hip -- the per-base scan of `gams gen` (first "next" row).
//
// Replaces the loop of src/cmd_gams/gen.rs:86-93: every base that is not one of
// A C G T a c g t is "ambiguous"; the valid set is the complement (gen.rs:100-102), then
// fill(fill-1) and excise(min) (gen.rs:103-104) and the --piece split (gen.rs:108-126).
//
// Device: one lane per 16-B chunk, coalesced 16-B loads, SWAR membership test (three
// exact zero-byte tests per dword: G/C class, A class, T class), 16-bit validity mask, and
// the positions where validity flips (a handful per megabase) appended to a small list.
// Host: sort the flips, pair them into spans, fill, excise, split into pieces -- O(#runs).

#include "common.hpp"

#include <algorithm>

namespace {

// 0x80 in every byte of x that equals `C` after clearing the case bit (0x20); exact for all
// byte values: t's bit 7 is set iff the low 7 bits (without bit 5) differ, x's bit 7 rules out
// bytes >= 0x80.
template <uint32_t C>
__device__ __forceinline__ uint32_t eq_nocase(uint32_t x) {
    constexpr uint32_t c4 = C * 0x01010101u;
    const uint32_t t = ((x & 0x5F5F5F5Fu) ^ c4) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

__device__ __forceinline__ uint32_t valid_flags(uint32_t x) {
    // G/C/g/c: (b & 0xDB) == 0x43 (see wave_kernels.hpp); A/a: 0x41; T/t: 0x54
    const uint32_t t = ((x & 0x5B5B5B5Bu) ^ 0x43434343u) + 0x7F7F7F7Fu;
    const uint32_t gc = ~(t | x | 0x7F7F7F7Fu);
    return gc | eq_nocase<0x41>(x) | eq_nocase<0x54>(x);
}

__device__ __forceinline__ uint32_t valid_mask16(const uint4 v) {
    uint32_t lo = __builtin_amdgcn_udot4(valid_flags(v.x), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(valid_flags(v.y), 0x80402010u, lo, false);
    uint32_t hi = __builtin_amdgcn_udot4(valid_flags(v.z), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(valid_flags(v.w), 0x80402010u, hi, false);
    return (lo >> 7) | (hi << 1);
}

// flips[] receives 2*pos + kind: kind 0 = a valid run starts at base pos (0-based),
// kind 1 = a valid run ends just before base pos.  Bases at or beyond len count as invalid.
__global__ __launch_bounds__(256) void gen_scan_kernel(const uint8_t *seq, uint64_t len, uint64_t n_chunks,
                                                       unsigned long long *flips, uint64_t cap,
                                                       unsigned long long *n_flips) {
    const uint64_t c = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_chunks) return;
    const uint4 v = load_once16(reinterpret_cast<const uint4 *>(seq) + c);   // a chromosome is scanned once
    uint32_t m = valid_mask16(v);
    const uint64_t b0 = c * 16u;
    if (b0 + 16u > len) m &= (1u << (uint32_t)(len - b0)) - 1u;      // tail of the sequence
    uint32_t prev = 0;                                                // validity of base b0 - 1
    if (c > 0) {
        const uint32_t pb = seq[b0 - 1u] & 0xDFu;
        prev = (pb == 0x41u || pb == 0x43u || pb == 0x47u || pb == 0x54u) ? 1u : 0u;
    }
    const uint32_t shifted = (m << 1) | prev;                         // validity of the base before each base
    uint32_t diff = (m ^ shifted) & 0xFFFFu;                          // bit j: base b0+j differs from base b0+j-1
    // (a flip at base b0+16 belongs to the next chunk)
    while (diff) {
        const uint32_t j = (uint32_t)__ffs((int)diff) - 1u;
        diff &= diff - 1u;
        const uint32_t starts = (m >> j) & 1u;                        // 1: run starts here, 0: run ended
        const unsigned long long at = atomicAdd(n_flips, 1ull);
        if (at < cap) flips[at] = 2ull * (b0 + j) + (starts ? 0ull : 1ull);
    }
    // the end of the sequence closes an open run; a partial last chunk already produced
    // that flip (its masked-off bases read as invalid), a full one has not
    if (b0 + 16u == len && (m >> 15) & 1u) {
        const unsigned long long at = atomicAdd(n_flips, 1ull);
        if (at < cap) flips[at] = 2ull * len + 1ull;
    }
}

}  // namespace

extern "C" int gams_gpu_valid_spans(gams_gpu_t *h, const uint8_t *seq, uint64_t len, int32_t fill, int32_t min_len,
                                    int32_t *span_lo, int32_t *span_hi, uint64_t cap, uint64_t *n_spans) {
    if (!h || !seq || !n_spans) return gams_fail(h, GAMS_EINVAL, "valid_spans: null argument");
    if (len == 0 || len > 0x7fffffffull) return gams_fail(h, GAMS_EINVAL, "valid_spans: length must be 1..2^31-1");
    GAMS_HIP(h, hipSetDevice(h->device));
    const uint64_t n_chunks = (len + 15) / 16;
    // device buffers from the handle's pool (one chromosome after another reuses them): the sequence, and
    // one block holding the flip counter (first 256 B) and the flip list behind it
    uint8_t *d_seq = nullptr, *d_blk = nullptr;
    size_t seq_cap = 0, blk_cap = 0;
    uint64_t fcap = 1u << 16;
    auto cleanup = [&]() {
        gams_pool_free(h, false, d_seq, seq_cap);
        gams_pool_free(h, false, d_blk, blk_cap);
        d_seq = d_blk = nullptr;
    };
#define G_HIP(call)                                                                            \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (void)hipStreamSynchronize(h->compute); /* nothing may still use the blocks */     \
            cleanup();                                                                         \
            return gams_fail(h, GAMS_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
        }                                                                                      \
    } while (0)
    G_HIP(gams_pool_alloc(h, false, n_chunks * 16 + 16, reinterpret_cast<void **>(&d_seq), &seq_cap));
    G_HIP(hipMemsetAsync(d_seq + (n_chunks - 1) * 16, 0, 32, h->compute));
    G_HIP(hipMemcpyAsync(d_seq, seq, len, hipMemcpyHostToDevice, h->compute));
    std::vector<unsigned long long> flips;
    for (int attempt = 0; attempt < 2; ++attempt) {
        G_HIP(gams_pool_alloc(h, false, 256 + fcap * sizeof(unsigned long long), reinterpret_cast<void **>(&d_blk), &blk_cap));
        unsigned long long *const d_n = reinterpret_cast<unsigned long long *>(d_blk);
        unsigned long long *const d_flips = reinterpret_cast<unsigned long long *>(d_blk + 256);
        G_HIP(hipMemsetAsync(d_n, 0, sizeof(unsigned long long), h->compute));
        G_HIP(hipEventRecord(h->k0, h->compute));
        hipLaunchKernelGGL(gen_scan_kernel, dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, h->compute, d_seq,
                           len, n_chunks, d_flips, fcap, d_n);
        G_HIP(hipGetLastError());
        G_HIP(hipEventRecord(h->k1, h->compute));
        h->k_valid = true;
        h->kq_used = 0;
        // the counter comes back through the handle's page-locked scratch word
        G_HIP(hipMemcpyAsync(h->pin_scratch, d_n, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->compute));
        G_HIP(hipStreamSynchronize(h->compute));
        const unsigned long long nf = h->pin_scratch[0];
        if (nf > fcap) {                 // more flips than the list holds: grow once and rescan
            gams_pool_free(h, false, d_blk, blk_cap);
            d_blk = nullptr;
            fcap = nf;
            continue;
        }
        flips.resize(nf);
        if (nf) G_HIP(hipMemcpy(flips.data(), d_flips, nf * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        break;
    }
#undef G_HIP
    cleanup();
    std::sort(flips.begin(), flips.end());
    if (flips.size() % 2 != 0) return gams_fail(h, GAMS_EHIP, "valid_spans: unpaired run boundary");
    // valid spans, 1-based inclusive (gen.rs:100-102)
    std::vector<std::pair<int64_t, int64_t>> spans;
    for (size_t i = 0; i < flips.size(); i += 2) {
        if ((flips[i] & 1ull) != 0 || (flips[i + 1] & 1ull) != 1)
            return gams_fail(h, GAMS_EHIP, "valid_spans: run boundaries out of order");
        spans.emplace_back((int64_t)(flips[i] >> 1) + 1, (int64_t)(flips[i + 1] >> 1));
    }
    // fill(fill - 1): holes of at most fill-1 bases are closed (gen.rs:103)
    std::vector<std::pair<int64_t, int64_t>> filled;
    for (auto &sp : spans) {
        if (!filled.empty() && sp.first - filled.back().second - 1 <= (int64_t)fill - 1)
            filled.back().second = sp.second;
        else
            filled.push_back(sp);
    }
    // excise(min): spans shorter than min are dropped (gen.rs:104)
    uint64_t n = 0;
    for (auto &sp : filled) {
        if (sp.second - sp.first + 1 < (int64_t)min_len) continue;
        if (span_lo && span_hi && n < cap) {
            span_lo[n] = (int32_t)sp.first;
            span_hi[n] = (int32_t)sp.second;
        }
        ++n;
    }
    *n_spans = n;
    return GAMS_OK;
}
