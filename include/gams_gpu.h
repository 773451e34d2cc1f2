/*
 * gams_gpu.h -- C ABI of the MI355X (gfx950) engine behind the `gams` hot path.
 *
 * This is the drop-in boundary: what a Rust host would bind with `extern "C"`
 * in place of the CPU loops inside the reference's per-ctg workers.  No torch
 * types, no C++ types: plain pointers and sizes.  All pointers are caller-owned
 * HOST memory unless a function says "device-resident"; the library owns every
 * byte of device memory, per handle.  A handle is not thread-safe; use one
 * handle per host worker thread (the reference runs one ctg per worker,
 * src/cmd_gams/wave.rs:288-299).  Every entry returns 0 on success or a
 * GAMS_E* code; gams_gpu_last_error() gives the message (the reference
 * panics instead: src/gams.rs:57, src/libs/stat.rs:30).
 *
 * Reference interfaces replaced (paths under wang-q/gams @ 2024-10-22):
 *   gams_gpu_wave*            src/cmd_gams/wave.rs:138-155  (sliding + gc_content + thresholding_algo)
 *                             = src/libs/window.rs:78-94, bio gc_content, src/libs/stat.rs:16-56
 *   gams_gpu_sw(_batch, _text) src/cmd_gams/sw.rs:141-184   (center_sw + cache_gc_content + cache_gc_stat; _text: + the rows' text, :152-190)
 *                             = src/libs/window.rs:3-56,96-124, src/libs/utils.rs:141-213
 *   gams_gpu_count            src/libs/utils.rs:24-36       (count_rg -> Lapper::count)
 *   gams_gpu_locate           src/libs/utils.rs:7-22        (find_one_idx -> Lapper::find().next())
 *   gams_gpu_cover            src/cmd_gams/anno.rs:128-139  (IntSpan intersect cardinalities)
 *   gams_gpu_valid_spans      src/cmd_gams/gen.rs:86-104    (ambiguous-base scan, fill, excise)
 *
 * Measurement and tuning entries (event stopwatch, phase stamps, guard-band and tile knobs, kernel
 * names) are NOT part of this surface: they live in gams_gpu_diag.h, which a host need not bind.
 *
 * Environment: the library neither reads nor writes the process environment.  Kernel arguments in
 * device memory (HIP_FORCE_DEV_KERNARG=1, set by the host BEFORE its first HIP call) shorten every
 * launch by the PCIe round trip of the argument fetch (12-Mb pass 9.5 -> 7.8 us); see INTEGRATION.md.
 */
#ifndef GAMS_GPU_H
#define GAMS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAMS_OK 0
#define GAMS_EINVAL 1       /* bad argument */
#define GAMS_ENODEV 2       /* no usable HIP device */
#define GAMS_ENOMEM 3       /* host or device allocation failed */
#define GAMS_EHIP 4         /* a HIP call failed (message has the HIP error) */
#define GAMS_ESHORT 5       /* a ctg has fewer windows than lag: the reference panics (stat.rs:30) */
#define GAMS_EUNSUPPORTED 6 /* outside what the kernels implement (sizes / lags beyond 65535, counts beyond 2^31) */
#define GAMS_ESTATE 7       /* call order violated (e.g. results read before run) */

typedef struct gams_gpu gams_gpu_t;
typedef struct gams_seqset gams_seqset_t;
typedef struct gams_wave_plan gams_wave_plan_t;
typedef struct gams_index gams_index_t;
typedef struct gams_spans gams_spans_t;

/* ---- handle ------------------------------------------------------------ */
int gams_gpu_create(int device, gams_gpu_t **h);
void gams_gpu_destroy(gams_gpu_t *h);
const char *gams_gpu_last_error(gams_gpu_t *h);
/* name (e.g. "gfx950"), CU count, HBM bytes of the device behind the handle */
int gams_gpu_device_info(gams_gpu_t *h, char *arch, size_t arch_len,
                         int32_t *compute_units, uint64_t *hbm_bytes);
/* block until everything queued on the handle's streams has finished */
int gams_gpu_sync(gams_gpu_t *h);
/* Freed HBM and pinned-host blocks stay cached on the handle for the next batch (at most 16
 * blocks, a quarter of the HBM, 1 GiB pinned); this returns them to the driver now.
 * cached_bytes (may be NULL) receives what was held. */
int gams_gpu_release_cached(gams_gpu_t *h, uint64_t *cached_bytes);
/* Page-locked host memory for query / result columns: gams_gpu_count, gams_gpu_locate and gams_gpu_cover
 * move their columns in chunks (copy in, search, copy out on three streams); from page-locked arrays the
 * three overlap and a call runs at the rate of the PCIe link.  Any host memory works. */
int gams_gpu_host_alloc(gams_gpu_t *h, uint64_t bytes, void **p);
/* Blocks come from (and go back to) the handle's pool of page-locked memory, so asking for the same columns batch
 * after batch does not pin them again.  Free with the handle that allocated, while it is alive (h == NULL: the
 * block is released to the system; that is also the way to free one after its handle is gone). */
void gams_gpu_host_free(gams_gpu_t *h, void *p);

/* window.rs:78-94: number of size/step windows over `len` bases (-1: bad size/step) */
int64_t gams_window_count(int64_t len, int32_t size, int32_t step);

/* ---- ctg sequences resident in HBM ------------------------------------- */
/* A seqset is the device image of a batch of `seq:{ctg}` values (gunzipped
 * bases, 1 B/base, redis.rs:142-146): one contiguous HBM buffer, every ctg
 * starting on a 256-B boundary.  lengths[i] = bases of ctg i. */
int gams_seqset_create(gams_gpu_t *h, uint32_t n_ctg, const uint32_t *lengths,
                       gams_seqset_t **s);
/* copy ctg `i` (lengths[i] bytes) host -> HBM on the handle's copy stream */
int gams_seqset_upload(gams_gpu_t *h, gams_seqset_t *s, uint32_t i,
                       const uint8_t *seq);
/* copy every ctg of the set (seqs[i] -> lengths[i] bytes; NULL allowed where lengths[i] == 0):
 * the batch form of the per-ctg GET loop (wave.rs:134-136).  Several host threads fill pinned
 * staging buffers in the device layout, one DMA per 16 MiB; returns once everything is queued. */
int gams_seqset_upload_all(gams_gpu_t *h, gams_seqset_t *s, const uint8_t *const *seqs);
/* Where the library put the ctgs: offsets[i] (n_ctg entries, may be NULL) = byte offset of ctg i in the
 * device buffer, *bytes (may be NULL) = the buffer's size in use.  A host that produces the bases itself
 * -- gunzip of the `seq:` values, redis.rs:142-161 -- writes them at these offsets of a page-locked
 * block of *bytes bytes (gams_gpu_host_alloc) and hands ranges of that image to
 * gams_seqset_upload_image: no staging copy between the inflate and the DMA. */
int gams_seqset_layout(gams_gpu_t *h, const gams_seqset_t *s, uint64_t *offsets, uint64_t *bytes);
/* DMA bytes [lo, hi) of `image` (a host image of the device buffer, laid out as gams_seqset_layout says)
 * to the same bytes of the device buffer, on the handle's copy stream; returns once queued.  The image must
 * stay untouched until the next gams_gpu_sync / a reader of the seqset has finished.  Page-locked images
 * (gams_gpu_host_alloc) go at the rate of the link; pageable ones work but stage through the driver.
 * Bytes between ctgs (alignment gaps) are never counted: they may hold anything. */
int gams_seqset_upload_image(gams_gpu_t *h, gams_seqset_t *s, const uint8_t *image, uint64_t lo, uint64_t hi);
void gams_seqset_destroy(gams_gpu_t *h, gams_seqset_t *s);

/* ---- wave (GC windows + smoothed z-score) ------------------------------- */
typedef struct {
    int32_t size;      /* --size      (wave.rs:125) */
    int32_t step;      /* --step      (wave.rs:126) */
    uint32_t lag;      /* --lag       (wave.rs:127) */
    float threshold;   /* --threshold (wave.rs:128) */
    float influence;   /* --influence (wave.rs:129) */
} gams_wave_params_t;

/* one signalled window, as wave.rs:170-187 collects them */
typedef struct {
    uint32_t ctg;      /* index into the seqset */
    uint32_t window;   /* k: bases [k*step, k*step+size) of the ctg */
    uint32_t gc_count; /* #{G,C,g,c}; gc_content = gc_count as f32 / size as f32 */
    int32_t signal;    /* +1 crest, -1 trough */
} gams_peak_t;

#define GAMS_WAVE_PEAKS 1u /* compacted (ctg, window, gc_count, signal) of signal != 0 */
#define GAMS_WAVE_DENSE 2u /* every window: gc_count u32 + signal i8 (--signal, wave.rs:158-168) */

/* Geometry + device buffers for running `params` over every ctg of `s`.
 * GAMS_ESHORT if any ctg has fewer than `lag` windows (the reference panics). */
int gams_wave_plan_create(gams_gpu_t *h, gams_seqset_t *s,
                          const gams_wave_params_t *params, uint32_t flags,
                          gams_wave_plan_t **p);
void gams_wave_plan_destroy(gams_gpu_t *h, gams_wave_plan_t *p);
/* total windows over all ctgs, and windows of ctg i */
uint64_t gams_wave_total_windows(const gams_wave_plan_t *p);
uint32_t gams_wave_ctg_windows(const gams_wave_plan_t *p, uint32_t i);
/* One pass: queue the kernels on the compute stream (asynchronous). */
int gams_wave_run(gams_gpu_t *h, gams_wave_plan_t *p);
/* n passes queued back to back (the loop a host would write, without its per-call overhead) */
int gams_wave_run_n(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t n);
/* Passes in flight.  With depth D (1..4, default 1) consecutive gams_wave_run calls of the plan
 * rotate over D HIP streams and D sets of output buffers, so that independent passes (batch
 * after batch of ctgs, wave.rs:288-299 hands them to workers the same way) overlap on the device:
 * a 12-Mb pass alone is launch-latency bound (8.2 us; 3.3 us each at depth 4).  The readers
 * (gams_wave_peaks, gams_wave_dense, gams_wave_exact_count) return the most recent run;
 * gams_wave_plan_select(age) points them at the run `age` runs earlier (age < depth).
 * Changing the depth drops the results held so far. */
int gams_wave_plan_set_depth(gams_gpu_t *h, gams_wave_plan_t *plan, uint32_t depth);
int gams_wave_plan_select(gams_gpu_t *h, gams_wave_plan_t *plan, uint32_t age);
/* Several plans in flight on one handle (batch k+1 computing while batch k is read back): a plan's
 * runs go to HIP stream (lane + way) % 4 of the handle, lane 0 by default.  Plans on different lanes
 * overlap on the device exactly like the ways of one plan; the readers wait for their own plan only.
 * Call while the plan is idle (it waits for the plan's queued runs). */
int gams_wave_plan_set_lane(gams_gpu_t *h, gams_wave_plan_t *plan, uint32_t lane);
/* Pipelined plans record an event behind every run, and gams_wave_peaks / gams_wave_dense wait
 * for that run only (so a host can keep several plans in flight on one handle: upload of batch
 * k+1 and its kernel overlap the readback and formatting of batch k).  Off by default: the
 * event costs a marker packet between back-to-back launches. */
int gams_wave_plan_set_pipelined(gams_gpu_t *h, gams_wave_plan_t *p, int enable);
/* Wait for the last run and fetch its compacted peaks, ordered by (ctg, window).
 * *peaks points into plan-owned host memory, valid until the next run. */
int gams_wave_peaks(gams_gpu_t *h, gams_wave_plan_t *p,
                    const gams_peak_t **peaks, uint64_t *n_peaks);
/* The rows of `gams wave` as TSV text, made on the device (wave.rs:157-252: crests and troughs separately,
 * overlapping windows merged into "{chr}(+):{min}-{max}", every row "...\t{gc_content}\t{signal}\n" in window
 * order, gc_content printed like Rust's `{}` of an f32) -- what the host otherwise derives from gams_wave_peaks.
 * setup: once per plan; chr[i] / chr_start[i] = chromosome name and first chromosome coordinate of ctg i
 *   (chr_id, chr_start of its Ctg record); coverage = --coverage.  GAMS_EUNSUPPORTED when that coverage links
 *   only some of the overlapping windows (> size / (size - step): merge gams_wave_peaks on the host then).
 * begin: queue the packing of the selected run's peaks, the rows and their copy to the host behind that run;
 *   returns at once, so several plans can have their rows in flight while later passes compute.
 * end: wait for this plan's rows.  *text (text_bytes bytes, no header line, no terminating NUL) and *ctg_off
 *   (n_ctg + 1 offsets: the rows of ctg i are text[ctg_off[i] .. ctg_off[i+1])) point into plan-owned
 *   page-locked memory, valid until the plan's next gams_wave_rows_begin. */
int gams_wave_rows_setup(gams_gpu_t *h, gams_wave_plan_t *p, const char *const *chr,
                         const int32_t *chr_start, float coverage);
int gams_wave_rows_begin(gams_gpu_t *h, gams_wave_plan_t *p);
int gams_wave_rows_end(gams_gpu_t *h, gams_wave_plan_t *p, const char **text, uint64_t *text_bytes,
                       const uint64_t **ctg_off);
/* `wave --signal` (wave.rs:158-168: a row for EVERY window, "{chr}:{start}-{end}\t{gc_content}\t{signal}\n") as text made
 * on the device from the dense rows of the selected run of a plan with GAMS_WAVE_DENSE.  chr[i] / chr_start[i] as in
 * gams_wave_rows_setup.  Waits for the run; *text (text_bytes bytes, no header line, no NUL) and *ctg_off (n_ctg + 1
 * offsets: the rows of ctg i are text[ctg_off[i] .. ctg_off[i+1])) point into plan-owned page-locked memory, valid
 * until the next call on this plan.  120 Mb at step 10 are 1.2e7 rows and 310 MB of text. */
int gams_wave_signal_text(gams_gpu_t *h, gams_wave_plan_t *p, const char *const *chr, const int32_t *chr_start,
                          const char **text, uint64_t *text_bytes, const uint64_t **ctg_off);
/* Wait for the last run and copy ctg i's dense rows (either pointer may be NULL). */
int gams_wave_dense(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t i,
                    uint32_t *gc_count, int8_t *signal);

/* Convenience, one ctg from host memory (what wave.rs:143-155 computes):
 * gc_count / signal receive n = gams_window_count(len,size,step) items. */
int gams_gpu_wave(gams_gpu_t *h, const uint8_t *seq, uint32_t len,
                  const gams_wave_params_t *params, uint32_t *gc_count,
                  int8_t *signal, uint32_t *n_windows);

/* ---- sw (windows around features) --------------------------------------- */
typedef struct {
    uint32_t feature;  /* index into the feature arrays */
    int32_t type;      /* 0 = M, 1 = L, 2 = R  (window.rs:13-15) */
    int32_t distance;  /* 0 for M, 1..max */
    int32_t start;     /* chromosome coordinates, 1-based inclusive */
    int32_t end;
    float gc_content;  /* round4 (utils.rs:161) */
    float gc_mean;     /* round4 (utils.rs:186) */
    float gc_stddev;
    float gc_cv;
} gams_sw_row_t;

/* Features are inclusive chromosome ranges whose middle lies inside the ctg (GAMS_EINVAL otherwise:
 * center_resize takes parent.index(mid), window.rs:98-111, undefined for a non-member); size and
 * resize >= 2.
 * rows per feature = 1 + nL + nR (window.rs:29-41); writes at most `cap` rows
 * in the reference's order (feature order, then M, L1.., R1..). ctg i of `s`
 * spans chromosome coordinates [chr_start, chr_start+len-1]. */
int gams_gpu_sw(gams_gpu_t *h, gams_seqset_t *s, uint32_t i, int32_t chr_start,
                const int32_t *feat_start, const int32_t *feat_end, uint32_t nf,
                int32_t size, int32_t max, int32_t resize, gams_sw_row_t *rows,
                uint64_t cap, uint64_t *n_rows);

/* The same over several ctgs of the seqset in ONE launch and one pair of transfers (a ctg's few hundred
 * features are a handful of workgroups: called per ctg the kernel is all launch latency).  ctg_index[k]
 * (k < n_sel) names a ctg of `s`, chr_start[k] its first chromosome coordinate; its features are
 * feat_start / feat_end [feat_off[k], feat_off[k+1]) (feat_off[0] = 0).  Rows come out in (k, feature, M,
 * L1.., R1..) order, `feature` counting from 0 inside each selected ctg; row_off (n_sel + 1 entries, may be
 * NULL) receives where the rows of selected ctg k begin.  rows == NULL or cap == 0: size query (n_rows and
 * row_off only, nothing runs on the device).  gams_gpu_sw is this call with n_sel = 1. */
int gams_gpu_sw_batch(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                      const int32_t *chr_start, const uint64_t *feat_off, const int32_t *feat_start,
                      const int32_t *feat_end, int32_t size, int32_t max, int32_t resize,
                      gams_sw_row_t *rows, uint64_t cap, uint64_t *row_off, uint64_t *n_rows);

/* The rows of gams_gpu_sw_batch as the TSV text of `gams sw` (sw.rs:152-190 and the Display of Sw, data.rs:58-83:
 * "sw:{feature id}:{serial}\t{chr}:{start}-{end}\t{M|L|R}\t{distance}\t{gc_content}\t{gc_mean}\t{gc_stddev}\t{gc_cv}\t\n",
 * the last field -- rg_count -- empty), formatted on the device.  chr[k] = chromosome name of selected ctg k,
 * feat_id[f] = id of feature f ("feature:{ctg}:{serial}", feature.rs:81-83), both NUL-terminated.  *text
 * (text_bytes bytes, no header, no NUL) and *ctg_off (n_sel + 1 offsets: the rows of selected ctg k are
 * text[ctg_off[k] .. ctg_off[k+1])) point into page-locked memory owned by the handle, valid until the handle's next
 * gams_gpu_sw_text.  GAMS_EUNSUPPORTED if a statistic is 1000 or more or a coordinate negative (the formatter
 * prints the four round4 values as m / 10^4 with the trailing zeros dropped, which is what Rust's `{}` prints for them
 * below 1000): format the rows of gams_gpu_sw_batch on the host then. */
int gams_gpu_sw_text(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                     const char *const *chr, const int32_t *chr_start, const uint64_t *feat_off,
                     const int32_t *feat_start, const int32_t *feat_end, const char *const *feat_id,
                     int32_t size, int32_t max, int32_t resize, const char **text, uint64_t *text_bytes,
                     const uint64_t **ctg_off, uint64_t *n_rows);

/* gc_content (round4, utils.rs:141-162) of n chromosome ranges inside ctg i: what `gams peak`
 * asks per merged peak (peak.rs:79; second "next" row of SURVEY section 8f). */
int gams_gpu_range_gc(gams_gpu_t *h, gams_seqset_t *s, uint32_t i, int32_t chr_start,
                      const int32_t *range_start, const int32_t *range_end, uint32_t n,
                      float *gc);
/* The same for the ranges of several ctgs of the seqset in one launch (ctg_index / chr_start / range_off as in
 * gams_gpu_sw_batch): gc[q] for range q of the concatenated list. */
int gams_gpu_range_gc_batch(gams_gpu_t *h, gams_seqset_t *s, uint32_t n_sel, const uint32_t *ctg_index,
                            const int32_t *chr_start, const uint64_t *range_off, const int32_t *range_start,
                            const int32_t *range_end, float *gc);

/* ---- sorted-interval index (replaces rust_lapper `idx:`) ---------------- */
/* m intervals [start, stop) (stop = end+1, redis.rs:245-248,291-294) grouped
 * in `n_groups` groups (one per ctg for idx:rg:, one per chr for idx:ctg:);
 * group g owns intervals [group_off[g], group_off[g+1]).  Any order inside a
 * group: the library sorts. */
int gams_index_create(gams_gpu_t *h, uint32_t n_groups, const uint64_t *group_off,
                      const uint32_t *starts, const uint32_t *stops,
                      gams_index_t **ix);
void gams_index_destroy(gams_gpu_t *h, gams_index_t *ix);
/* Lapper::count(qs,qe) per query against its group (utils.rs:35).
 * group[q] >= n_groups (e.g. UINT32_MAX) -> 0 (utils.rs:29-32). */
int gams_gpu_count(gams_gpu_t *h, gams_index_t *ix, const uint32_t *group,
                   const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                   int32_t *count);
/* Lapper::find(qs,qe).next() per query (utils.rs:16): index (into the
 * caller's original interval order) of the first hit in (start,stop) order,
 * or -1. */
int gams_gpu_locate(gams_gpu_t *h, gams_index_t *ix, const uint32_t *group,
                    const uint32_t *qs, const uint32_t *qe, uint64_t nq,
                    int64_t *hit);

/* ---- runlist coverage (anno) -------------------------------------------- */
/* Sorted, disjoint, inclusive spans per chr group (intspan runlists). */
int gams_spans_create(gams_gpu_t *h, uint32_t n_groups, const uint64_t *group_off,
                      const int32_t *lo, const int32_t *hi, gams_spans_t **sp);
void gams_spans_destroy(gams_gpu_t *h, gams_spans_t *sp);
/* anno.rs:128-139: prop = |set[group] & [clip_lo,clip_hi] & [qs,qe]| as f32 /
 * |[qs,qe]| as f32; group >= n_groups -> 0.0 */
int gams_gpu_cover(gams_gpu_t *h, gams_spans_t *sp, const uint32_t *group,
                   const int32_t *clip_lo, const int32_t *clip_hi,
                   const int32_t *qs, const int32_t *qe, uint64_t nq,
                   float *prop);

/* ---- gen: valid regions of a chromosome (first "next" row of SURVEY section 8f) ---- */
/* gen.rs:86-104: bases other than A C G T a c g t are ambiguous; the valid set is their
 * complement after fill(fill-1) and excise(min_len).  Writes up to `cap` spans (1-based,
 * inclusive, ascending); *n_spans is the full count (call with cap = 0 to size). */
int gams_gpu_valid_spans(gams_gpu_t *h, const uint8_t *seq, uint64_t len, int32_t fill,
                         int32_t min_len, int32_t *span_lo, int32_t *span_hi, uint64_t cap,
                         uint64_t *n_spans);

#ifdef __cplusplus
}
#endif
#endif
