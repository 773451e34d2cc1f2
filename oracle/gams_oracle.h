/*
 * gams_oracle.h -- CPU restatement of the wang-q/gams hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain, scalar C restatement of the reference's
 * Rust algorithm, evaluated in the reference's own order (f32, left to right).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product path (gams_amd/csrc) never links or calls anything here.
 *
 * Citations are file:line under /root/reference (wang-q/gams @ 2024-10-22).
 * Off-tree crates restated from their published behaviour (no Cargo.lock in
 * the reference, caret requirements from Cargo.toml:15-41):
 *   bio ^1.5.0         seq_analysis::gc::gc_content   (pinned by I.peaks.tsv)
 *   intspan ^0.7.7     IntSpan/Range on single-span parents (pinned by window.rs tests)
 *   rust-lapper ^1.1.0 Lapper::find / Lapper::count  (interior hits pinned;
 *                      half-open query end + ctg-start point miss: PARITY UNPINNED)
 *   petgraph ^0.6.3    connected components (pinned for coverage <= 1 by I.peaks.tsv;
 *                      coverage > 1: PARITY UNPINNED)
 * Pinning: tests/test_oracle_golden.py checks this file against every golden
 * vector the reference holds for the path.
 */
#ifndef GAMS_ORACLE_H
#define GAMS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- libs/window.rs ---------------------------------------------------- */
/* sliding (window.rs:78-94): number of windows over a parent of parent_size. */
int64_t ora_sliding_count(int64_t parent_size, int32_t size, int32_t step);
/* center_resize (window.rs:96-124) for a single-span parent [ps,pe] and a
 * single-span intspan [is,ie]; outputs chromosome coordinates. */
void ora_center_resize(int32_t ps, int32_t pe, int32_t is, int32_t ie,
                       int32_t resize, int32_t *out_s, int32_t *out_e);
/* center_sw (window.rs:3-56).  Writes up to 1+2*max rows, returns the count.
 * type: 0 = M, 1 = L, 2 = R. */
int32_t ora_center_sw(int32_t ps, int32_t pe, int32_t start, int32_t end,
                      int32_t size, int32_t max, int32_t *w_start,
                      int32_t *w_end, int32_t *w_type, int32_t *w_dist);

/* ---- bio::seq_analysis::gc::gc_content (wave.rs:151, utils.rs:157) ------ */
uint32_t ora_gc_count(const uint8_t *s, size_t n);
float ora_gc_content(const uint8_t *s, size_t n);

/* ---- libs/stat.rs ------------------------------------------------------ */
float ora_mean(const float *d, size_t n);   /* stat.rs:1-6   */
float ora_stddev(const float *d, size_t n); /* stat.rs:8-14  */
/* thresholding_algo (stat.rs:16-56).  Returns 0, or -1 where the reference
 * panics (n < lag or lag == 0). */
int ora_thresholding_algo(const float *data, size_t n, size_t lag,
                          float threshold, float influence, int32_t *signals);

/* ---- libs/utils.rs ----------------------------------------------------- */
float ora_round(float x, uint32_t decimals);                      /* utils.rs:135-138 */
void ora_gc_stat(const float *gcs, size_t n, float *mean, float *stddev,
                 float *cv);                                      /* utils.rs:164-187 */
/* cache_gc_content (utils.rs:141-162) without the memo: gc of chromosome range
 * [rs,re] inside the ctg starting at chr_start, rounded to 4. */
float ora_range_gc_content(const uint8_t *seq, int32_t chr_start, int32_t rs,
                           int32_t re);
/* cache_gc_stat (utils.rs:189-213). */
void ora_range_gc_stat(const uint8_t *seq, int32_t chr_start, int32_t rs,
                       int32_t re, int32_t size, int32_t step, float *mean,
                       float *stddev, float *cv);

/* ---- cmd_gams/wave.rs -------------------------------------------------- */
/* Per-window gc counts + signals of one ctg, the reference way:
 * per-window O(size) byte fold (wave.rs:143-153), then thresholding_algo
 * (wave.rs:155).  gc_count / gc / signals hold n = ora_sliding_count() items;
 * any of the three may be NULL.  Returns n, or -1 where the reference panics. */
int64_t ora_wave_windows(const uint8_t *seq, int64_t len, int32_t size,
                         int32_t step, size_t lag, float threshold,
                         float influence, uint32_t *gc_count, float *gc,
                         int32_t *signals);
/* proc_ctg (wave.rs:121-215): TSV rows of one ctg (no header).  malloc'd,
 * free with ora_free.  NULL where the reference panics. */
char *ora_wave_proc_ctg(const char *chr_id, int32_t chr_start, int32_t chr_end,
                        const uint8_t *seq, int32_t size, int32_t step,
                        size_t lag, float threshold, float influence,
                        float coverage, int is_signal);

/* ---- cmd_gams/sw.rs + libs/data.rs:45-83 ------------------------------- */
/* proc_ctg (sw.rs:108-194) for features given as (id string, start, end). */
char *ora_sw_proc_ctg(const char *chr_id, int32_t chr_start, int32_t chr_end,
                      const uint8_t *seq, const char *const *feature_ids,
                      const int32_t *f_start, const int32_t *f_end, size_t nf,
                      int32_t size, int32_t max, int32_t resize);

/* ---- rust-lapper (utils.rs:7-36, redis.rs:236-324) --------------------- */
/* Lapper::find(qs,qe).next(): intervals sorted by (start,stop); returns the
 * index of the first with start < qe && stop > qs, or -1. */
int64_t ora_lapper_find_first(const uint32_t *starts, const uint32_t *stops,
                              size_t m, uint32_t qs, uint32_t qe);
/* Lapper::count(qs,qe): sorted_starts / sorted_stops are the two
 * independently sorted arrays Lapper keeps. */
int32_t ora_lapper_count(const uint32_t *sorted_starts,
                         const uint32_t *sorted_stops, size_t m, uint32_t qs,
                         uint32_t qe);

/* ---- cmd_gams/anno.rs:128-139 ------------------------------------------ */
/* |set[chr] & [ctg_s,ctg_e] & [rs,re]| / |[rs,re]| as f32; spans sorted,
 * disjoint, inclusive. */
float ora_anno_prop(const int32_t *span_lo, const int32_t *span_hi, size_t ns,
                    int32_t ctg_s, int32_t ctg_e, int32_t rs, int32_t re);

/* ---- cmd_gams/gen.rs:81-126 ------------------------------------------- */
/* ctg regions of one chromosome: ambiguous-base scan, fill(fill-1), excise(min), --piece
 * split (last piece absorbs the remainder).  1-based inclusive; returns the count. */
int64_t ora_gen_regions(const uint8_t *seq, int64_t len, int32_t piece, int32_t fill,
                        int32_t min_len, int32_t *out_start, int32_t *out_end, int64_t cap);

/* ---- cmd_gams/peak.rs:65-158 ------------------------------------------ */
/* TSV rows of the Peak records of one ctg (fields of data.rs:30-43), peaks in bucket order.
 * PARITY UNPINNED upstream: the reference's tests only check a stderr line (tests/cli.rs:366-382). */
char *ora_peak_rows(const char *ctg_id, const char *chr_id, int32_t chr_start, int32_t chr_end,
                    const uint8_t *seq, const int32_t *p_start, const int32_t *p_end,
                    const char *const *p_signal, size_t np);

/* ---- Rust float Display ------------------------------------------------ */
/* `{}` for f32: shortest round-trip digits, positional.  Returns length. */
int ora_fmt_f32(float v, char *out /* >= 64 bytes */);

void ora_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
