/*
 * gams_ref.h -- CPU twins of the C ABI's compute entries (TEST INFRASTRUCTURE ONLY, like gams_oracle.h).
 *
 * "every entry has a CPU twin (gams_ref_*, same signature) = the oracle".  Each function
 * here takes the arguments of its gams_gpu_* / gams_wave_* counterpart in include/gams_gpu.h, with the device
 * objects replaced by what they stand for (a handle disappears; a seqset + ctg index becomes the ctg's bases; an
 * index / span object becomes the arrays it was created from), and computes the answer with the oracle's
 * restatement of the reference (gams_oracle.c, which every twin cites through the ora_* function it calls).
 * The parity tests call twin and ABI entry with the same arrays and compare the outputs.
 * Return codes are the ABI's (GAMS_OK, GAMS_EINVAL, GAMS_ESHORT ...).  Nothing under gams_amd/ links this.
 */
#ifndef GAMS_REF_H
#define GAMS_REF_H

#include "../include/gams_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gams_gpu_wave: per-window gc counts + signals of one ctg (wave.rs:143-155) */
int gams_ref_wave(const uint8_t *seq, uint32_t len, const gams_wave_params_t *params, uint32_t *gc_count,
                  int8_t *signal, uint32_t *n_windows);
/* gams_wave_peaks over a batch: the compacted (ctg, window, gc_count, signal) of signal != 0, ordered by
 * (ctg, window); writes at most cap records, *n_peaks = how many there are */
int gams_ref_wave_peaks(uint32_t n_ctg, const uint8_t *const *seqs, const uint32_t *lengths,
                        const gams_wave_params_t *params, gams_peak_t *peaks, uint64_t cap, uint64_t *n_peaks);
/* gams_wave_rows_*: the TSV rows of one ctg (wave.rs:157-252); malloc'd, free with gams_ref_free */
int gams_ref_wave_rows(const char *chr, int32_t chr_start, const uint8_t *seq, uint32_t len,
                       const gams_wave_params_t *params, float coverage, char **text, uint64_t *text_bytes);
/* gams_wave_signal_text: the `--signal` rows of one ctg, a row per window (wave.rs:158-168); malloc'd */
int gams_ref_wave_signal_text(const char *chr, int32_t chr_start, const uint8_t *seq, uint32_t len,
                              const gams_wave_params_t *params, char **text, uint64_t *text_bytes);
/* gams_gpu_sw_text for one ctg: the rows as `Sw`'s Display prints them (sw.rs:152-190, data.rs:58-83); malloc'd */
int gams_ref_sw_text(const char *chr, const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *feat_start,
                     const int32_t *feat_end, const char *const *feat_id, uint32_t nf, int32_t size, int32_t max,
                     int32_t resize, char **text, uint64_t *text_bytes);
/* gams_gpu_sw for one ctg whose first base sits at chr_start (sw.rs:141-184) */
int gams_ref_sw(const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *feat_start,
                const int32_t *feat_end, uint32_t nf, int32_t size, int32_t max, int32_t resize,
                gams_sw_row_t *rows, uint64_t cap, uint64_t *n_rows);
/* gams_gpu_range_gc (utils.rs:141-162) */
int gams_ref_range_gc(const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *range_start,
                      const int32_t *range_end, uint32_t n, float *gc);
/* gams_index_create + gams_gpu_count / gams_gpu_locate (redis.rs:236-324, utils.rs:7-36) */
int gams_ref_count(uint32_t n_groups, const uint64_t *group_off, const uint32_t *starts, const uint32_t *stops,
                   const uint32_t *group, const uint32_t *qs, const uint32_t *qe, uint64_t nq, int32_t *count);
int gams_ref_locate(uint32_t n_groups, const uint64_t *group_off, const uint32_t *starts, const uint32_t *stops,
                    const uint32_t *group, const uint32_t *qs, const uint32_t *qe, uint64_t nq, int64_t *hit);
/* gams_spans_create + gams_gpu_cover (anno.rs:128-139) */
int gams_ref_cover(uint32_t n_groups, const uint64_t *group_off, const int32_t *lo, const int32_t *hi,
                   const uint32_t *group, const int32_t *clip_lo, const int32_t *clip_hi, const int32_t *qs,
                   const int32_t *qe, uint64_t nq, float *prop);
/* gams_gpu_valid_spans (gen.rs:86-104) */
int gams_ref_valid_spans(const uint8_t *seq, uint64_t len, int32_t fill, int32_t min_len, int32_t *span_lo,
                         int32_t *span_hi, uint64_t cap, uint64_t *n_spans);
void gams_ref_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
