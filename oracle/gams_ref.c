/* gams_ref.c -- see gams_ref.h.  TEST INFRASTRUCTURE ONLY: thin adapters from the C ABI's argument lists to the
 * oracle's restatement of the reference (gams_oracle.c). */
#include "gams_ref.h"

#include <stdlib.h>
#include <string.h>

#include "gams_oracle.h"

int gams_ref_wave(const uint8_t *seq, uint32_t len, const gams_wave_params_t *p, uint32_t *gc_count, int8_t *signal,
                  uint32_t *n_windows) {
    if (!seq || !p || !n_windows) return GAMS_EINVAL;
    if (p->size <= 0 || p->step <= 0) return GAMS_EINVAL;
    const int64_t n = ora_sliding_count(len, p->size, p->step);
    int32_t *sig = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    if (!sig) return GAMS_ENOMEM;
    const int64_t got = ora_wave_windows(seq, len, p->size, p->step, p->lag, p->threshold, p->influence, gc_count,
                                         NULL, sig);               /* wave.rs:143-155 */
    if (got < 0) {
        free(sig);
        return GAMS_ESHORT;                                        /* the reference panics (stat.rs:30) */
    }
    if (signal)
        for (int64_t i = 0; i < got; ++i) signal[i] = (int8_t)sig[i];
    free(sig);
    *n_windows = (uint32_t)got;
    return GAMS_OK;
}

int gams_ref_wave_peaks(uint32_t n_ctg, const uint8_t *const *seqs, const uint32_t *lengths,
                        const gams_wave_params_t *p, gams_peak_t *peaks, uint64_t cap, uint64_t *n_peaks) {
    if (!p || !n_peaks || (n_ctg && (!seqs || !lengths))) return GAMS_EINVAL;
    uint64_t at = 0;
    for (uint32_t c = 0; c < n_ctg; ++c) {
        const int64_t n = ora_sliding_count(lengths[c], p->size, p->step);
        uint32_t *cnt = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
        int8_t *sig = (int8_t *)malloc((size_t)(n > 0 ? n : 1));
        uint32_t nw = 0;
        const int rc = (cnt && sig) ? gams_ref_wave(seqs[c], lengths[c], p, cnt, sig, &nw) : GAMS_ENOMEM;
        if (rc == GAMS_OK)
            for (uint32_t i = 0; i < nw; ++i)
                if (sig[i]) {                                      /* wave.rs:170-187 */
                    if (peaks && at < cap) {
                        peaks[at].ctg = c;
                        peaks[at].window = i;
                        peaks[at].gc_count = cnt[i];
                        peaks[at].signal = sig[i];
                    }
                    ++at;
                }
        free(cnt);
        free(sig);
        if (rc != GAMS_OK) return rc;
    }
    *n_peaks = at;
    return GAMS_OK;
}

int gams_ref_wave_rows(const char *chr, int32_t chr_start, const uint8_t *seq, uint32_t len,
                       const gams_wave_params_t *p, float coverage, char **text, uint64_t *text_bytes) {
    if (!chr || !seq || !p || !text || !text_bytes) return GAMS_EINVAL;
    char *t = ora_wave_proc_ctg(chr, chr_start, chr_start + (int32_t)len - 1, seq, p->size, p->step, p->lag,
                                p->threshold, p->influence, coverage, 0);     /* wave.rs:121-215 */
    if (!t) return GAMS_ESHORT;
    *text = t;
    *text_bytes = strlen(t);
    return GAMS_OK;
}

int gams_ref_wave_signal_text(const char *chr, int32_t chr_start, const uint8_t *seq, uint32_t len,
                              const gams_wave_params_t *p, char **text, uint64_t *text_bytes) {
    if (!chr || !seq || !p || !text || !text_bytes) return GAMS_EINVAL;
    char *t = ora_wave_proc_ctg(chr, chr_start, chr_start + (int32_t)len - 1, seq, p->size, p->step, p->lag,
                                p->threshold, p->influence, 0.2f, 1);         /* wave.rs:158-168 */
    if (!t) return GAMS_ESHORT;
    *text = t;
    *text_bytes = strlen(t);
    return GAMS_OK;
}

int gams_ref_sw_text(const char *chr, const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *fs,
                     const int32_t *fe, const char *const *fid, uint32_t nf, int32_t size, int32_t max, int32_t resize,
                     char **text, uint64_t *text_bytes) {
    if (!chr || !seq || !text || !text_bytes || (nf && (!fs || !fe || !fid))) return GAMS_EINVAL;
    char *t = ora_sw_proc_ctg(chr, chr_start, chr_start + (int32_t)len - 1, seq, fid, fs, fe, nf, size, max, resize);  /* sw.rs:108-194 */
    if (!t) return GAMS_ENOMEM;
    *text = t;
    *text_bytes = strlen(t);
    return GAMS_OK;
}

int gams_ref_sw(const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *fs, const int32_t *fe, uint32_t nf,
                int32_t size, int32_t max, int32_t resize, gams_sw_row_t *rows, uint64_t cap, uint64_t *n_rows) {
    if (!seq || !n_rows || (nf && (!fs || !fe))) return GAMS_EINVAL;
    const int32_t chr_end = chr_start + (int32_t)len - 1;
    const size_t wcap = (size_t)(1 + 2 * (max > 0 ? max : 0));
    int32_t *ws = (int32_t *)malloc(sizeof(int32_t) * wcap * 4), *we = ws + wcap, *wt = we + wcap, *wd = wt + wcap;
    if (!ws) return GAMS_ENOMEM;
    uint64_t at = 0;
    for (uint32_t f = 0; f < nf; ++f) {                                             /* sw.rs:141 */
        const int32_t nw = ora_center_sw(chr_start, chr_end, fs[f], fe[f], size, max, ws, we, wt, wd);   /* :150 */
        for (int32_t k = 0; k < nw; ++k, ++at) {
            if (!rows || at >= cap) continue;
            gams_sw_row_t *r = rows + at;
            r->feature = f;
            r->type = wt[k];
            r->distance = wd[k];
            r->start = ws[k];
            r->end = we[k];
            r->gc_content = ora_range_gc_content(seq, chr_start, ws[k], we[k]);     /* :168 */
            int32_t rs, re;
            ora_center_resize(chr_start, chr_end, ws[k], we[k], resize, &rs, &re);  /* :175 */
            ora_range_gc_stat(seq, chr_start, rs, re, size, size, &r->gc_mean, &r->gc_stddev, &r->gc_cv);   /* :177-178 */
        }
    }
    free(ws);
    *n_rows = at;
    return GAMS_OK;
}

int gams_ref_range_gc(const uint8_t *seq, uint32_t len, int32_t chr_start, const int32_t *rs, const int32_t *re,
                      uint32_t n, float *gc) {
    if (!seq || (n && (!rs || !re || !gc))) return GAMS_EINVAL;
    for (uint32_t q = 0; q < n; ++q) {
        if (rs[q] < chr_start || re[q] > chr_start + (int32_t)len - 1 || re[q] < rs[q]) return GAMS_EINVAL;
        gc[q] = ora_range_gc_content(seq, chr_start, rs[q], re[q]);                 /* utils.rs:141-162 */
    }
    return GAMS_OK;
}

/* Lapper::new of one group: intervals.sort() by (start, stop) keeping the caller's order on ties, the starts and the
 * stops sorted on their own (redis.rs:253,299; rust-lapper lib.rs) */
typedef struct { uint32_t start, stop; uint64_t orig; } ref_iv;
static int cmp_iv(const void *a, const void *b) {
    const ref_iv *x = (const ref_iv *)a, *y = (const ref_iv *)b;
    if (x->start != y->start) return x->start < y->start ? -1 : 1;
    if (x->stop != y->stop) return x->stop < y->stop ? -1 : 1;
    return x->orig < y->orig ? -1 : (x->orig > y->orig ? 1 : 0);
}
static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
typedef struct { ref_iv *iv; uint32_t *ss, *ts, *ls, *lt; size_t n; } ref_group;
static int build_group(ref_group *g, uint64_t lo, uint64_t hi, const uint32_t *starts, const uint32_t *stops) {
    g->n = (size_t)(hi - lo);
    const size_t n1 = g->n ? g->n : 1;
    g->iv = (ref_iv *)malloc(sizeof(ref_iv) * n1);
    g->ss = (uint32_t *)malloc(sizeof(uint32_t) * n1 * 4);
    if (!g->iv || !g->ss) return -1;
    g->ts = g->ss + n1;
    g->ls = g->ts + n1;
    g->lt = g->ls + n1;
    for (size_t i = 0; i < g->n; ++i) {
        g->iv[i].start = starts[lo + i];
        g->iv[i].stop = stops[lo + i];
        g->iv[i].orig = lo + i;
        g->ss[i] = starts[lo + i];
        g->ts[i] = stops[lo + i];
    }
    qsort(g->iv, g->n, sizeof(ref_iv), cmp_iv);
    qsort(g->ss, g->n, sizeof(uint32_t), cmp_u32);
    qsort(g->ts, g->n, sizeof(uint32_t), cmp_u32);
    for (size_t i = 0; i < g->n; ++i) {
        g->ls[i] = g->iv[i].start;
        g->lt[i] = g->iv[i].stop;
    }
    return 0;
}
static void free_groups(ref_group *g, uint32_t n) {
    if (!g) return;
    for (uint32_t i = 0; i < n; ++i) {
        free(g[i].iv);
        free(g[i].ss);
    }
    free(g);
}
static ref_group *build_groups(uint32_t n_groups, const uint64_t *off, const uint32_t *starts, const uint32_t *stops) {
    ref_group *g = (ref_group *)calloc(n_groups ? n_groups : 1, sizeof(ref_group));
    if (!g) return NULL;
    for (uint32_t i = 0; i < n_groups; ++i)
        if (build_group(g + i, off[i], off[i + 1], starts, stops) != 0) {
            free_groups(g, n_groups);
            return NULL;
        }
    return g;
}

int gams_ref_count(uint32_t n_groups, const uint64_t *off, const uint32_t *starts, const uint32_t *stops,
                   const uint32_t *group, const uint32_t *qs, const uint32_t *qe, uint64_t nq, int32_t *count) {
    if (!off || (nq && (!group || !qs || !qe || !count))) return GAMS_EINVAL;
    ref_group *g = build_groups(n_groups, off, starts, stops);
    if (!g) return GAMS_ENOMEM;
    for (uint64_t q = 0; q < nq; ++q)                                 /* utils.rs:24-36: unknown ctg -> 0 */
        count[q] = group[q] < n_groups ? ora_lapper_count(g[group[q]].ss, g[group[q]].ts, g[group[q]].n, qs[q], qe[q]) : 0;
    free_groups(g, n_groups);
    return GAMS_OK;
}

int gams_ref_locate(uint32_t n_groups, const uint64_t *off, const uint32_t *starts, const uint32_t *stops,
                    const uint32_t *group, const uint32_t *qs, const uint32_t *qe, uint64_t nq, int64_t *hit) {
    if (!off || (nq && (!group || !qs || !qe || !hit))) return GAMS_EINVAL;
    ref_group *g = build_groups(n_groups, off, starts, stops);
    if (!g) return GAMS_ENOMEM;
    for (uint64_t q = 0; q < nq; ++q) {                               /* utils.rs:7-22 */
        hit[q] = -1;
        if (group[q] >= n_groups) continue;
        const ref_group *gg = g + group[q];
        const int64_t k = ora_lapper_find_first(gg->ls, gg->lt, gg->n, qs[q], qe[q]);
        if (k >= 0) hit[q] = (int64_t)gg->iv[k].orig;                 /* index in the caller's order */
    }
    free_groups(g, n_groups);
    return GAMS_OK;
}

int gams_ref_cover(uint32_t n_groups, const uint64_t *off, const int32_t *lo, const int32_t *hi, const uint32_t *group,
                   const int32_t *clip_lo, const int32_t *clip_hi, const int32_t *qs, const int32_t *qe, uint64_t nq,
                   float *prop) {
    if (!off || (nq && (!group || !clip_lo || !clip_hi || !qs || !qe || !prop))) return GAMS_EINVAL;
    for (uint64_t q = 0; q < nq; ++q) {                               /* anno.rs:128-139: chr not in the set -> 0 */
        if (group[q] >= n_groups) {
            prop[q] = 0.0f;
            continue;
        }
        const uint64_t a = off[group[q]], b = off[group[q] + 1];
        prop[q] = ora_anno_prop(lo + a, hi + a, (size_t)(b - a), clip_lo[q], clip_hi[q], qs[q], qe[q]);
    }
    return GAMS_OK;
}

int gams_ref_valid_spans(const uint8_t *seq, uint64_t len, int32_t fill, int32_t min_len, int32_t *span_lo,
                         int32_t *span_hi, uint64_t cap, uint64_t *n_spans) {
    if (!seq || !n_spans || len == 0 || len > 0x7fffffffull) return GAMS_EINVAL;
    /* gen.rs:86-104 = the regions of gen.rs:81-126 before the --piece split: a piece larger than any chromosome */
    const int64_t n = ora_gen_regions(seq, (int64_t)len, 0x7fffffff, fill, min_len, span_lo, span_hi, (int64_t)cap);
    if (n < 0) return GAMS_EINVAL;
    *n_spans = (uint64_t)n;
    return GAMS_OK;
}

void gams_ref_free(void *p) { ora_free(p); }
