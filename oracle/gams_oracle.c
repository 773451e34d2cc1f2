/*
 * gams_oracle.c -- CPU restatement of the wang-q/gams hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see gams_oracle.h).  Scalar C, f32 arithmetic in
 * the reference's evaluation order; build with -ffp-contract=off so that no
 * multiply-add is fused (Rust does not contract).  Every function cites the
 * reference lines it follows (paths under /root/reference).
 */
#include "gams_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* growable string                                                          */
/* ------------------------------------------------------------------------ */
typedef struct {
    char *p;
    size_t len, cap;
} sbuf;

static void sb_reserve(sbuf *b, size_t extra) {
    if (b->len + extra + 1 > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 4096;
        while (nc < b->len + extra + 1) nc *= 2;
        b->p = (char *)realloc(b->p, nc);
        b->cap = nc;
    }
}
static void sb_puts(sbuf *b, const char *s) {
    size_t n = strlen(s);
    sb_reserve(b, n);
    memcpy(b->p + b->len, s, n);
    b->len += n;
    b->p[b->len] = 0;
}
static void sb_puti(sbuf *b, long long v) {
    char t[32];
    snprintf(t, sizeof t, "%lld", v);
    sb_puts(b, t);
}
/* IntSpan::runlist of a single span: "a-b", or "a" when a == b */
static void sb_runlist(sbuf *b, long long s, long long e) {
    sb_puti(b, s);
    if (e != s) {
        sb_puts(b, "-");
        sb_puti(b, e);
    }
}

void ora_free(void *p) { free(p); }

/* ------------------------------------------------------------------------ */
/* Rust `{}` for f32: shortest digits that round-trip, positional notation. */
/* ------------------------------------------------------------------------ */
int ora_fmt_f32(float v, char *out) {
    if (isnan(v)) return sprintf(out, "NaN");
    if (isinf(v)) return sprintf(out, v < 0 ? "-inf" : "inf");
    if (v == 0.0f) return sprintf(out, signbit(v) ? "-0" : "0");
    char e[64];
    int prec;
    for (prec = 1; prec <= 9; ++prec) {
        snprintf(e, sizeof e, "%.*e", prec - 1, (double)v);
        if (strtof(e, NULL) == v) break;
    }
    /* e = [-]d[.ddd]e[+-]XX */
    char digits[16];
    int nd = 0, neg = 0;
    const char *c = e;
    if (*c == '-') {
        neg = 1;
        ++c;
    }
    for (; *c && *c != 'e'; ++c)
        if (*c >= '0' && *c <= '9') digits[nd++] = *c;
    int ex = atoi(c + 1);
    while (nd > 1 && digits[nd - 1] == '0') --nd; /* cannot happen for shortest, be safe */
    char *o = out;
    if (neg) *o++ = '-';
    if (ex >= 0) {
        /* integer part has ex+1 digits */
        for (int i = 0; i <= ex; ++i) *o++ = i < nd ? digits[i] : '0';
        if (nd > ex + 1) {
            *o++ = '.';
            for (int i = ex + 1; i < nd; ++i) *o++ = digits[i];
        }
    } else {
        *o++ = '0';
        *o++ = '.';
        for (int i = 0; i < -ex - 1; ++i) *o++ = '0';
        for (int i = 0; i < nd; ++i) *o++ = digits[i];
    }
    *o = 0;
    return (int)(o - out);
}

static void sb_putf(sbuf *b, float v) {
    char t[64];
    ora_fmt_f32(v, t);
    sb_puts(b, t);
}

/* ------------------------------------------------------------------------ */
/* libs/window.rs                                                           */
/* ------------------------------------------------------------------------ */

/* window.rs:78-94: start = 1; loop { end = start+size-1; if end > parent.size
 * break; push; start += step } */
int64_t ora_sliding_count(int64_t parent_size, int32_t size, int32_t step) {
    int64_t n = 0;
    int64_t start = 1;
    if (step <= 0 || size <= 0) return -1; /* reference would loop forever / slice oddly */
    for (;;) {
        int64_t end = start + size - 1;
        if (end > parent_size) break;
        start += step;
        ++n;
    }
    return n;
}

/* window.rs:96-124.  Single-span parent [ps,pe]: index(x) = x-ps+1,
 * slice(a,b) = [ps+a-1, ps+b-1]; single-span intspan [is,ie]: at(i) = is+i-1. */
void ora_center_resize(int32_t ps, int32_t pe, int32_t is, int32_t ie,
                       int32_t resize, int32_t *out_s, int32_t *out_e) {
    int32_t psize = pe - ps + 1;
    int32_t isize = ie - is + 1;
    int32_t half_size = isize / 2;                                   /* :98  */
    int32_t mid_left = half_size == 0 ? is : is + half_size - 1;     /* :99-103  */
    int32_t mid_right = half_size == 0 ? is : is + half_size;        /* :104-108 */
    int32_t mid_left_idx = mid_left - ps + 1;                        /* :109 */
    int32_t mid_right_idx = mid_right - ps + 1;                      /* :110 */
    int32_t half_resize = resize / 2;                                /* :113 */
    int32_t left_idx = mid_left_idx - half_resize + 1;               /* :114 */
    if (left_idx < 1) left_idx = 1;
    int32_t right_idx = mid_right_idx + half_resize - 1;             /* :118 */
    if (right_idx > psize) right_idx = psize;
    *out_s = ps + left_idx - 1;                                      /* :123 */
    *out_e = ps + right_idx - 1;
}

/* window.rs:3-56 */
int32_t ora_center_sw(int32_t ps, int32_t pe, int32_t start, int32_t end,
                      int32_t size, int32_t max, int32_t *w_start,
                      int32_t *w_end, int32_t *w_type, int32_t *w_dist) {
    int32_t psize = pe - ps + 1;
    int32_t n = 0;
    int32_t m_s, m_e;
    ora_center_resize(ps, pe, start, end, size, &m_s, &m_e);         /* :12 */
    w_start[n] = m_s;
    w_end[n] = m_e;
    w_type[n] = 0;
    w_dist[n] = 0;
    ++n;
    for (int t = 1; t <= 2; ++t) { /* ["L","R"] :15 */
        int32_t sw_start, sw_end;
        if (t == 2) {
            sw_start = (m_e - ps + 1) + 1;                           /* :21 */
            sw_end = sw_start + size - 1;
        } else {
            sw_end = (m_s - ps + 1) - 1;                             /* :24 */
            sw_start = sw_end - size + 1;
        }
        for (int32_t d = 1; d <= max; ++d) {                         /* :29 */
            if (sw_start < 1) break;
            if (sw_end > psize) break;
            int32_t s = ps + sw_start - 1, e = ps + sw_end - 1;      /* slice :37 */
            if (e - s + 1 < size) break;                             /* :39 */
            w_start[n] = s;
            w_end[n] = e;
            w_type[n] = t;
            w_dist[n] = d;
            ++n;
            if (t == 2) {
                sw_start = sw_end + 1;
                sw_end = sw_start + size - 1;
            } else {
                sw_end = sw_start - 1;
                sw_start = sw_end - size + 1;
            }
        }
    }
    return n;
}

/* ------------------------------------------------------------------------ */
/* bio::seq_analysis::gc::gc_content: count of G,C,g,c over the length.     */
/* ------------------------------------------------------------------------ */
uint32_t ora_gc_count(const uint8_t *s, size_t n) {
    uint32_t c = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t b = s[i];
        c += (b == 'G' || b == 'C' || b == 'g' || b == 'c');
    }
    return c;
}

float ora_gc_content(const uint8_t *s, size_t n) {
    return (float)ora_gc_count(s, n) / (float)n;
}

/* ------------------------------------------------------------------------ */
/* libs/stat.rs                                                             */
/* ------------------------------------------------------------------------ */
float ora_mean(const float *d, size_t n) { /* stat.rs:1-6 */
    float len = (float)n;
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) sum = sum + d[i];
    return sum / len;
}

float ora_stddev(const float *d, size_t n) { /* stat.rs:8-14 */
    float len = (float)n;
    float mean = ora_mean(d, n);
    float sq = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float t = (d[i] - mean) * (d[i] - mean);
        sq = sq + t;
    }
    return sqrtf(sq / (len - 1.0f));
}

int ora_thresholding_algo(const float *data, size_t n, size_t lag,
                          float threshold, float influence, int32_t *signals) {
    if (lag == 0 || n < lag) return -1; /* stat.rs:30 slice panic */
    float *filtered = (float *)malloc(sizeof(float) * (n ? n : 1));
    float *avg = (float *)calloc(n ? n : 1, sizeof(float));
    float *sd = (float *)calloc(n ? n : 1, sizeof(float));
    memcpy(filtered, data, sizeof(float) * n);                       /* :21 */
    for (size_t i = 0; i < n; ++i) signals[i] = 0;                   /* :18 */
    avg[lag - 1] = ora_mean(data, lag);                              /* :30 */
    sd[lag - 1] = ora_stddev(data, lag);                             /* :31 */
    for (size_t i = lag; i < n; ++i) {                               /* :34 */
        if (fabsf(data[i] - avg[i - 1]) > threshold * sd[i - 1]) {   /* :36 */
            signals[i] = data[i] > avg[i - 1] ? 1 : -1;              /* :38 */
            filtered[i] = influence * data[i] +
                          (1.0f - influence) * filtered[i - 1];      /* :42 */
        } else {
            signals[i] = 0;
            filtered[i] = data[i];                                   /* :47 */
        }
        avg[i] = ora_mean(filtered + (i - lag), lag);                /* :51 */
        sd[i] = ora_stddev(filtered + (i - lag), lag);               /* :52 */
    }
    free(filtered);
    free(avg);
    free(sd);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* libs/utils.rs                                                            */
/* ------------------------------------------------------------------------ */
float ora_round(float x, uint32_t decimals) { /* utils.rs:135-138 */
    int32_t p = 1;
    for (uint32_t i = 0; i < decimals; ++i) p *= 10;
    float y = (float)p;
    return roundf(x * y) / y; /* f32::round = half away from zero = roundf */
}

void ora_gc_stat(const float *gcs, size_t n, float *mean, float *stddev,
                 float *cv) { /* utils.rs:164-187 */
    float m = ora_mean(gcs, n);
    float s = ora_stddev(gcs, n);
    float c;
    if (m == 0.0f || m == 1.0f)
        c = 0.0f;
    else if (m <= 0.5f)
        c = s / m;
    else
        c = s / (1.0f - m);
    *mean = ora_round(m, 4);
    *stddev = ora_round(s, 4);
    *cv = ora_round(c, 4);
}

float ora_range_gc_content(const uint8_t *seq, int32_t chr_start, int32_t rs,
                           int32_t re) { /* utils.rs:141-162 */
    int64_t from = (int64_t)rs - chr_start + 1; /* parent.index */
    int64_t to = (int64_t)re - chr_start + 1;
    float gc = ora_gc_content(seq + (from - 1), (size_t)(to - from + 1));
    return ora_round(gc, 4);
}

void ora_range_gc_stat(const uint8_t *seq, int32_t chr_start, int32_t rs,
                       int32_t re, int32_t size, int32_t step, float *mean,
                       float *stddev, float *cv) { /* utils.rs:189-213 */
    int64_t n = ora_sliding_count((int64_t)re - rs + 1, size, step);
    if (n < 0) n = 0;
    float *gcs = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    for (int64_t k = 0; k < n; ++k) {
        int32_t ws = rs + (int32_t)(k * step);
        int32_t we = ws + size - 1;
        gcs[k] = ora_range_gc_content(seq, chr_start, ws, we);
    }
    ora_gc_stat(gcs, (size_t)n, mean, stddev, cv);
    free(gcs);
}

/* ------------------------------------------------------------------------ */
/* cmd_gams/wave.rs                                                         */
/* ------------------------------------------------------------------------ */
int64_t ora_wave_windows(const uint8_t *seq, int64_t len, int32_t size,
                         int32_t step, size_t lag, float threshold,
                         float influence, uint32_t *gc_count, float *gc,
                         int32_t *signals) {
    int64_t n = ora_sliding_count(len, size, step);                  /* wave.rs:139 */
    if (n < 0) return -1;
    float *gcs = gc ? gc : (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    for (int64_t k = 0; k < n; ++k) {                                /* wave.rs:144-153 */
        int64_t from = 1 + k * step; /* 1-based ctg index of window.min() */
        uint32_t c = ora_gc_count(seq + (from - 1), (size_t)size);
        if (gc_count) gc_count[k] = c;
        gcs[k] = (float)c / (float)size;
    }
    int rc = 0;
    if (signals) {
        rc = ora_thresholding_algo(gcs, (size_t)n, lag, threshold, influence,
                                   signals);                         /* wave.rs:155 */
    } else if (lag == 0 || (size_t)n < lag) {
        rc = -1;
    }
    if (!gc) free(gcs);
    return rc ? -1 : n;
}

/* union-find for the connected components of merge_ints (wave.rs:217-252) */
static size_t uf_find(size_t *p, size_t x) {
    while (p[x] != x) {
        p[x] = p[p[x]];
        x = p[x];
    }
    return x;
}

/* merge_ints for one sign.  idx[0..p) = window indices (ascending) of the
 * peaks; all windows have the same size and start = chr_start + idx*step.
 * On return comp_min/comp_max[i] hold the merged span of i's component and
 * in_graph[i] says whether i has at least one edge (wave.rs:232: only nodes
 * with an edge enter the graph). */
static void merge_ints(const int64_t *idx, size_t p, int32_t chr_start,
                       int32_t size, int32_t step, float coverage,
                       int64_t *comp_min, int64_t *comp_max, char *in_graph) {
    size_t *par = (size_t *)malloc(sizeof(size_t) * (p ? p : 1));
    for (size_t i = 0; i < p; ++i) {
        par[i] = i;
        in_graph[i] = 0;
    }
    for (size_t i = 0; i < p; ++i) {                                 /* :223 */
        int64_t si = chr_start + idx[i] * step, ei = si + size - 1;
        for (size_t j = i + 1; j < p; ++j) {                         /* :225 */
            int64_t sj = chr_start + idx[j] * step, ej = sj + size - 1;
            int64_t lo = si > sj ? si : sj, hi = ei < ej ? ei : ej;
            if (hi < lo) continue;                                   /* :228 is_empty */
            float inter = (float)(int32_t)(hi - lo + 1);
            float cov_i = (float)size / inter;                       /* :229 */
            float cov_j = (float)size / inter;                       /* :230 */
            if (cov_i >= coverage && cov_j >= coverage) {            /* :231 */
                size_t a = uf_find(par, i), b = uf_find(par, j);
                if (a != b) par[b] = a;
                in_graph[i] = in_graph[j] = 1;
            }
        }
    }
    /* union of the member spans per component (:240-248); members are linked
     * through non-empty intersections, so the union is one span */
    for (size_t i = 0; i < p; ++i) {
        comp_min[i] = INT64_MAX;
        comp_max[i] = INT64_MIN;
    }
    for (size_t i = 0; i < p; ++i) {
        size_t r = uf_find(par, i);
        int64_t si = chr_start + idx[i] * step, ei = si + size - 1;
        if (si < comp_min[r]) comp_min[r] = si;
        if (ei > comp_max[r]) comp_max[r] = ei;
    }
    for (size_t i = 0; i < p; ++i) {
        size_t r = uf_find(par, i);
        comp_min[i] = comp_min[r];
        comp_max[i] = comp_max[r];
    }
    free(par);
}

char *ora_wave_proc_ctg(const char *chr_id, int32_t chr_start, int32_t chr_end,
                        const uint8_t *seq, int32_t size, int32_t step,
                        size_t lag, float threshold, float influence,
                        float coverage, int is_signal) {
    int64_t len = (int64_t)chr_end - chr_start + 1;
    int64_t n = ora_sliding_count(len, size, step);
    if (n < 0) return NULL;
    float *gcs = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1));
    int32_t *sig = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    if (ora_wave_windows(seq, len, size, step, lag, threshold, influence, NULL,
                         gcs, sig) < 0) {
        free(gcs);
        free(sig);
        return NULL;
    }
    sbuf out = {0};
    sb_reserve(&out, 16);
    out.p[0] = 0;
    if (is_signal) {                                                 /* wave.rs:158-168 */
        for (int64_t i = 0; i < n; ++i) {
            int64_t s = chr_start + i * step;
            sb_puts(&out, chr_id);
            sb_puts(&out, ":");
            sb_runlist(&out, s, s + size - 1);
            sb_puts(&out, "\t");
            sb_putf(&out, gcs[i]);
            sb_puts(&out, "\t");
            sb_puti(&out, sig[i]);
            sb_puts(&out, "\n");
        }
    } else {                                                         /* wave.rs:169-211 */
        /* comp span + membership per window, filled per sign */
        int64_t *cmin = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        int64_t *cmax = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
        char *merged = (char *)calloc((size_t)(n ? n : 1), 1);
        for (int sgn = 1; sgn >= -1; sgn -= 2) { /* crests then troughs :172-186 */
            size_t p = 0;
            for (int64_t i = 0; i < n; ++i) p += sig[i] == sgn;
            int64_t *idx = (int64_t *)malloc(sizeof(int64_t) * (p ? p : 1));
            int64_t *mn = (int64_t *)malloc(sizeof(int64_t) * (p ? p : 1));
            int64_t *mx = (int64_t *)malloc(sizeof(int64_t) * (p ? p : 1));
            char *ing = (char *)malloc(p ? p : 1);
            size_t q = 0;
            for (int64_t i = 0; i < n; ++i)
                if (sig[i] == sgn) idx[q++] = i;
            merge_ints(idx, p, chr_start, size, step, coverage, mn, mx, ing);
            for (size_t k = 0; k < p; ++k) {
                cmin[idx[k]] = mn[k];
                cmax[idx[k]] = mx[k];
                merged[idx[k]] = ing[k];
            }
            free(idx);
            free(mn);
            free(mx);
            free(ing);
        }
        /* outputs :190-211.  `seen` is keyed by the merged runlist string;
         * distinct components have distinct merged spans (a window is either a
         * crest or a trough), so "not seen yet" == "first member of its
         * component in window order" == "window start equals the span min". */
        for (int64_t i = 0; i < n; ++i) {
            if (sig[i] == 0) continue;
            int64_t s = chr_start + i * step, e = s + size - 1;
            if (merged[i]) {
                if (s != cmin[i]) continue;
                sb_puts(&out, chr_id);
                sb_puts(&out, "(+):");
                sb_runlist(&out, cmin[i], cmax[i]);
            } else {
                sb_puts(&out, chr_id);
                sb_puts(&out, ":");
                sb_runlist(&out, s, e);
            }
            sb_puts(&out, "\t");
            sb_putf(&out, gcs[i]);
            sb_puts(&out, "\t");
            sb_puti(&out, sig[i]);
            sb_puts(&out, "\n");
        }
        free(cmin);
        free(cmax);
        free(merged);
    }
    free(gcs);
    free(sig);
    return out.p;
}

/* ------------------------------------------------------------------------ */
/* cmd_gams/sw.rs:108-194 + Sw Display (libs/data.rs:58-83)                 */
/* ------------------------------------------------------------------------ */
static void sb_range(sbuf *b, const char *chr, int32_t s, int32_t e) {
    sb_puts(b, chr);
    sb_puts(b, ":");
    sb_runlist(b, s, e);
}

char *ora_sw_proc_ctg(const char *chr_id, int32_t chr_start, int32_t chr_end,
                      const uint8_t *seq, const char *const *feature_ids,
                      const int32_t *f_start, const int32_t *f_end, size_t nf,
                      int32_t size, int32_t max, int32_t resize) {
    sbuf out = {0};
    sb_reserve(&out, 16);
    out.p[0] = 0;
    size_t cap = (size_t)(1 + 2 * (max > 0 ? max : 0));
    int32_t *ws = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *we = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *wt = (int32_t *)malloc(sizeof(int32_t) * cap);
    int32_t *wd = (int32_t *)malloc(sizeof(int32_t) * cap);
    static const char *TYPES[3] = {"M", "L", "R"};
    for (size_t f = 0; f < nf; ++f) {                                /* sw.rs:141 */
        int32_t nw = ora_center_sw(chr_start, chr_end, f_start[f], f_end[f],
                                   size, max, ws, we, wt, wd);       /* :150 */
        for (int32_t k = 0; k < nw; ++k) {                           /* :152 */
            float gc = ora_range_gc_content(seq, chr_start, ws[k], we[k]); /* :168 */
            int32_t rs, re;
            ora_center_resize(chr_start, chr_end, ws[k], we[k], resize, &rs,
                              &re);                                  /* :175 */
            float m, s, c;
            ora_range_gc_stat(seq, chr_start, rs, re, size, size, &m, &s,
                              &c);                                   /* :177-178 */
            sb_puts(&out, "sw:");                                    /* :153 */
            sb_puts(&out, feature_ids[f]);
            sb_puts(&out, ":");
            sb_puti(&out, k + 1);
            sb_puts(&out, "\t");
            sb_range(&out, chr_id, ws[k], we[k]);                    /* :157 */
            sb_puts(&out, "\t");
            sb_puts(&out, TYPES[wt[k]]);
            sb_puts(&out, "\t");
            sb_puti(&out, wd[k]);
            sb_puts(&out, "\t");
            sb_putf(&out, gc);                                       /* data.rs:61-67 */
            sb_puts(&out, "\t");
            sb_putf(&out, m);
            sb_puts(&out, "\t");
            sb_putf(&out, s);
            sb_puts(&out, "\t");
            sb_putf(&out, c);
            sb_puts(&out, "\t\n"); /* empty rg_count field, data.rs:71-80 */
        }
    }
    free(ws);
    free(we);
    free(wt);
    free(wd);
    return out.p;
}

/* ------------------------------------------------------------------------ */
/* rust-lapper                                                              */
/* ------------------------------------------------------------------------ */
int64_t ora_lapper_find_first(const uint32_t *starts, const uint32_t *stops,
                              size_t m, uint32_t qs, uint32_t qe) {
    /* Lapper::find starts at lower_bound(qs - max_len) and walks forward
     * (rust-lapper 1.1.0 lib.rs, IterFind::next); nothing before that offset
     * can overlap, so a scan from 0 returns the same first hit. */
    for (size_t i = 0; i < m; ++i) {
        if (starts[i] < qe && stops[i] > qs) return (int64_t)i; /* Interval::overlap */
        if (starts[i] >= qe) break;
    }
    return -1;
}

/* number of elements < key (Lapper::bsearch_seq) */
static size_t lower_bound_u32(const uint32_t *a, size_t m, uint64_t key) {
    size_t lo = 0, hi = m;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if ((uint64_t)a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

int32_t ora_lapper_count(const uint32_t *sorted_starts,
                         const uint32_t *sorted_stops, size_t m, uint32_t qs,
                         uint32_t qe) {
    /* Lapper::count: first = bsearch_seq(start+1, stops); last =
     * bsearch_seq(stop, starts); len - first - (len - last) */
    size_t first = lower_bound_u32(sorted_stops, m, (uint64_t)qs + 1);
    size_t last = lower_bound_u32(sorted_starts, m, (uint64_t)qe);
    return (int32_t)((int64_t)last - (int64_t)first);
}

/* ------------------------------------------------------------------------ */
/* cmd_gams/anno.rs:128-139                                                 */
/* ------------------------------------------------------------------------ */
float ora_anno_prop(const int32_t *span_lo, const int32_t *span_hi, size_t ns,
                    int32_t ctg_s, int32_t ctg_e, int32_t rs, int32_t re) {
    int64_t card = 0;
    int64_t lo0 = rs > ctg_s ? rs : ctg_s;
    int64_t hi0 = re < ctg_e ? re : ctg_e;
    for (size_t i = 0; i < ns; ++i) {
        int64_t lo = span_lo[i] > lo0 ? span_lo[i] : lo0;
        int64_t hi = span_hi[i] < hi0 ? span_hi[i] : hi0;
        if (hi >= lo) card += hi - lo + 1;
    }
    int64_t total = (int64_t)re - rs + 1;
    return (float)(int32_t)card / (float)(int32_t)total;             /* :138 */
}

/* ------------------------------------------------------------------------ */
/* cmd_gams/gen.rs:81-126: valid regions -> --piece chunks                   */
/* ------------------------------------------------------------------------ */
int64_t ora_gen_regions(const uint8_t *seq, int64_t len, int32_t piece, int32_t fill,
                        int32_t min_len, int32_t *out_start, int32_t *out_end, int64_t cap) {
    /* valid spans = complement of the ambiguous bases (gen.rs:86-102) */
    int64_t nsp = 0, csp = 1024;
    int64_t *lo = (int64_t *)malloc(sizeof(int64_t) * csp), *hi = (int64_t *)malloc(sizeof(int64_t) * csp);
    int64_t run = -1;
    for (int64_t i = 0; i <= len; ++i) {
        int ok = 0;
        if (i < len) {
            uint8_t b = seq[i];
            ok = b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'a' || b == 'c' || b == 'g' || b == 't';
        }
        if (ok && run < 0) run = i;
        if (!ok && run >= 0) {
            /* fill(fill-1): a hole of at most fill-1 bases joins the previous span (gen.rs:103) */
            if (nsp > 0 && (run + 1) - hi[nsp - 1] - 1 <= (int64_t)fill - 1) {
                hi[nsp - 1] = i;
            } else {
                if (nsp == csp) {
                    csp *= 2;
                    lo = (int64_t *)realloc(lo, sizeof(int64_t) * csp);
                    hi = (int64_t *)realloc(hi, sizeof(int64_t) * csp);
                }
                lo[nsp] = run + 1;
                hi[nsp] = i;
                ++nsp;
            }
            run = -1;
        }
    }
    int64_t n = 0;
    for (int64_t s = 0; s < nsp; ++s) {
        if (hi[s] - lo[s] + 1 < (int64_t)min_len) continue; /* excise(min): gen.rs:104 */
        int64_t pos = lo[s], max = hi[s];
        int64_t first = n;
        while (max - pos + 1 > piece) { /* gen.rs:112-116 */
            if (n < cap) {
                out_start[n] = (int32_t)pos;
                out_end[n] = (int32_t)(pos + piece - 1);
            }
            ++n;
            pos += piece;
        }
        if (n == first) { /* gen.rs:118-120 */
            if (n < cap) {
                out_start[n] = (int32_t)pos;
                out_end[n] = (int32_t)max;
            }
            ++n;
        } else if (n - 1 < cap) { /* gen.rs:121-123: the last piece absorbs the remainder */
            out_end[n - 1] = (int32_t)max;
        }
    }
    free(lo);
    free(hi);
    return n;
}

/* ------------------------------------------------------------------------ */
/* cmd_gams/peak.rs:65-158 for one ctg: peaks given in bucket order          */
/* ------------------------------------------------------------------------ */
char *ora_peak_rows(const char *ctg_id, const char *chr_id, int32_t chr_start, int32_t chr_end,
                    const uint8_t *seq, const int32_t *p_start, const int32_t *p_end,
                    const char *const *p_signal, size_t np) {
    sbuf out = {0};
    sb_reserve(&out, 16);
    out.p[0] = 0;
    if (np == 0) return out.p;
    float *gc = (float *)malloc(sizeof(float) * np);
    int32_t *lw = (int32_t *)malloc(sizeof(int32_t) * np), *rw = (int32_t *)malloc(sizeof(int32_t) * np);
    float *la = (float *)malloc(sizeof(float) * np), *ra = (float *)malloc(sizeof(float) * np);
    const char **ls = (const char **)malloc(sizeof(char *) * np), **rs = (const char **)malloc(sizeof(char *) * np);
    for (size_t i = 0; i < np; ++i) gc[i] = ora_range_gc_content(seq, chr_start, p_start[i], p_end[i]); /* :79 */
    const char *prev_signal = p_signal[0];                             /* :112-133 */
    float prev_gc = gc[0];
    int32_t prev_end = chr_start;
    for (size_t i = 0; i < np; ++i) {
        lw[i] = p_start[i] - prev_end + 1;
        la[i] = fabsf(gc[i] - prev_gc);
        ls[i] = prev_signal;
        prev_signal = p_signal[i];
        prev_end = p_end[i];
        prev_gc = gc[i];
    }
    const char *next_signal = p_signal[np - 1];                        /* :135-157 */
    float next_gc = gc[np - 1];
    int32_t next_start = chr_end;
    for (size_t i = np; i-- > 0;) {
        rw[i] = next_start - p_end[i] + 1;
        ra[i] = fabsf(gc[i] - next_gc);
        rs[i] = next_signal;
        next_signal = p_signal[i];
        next_start = p_start[i];
        next_gc = gc[i];
    }
    for (size_t i = 0; i < np; ++i) {
        sb_puts(&out, "peak:");
        sb_puts(&out, ctg_id);
        sb_puts(&out, ":");
        sb_puti(&out, (long long)i + 1);
        sb_puts(&out, "\t");
        sb_range(&out, chr_id, p_start[i], p_end[i]);
        sb_puts(&out, "\t");
        sb_puti(&out, p_end[i] - p_start[i] + 1);
        sb_puts(&out, "\t");
        sb_putf(&out, gc[i]);
        sb_puts(&out, "\t");
        sb_puts(&out, p_signal[i]);
        sb_puts(&out, "\t");
        sb_puti(&out, lw[i]);
        sb_puts(&out, "\t");
        sb_putf(&out, la[i]);
        sb_puts(&out, "\t");
        sb_puts(&out, ls[i]);
        sb_puts(&out, "\t");
        sb_puti(&out, rw[i]);
        sb_puts(&out, "\t");
        sb_putf(&out, ra[i]);
        sb_puts(&out, "\t");
        sb_puts(&out, rs[i]);
        sb_puts(&out, "\n");
    }
    free(gc); free(lw); free(rw); free(la); free(ra); free(ls); free(rs);
    return out.p;
}
