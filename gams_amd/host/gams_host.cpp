// gams_host.cpp -- see gams_host.hpp.  Citations are file:line under the reference.
#include "gams_host.hpp"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <charconv>
#include <string_view>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <numeric>
#include <queue>
#include <thread>

#include <dlfcn.h>
#include <zlib.h>
#include <chrono>

namespace gams {

namespace {

void check(gams_gpu_t *h, int rc) {
    if (rc != GAMS_OK) throw Error(rc, gams_gpu_last_error(h));
}

bool is_word(char c) {
    return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}

struct SeqSetGuard {
    gams_gpu_t *h;
    gams_seqset_t *s = nullptr;
    ~SeqSetGuard() {
        if (s) gams_seqset_destroy(h, s);
    }
};
struct PlanGuard {
    gams_gpu_t *h;
    gams_wave_plan_t *p = nullptr;
    ~PlanGuard() {
        if (p) gams_wave_plan_destroy(h, p);
    }
};

}  // namespace

// ---------------------------------------------------------------------------
// formatting
// ---------------------------------------------------------------------------
std::string fmt_f32(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    if (v == 0.0f) return std::signbit(v) ? "-0" : "0";
    // shortest round-trip digits (Ryu-style, like Rust), then positional notation
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr);           // [-]d[.ddd]e[+-]XX
    std::string out;
    size_t i = 0;
    if (sci[0] == '-') {
        out += '-';
        i = 1;
    }
    std::string digits;
    for (; i < sci.size() && sci[i] != 'e'; ++i)
        if (sci[i] != '.') digits += sci[i];
    const int ex = std::atoi(sci.c_str() + i + 1);
    const int nd = (int)digits.size();
    if (ex >= 0) {
        for (int k = 0; k <= ex; ++k) out += k < nd ? digits[k] : '0';
        if (nd > ex + 1) {
            out += '.';
            out.append(digits, ex + 1, std::string::npos);
        }
    } else {
        out += "0.";
        out.append((size_t)(-ex - 1), '0');
        out += digits;
    }
    return out;
}

std::string runlist(int64_t s, int64_t e) {
    std::string o = std::to_string(s);
    if (e != s) {
        o += '-';
        o += std::to_string(e);
    }
    return o;
}

// intspan::Range::from_str: [name.]chr[(strand)]:start[-end]
Range Range::from_str(const std::string &in) {
    // (views into `in` until the end: the text paths call this once per line, 10^6-10^8 times)
    Range r;
    std::string_view s(in);
    while (!s.empty() && (s.back() == '\r' || s.back() == '\n' || s.back() == ' ')) s.remove_suffix(1);
    const size_t colon = s.rfind(':');
    if (colon == std::string_view::npos || colon == 0 || colon + 1 >= s.size()) return r;
    std::string_view head = s.substr(0, colon);
    const std::string_view tail = s.substr(colon + 1);
    auto number = [](std::string_view d) {             // <= 10 digits: no overflow of long long
        long long v = 0;
        for (char c : d) v = v * 10 + (c - '0');
        return v;
    };
    // start[-_]end
    size_t i = 0;
    while (i < tail.size() && tail[i] >= '0' && tail[i] <= '9') ++i;
    if (i == 0 || i > 10) return r;
    long long st = number(tail.substr(0, i)), en = st;
    if (i < tail.size()) {
        size_t j = i;
        while (j < tail.size() && (tail[j] == '-' || tail[j] == '_')) ++j;
        if (j == i) return r;
        size_t k = j;
        while (k < tail.size() && tail[k] >= '0' && tail[k] <= '9') ++k;
        if (k == j || k != tail.size() || k - j > 10) return r;
        en = number(tail.substr(j, k - j));
    }
    if (st > INT32_MAX || en > INT32_MAX) return r;
    // head: [name.]chr[(strand)]
    std::string_view strand, name;
    if (!head.empty() && head.back() == ')') {
        const size_t open = head.rfind('(');
        if (open == std::string_view::npos) return r;
        strand = head.substr(open + 1, head.size() - open - 2);
        head = head.substr(0, open);
    }
    const size_t dot = head.find('.');
    if (dot != std::string_view::npos) {
        name = head.substr(0, dot);
        head = head.substr(dot + 1);
    }
    if (head.empty()) return r;
    for (char c : head)
        if (!(is_word(c) || c == '-' || c == '/')) return r;
    r.strand.assign(strand);
    r.name.assign(name);
    r.chr.assign(head);
    r.start = (int32_t)st;
    r.end = (int32_t)en;
    r.valid = true;
    return r;
}

std::string Range::to_string() const {
    std::string o;
    if (!name.empty()) o += name + ".";
    o += chr;
    if (!strand.empty()) o += "(" + strand + ")";
    o += ":" + runlist(start, end);
    return o;
}

// ---------------------------------------------------------------------------
// wave
// ---------------------------------------------------------------------------
namespace {

// merge_ints (wave.rs:217-252) for the peaks of one sign.  All windows have the
// same size and start at chr_start + k*step, so the intersection of windows k_i
// < k_j has size - (k_j-k_i)*step bases and the edge test (:229-231) depends on
// d = k_j - k_i alone: d in [dmin, dmax].  Components by union-find.
struct Merge {
    std::vector<int64_t> cmin, cmax;
    std::vector<char> in_graph;
};

Merge merge_ints(const std::vector<uint32_t> &w, int32_t chr_start, int32_t size, int32_t step, float coverage) {
    const size_t p = w.size();
    Merge m;
    m.cmin.resize(p);
    m.cmax.resize(p);
    m.in_graph.assign(p, 0);
    std::vector<size_t> par(p);
    std::iota(par.begin(), par.end(), (size_t)0);
    auto find = [&](size_t x) {
        while (par[x] != x) {
            par[x] = par[par[x]];
            x = par[x];
        }
        return x;
    };
    // d range with a non-empty intersection that passes both coverage tests
    const int64_t dmax = ((int64_t)size + step - 1) / step - 1;  // largest d with d*step < size
    int64_t dmin = dmax + 1;
    for (int64_t d = 1; d <= dmax; ++d) {
        const float inter = (float)(int32_t)(size - d * step);
        const float cov = (float)size / inter;                   // cov_i == cov_j (:229-230)
        if (cov >= coverage) {                                   // monotone in d
            dmin = d;
            break;
        }
    }
    if (dmin == 1) {
        // every overlap links (coverage <= 1, the usual case): components are the maximal runs whose
        // consecutive windows are at most dmax apart -- one sweep instead of p*dmax pair tests
        // (step 1: dmax = 99 and tens of thousands of peaks per ctg)
        for (size_t i = 1; i < p; ++i) {
            if ((int64_t)w[i] - (int64_t)w[i - 1] <= dmax) {
                par[i] = find(i - 1);
                m.in_graph[i] = m.in_graph[i - 1] = 1;
            }
        }
    } else {
        for (size_t i = 0; i < p; ++i) {
            for (size_t j = i + 1; j < p; ++j) {
                const int64_t d = (int64_t)w[j] - (int64_t)w[i];
                if (d > dmax) break;
                if (d >= dmin) {
                    size_t a = find(i), b = find(j);
                    if (a != b) par[b] = a;
                    m.in_graph[i] = m.in_graph[j] = 1;
                }
            }
        }
    }
    std::vector<int64_t> mn(p, INT64_MAX), mx(p, INT64_MIN);
    for (size_t i = 0; i < p; ++i) {
        const size_t r = find(i);
        const int64_t s = (int64_t)chr_start + (int64_t)w[i] * step, e = s + size - 1;
        mn[r] = std::min(mn[r], s);
        mx[r] = std::max(mx[r], e);
    }
    for (size_t i = 0; i < p; ++i) {
        const size_t r = find(i);
        m.cmin[i] = mn[r];
        m.cmax[i] = mx[r];
    }
    return m;
}

}  // namespace

void merge_windows(const uint32_t *w, size_t n, int32_t chr_start, int32_t size, int32_t step, float coverage,
                   int64_t *cmin, int64_t *cmax, char *in_graph) {
    const Merge m = merge_ints(std::vector<uint32_t>(w, w + n), chr_start, size, step, coverage);
    for (size_t i = 0; i < n; ++i) {
        cmin[i] = m.cmin[i];
        cmax[i] = m.cmax[i];
        in_graph[i] = m.in_graph[i];
    }
}

namespace {

// One batch of ctgs in flight on one handle: start() queues the uploads (copy stream) and the
// kernels (compute stream, behind the upload event) and returns; finish() waits, fetches and
// formats.  Starting batch k+1 before finishing batch k overlaps its upload with k's kernel and
// k's host-side merge/formatting with k+1's kernel.
// fn(i) for i in [0,n) on up to 8 host threads; the first exception is rethrown on the caller
template <typename F>
void parallel_for(uint32_t n, F fn) {
    const unsigned T = std::max(1u, std::min({8u, std::thread::hardware_concurrency(), n / 4u}));
    if (T <= 1) {
        for (uint32_t i = 0; i < n; ++i) fn(i);
        return;
    }
    std::atomic<uint32_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t)
        pool.emplace_back([&] {
            try {
                for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i);
            } catch (...) {
                std::lock_guard<std::mutex> lk(mu);
                if (!err) err = std::current_exception();
                next.store(n);
            }
        });
    for (auto &th : pool) th.join();
    if (err) std::rethrow_exception(err);
}

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct WaveJob {
    gams_gpu_t *h;
    std::vector<Ctg> ctgs;
    WaveArgs a;
    SeqSetGuard sg;
    PlanGuard pg;
    uint8_t *image = nullptr;      // start_gz: page-locked image of the seqset the workers inflate into
    WaveStages *st = nullptr;      // optional stage clock (wave_proc_ctgs*, see gams_host.hpp)
    double t_mark = 0;
    WaveJob(gams_gpu_t *h_, std::vector<Ctg> c, const WaveArgs &a_) : h(h_), ctgs(std::move(c)), a(a_), sg{h_}, pg{h_} {}
    ~WaveJob() {
        if (image) {
            (void)gams_gpu_sync(h);            // a DMA may still be reading it
            gams_gpu_host_free(h, image);
        }
    }
    void start(const std::vector<const uint8_t *> &seqs);
    void start_gz(const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len, unsigned threads);
    std::vector<std::string> finish();
    // stage clock: with st->sync the device is drained first, so the interval belongs to the stage just queued
    void mark(double WaveStages::*slot) {
        if (!st) return;
        if (st->sync) check(h, gams_gpu_sync(h));
        const double t = now_ms();
        st->*slot += t - t_mark;
        t_mark = t;
    }
    void plan_and_run();
    bool device_rows = false;      // the rows come as text from the device (gams_wave_rows_*): coverage links every overlap
};

void WaveJob::plan_and_run() {
    gams_wave_params_t prm{a.size, a.step, a.lag, a.threshold, a.influence};
    check(h, gams_wave_plan_create(h, sg.s, &prm, a.signal ? GAMS_WAVE_DENSE : GAMS_WAVE_PEAKS, &pg.p));
    check(h, gams_wave_plan_set_pipelined(h, pg.p, 1));   // finish() waits for this job, not the stream
    if (!a.signal) {
        // merged rows on the device when --coverage links every overlap (every value up to 1; the default is 0.2);
        // otherwise the peaks come back and merge_ints runs here
        std::vector<const char *> chr(ctgs.size());
        std::vector<int32_t> cs(ctgs.size());
        for (size_t c = 0; c < ctgs.size(); ++c) {
            chr[c] = ctgs[c].chr_id.c_str();
            cs[c] = ctgs[c].chr_start;
        }
        const int rc = gams_wave_rows_setup(h, pg.p, chr.data(), cs.data(), a.coverage);
        if (rc == GAMS_OK)
            device_rows = true;
        else if (rc != GAMS_EUNSUPPORTED)
            check(h, rc);
    }
    mark(&WaveStages::plan_ms);
    check(h, gams_wave_run(h, pg.p));
    if (device_rows) check(h, gams_wave_rows_begin(h, pg.p));   // packing, merging, formatting and the copy queue behind the pass
    mark(&WaveStages::kernel_ms);
}

void WaveJob::start(const std::vector<const uint8_t *> &seqs) {
    const uint32_t n = (uint32_t)ctgs.size();
    if (n == 0) return;
    t_mark = now_ms();
    std::vector<uint32_t> lens(n);
    for (uint32_t c = 0; c < n; ++c) lens[c] = (uint32_t)(ctgs[c].chr_end - ctgs[c].chr_start + 1);
    check(h, gams_seqset_create(h, n, lens.data(), &sg.s));
    check(h, gams_seqset_upload_all(h, sg.s, seqs.data()));
    mark(&WaveStages::upload_ms);
    plan_and_run();
}

// The `seq:` values as the store holds them (gzip members, redis.rs:149-161): `threads` workers take the
// ctgs in order -- one ctg per worker at a time, like the reference's --parallel workers (wave.rs:288-299
// call get_seq -> decode_gz inside the worker) -- and inflate each one straight into a page-locked image
// of the device buffer; the calling thread follows them and hands every finished stretch of >= 8 MiB to
// the DMA engine (gams_seqset_upload_image): no staging copy, and the upload hides behind the inflate.
void WaveJob::start_gz(const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len,
                       unsigned threads) {
    const uint32_t n = (uint32_t)ctgs.size();
    if (n == 0) return;
    t_mark = now_ms();
    std::vector<uint32_t> lens(n);
    for (uint32_t c = 0; c < n; ++c) lens[c] = (uint32_t)(ctgs[c].chr_end - ctgs[c].chr_start + 1);
    check(h, gams_seqset_create(h, n, lens.data(), &sg.s));
    std::vector<uint64_t> off(n);
    uint64_t bytes = 0;
    check(h, gams_seqset_layout(h, sg.s, off.data(), &bytes));
    void *blk = nullptr;
    check(h, gams_gpu_host_alloc(h, bytes, &blk));
    image = static_cast<uint8_t *>(blk);
    const unsigned T = std::max(1u, std::min(threads ? threads : 16u, n));
    if (st) st->threads = T;
    std::vector<std::atomic<uint8_t>> done(n);
    for (auto &d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<uint32_t> next{0};
    std::atomic<bool> failed{false};
    std::exception_ptr err;
    std::mutex mu;
    auto work = [&] {
        try {
            for (uint32_t i = next.fetch_add(1); i < n && !failed.load(); i = next.fetch_add(1)) {
                const size_t got = decode_gz_into(blobs[i], (size_t)blob_len[i], image + off[i], lens[i]);
                if (got != lens[i])
                    throw Error(GAMS_EINVAL, "seq:" + ctgs[i].id + " inflates to " + std::to_string(got) +
                                                 " bases, the ctg record says " + std::to_string(lens[i]));
                // the alignment gap behind the ctg (never counted; kept defined)
                const uint64_t end = off[i] + lens[i], stop = i + 1 < n ? off[i + 1] : end;
                if (stop > end) std::memset(image + end, 0, stop - end);
                done[i].store(1, std::memory_order_release);
            }
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!err) err = std::current_exception();
            failed.store(true);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < T; ++t) pool.emplace_back(work);
    // follow the workers: ctgs are taken in order, so they also finish roughly in order
    constexpr uint64_t kPiece = 8ull << 20;
    uint64_t lo = 0;
    int rc = GAMS_OK;
    for (uint32_t i = 0; i < n && rc == GAMS_OK; ++i) {
        while (!done[i].load(std::memory_order_acquire) && !failed.load()) std::this_thread::yield();
        if (failed.load()) break;
        const uint64_t hi = i + 1 < n ? off[i + 1] : off[i] + lens[i];
        if (hi - lo >= kPiece || i + 1 == n) {
            rc = gams_seqset_upload_image(h, sg.s, image, lo, hi);
            lo = hi;
        }
    }
    for (auto &th : pool) th.join();
    if (err) std::rethrow_exception(err);
    check(h, rc);
    mark(&WaveStages::inflate_upload_ms);
    plan_and_run();
}

std::vector<std::string> WaveJob::finish() {
    const uint32_t n = (uint32_t)ctgs.size();
    std::vector<std::string> out(n);
    if (n == 0) return out;
    const float fsize = (float)a.size;
    if (a.signal) {                                                     // wave.rs:158-168
        // the rows as text from the device (gams_wave_signal_text); what it does not cover goes the host way below
        {
            std::vector<const char *> chr(n);
            std::vector<int32_t> cst(n);
            for (uint32_t c = 0; c < n; ++c) {
                chr[c] = ctgs[c].chr_id.c_str();
                cst[c] = ctgs[c].chr_start;
            }
            const char *text = nullptr;
            uint64_t bytes = 0;
            const uint64_t *off = nullptr;
            const int rc = gams_wave_signal_text(h, pg.p, chr.data(), cst.data(), &text, &bytes, &off);
            if (rc == GAMS_OK) {
                mark(&WaveStages::peaks_ms);
                const unsigned T = (unsigned)std::max<uint64_t>(
                    1, std::min<uint64_t>({16, std::thread::hardware_concurrency(), (uint64_t)n, bytes / (4u << 20) + 1}));
                std::atomic<uint32_t> next{0};
                std::vector<std::exception_ptr> errs(T);
                auto work = [&](unsigned t) {
                    try {
                        for (uint32_t c = next.fetch_add(1); c < n; c = next.fetch_add(1)) out[c].assign(text + off[c], text + off[c + 1]);
                    } catch (...) {
                        errs[t] = std::current_exception();
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
                work(0);
                for (auto &th : pool) th.join();
                for (auto &er : errs)
                    if (er) std::rethrow_exception(er);
                if (st) st->peaks = bytes;
                mark(&WaveStages::format_ms);
                return out;
            }
            if (rc != GAMS_EUNSUPPORTED && rc != GAMS_EINVAL) check(h, rc);
        }
        // fetch the dense rows on this thread (the handle's), format the ctgs on several
        std::vector<std::vector<uint32_t>> cnts(n);
        std::vector<std::vector<int8_t>> sigs(n);
        for (uint32_t c = 0; c < n; ++c) {
            const uint32_t nw = gams_wave_ctg_windows(pg.p, c);
            cnts[c].resize(nw ? nw : 1);
            sigs[c].resize(nw ? nw : 1);
            check(h, gams_wave_dense(h, pg.p, c, cnts[c].data(), sigs[c].data()));
        }
        parallel_for(n, [&](uint32_t c) {
            const uint32_t nw = gams_wave_ctg_windows(pg.p, c);
            const std::vector<uint32_t> &cnt = cnts[c];
            const std::vector<int8_t> &sig = sigs[c];
            std::string &o = out[c];
            o.reserve((size_t)nw * 24);
            for (uint32_t i = 0; i < nw; ++i) {
                const int64_t s = (int64_t)ctgs[c].chr_start + (int64_t)i * a.step;
                o += ctgs[c].chr_id;
                o += ':';
                o += runlist(s, s + a.size - 1);
                o += '\t';
                o += fmt_f32((float)cnt[i] / fsize);                    // gc_content: count as f32 / len as f32
                o += '\t';
                o += std::to_string((int)sig[i]);
                o += '\n';
            }
        });
        return out;
    }
    if (device_rows) {
        const char *text = nullptr;
        uint64_t bytes = 0;
        const uint64_t *off = nullptr;
        check(h, gams_wave_rows_end(h, pg.p, &text, &bytes, &off));
        mark(&WaveStages::peaks_ms);
        for (uint32_t c = 0; c < n; ++c) out[c].assign(text + off[c], text + off[c + 1]);
        if (st) st->peaks = bytes;          // (text bytes fetched: the peak records stay on the device)
        mark(&WaveStages::format_ms);
        return out;
    }
    const gams_peak_t *pk = nullptr;
    uint64_t np = 0;
    check(h, gams_wave_peaks(h, pg.p, &pk, &np));
    mark(&WaveStages::peaks_ms);
    if (st) st->peaks = np;
    // peaks arrive ordered by (ctg, window): slice per ctg, then merge + format the ctgs on a few
    // host threads (the rows of one ctg depend on nothing else)
    std::vector<uint64_t> first(n + 1, np);
    {
        uint64_t q = 0;
        for (uint32_t c = 0; c < n; ++c) {
            first[c] = q;
            while (q < np && pk[q].ctg == c) ++q;
        }
        first[n] = q;
    }
    parallel_for(n, [&](uint32_t c) {                                   // wave.rs:169-211
        const uint64_t q = first[c], q1 = first[c + 1];
        // crests and troughs separately (:172-186)
        std::vector<uint32_t> w[2];
        std::vector<uint64_t> src[2];
        for (uint64_t i = q; i < q1; ++i) {
            const int s = pk[i].signal == 1 ? 0 : 1;
            w[s].push_back(pk[i].window);
            src[s].push_back(i);
        }
        std::vector<int64_t> cmin(q1 - q), cmax(q1 - q);
        std::vector<char> merged(q1 - q, 0);
        for (int s = 0; s < 2; ++s) {
            Merge m = merge_ints(w[s], ctgs[c].chr_start, a.size, a.step, a.coverage);
            for (size_t k = 0; k < w[s].size(); ++k) {
                cmin[src[s][k] - q] = m.cmin[k];
                cmax[src[s][k] - q] = m.cmax[k];
                merged[src[s][k] - q] = m.in_graph[k];
            }
        }
        std::string &o = out[c];
        o.reserve((size_t)(q1 - q) * 24);
        for (uint64_t i = q; i < q1; ++i) {                             // window order (:191)
            const int64_t s = (int64_t)ctgs[c].chr_start + (int64_t)pk[i].window * a.step, e = s + a.size - 1;
            if (merged[i - q]) {
                // printed once, at the first member of the component (:201-206)
                if (s != cmin[i - q]) continue;
                o += ctgs[c].chr_id;
                o += "(+):";
                o += runlist(cmin[i - q], cmax[i - q]);
            } else {
                o += ctgs[c].chr_id;
                o += ':';
                o += runlist(s, e);
            }
            o += '\t';
            o += fmt_f32((float)pk[i].gc_count / fsize);
            o += '\t';
            o += std::to_string(pk[i].signal);
            o += '\n';
        }
    });
    mark(&WaveStages::format_ms);
    return out;
}

}  // namespace

std::vector<std::string> wave_proc_ctgs(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                        const std::vector<const uint8_t *> &seqs, const WaveArgs &a, WaveStages *stages) {
    if (ctgs.size() != seqs.size()) throw Error(GAMS_EINVAL, "wave_proc_ctgs: ctgs/seqs size mismatch");
    const double t0 = now_ms();
    WaveJob job(h, ctgs, a);
    job.st = stages;
    job.start(seqs);
    std::vector<std::string> out = job.finish();
    if (stages) stages->total_ms += now_ms() - t0;
    return out;
}

std::vector<std::string> wave_proc_ctgs_gz(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                           const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len,
                                           const WaveArgs &a, unsigned threads, WaveStages *stages) {
    if (ctgs.size() != blobs.size() || ctgs.size() != blob_len.size())
        throw Error(GAMS_EINVAL, "wave_proc_ctgs_gz: ctgs/blobs size mismatch");
    const double t0 = now_ms();
    WaveJob job(h, ctgs, a);
    job.st = stages;
    job.start_gz(blobs, blob_len, threads);
    std::vector<std::string> out = job.finish();
    if (stages) stages->total_ms += now_ms() - t0;
    return out;
}

std::string wave_proc_ctg(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq, const WaveArgs &a) {
    return wave_proc_ctgs(h, {ctg}, {seq}, a)[0];
}

std::vector<uint32_t> lpt_assign(const std::vector<uint64_t> &weights, uint32_t n_owners) {
    std::vector<size_t> order(weights.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] > weights[b]; });
    using Load = std::pair<uint64_t, uint32_t>;  // (load, owner): smallest load first, then smallest owner
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (uint32_t o = 0; o < n_owners; ++o) heap.push({0, o});
    std::vector<uint32_t> owner(weights.size(), 0);
    for (size_t i : order) {
        Load l = heap.top();
        heap.pop();
        owner[i] = l.second;
        heap.push({l.first + weights[i], l.second});
    }
    return owner;
}

namespace {

// what one device owns of a sharded interval problem: its groups (renumbered 0..), its queries
struct IntervalShare {
    std::vector<uint32_t> groups;       // global group ids, ascending
    std::vector<uint64_t> off;          // local group_off
    std::vector<uint64_t> q;            // global query indices, in query order
};

// groups -> owners by LPT on their sizes; queries follow their group
std::vector<IntervalShare> shard_intervals(uint32_t n_owners, uint32_t n_groups, const uint64_t *group_off,
                                           const uint32_t *q_group, uint64_t nq, std::vector<uint32_t> &local_id) {
    std::vector<uint64_t> weight(n_groups);
    for (uint32_t g = 0; g < n_groups; ++g) weight[g] = group_off[g + 1] - group_off[g] + 1;
    const std::vector<uint32_t> owner = lpt_assign(weight, n_owners);
    std::vector<IntervalShare> sh(n_owners);
    local_id.assign(n_groups, 0);
    for (uint32_t g = 0; g < n_groups; ++g) {
        IntervalShare &s = sh[owner[g]];
        local_id[g] = (uint32_t)s.groups.size();
        if (s.off.empty()) s.off.push_back(0);
        s.groups.push_back(g);
        s.off.push_back(s.off.back() + (group_off[g + 1] - group_off[g]));
    }
    for (auto &s : sh)
        if (s.off.empty()) s.off.push_back(0);
    for (uint64_t i = 0; i < nq; ++i)
        if (q_group[i] < n_groups) sh[owner[q_group[i]]].q.push_back(i);
    return sh;
}

template <typename F>
void run_shares(uint32_t n, F fn) {
    std::vector<std::exception_ptr> errs(n);
    std::vector<std::thread> pool;
    for (uint32_t g = 0; g < n; ++g)
        pool.emplace_back([&, g] {
            try {
                fn(g);
            } catch (...) {
                errs[g] = std::current_exception();
            }
        });
    for (auto &t : pool) t.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}

}  // namespace

std::vector<std::string> sw_proc_ctgs_multi(const std::vector<gams_gpu_t *> &handles, const std::vector<Ctg> &ctgs,
                                            const std::vector<const uint8_t *> &seqs,
                                            const std::vector<std::vector<Feature>> &features, const SwArgs &a) {
    if (handles.empty()) throw Error(GAMS_EINVAL, "sw_proc_ctgs_multi: no handles");
    if (ctgs.size() != seqs.size() || ctgs.size() != features.size())
        throw Error(GAMS_EINVAL, "sw_proc_ctgs_multi: ctgs / seqs / features size mismatch");
    std::vector<uint64_t> weight(ctgs.size());
    for (size_t c = 0; c < ctgs.size(); ++c) weight[c] = features[c].size() + 1;
    const std::vector<uint32_t> owner = lpt_assign(weight, (uint32_t)handles.size());
    std::vector<std::string> out(ctgs.size());
    run_shares((uint32_t)handles.size(), [&](uint32_t d) {
        std::vector<Ctg> dc;
        std::vector<const uint8_t *> ds;
        std::vector<std::vector<Feature>> df;
        std::vector<size_t> where;
        for (size_t c = 0; c < ctgs.size(); ++c)
            if (owner[c] == d && !features[c].empty()) {
                dc.push_back(ctgs[c]);
                ds.push_back(seqs[c]);
                df.push_back(features[c]);
                where.push_back(c);
            }
        std::vector<std::string> rows = sw_proc_ctgs(handles[d], dc, ds, df, a);
        for (size_t k = 0; k < where.size(); ++k) out[where[k]] = std::move(rows[k]);
    });
    return out;
}

void count_multi(const std::vector<gams_gpu_t *> &handles, uint32_t n_groups, const uint64_t *group_off,
                 const uint32_t *starts, const uint32_t *stops, const uint32_t *q_group, const uint32_t *qs,
                 const uint32_t *qe, uint64_t nq, int32_t *count) {
    if (handles.empty()) throw Error(GAMS_EINVAL, "count_multi: no handles");
    std::vector<uint32_t> local_id;
    const std::vector<IntervalShare> sh = shard_intervals((uint32_t)handles.size(), n_groups, group_off, q_group, nq, local_id);
    for (uint64_t i = 0; i < nq; ++i)
        if (q_group[i] >= n_groups) count[i] = 0;                        // utils.rs:29-32: ctg without an index
    run_shares((uint32_t)handles.size(), [&](uint32_t d) {
        const IntervalShare &s = sh[d];
        if (s.q.empty()) return;
        gams_gpu_t *h = handles[d];
        std::vector<uint32_t> st, sp;
        for (uint32_t g : s.groups) {
            st.insert(st.end(), starts + group_off[g], starts + group_off[g + 1]);
            sp.insert(sp.end(), stops + group_off[g], stops + group_off[g + 1]);
        }
        gams_index_t *ix = nullptr;
        check(h, gams_index_create(h, (uint32_t)s.groups.size(), s.off.data(), st.data(), sp.data(), &ix));
        std::vector<uint32_t> g(s.q.size()), a(s.q.size()), b(s.q.size());
        for (size_t k = 0; k < s.q.size(); ++k) {
            g[k] = local_id[q_group[s.q[k]]];
            a[k] = qs[s.q[k]];
            b[k] = qe[s.q[k]];
        }
        std::vector<int32_t> out(s.q.size());
        const int rc = gams_gpu_count(h, ix, g.data(), a.data(), b.data(), s.q.size(), out.data());
        gams_index_destroy(h, ix);
        check(h, rc);
        for (size_t k = 0; k < s.q.size(); ++k) count[s.q[k]] = out[k];
    });
}

void cover_multi(const std::vector<gams_gpu_t *> &handles, uint32_t n_groups, const uint64_t *group_off,
                 const int32_t *lo, const int32_t *hi, const uint32_t *q_group, const int32_t *clip_lo,
                 const int32_t *clip_hi, const int32_t *qs, const int32_t *qe, uint64_t nq, float *prop) {
    if (handles.empty()) throw Error(GAMS_EINVAL, "cover_multi: no handles");
    std::vector<uint32_t> local_id;
    const std::vector<IntervalShare> sh = shard_intervals((uint32_t)handles.size(), n_groups, group_off, q_group, nq, local_id);
    for (uint64_t i = 0; i < nq; ++i)
        if (q_group[i] >= n_groups) prop[i] = 0.0f;                      // anno.rs:128: chr absent from the set
    run_shares((uint32_t)handles.size(), [&](uint32_t d) {
        const IntervalShare &s = sh[d];
        if (s.q.empty()) return;
        gams_gpu_t *h = handles[d];
        std::vector<int32_t> l, u;
        for (uint32_t g : s.groups) {
            l.insert(l.end(), lo + group_off[g], lo + group_off[g + 1]);
            u.insert(u.end(), hi + group_off[g], hi + group_off[g + 1]);
        }
        gams_spans_t *sp = nullptr;
        check(h, gams_spans_create(h, (uint32_t)s.groups.size(), s.off.data(), l.data(), u.data(), &sp));
        const size_t m = s.q.size();
        std::vector<uint32_t> g(m);
        std::vector<int32_t> cl(m), ch(m), a(m), b(m);
        for (size_t k = 0; k < m; ++k) {
            const uint64_t i = s.q[k];
            g[k] = local_id[q_group[i]];
            cl[k] = clip_lo[i];
            ch[k] = clip_hi[i];
            a[k] = qs[i];
            b[k] = qe[i];
        }
        std::vector<float> out(m);
        const int rc = gams_gpu_cover(h, sp, g.data(), cl.data(), ch.data(), a.data(), b.data(), m, out.data());
        gams_spans_destroy(h, sp);
        check(h, rc);
        for (size_t k = 0; k < m; ++k) prop[s.q[k]] = out[k];
    });
}

std::vector<std::string> wave_proc_ctgs_multi(const std::vector<gams_gpu_t *> &handles, const std::vector<Ctg> &ctgs,
                                              const std::vector<const uint8_t *> &seqs, const WaveArgs &a,
                                              uint64_t batch_bytes) {
    if (handles.empty()) throw Error(GAMS_EINVAL, "wave_proc_ctgs_multi: no handles");
    if (ctgs.size() != seqs.size()) throw Error(GAMS_EINVAL, "wave_proc_ctgs_multi: ctgs/seqs size mismatch");
    const uint32_t G = (uint32_t)handles.size();
    std::vector<uint64_t> weight(ctgs.size());
    for (size_t c = 0; c < ctgs.size(); ++c) {
        const int64_t n = gams_window_count((int64_t)ctgs[c].chr_end - ctgs[c].chr_start + 1, a.size, a.step);
        weight[c] = n > 0 ? (uint64_t)n : 0;
    }
    const std::vector<uint32_t> owner = lpt_assign(weight, G);
    std::vector<std::string> out(ctgs.size());
    std::vector<std::exception_ptr> errs(G);
    auto work = [&](uint32_t g) {
        try {
            // this device's ctgs, in ctg order, cut into batches of <= batch_bytes bases
            std::vector<size_t> mine;
            for (size_t c = 0; c < ctgs.size(); ++c)
                if (owner[c] == g) mine.push_back(c);
            // cut this device's share into batches, keep two jobs in flight
            std::vector<std::pair<size_t, size_t>> batches;   // [b, e) into `mine`
            for (size_t b = 0; b < mine.size();) {
                uint64_t bytes = 0;
                size_t e = b;
                while (e < mine.size()) {
                    const uint64_t len = (uint64_t)(ctgs[mine[e]].chr_end - ctgs[mine[e]].chr_start + 1);
                    if (e > b && bytes + len > batch_bytes) break;
                    bytes += len;
                    ++e;
                }
                batches.emplace_back(b, e);
                b = e;
            }
            auto make_job = [&](size_t k) {
                std::vector<Ctg> bc;
                std::vector<const uint8_t *> bs;
                for (size_t i = batches[k].first; i < batches[k].second; ++i) {
                    bc.push_back(ctgs[mine[i]]);
                    bs.push_back(seqs[mine[i]]);
                }
                std::unique_ptr<WaveJob> job(new WaveJob(handles[g], std::move(bc), a));
                job->start(bs);
                return job;
            };
            std::unique_ptr<WaveJob> cur, next;
            if (!batches.empty()) cur = make_job(0);
            for (size_t k = 0; k < batches.size(); ++k) {
                if (k + 1 < batches.size()) next = make_job(k + 1);     // queued behind batch k on the device
                std::vector<std::string> rows = cur->finish();
                for (size_t i = batches[k].first; i < batches[k].second; ++i)
                    out[mine[i]] = std::move(rows[i - batches[k].first]);
                cur = std::move(next);
            }
        } catch (...) {
            errs[g] = std::current_exception();
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t g = 1; g < G; ++g) threads.emplace_back(work, g);
    work(0);
    for (auto &t : threads) t.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
    return out;
}

// ---------------------------------------------------------------------------
// sw
// ---------------------------------------------------------------------------
namespace {
// TSV text of the rows of ONE ctg (sw.rs:152-190), `rows` in the device's order (feature, then M, L1.., R1..)
std::string sw_format_rows(const gams_sw_row_t *rows, uint64_t nrows, const Ctg &ctg, const std::vector<Feature> &features,
                           unsigned max_threads) {
    static const char *TYPES[3] = {"M", "L", "R"};
    // text of rows [r0, r1), r0 on a feature boundary (the serial number restarts per feature)
    auto format = [&](uint64_t r0, uint64_t r1, std::string &o) {
        o.reserve((r1 - r0) * 80);
        uint32_t cur = UINT32_MAX, sn = 0;
        for (uint64_t r = r0; r < r1; ++r) {                            // sw.rs:152-190
            const gams_sw_row_t &w = rows[r];
            if (w.feature != cur) {
                cur = w.feature;
                sn = 1;                                                 // sw.rs:148
            }
            o += "sw:";                                                 // sw.rs:153
            o += features[w.feature].id;
            o += ':';
            o += std::to_string(sn++);
            o += '\t';
            o += ctg.chr_id;                                            // sw.rs:157 Range::from(chr, min, max)
            o += ':';
            o += runlist(w.start, w.end);
            o += '\t';
            o += TYPES[w.type];
            o += '\t';
            o += std::to_string(w.distance);
            o += '\t';
            o += fmt_f32(w.gc_content);                                 // data.rs:61-67
            o += '\t';
            o += fmt_f32(w.gc_mean);
            o += '\t';
            o += fmt_f32(w.gc_stddev);
            o += '\t';
            o += fmt_f32(w.gc_cv);
            o += "\t\n";                                                // empty rg_count (data.rs:71-80)
        }
    };
    std::string out;
    // a few host threads, each a run of whole features
    const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({max_threads, std::thread::hardware_concurrency(), nrows / 20000}));
    if (T <= 1) {
        format(0, nrows, out);
        return out;
    }
    std::vector<uint64_t> cut(T + 1, nrows);
    cut[0] = 0;
    for (unsigned t = 1; t < T; ++t) {
        uint64_t b = std::max(cut[t - 1], nrows * t / T);
        while (b < nrows && b > 0 && rows[b].feature == rows[b - 1].feature) ++b;
        cut[t] = b;
    }
    std::vector<std::string> part(T);
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back([&, t] { format(cut[t], cut[t + 1], part[t]); });
    format(cut[0], cut[1], part[0]);
    for (auto &th : pool) th.join();
    size_t total = 0;
    for (auto &x : part) total += x.size();
    out.reserve(total);
    for (auto &x : part) out += x;
    return out;
}
}  // namespace

std::string sw_proc_ctg(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq, const std::vector<Feature> &features,
                        const SwArgs &a) {
    const uint32_t nf = (uint32_t)features.size();
    if (nf == 0) return std::string();
    uint32_t len = (uint32_t)(ctg.chr_end - ctg.chr_start + 1);
    SeqSetGuard sg{h};
    check(h, gams_seqset_create(h, 1, &len, &sg.s));
    check(h, gams_seqset_upload(h, sg.s, 0, seq));
    std::vector<int32_t> fs(nf), fe(nf);
    for (uint32_t f = 0; f < nf; ++f) {
        fs[f] = features[f].start;
        fe[f] = features[f].end;
    }
    uint64_t nrows = 0;
    check(h, gams_gpu_sw(h, sg.s, 0, ctg.chr_start, fs.data(), fe.data(), nf, a.size, a.max, a.resize, nullptr, 0,
                         &nrows));
    std::vector<gams_sw_row_t> rows(nrows ? nrows : 1);
    check(h, gams_gpu_sw(h, sg.s, 0, ctg.chr_start, fs.data(), fe.data(), nf, a.size, a.max, a.resize,
                         rows.data(), nrows, &nrows));
    return sw_format_rows(rows.data(), nrows, ctg, features, 8);
}

// Several ctgs on one handle: their sequences go into one seqset per batch of <= batch_bytes bases, all
// features of a batch through ONE gams_gpu_sw_batch call (one upload, one gc index, one launch, one readback
// into page-locked memory), and the rows of the batch's ctgs are formatted on host threads, a ctg each.
std::vector<std::string> sw_proc_ctgs(gams_gpu_t *h, const std::vector<Ctg> &ctgs, const std::vector<const uint8_t *> &seqs,
                                      const std::vector<std::vector<Feature>> &features, const SwArgs &a,
                                      uint64_t batch_bytes) {
    if (ctgs.size() != seqs.size() || ctgs.size() != features.size())
        throw Error(GAMS_EINVAL, "sw_proc_ctgs: ctgs / seqs / features size mismatch");
    std::vector<std::string> out(ctgs.size());
    std::vector<size_t> todo;                                           // ctgs with features, in ctg order
    for (size_t c = 0; c < ctgs.size(); ++c)
        if (!features[c].empty()) todo.push_back(c);
    for (size_t b = 0; b < todo.size();) {
        uint64_t bytes = 0;
        size_t e = b;
        while (e < todo.size()) {
            const uint64_t len = (uint64_t)(ctgs[todo[e]].chr_end - ctgs[todo[e]].chr_start + 1);
            if (e > b && bytes + len > batch_bytes) break;
            bytes += len;
            ++e;
        }
        const uint32_t n = (uint32_t)(e - b);
        std::vector<uint32_t> lens(n), index(n);
        std::vector<const uint8_t *> ptrs(n);
        std::vector<int32_t> chr_start(n);
        std::vector<uint64_t> feat_off(n + 1, 0);
        for (uint32_t k = 0; k < n; ++k) {
            const Ctg &c = ctgs[todo[b + k]];
            lens[k] = (uint32_t)(c.chr_end - c.chr_start + 1);
            ptrs[k] = seqs[todo[b + k]];
            index[k] = k;
            chr_start[k] = c.chr_start;
            feat_off[k + 1] = feat_off[k] + features[todo[b + k]].size();
        }
        std::vector<int32_t> fs(feat_off[n]), fe(feat_off[n]);
        for (uint32_t k = 0; k < n; ++k) {
            const std::vector<Feature> &fv = features[todo[b + k]];
            for (size_t f = 0; f < fv.size(); ++f) {
                fs[feat_off[k] + f] = fv[f].start;
                fe[feat_off[k] + f] = fv[f].end;
            }
        }
        SeqSetGuard sg{h};
        check(h, gams_seqset_create(h, n, lens.data(), &sg.s));
        check(h, gams_seqset_upload_all(h, sg.s, ptrs.data()));
        std::vector<uint64_t> row_off(n + 1, 0);
        uint64_t nrows = 0;
        {
            // the rows' text straight from the device (gams_gpu_sw_text); values it does not cover send the batch
            // down the host formatter below
            std::vector<const char *> chr(n), ids(feat_off[n]);
            for (uint32_t k = 0; k < n; ++k) {
                chr[k] = ctgs[todo[b + k]].chr_id.c_str();
                const std::vector<Feature> &fv = features[todo[b + k]];
                for (size_t f = 0; f < fv.size(); ++f) ids[feat_off[k] + f] = fv[f].id.c_str();
            }
            const char *text = nullptr;
            uint64_t tbytes = 0;
            const uint64_t *toff = nullptr;
            const int rc = gams_gpu_sw_text(h, sg.s, n, index.data(), chr.data(), chr_start.data(), feat_off.data(), fs.data(),
                                            fe.data(), ids.data(), a.size, a.max, a.resize, &text, &tbytes, &toff, &nrows);
            if (rc == GAMS_OK) {
                // proc_ctg's Strings: slices of the page-locked text, copied by a few host threads (one thread moves
                // ~10 GB/s; 313 MB for the 4.1 M rows of a 30-Mb chromosome)
                const unsigned T = (unsigned)std::max<uint64_t>(
                    1, std::min<uint64_t>({16, std::thread::hardware_concurrency(), (uint64_t)n, tbytes / (4u << 20) + 1}));
                std::atomic<uint32_t> next{0};
                std::vector<std::exception_ptr> errs(T);
                auto work = [&](unsigned t) {
                    try {
                        for (uint32_t k = next.fetch_add(1); k < n; k = next.fetch_add(1))
                            out[todo[b + k]].assign(text + toff[k], text + toff[k + 1]);
                    } catch (...) {
                        errs[t] = std::current_exception();
                    }
                };
                std::vector<std::thread> pool;
                for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
                work(0);
                for (auto &th : pool) th.join();
                for (auto &er : errs)
                    if (er) std::rethrow_exception(er);
                b = e;
                continue;
            }
            if (rc != GAMS_EUNSUPPORTED) check(h, rc);
        }
        check(h, gams_gpu_sw_batch(h, sg.s, n, index.data(), chr_start.data(), feat_off.data(), fs.data(), fe.data(),
                                   a.size, a.max, a.resize, nullptr, 0, row_off.data(), &nrows));
        // rows land in page-locked memory: the readback runs at the rate of the link
        struct Pinned {
            gams_gpu_t *h;
            void *p = nullptr;
            ~Pinned() { if (p) gams_gpu_host_free(h, p); }
        } pin{h};
        check(h, gams_gpu_host_alloc(h, std::max<uint64_t>(nrows, 1) * sizeof(gams_sw_row_t), &pin.p));
        gams_sw_row_t *rows = static_cast<gams_sw_row_t *>(pin.p);
        check(h, gams_gpu_sw_batch(h, sg.s, n, index.data(), chr_start.data(), feat_off.data(), fs.data(), fe.data(),
                                   a.size, a.max, a.resize, rows, nrows, row_off.data(), &nrows));
        // format: ctgs of the batch dealt to a few host threads
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>({16, std::thread::hardware_concurrency(), (uint64_t)n}));
        std::atomic<uint32_t> next{0};
        std::vector<std::exception_ptr> errs(T);
        auto work = [&](unsigned t) {
            try {
                for (uint32_t k = next.fetch_add(1); k < n; k = next.fetch_add(1))
                    out[todo[b + k]] = sw_format_rows(rows + row_off[k], row_off[k + 1] - row_off[k], ctgs[todo[b + k]],
                                                      features[todo[b + k]], T > 1 ? 1 : 8);
            } catch (...) {
                errs[t] = std::current_exception();
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < T; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &th : pool) th.join();
        for (auto &er : errs)
            if (er) std::rethrow_exception(er);
        b = e;
    }
    return out;
}

// ---------------------------------------------------------------------------
// locate
// ---------------------------------------------------------------------------
Locator::Locator(gams_gpu_t *h, const std::vector<Ctg> &ctgs) : h_(h), ctgs_(ctgs) {
    // redis.rs:236-258: per chr a Lapper of (chr_start, chr_end + 1, ctg_id)
    std::map<std::string, std::vector<uint32_t>> by_chr;
    for (uint32_t i = 0; i < ctgs_.size(); ++i) {
        by_chr[ctgs_[i].chr_id].push_back(i);
        ctg_slot_[ctgs_[i].id] = i;
    }
    std::vector<uint64_t> off{0};
    std::vector<uint32_t> st, sp;
    std::vector<Ctg> ordered;
    uint32_t g = 0;
    for (auto &kv : by_chr) {
        chr_group_[kv.first] = g++;
        for (uint32_t i : kv.second) {
            st.push_back((uint32_t)ctgs_[i].chr_start);
            sp.push_back((uint32_t)ctgs_[i].chr_end + 1u);
            ordered.push_back(ctgs_[i]);
        }
        off.push_back(st.size());
    }
    ctgs_ = ordered;  // hit indices refer to this order
    ctg_slot_.clear();
    for (uint32_t i = 0; i < ctgs_.size(); ++i) ctg_slot_[ctgs_[i].id] = i;
    check(h_, gams_index_create(h_, g, off.data(), st.data(), sp.data(), &ctg_ix_));
}

Locator::~Locator() {
    if (ctg_ix_) gams_index_destroy(h_, ctg_ix_);
    if (rg_ix_) gams_index_destroy(h_, rg_ix_);
}

const Ctg *Locator::ctg(const std::string &id) const {
    auto it = ctg_slot_.find(id);
    return it == ctg_slot_.end() ? nullptr : &ctgs_[it->second];
}

void Locator::set_rg_index(const std::map<std::string, std::vector<Range>> &rg_of_ctg) {
    if (rg_ix_) {
        gams_index_destroy(h_, rg_ix_);
        rg_ix_ = nullptr;
    }
    rg_group_.clear();
    std::vector<uint64_t> off{0};
    std::vector<uint32_t> st, sp;
    uint32_t g = 0;
    for (auto &kv : rg_of_ctg) {                                        // redis.rs:288-299
        rg_group_[kv.first] = g++;
        for (const Range &r : kv.second) {
            st.push_back((uint32_t)r.start);
            sp.push_back((uint32_t)r.end + 1u);
        }
        off.push_back(st.size());
    }
    check(h_, gams_index_create(h_, g, off.data(), st.data(), sp.data(), &rg_ix_));
}

std::vector<std::string> Locator::find(const std::vector<Range> &rgs) {
    const uint64_t nq = rgs.size();
    std::vector<uint32_t> grp(nq), qs(nq), qe(nq);
    for (uint64_t i = 0; i < nq; ++i) {
        auto it = chr_group_.find(rgs[i].chr);                          // utils.rs:11-13
        grp[i] = it == chr_group_.end() ? UINT32_MAX : it->second;
        qs[i] = (uint32_t)rgs[i].start;                                 // utils.rs:16: find(start, end)
        qe[i] = (uint32_t)rgs[i].end;
    }
    std::vector<int64_t> hit(nq ? nq : 1);
    check(h_, gams_gpu_locate(h_, ctg_ix_, grp.data(), qs.data(), qe.data(), nq, hit.data()));
    std::vector<std::string> out(nq);
    for (uint64_t i = 0; i < nq; ++i)
        if (hit[i] >= 0) out[i] = ctgs_[(size_t)hit[i]].id;
    return out;
}

std::string Locator::locate(const std::vector<std::string> &rgs, bool is_count) {
    // Range strings are parsed and rows are formatted on several host threads (contiguous shares of the lines, joined
    // in order); the lookups in between are one device call each.  Nothing string-valued is kept per line between the
    // stages: a share keeps (chromosome group, start, end, line number) of its valid ranges, the ctg of a hit is an index
    // into the ctg table, and the rg group of a ctg is a table by that index (1e6 lines: 169 -> ~40 ms, most of it
    // Range::from_str).
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), rgs.size() / 20000}));
    auto cut = [&](size_t n, unsigned t) { return n * t / T; };
    struct Share {
        std::vector<uint32_t> grp, qs, qe;
        std::vector<size_t> src;
        size_t first = 0;               // position of the share's first valid range in the joined arrays
        std::string out;
    };
    std::vector<Share> sh(T);
    run_shares(T, [&](uint32_t t) {
        Share &m = sh[t];
        const std::string *last_chr = nullptr;
        uint32_t last_grp = UINT32_MAX;
        std::string keep;
        for (size_t i = cut(rgs.size(), t); i < cut(rgs.size(), t + 1); ++i) {   // locate.rs:111-116
            const Range r = Range::from_str(rgs[i]);
            if (!r.valid) continue;
            if (!last_chr || r.chr != keep) {                                   // utils.rs:11-13 (runs of one chromosome are the rule)
                auto it = chr_group_.find(r.chr);
                last_grp = it == chr_group_.end() ? UINT32_MAX : it->second;
                keep = r.chr;
                last_chr = &keep;
            }
            m.grp.push_back(last_grp);
            m.qs.push_back((uint32_t)r.start);                                  // utils.rs:16: find(start, end)
            m.qe.push_back((uint32_t)r.end);
            m.src.push_back(i);
        }
    });
    size_t nq = 0;
    for (Share &m : sh) {
        m.first = nq;
        nq += m.src.size();
    }
    std::vector<uint32_t> grp(nq), qs(nq), qe(nq);
    run_shares(T, [&](uint32_t t) {
        const Share &m = sh[t];
        std::copy(m.grp.begin(), m.grp.end(), grp.begin() + m.first);
        std::copy(m.qs.begin(), m.qs.end(), qs.begin() + m.first);
        std::copy(m.qe.begin(), m.qe.end(), qe.begin() + m.first);
    });
    std::vector<int64_t> hit(nq ? nq : 1);
    check(h_, gams_gpu_locate(h_, ctg_ix_, grp.data(), qs.data(), qe.data(), nq, hit.data()));
    auto join = [&] {
        std::string out;
        size_t bytes = 0;
        for (auto &m : sh) bytes += m.out.size();
        out.reserve(bytes);
        for (auto &m : sh) out += m.out;
        return out;
    };
    if (!is_count) {
        run_shares(T, [&](uint32_t t) {
            Share &m = sh[t];
            for (size_t j = 0; j < m.src.size(); ++j) {
                const int64_t c = hit[m.first + j];
                if (c < 0) continue;                                            // locate.rs:120-122
                m.out += rgs[m.src[j]];
                m.out += '\t';
                m.out += ctgs_[(size_t)c].id;                                   // locate.rs:139
                m.out += '\n';
            }
        });
        return join();
    }
    if (!rg_ix_) throw Error(GAMS_ESTATE, "locate --count: no rg index loaded");
    // rg group of every ctg of the table (UINT32_MAX: the ctg has no entry in the index)
    std::vector<uint32_t> group_of(ctgs_.size(), UINT32_MAX);
    for (size_t c = 0; c < ctgs_.size(); ++c) {
        auto it = rg_group_.find(ctgs_[c].id);
        if (it != rg_group_.end()) group_of[c] = it->second;
    }
    // the located ranges, in line order: the reference counts only those (locate.rs:120-122, :135-137)
    std::vector<size_t> kept(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) {
        size_t n = 0;
        for (size_t j = 0; j < sh[t].src.size(); ++j) n += hit[sh[t].first + j] >= 0;
        kept[t + 1] = kept[t] + n;
    }
    const size_t nk = kept[T];
    std::vector<uint32_t> cg(nk), cs(nk), ce(nk);
    run_shares(T, [&](uint32_t t) {
        const Share &m = sh[t];
        size_t o = kept[t];
        for (size_t j = 0; j < m.src.size(); ++j) {
            const int64_t c = hit[m.first + j];
            if (c < 0) continue;
            cg[o] = group_of[(size_t)c];
            cs[o] = qs[m.first + j];
            ce[o] = qe[m.first + j];
            ++o;
        }
    });
    for (size_t o = 0; o < nk; ++o)                                             // utils.rs:30 (rare: serial is fine)
        if (cg[o] == UINT32_MAX) {
            // (which ctg: recomputed only on this path)
            size_t seen = 0;
            for (unsigned t = 0; t < T && seen <= o; ++t)
                for (size_t j = 0; j < sh[t].src.size(); ++j) {
                    const int64_t c = hit[sh[t].first + j];
                    if (c < 0) continue;
                    if (seen++ == o) fprintf(stderr, "%s not found in idx\n", ctgs_[(size_t)c].id.c_str());
                }
        }
    std::vector<int32_t> cnt(nk ? nk : 1);
    check(h_, gams_gpu_count(h_, rg_ix_, cg.data(), cs.data(), ce.data(), nk, cnt.data()));
    run_shares(T, [&](uint32_t t) {
        Share &m = sh[t];
        size_t o = kept[t];
        char num[16];
        for (size_t j = 0; j < m.src.size(); ++j) {
            if (hit[m.first + j] < 0) continue;
            m.out += rgs[m.src[j]];
            m.out += '\t';
            auto r = std::to_chars(num, num + sizeof num, cnt[o++]);                // locate.rs:137
            m.out.append(num, r.ptr);
            m.out += '\n';
        }
    });
    return join();
}

std::string Locator::locate_seq(const std::vector<std::string> &rgs,
                                const std::map<std::string, std::string> &seq_of) {
    std::vector<Range> valid;
    std::vector<size_t> src;
    for (size_t i = 0; i < rgs.size(); ++i) {
        Range r = Range::from_str(rgs[i]);
        if (!r.valid) continue;
        r.strand.clear();
        valid.push_back(r);
        src.push_back(i);
    }
    std::vector<std::string> ctg_ids = find(valid);
    std::string out;
    for (size_t k = 0; k < valid.size(); ++k) {
        if (ctg_ids[k].empty()) continue;
        const Ctg *c = ctg(ctg_ids[k]);
        auto it = seq_of.find(ctg_ids[k]);
        if (!c || it == seq_of.end()) throw Error(GAMS_EINVAL, "locate --seq: no sequence for " + ctg_ids[k]);
        const int64_t from = (int64_t)valid[k].start - c->chr_start + 1;   // locate.rs:128-129
        const int64_t to = (int64_t)valid[k].end - c->chr_start + 1;
        if (from < 1 || to > (int64_t)it->second.size() || to < from)
            throw Error(GAMS_EINVAL, "locate --seq: range outside " + ctg_ids[k] +
                                         " (the reference panics, locate.rs:133)");
        out += ">" + rgs[src[k]] + "\n" + it->second.substr((size_t)from - 1, (size_t)(to - from + 1)) + "\n";
    }
    return out;
}

// ---------------------------------------------------------------------------
// loaders' bucketing (utils.rs:39-67) and the gzip framing of `seq:` (redis.rs:149-161)
// ---------------------------------------------------------------------------
std::map<std::string, std::vector<Range>> read_range(Locator &loc, const std::vector<std::string> &lines) {
    std::vector<Range> valid;
    for (const std::string &ln : lines) {
        Range r = Range::from_str(ln);                                  // utils.rs:50-53
        if (r.valid) valid.push_back(r);
    }
    std::vector<std::string> ids = loc.find(valid);                     // utils.rs:55 (strand is not used by the lookup)
    std::map<std::string, std::vector<Range>> ranges_of;
    for (size_t k = 0; k < valid.size(); ++k) {
        if (ids[k].empty()) continue;                                   // utils.rs:56-58
        auto it = ranges_of.find(ids[k]);
        if (it == ranges_of.end())
            ranges_of[ids[k]];                                          // or_default(): the first range is dropped
        else
            it->second.push_back(valid[k]);                             // and_modify(push)
    }
    return ranges_of;
}

std::map<std::string, std::vector<std::pair<Range, std::string>>> read_peak(Locator &loc,
                                                                          const std::vector<std::string> &lines) {
    std::vector<Range> valid;
    std::vector<std::string> sig;
    for (const std::string &ln : lines) {
        std::vector<std::string> parts;
        size_t b = 0;
        for (;;) {
            size_t e = ln.find('\t', b);
            parts.push_back(ln.substr(b, e == std::string::npos ? std::string::npos : e - b));
            if (e == std::string::npos) break;
            b = e + 1;
        }
        Range r = Range::from_str(parts[0]);                            // utils.rs:96-99
        if (!r.valid) continue;
        if (parts.size() < 3) throw Error(GAMS_EINVAL, "read_peak: a row has no signal column (the reference panics, utils.rs:102)");
        r.strand.clear();                                               // utils.rs:100
        valid.push_back(r);
        sig.push_back(parts[2]);
    }
    std::vector<std::string> ids = loc.find(valid);
    std::map<std::string, std::vector<std::pair<Range, std::string>>> peaks_of;
    for (size_t k = 0; k < valid.size(); ++k) {
        if (ids[k].empty()) continue;
        auto it = peaks_of.find(ids[k]);
        if (it == peaks_of.end())
            peaks_of[ids[k]];                                           // utils.rs:109-112: first one dropped
        else
            it->second.emplace_back(valid[k], sig[k]);
    }
    return peaks_of;
}

namespace {
void peak_check_inside(const Ctg &ctg, const std::vector<std::pair<Range, std::string>> &peaks) {
    for (const auto &pk : peaks)
        if (pk.first.start < ctg.chr_start || pk.first.end > ctg.chr_end || pk.first.end < pk.first.start)
            throw Error(GAMS_EINVAL, "peak: " + pk.first.to_string() + " is not inside " + ctg.id +
                                         " (the reference panics on the slice, utils.rs:155)");
}

// the Peak records of one ctg from its merged ranges and their gc (peak.rs:65-158)
std::vector<Peak> peak_fill(const Ctg &ctg, const std::vector<std::pair<Range, std::string>> &peaks, const float *gc) {
    std::vector<Peak> out(peaks.size());
    if (peaks.empty()) return out;
    for (size_t i = 0; i < peaks.size(); ++i) {                         // peak.rs:65-95
        Peak &p = out[i];
        p.id = "peak:" + ctg.id + ":" + std::to_string(i + 1);
        p.range = peaks[i].first.to_string();
        p.length = peaks[i].first.end - peaks[i].first.start + 1;
        p.signal = peaks[i].second;
        p.gc = gc[i];
    }
    // left (peak.rs:112-133)
    std::string prev_signal = out.front().signal;
    float prev_gc = out.front().gc;
    int32_t prev_end = ctg.chr_start;
    for (size_t i = 0; i < out.size(); ++i) {
        out[i].left_wave_length = peaks[i].first.start - prev_end + 1;
        out[i].left_amplitude = std::fabs(out[i].gc - prev_gc);
        out[i].left_signal = prev_signal;
        prev_signal = out[i].signal;
        prev_end = peaks[i].first.end;
        prev_gc = out[i].gc;
    }
    // right (peak.rs:135-157)
    std::string next_signal = out.back().signal;
    float next_gc = out.back().gc;
    int32_t next_start = ctg.chr_end;
    for (size_t i = out.size(); i-- > 0;) {
        out[i].right_wave_length = next_start - peaks[i].first.end + 1;
        out[i].right_amplitude = std::fabs(out[i].gc - next_gc);
        out[i].right_signal = next_signal;
        next_signal = out[i].signal;
        next_start = peaks[i].first.start;
        next_gc = out[i].gc;
    }
    return out;
}
}  // namespace

std::vector<Peak> peak_records(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq,
                               const std::vector<std::pair<Range, std::string>> &peaks) {
    if (peaks.empty()) return {};
    peak_check_inside(ctg, peaks);
    uint32_t len = (uint32_t)(ctg.chr_end - ctg.chr_start + 1);
    SeqSetGuard sg{h};
    check(h, gams_seqset_create(h, 1, &len, &sg.s));
    check(h, gams_seqset_upload(h, sg.s, 0, seq));
    std::vector<int32_t> rs(peaks.size()), re(peaks.size());
    for (size_t i = 0; i < peaks.size(); ++i) {
        rs[i] = peaks[i].first.start;
        re[i] = peaks[i].first.end;
    }
    std::vector<float> gc(peaks.size());
    check(h, gams_gpu_range_gc(h, sg.s, 0, ctg.chr_start, rs.data(), re.data(), (uint32_t)peaks.size(), gc.data()));
    return peak_fill(ctg, peaks, gc.data());
}

// several ctgs: one seqset per batch of <= batch_bytes bases, the merged ranges of all of its ctgs through ONE
// gams_gpu_range_gc_batch call (a ctg's few hundred ranges are one or two workgroups)
std::vector<std::vector<Peak>> peak_records_batch(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                                  const std::vector<const uint8_t *> &seqs,
                                                  const std::vector<std::vector<std::pair<Range, std::string>>> &peaks,
                                                  uint64_t batch_bytes) {
    if (ctgs.size() != seqs.size() || ctgs.size() != peaks.size())
        throw Error(GAMS_EINVAL, "peak_records_batch: ctgs / seqs / peaks size mismatch");
    std::vector<std::vector<Peak>> out(ctgs.size());
    std::vector<size_t> todo;
    for (size_t c = 0; c < ctgs.size(); ++c)
        if (!peaks[c].empty()) {
            peak_check_inside(ctgs[c], peaks[c]);
            todo.push_back(c);
        }
    for (size_t b = 0; b < todo.size();) {
        uint64_t bytes = 0;
        size_t e = b;
        while (e < todo.size()) {
            const uint64_t len = (uint64_t)(ctgs[todo[e]].chr_end - ctgs[todo[e]].chr_start + 1);
            if (e > b && bytes + len > batch_bytes) break;
            bytes += len;
            ++e;
        }
        const uint32_t n = (uint32_t)(e - b);
        std::vector<uint32_t> lens(n), index(n);
        std::vector<const uint8_t *> ptrs(n);
        std::vector<int32_t> chr_start(n);
        std::vector<uint64_t> off(n + 1, 0);
        for (uint32_t k = 0; k < n; ++k) {
            const Ctg &c = ctgs[todo[b + k]];
            lens[k] = (uint32_t)(c.chr_end - c.chr_start + 1);
            ptrs[k] = seqs[todo[b + k]];
            index[k] = k;
            chr_start[k] = c.chr_start;
            off[k + 1] = off[k] + peaks[todo[b + k]].size();
        }
        std::vector<int32_t> rs(off[n]), re(off[n]);
        for (uint32_t k = 0; k < n; ++k) {
            const auto &pv = peaks[todo[b + k]];
            for (size_t i = 0; i < pv.size(); ++i) {
                rs[off[k] + i] = pv[i].first.start;
                re[off[k] + i] = pv[i].first.end;
            }
        }
        SeqSetGuard sg{h};
        check(h, gams_seqset_create(h, n, lens.data(), &sg.s));
        check(h, gams_seqset_upload_all(h, sg.s, ptrs.data()));
        std::vector<float> gc(off[n]);
        check(h, gams_gpu_range_gc_batch(h, sg.s, n, index.data(), chr_start.data(), off.data(), rs.data(), re.data(),
                                         gc.data()));
        for (uint32_t k = 0; k < n; ++k) out[todo[b + k]] = peak_fill(ctgs[todo[b + k]], peaks[todo[b + k]], gc.data() + off[k]);
        b = e;
    }
    return out;
}

namespace {
std::string json_escape(const std::string &v) {
    std::string o;
    for (char c : v) {
        if (c == '"' || c == '\\') o += '\\';
        o += c;
    }
    return o;
}
}  // namespace

std::string ctg_json(const Ctg &c) {
    return "{\"id\":\"" + json_escape(c.id) + "\",\"range\":\"" + json_escape(c.range) + "\",\"chr_id\":\"" +
           json_escape(c.chr_id) + "\",\"chr_start\":" + std::to_string(c.chr_start) + ",\"chr_end\":" +
           std::to_string(c.chr_end) + ",\"chr_strand\":\"" + json_escape(c.chr_strand) + "\",\"length\":" +
           std::to_string(c.length) + "}";
}

Ctg ctg_from_json(const std::string &json) {
    // flat object of strings and integers, any field order, whitespace tolerated
    Ctg c;
    size_t pos = 0;
    auto skip = [&] {
        while (pos < json.size() && (json[pos] == ' ' || json[pos] == '\n' || json[pos] == '\t' || json[pos] == '\r')) ++pos;
    };
    auto str = [&]() {
        std::string v;
        if (pos >= json.size() || json[pos] != '"') throw Error(GAMS_EINVAL, "ctg_from_json: expected a string");
        for (++pos; pos < json.size() && json[pos] != '"'; ++pos) {
            if (json[pos] == '\\' && pos + 1 < json.size()) ++pos;
            v += json[pos];
        }
        if (pos >= json.size()) throw Error(GAMS_EINVAL, "ctg_from_json: unterminated string");
        ++pos;
        return v;
    };
    skip();
    if (pos >= json.size() || json[pos] != '{') throw Error(GAMS_EINVAL, "ctg_from_json: not an object");
    ++pos;
    for (;;) {
        skip();
        if (pos < json.size() && json[pos] == '}') break;
        const std::string key = str();
        skip();
        if (pos >= json.size() || json[pos] != ':') throw Error(GAMS_EINVAL, "ctg_from_json: expected ':'");
        ++pos;
        skip();
        if (pos < json.size() && json[pos] == '"') {
            const std::string v = str();
            if (key == "id") c.id = v;
            else if (key == "range") c.range = v;
            else if (key == "chr_id") c.chr_id = v;
            else if (key == "chr_strand") c.chr_strand = v;
        } else {
            size_t e = pos;
            while (e < json.size() && (json[e] == '-' || (json[e] >= '0' && json[e] <= '9'))) ++e;
            if (e == pos) throw Error(GAMS_EINVAL, "ctg_from_json: expected a number");
            const int32_t v = (int32_t)std::stol(json.substr(pos, e - pos));
            pos = e;
            if (key == "chr_start") c.chr_start = v;
            else if (key == "chr_end") c.chr_end = v;
            else if (key == "length") c.length = v;
        }
        skip();
        if (pos < json.size() && json[pos] == ',') ++pos;
    }
    if (c.id.empty() || c.chr_id.empty()) throw Error(GAMS_EINVAL, "ctg_from_json: id / chr_id missing");
    return c;
}

std::string tsv_ctgs(const std::vector<Ctg> &ctgs) {
    std::string out = "id\trange\tchr_id\tchr_start\tchr_end\tchr_strand\tlength\n";   // data.rs:5-14
    for (const Ctg &c : ctgs)
        out += c.id + "\t" + c.range + "\t" + c.chr_id + "\t" + std::to_string(c.chr_start) + "\t" +
               std::to_string(c.chr_end) + "\t" + c.chr_strand + "\t" + std::to_string(c.length) + "\n";
    return out;
}

std::string tsv_records(const std::vector<Record> &records, bool features) {
    std::string out = features ? "id\trange\tlength\ttag\n" : "id\trange\n";           // data.rs:16-28
    for (const Record &r : records) {
        // values of the flat JSON object this library wrote itself, in field order
        std::string row;
        size_t pos = 0;
        while ((pos = r.json.find("\":", pos)) != std::string::npos) {
            pos += 2;
            std::string v;
            if (r.json[pos] == '"') {
                for (++pos; pos < r.json.size() && r.json[pos] != '"'; ++pos) {
                    if (r.json[pos] == '\\' && pos + 1 < r.json.size()) ++pos;
                    v += r.json[pos];
                }
            } else {
                while (pos < r.json.size() && r.json[pos] != ',' && r.json[pos] != '}') v += r.json[pos++];
            }
            if (!row.empty()) row += '\t';
            row += v;
        }
        out += row + "\n";
    }
    return out;
}

std::vector<Record> rg_records(Locator &loc, const std::vector<std::string> &lines) {
    std::vector<Record> out;
    for (auto &kv : read_range(loc, lines)) {                           // BTreeMap order = ctg id order
        int32_t serial = 0;
        for (const Range &r : kv.second) {
            const std::string id = "rg:" + kv.first + ":" + std::to_string(++serial);   // rg.rs:64-66
            out.push_back({id, "{\"id\":\"" + json_escape(id) + "\",\"range\":\"" + json_escape(r.to_string()) + "\"}"});
        }
    }
    return out;
}

std::vector<Record> feature_records(Locator &loc, const std::vector<std::string> &lines, const std::string &tag) {
    std::vector<Record> out;
    for (auto &kv : read_range(loc, lines)) {
        int32_t serial = 0;
        for (const Range &r : kv.second) {
            const std::string id = "feature:" + kv.first + ":" + std::to_string(++serial);   // feature.rs:81-83
            out.push_back({id, "{\"id\":\"" + json_escape(id) + "\",\"range\":\"" + json_escape(r.to_string()) +
                                   "\",\"length\":" + std::to_string(r.end - r.start + 1) + ",\"tag\":\"" +
                                   json_escape(tag) + "\"}"});                               // feature.rs:85-92
        }
    }
    return out;
}

namespace {
// libdeflate (the runtime library ships with the image: libdeflate.so.0, no headers) inflates DNA text about
// three times as fast as zlib; bound by hand to its stable C API, zlib when it is absent.
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*gunzip)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    LibDeflate() {
        if (void *so = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL)) {
            alloc = reinterpret_cast<void *(*)()>(dlsym(so, "libdeflate_alloc_decompressor"));
            gunzip = reinterpret_cast<int (*)(void *, const void *, size_t, void *, size_t, size_t *)>(
                dlsym(so, "libdeflate_gzip_decompress"));
            release = reinterpret_cast<void (*)(void *)>(dlsym(so, "libdeflate_free_decompressor"));
            if (!alloc || !gunzip || !release) alloc = nullptr;
        }
    }
    bool ok() const { return alloc != nullptr; }
};
const LibDeflate &libdeflate() {
    static const LibDeflate L;
    return L;
}
struct Decompressor {   // one per thread, reused
    void *d = nullptr;
    ~Decompressor() {
        if (d) libdeflate().release(d);
    }
};

size_t zlib_gunzip_into(const uint8_t *bytes, size_t n, uint8_t *dst, size_t cap, bool *overflow) {
    z_stream zs{};
    if (inflateInit2(&zs, 15 + 16) != Z_OK) throw Error(GAMS_EINVAL, "decode_gz: inflateInit2 failed");
    zs.next_in = const_cast<Bytef *>(bytes);
    zs.avail_in = (uInt)n;
    size_t at = 0;
    int rc;
    do {
        zs.next_out = dst + at;
        zs.avail_out = (uInt)std::min<size_t>(cap - at, 1u << 30);
        const size_t room = zs.avail_out;
        rc = inflate(&zs, Z_NO_FLUSH);
        at += room - zs.avail_out;
        if (rc != Z_OK && rc != Z_STREAM_END) {
            inflateEnd(&zs);
            throw Error(GAMS_EINVAL, "decode_gz: corrupt gzip member");
        }
        if (rc == Z_OK && at == cap) {          // output full before the member ended
            inflateEnd(&zs);
            *overflow = true;
            return at;
        }
    } while (rc != Z_STREAM_END);
    inflateEnd(&zs);
    return at;
}
}  // namespace

// The first gzip member of `bytes` (flate2's GzDecoder reads one member, redis.rs:156-161) inflated into
// dst[0, cap); returns the bases written.  A member longer than cap is an error.
size_t decode_gz_into(const uint8_t *bytes, size_t n, uint8_t *dst, size_t cap) {
    if (n < 18) throw Error(GAMS_EINVAL, "decode_gz: corrupt gzip member");
    if (libdeflate().ok()) {
        static thread_local Decompressor dc;
        if (!dc.d) dc.d = libdeflate().alloc();
        if (dc.d) {
            size_t got = 0;
            const int rc = libdeflate().gunzip(dc.d, bytes, n, dst, cap, &got);
            if (rc == 0) return got;
            if (rc == 3) throw Error(GAMS_EINVAL, "decode_gz: the member inflates to more than " + std::to_string(cap) + " bytes");
            throw Error(GAMS_EINVAL, "decode_gz: corrupt gzip member");
        }
    }
    bool overflow = false;
    const size_t got = zlib_gunzip_into(bytes, n, dst, cap, &overflow);
    if (overflow) throw Error(GAMS_EINVAL, "decode_gz: the member inflates to more than " + std::to_string(cap) + " bytes");
    return got;
}

std::string decode_gz(const uint8_t *bytes, size_t n) {
    if (n < 18) throw Error(GAMS_EINVAL, "decode_gz: corrupt gzip member");
    // ISIZE (RFC 1952: uncompressed size mod 2^32 in the last four bytes) sizes the output in one piece for a
    // single-member value, which is what the reference stores; anything else grows the buffer
    uint32_t isize;
    std::memcpy(&isize, bytes + n - 4, 4);
    std::string out;
    // (deflate cannot expand beyond 1032:1, so a trailer claiming more is not believed)
    size_t cap = std::max<size_t>(std::min<uint64_t>(isize, (uint64_t)n * 1032u + 64u), 64);
    for (int attempt = 0; attempt < 40; ++attempt) {
        out.resize(cap);
        if (libdeflate().ok()) {
            static thread_local Decompressor dc;
            if (!dc.d) dc.d = libdeflate().alloc();
            if (dc.d) {
                size_t got = 0;
                const int rc = libdeflate().gunzip(dc.d, bytes, n, &out[0], cap, &got);
                if (rc == 0) {
                    out.resize(got);
                    return out;
                }
                if (rc != 3) throw Error(GAMS_EINVAL, "decode_gz: corrupt gzip member");
                cap *= 2;
                continue;
            }
        }
        bool overflow = false;
        const size_t got = zlib_gunzip_into(bytes, n, reinterpret_cast<uint8_t *>(&out[0]), cap, &overflow);
        if (!overflow) {
            out.resize(got);
            return out;
        }
        cap *= 2;
    }
    throw Error(GAMS_EINVAL, "decode_gz: output too large");
}

// get_seq + decode_gz for a batch of values on `threads` host threads, one ctg per worker at a time
// (redis.rs:142-161 inside the workers of wave.rs:288-299)
std::vector<std::string> decode_gz_many(const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len,
                                        unsigned threads) {
    if (blobs.size() != blob_len.size()) throw Error(GAMS_EINVAL, "decode_gz_many: blobs/lengths size mismatch");
    const uint32_t n = (uint32_t)blobs.size();
    std::vector<std::string> out(n);
    const unsigned T = std::max(1u, std::min(threads ? threads : 16u, std::max(n, 1u)));
    std::atomic<uint32_t> next{0};
    std::exception_ptr err;
    std::mutex mu;
    auto work = [&] {
        try {
            for (uint32_t i = next.fetch_add(1); i < n; i = next.fetch_add(1))
                out[i] = decode_gz(blobs[i], (size_t)blob_len[i]);
        } catch (...) {
            std::lock_guard<std::mutex> lk(mu);
            if (!err) err = std::current_exception();
            next.store(n);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < T; ++t) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    if (err) std::rethrow_exception(err);
    return out;
}

std::string encode_gz(const uint8_t *bytes, size_t n) {
    z_stream zs{};
    if (deflateInit2(&zs, Z_BEST_SPEED, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
        throw Error(GAMS_EINVAL, "encode_gz: deflateInit2 failed");
    zs.next_in = const_cast<Bytef *>(bytes);
    zs.avail_in = (uInt)n;
    std::string out;
    char buf[1 << 16];
    int rc;
    do {
        zs.next_out = reinterpret_cast<Bytef *>(buf);
        zs.avail_out = sizeof buf;
        rc = deflate(&zs, Z_FINISH);
        out.append(buf, sizeof buf - zs.avail_out);
    } while (rc != Z_STREAM_END);
    deflateEnd(&zs);
    return out;
}

// ---------------------------------------------------------------------------
// gen
// ---------------------------------------------------------------------------
std::vector<Ctg> gen_ctgs(gams_gpu_t *h, const std::string &chr_id, const uint8_t *seq, uint64_t len,
                          const GenArgs &a) {
    // one scan of the chromosome; a second only if it has more valid regions than a generous first guess
    uint64_t n = 0;
    std::vector<int32_t> lo(4096), hi(4096);
    check(h, gams_gpu_valid_spans(h, seq, len, a.fill, a.min, lo.data(), hi.data(), lo.size(), &n));
    if (n > lo.size()) {
        lo.resize(n);
        hi.resize(n);
        check(h, gams_gpu_valid_spans(h, seq, len, a.fill, a.min, lo.data(), hi.data(), n, &n));
    }
    std::vector<Ctg> out;
    int32_t serial = 0;
    for (uint64_t i = 0; i < n; ++i) {                                 // gen.rs:108-126
        int64_t pos = lo[i];
        const int64_t max = hi[i];
        std::vector<std::pair<int64_t, int64_t>> cur;
        while (max - pos + 1 > a.piece) {
            cur.emplace_back(pos, pos + a.piece - 1);
            pos += a.piece;
        }
        if (cur.empty())
            cur.emplace_back(pos, max);
        else
            cur.back().second = max;                                   // the last piece absorbs the remainder
        for (auto &r : cur) {                                          // gen.rs:131-150
            Ctg c;
            c.id = "ctg:" + chr_id + ":" + std::to_string(++serial);
            c.chr_id = chr_id;
            c.chr_start = (int32_t)r.first;
            c.chr_end = (int32_t)r.second;
            c.chr_strand = "+";
            c.length = c.chr_end - c.chr_start + 1;
            c.range = chr_id + ":" + runlist(r.first, r.second);
            out.push_back(c);
        }
    }
    return out;
}

// ---------------------------------------------------------------------------
// anno
// ---------------------------------------------------------------------------
namespace {

// utils.rs:118-129: first match of (?i)ctg:[\w_]+:\d+
bool extract_ctg_id(const std::string &s, std::string &id) {
    for (size_t i = 0; i + 4 <= s.size(); ++i) {
        if ((s[i] == 'c' || s[i] == 'C') && (s[i + 1] == 't' || s[i + 1] == 'T') &&
            (s[i + 2] == 'g' || s[i + 2] == 'G') && s[i + 3] == ':') {
            size_t j = i + 4;
            while (j < s.size() && is_word(s[j])) ++j;
            if (j == i + 4 || j >= s.size() || s[j] != ':') continue;
            size_t k = j + 1;
            while (k < s.size() && s[k] >= '0' && s[k] <= '9') ++k;
            if (k == j + 1) continue;
            id = s.substr(i, k - i);
            return true;
        }
    }
    return false;
}

}  // namespace

std::string anno(gams_gpu_t *h, const std::map<std::string, Runlist> &sets, const std::vector<Ctg> &ctgs,
                 const std::vector<std::string> &lines, bool header, const std::string &prefix, size_t idx_id,
                 size_t idx_range) {
    std::map<std::string, const Ctg *> ctg_of;
    for (const Ctg &c : ctgs) ctg_of[c.id] = &c;
    // device image of the runlists: one group per chr
    std::map<std::string, uint32_t> group_of;
    std::vector<uint64_t> off{0};
    std::vector<int32_t> lo, hi;
    uint32_t g = 0;
    for (auto &kv : sets) {
        group_of[kv.first] = g++;
        lo.insert(lo.end(), kv.second.lo.begin(), kv.second.lo.end());
        hi.insert(hi.end(), kv.second.hi.begin(), kv.second.hi.end());
        off.push_back(lo.size());
    }
    gams_spans_t *sp = nullptr;
    check(h, gams_spans_create(h, g, off.data(), lo.data(), hi.data(), &sp));
    struct Guard {
        gams_gpu_t *h;
        gams_spans_t *sp;
        ~Guard() { gams_spans_destroy(h, sp); }
    } guard{h, sp};

    // parse the lines on several threads (each a contiguous run, results concatenated in order),
    // one device call, then format the rows the same way
    struct Part {
        std::vector<size_t> keep;  // lines that produce a row
        std::vector<uint32_t> grp;
        std::vector<int32_t> cl, ch, qs, qe;
        std::string out;
    };
    const size_t first = header ? 1 : 0;
    const size_t n_lines = lines.size() > first ? lines.size() - first : 0;
    const unsigned T = (unsigned)std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), n_lines / 20000}));
    std::vector<Part> part(T);
    run_shares(T, [&](uint32_t t) {
        Part &P = part[t];
        const size_t b0 = first + n_lines * t / T, b1 = first + n_lines * (t + 1) / T;
        for (size_t i = b0; i < b1; ++i) {                             // anno.rs:97
            // the two tab-separated fields the line is asked for (no vector of all its fields)
            const std::string &ln = lines[i];
            size_t n_fields = 0, fb = 0, id_b = 0, id_e = 0, rg_b = 0, rg_e = 0;
            for (;;) {
                const size_t fe = std::min(ln.find('\t', fb), ln.size());
                ++n_fields;
                if (n_fields == idx_id) id_b = fb, id_e = fe;
                if (n_fields == idx_range) rg_b = fb, rg_e = fe;
                if (fe == ln.size()) break;
                fb = fe + 1;
            }
            if (idx_id == 0 || idx_range == 0 || idx_id > n_fields || idx_range > n_fields)
                throw Error(GAMS_EINVAL, "anno: field index out of range (the reference panics, anno.rs:115)");
            std::string ctg_id;
            if (!extract_ctg_id(ln.substr(id_b, id_e - id_b), ctg_id)) continue;   // anno.rs:116-119
            Range r = Range::from_str(ln.substr(rg_b, rg_e - rg_b));
            if (!r.valid) continue;                                     // anno.rs:123-125
            auto gi = group_of.find(r.chr);
            uint32_t gq = UINT32_MAX;
            int32_t c0 = 0, c1 = 0;
            if (gi != group_of.end()) {                                 // anno.rs:129
                auto ci = ctg_of.find(ctg_id);
                if (ci == ctg_of.end())
                    throw Error(GAMS_EINVAL, "anno: unknown " + ctg_id + " (the reference panics, redis.rs:133-134)");
                gq = gi->second;
                c0 = ci->second->chr_start;
                c1 = ci->second->chr_end;
            }
            P.keep.push_back(i);
            P.grp.push_back(gq);
            P.cl.push_back(c0);
            P.ch.push_back(c1);
            P.qs.push_back(r.start);
            P.qe.push_back(r.end);
        }
    });
    std::vector<size_t> base(T + 1, 0);
    for (unsigned t = 0; t < T; ++t) base[t + 1] = base[t] + part[t].keep.size();
    const size_t nk = base[T];
    std::vector<uint32_t> grp(nk ? nk : 1);
    std::vector<int32_t> cl(nk ? nk : 1), ch(nk ? nk : 1), qs(nk ? nk : 1), qe(nk ? nk : 1);
    for (unsigned t = 0; t < T; ++t) {
        std::copy(part[t].grp.begin(), part[t].grp.end(), grp.begin() + base[t]);
        std::copy(part[t].cl.begin(), part[t].cl.end(), cl.begin() + base[t]);
        std::copy(part[t].ch.begin(), part[t].ch.end(), ch.begin() + base[t]);
        std::copy(part[t].qs.begin(), part[t].qs.end(), qs.begin() + base[t]);
        std::copy(part[t].qe.begin(), part[t].qe.end(), qe.begin() + base[t]);
    }
    std::vector<float> prop(nk ? nk : 1);
    check(h, gams_gpu_cover(h, sp, grp.data(), cl.data(), ch.data(), qs.data(), qe.data(), nk, prop.data()));
    run_shares(T, [&](uint32_t t) {
        Part &P = part[t];
        char buf[64];
        size_t bytes = 0;
        for (size_t i : P.keep) bytes += lines[i].size() + 9;
        P.out.reserve(bytes);
        for (size_t k = 0; k < P.keep.size(); ++k) {
            snprintf(buf, sizeof buf, "%.4f", (double)prop[base[t] + k]);   // anno.rs:140 {:.4}
            P.out += lines[P.keep[k]];
            P.out += '\t';
            P.out += buf;
            P.out += '\n';
        }
    });
    std::string out;
    if (header && !lines.empty()) out += lines[0] + "\t" + prefix + "Prop\n";  // anno.rs:108
    for (unsigned t = 0; t < T; ++t) out += part[t].out;
    return out;
}

}  // namespace gams
