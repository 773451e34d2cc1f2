// gams_host_c.cpp -- flat C entry points of the host layer, for tests (ctypes) and for a
// non-C++ host.  Strings returned are malloc'd: release with gams_host_free.  On error the
// functions return NULL and gams_host_last_error() has the message.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "gams_host.hpp"

namespace {
thread_local std::string g_err;
thread_local int g_code = 0;

char *dup(const std::string &s) {
    char *p = (char *)std::malloc(s.size() + 1);
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

template <typename F>
char *guarded(F &&f) {
    try {
        g_err.clear();
        g_code = 0;
        return dup(f());
    } catch (const gams::Error &e) {
        g_err = e.what();
        g_code = e.code;
    } catch (const std::exception &e) {
        g_err = e.what();
        g_code = -1;
    }
    return nullptr;
}

char *dup_bytes(const std::string &s, uint64_t *out_len) {
    if (out_len) *out_len = s.size();
    char *p = (char *)std::malloc(s.size() + 1);
    if (p) {
        std::memcpy(p, s.data(), s.size());
        p[s.size()] = 0;
    }
    return p;
}
template <typename F>
char *guarded_bytes(uint64_t *out_len, F &&f) {
    try {
        g_err.clear();
        g_code = 0;
        return dup_bytes(f(), out_len);
    } catch (const gams::Error &e) {
        g_err = e.what();
        g_code = e.code;
    } catch (const std::exception &e) {
        g_err = e.what();
        g_code = -1;
    }
    return nullptr;
}


std::vector<gams::Ctg> make_ctgs(uint32_t n, const char *const *ids, const char *const *chrs, const int32_t *starts,
                                 const int32_t *ends) {
    std::vector<gams::Ctg> v(n);
    for (uint32_t i = 0; i < n; ++i) {
        v[i].id = ids[i];
        v[i].chr_id = chrs[i];
        v[i].chr_start = starts[i];
        v[i].chr_end = ends[i];
        v[i].length = ends[i] - starts[i] + 1;
        v[i].range = v[i].chr_id + ":" + gams::runlist(starts[i], ends[i]);
    }
    return v;
}

std::vector<std::string> split_lines(const char *text) {
    std::vector<std::string> lines;
    std::istringstream is(text ? text : "");
    std::string ln;
    while (std::getline(is, ln)) lines.push_back(ln);
    return lines;
}
}  // namespace

extern "C" {

const char *gams_host_last_error() { return g_err.c_str(); }
int gams_host_last_code() { return g_code; }
void gams_host_free(void *p) { std::free(p); }

// wave.rs:121-215 over n ctgs in one device pass; the per-ctg Strings are concatenated in ctg order
char *gams_host_wave(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                     const int32_t *starts, const int32_t *ends, const uint8_t *const *seqs, int32_t size,
                     int32_t step, uint32_t lag, float threshold, float influence, float coverage, int is_signal) {
    return guarded([&] {
        gams::WaveArgs a;
        a.size = size;
        a.step = step;
        a.lag = lag;
        a.threshold = threshold;
        a.influence = influence;
        a.coverage = coverage;
        a.signal = is_signal != 0;
        std::vector<const uint8_t *> sp(seqs, seqs + n);
        std::string out;
        for (auto &s : gams::wave_proc_ctgs(h, make_ctgs(n, ids, chrs, starts, ends), sp, a)) out += s;
        return out;
    });
}

namespace {
// stages[0..9] = inflate_upload, upload, plan, kernel, peaks, format, total ms, threads, peaks fetched, reserved
void put_stages(const gams::WaveStages &st, double *stages) {
    if (!stages) return;
    stages[0] = st.inflate_upload_ms;
    stages[1] = st.upload_ms;
    stages[2] = st.plan_ms;
    stages[3] = st.kernel_ms;
    stages[4] = st.peaks_ms;
    stages[5] = st.format_ms;
    stages[6] = st.total_ms;
    stages[7] = (double)st.threads;
    stages[8] = (double)st.peaks;
    stages[9] = 0;
}
}  // namespace

// gams_host_wave with the stage clock of gams::WaveStages (sync bit 0: the device is drained at every stage
// boundary; bit 1: `--signal`, a row for every window); *out_len receives the length of the text
char *gams_host_wave_timed(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                           const int32_t *starts, const int32_t *ends, const uint8_t *const *seqs, int32_t size,
                           int32_t step, uint32_t lag, float threshold, float influence, float coverage, int sync,
                           double *stages, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        gams::WaveArgs a;
        a.size = size;
        a.step = step;
        a.lag = lag;
        a.threshold = threshold;
        a.influence = influence;
        a.coverage = coverage;
        a.signal = (sync & 2) != 0;
        gams::WaveStages st;
        st.sync = (sync & 1) != 0;
        std::vector<const uint8_t *> sp(seqs, seqs + n);
        std::vector<std::string> rows = gams::wave_proc_ctgs(h, make_ctgs(n, ids, chrs, starts, ends), sp, a, &st);
        put_stages(st, stages);
        size_t total = 0;
        for (auto &s : rows) total += s.size();
        std::string out;
        out.reserve(total);
        for (auto &s : rows) out += s;
        return out;
    });
}

// the same from the gzip'd `seq:` values (gams::wave_proc_ctgs_gz): blobs[i] / blob_len[i] = value of ctg i
char *gams_host_wave_gz(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                        const int32_t *starts, const int32_t *ends, const uint8_t *const *blobs,
                        const uint64_t *blob_len, int32_t size, int32_t step, uint32_t lag, float threshold,
                        float influence, float coverage, uint32_t threads, int sync, double *stages, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        gams::WaveArgs a;
        a.size = size;
        a.step = step;
        a.lag = lag;
        a.threshold = threshold;
        a.influence = influence;
        a.coverage = coverage;
        gams::WaveStages st;
        st.sync = sync != 0;
        std::vector<const uint8_t *> bp(blobs, blobs + n);
        std::vector<uint64_t> bl(blob_len, blob_len + n);
        std::vector<std::string> rows =
            gams::wave_proc_ctgs_gz(h, make_ctgs(n, ids, chrs, starts, ends), bp, bl, a, threads, &st);
        put_stages(st, stages);
        size_t total = 0;
        for (auto &s : rows) total += s.size();
        std::string out;
        out.reserve(total);
        for (auto &s : rows) out += s;
        return out;
    });
}

// decode_gz of n values on `threads` host threads into caller buffers: dst[i] has room for dst_cap[i] bytes,
// got[i] receives the bytes written.  Returns 0, or -1 with the message in gams_host_last_error().
int gams_host_decode_gz_many(uint32_t n, const uint8_t *const *blobs, const uint64_t *blob_len, uint8_t *const *dst,
                             const uint64_t *dst_cap, uint64_t *got, uint32_t threads) {
    try {
        g_err.clear();
        g_code = 0;
        std::vector<const uint8_t *> bp(blobs, blobs + n);
        std::vector<uint64_t> bl(blob_len, blob_len + n);
        std::vector<std::string> out = gams::decode_gz_many(bp, bl, threads);
        for (uint32_t i = 0; i < n; ++i) {
            if (out[i].size() > dst_cap[i]) throw gams::Error(GAMS_EINVAL, "decode_gz_many: value " + std::to_string(i) + " does not fit");
            std::memcpy(dst[i], out[i].data(), out[i].size());
            got[i] = out[i].size();
        }
        return 0;
    } catch (const gams::Error &e) {
        g_err = e.what();
        g_code = e.code;
    } catch (const std::exception &e) {
        g_err = e.what();
        g_code = -1;
    }
    return -1;
}

// the same over several handles (one per device; tests pass two handles of one device)
char *gams_host_wave_multi(gams_gpu_t *const *handles, uint32_t n_handles, uint32_t n, const char *const *ids,
                           const char *const *chrs, const int32_t *starts, const int32_t *ends,
                           const uint8_t *const *seqs, int32_t size, int32_t step, uint32_t lag, float threshold,
                           float influence, float coverage, int is_signal, uint64_t batch_bytes) {
    return guarded([&] {
        gams::WaveArgs a;
        a.size = size;
        a.step = step;
        a.lag = lag;
        a.threshold = threshold;
        a.influence = influence;
        a.coverage = coverage;
        a.signal = is_signal != 0;
        std::vector<gams_gpu_t *> hs(handles, handles + n_handles);
        std::vector<const uint8_t *> sp(seqs, seqs + n);
        std::string out;
        for (auto &s : gams::wave_proc_ctgs_multi(hs, make_ctgs(n, ids, chrs, starts, ends), sp, a, batch_bytes)) out += s;
        return out;
    });
}

// sw.rs:108-194 for one ctg
char *gams_host_sw(gams_gpu_t *h, const char *ctg_id, const char *chr, int32_t chr_start, int32_t chr_end,
                   const uint8_t *seq, uint32_t nf, const char *const *feature_ids, const int32_t *fs,
                   const int32_t *fe, int32_t size, int32_t max, int32_t resize) {
    return guarded([&] {
        const char *ids[1] = {ctg_id}, *chrs[1] = {chr};
        gams::Ctg c = make_ctgs(1, ids, chrs, &chr_start, &chr_end)[0];
        std::vector<gams::Feature> f(nf);
        for (uint32_t i = 0; i < nf; ++i) f[i] = gams::Feature{feature_ids[i], fs[i], fe[i]};
        gams::SwArgs a;
        a.size = size;
        a.max = max;
        a.resize = resize;
        return gams::sw_proc_ctg(h, c, seq, f, a);
    });
}

// `gams sw` over several handles.  features: rows "ctg_index\tfeature_id\tstart\tend"
char *gams_host_sw_multi(gams_gpu_t *const *handles, uint32_t n_handles, uint32_t n, const char *const *ids,
                         const char *const *chrs, const int32_t *starts, const int32_t *ends,
                         const uint8_t *const *seqs, const char *features, int32_t size, int32_t max, int32_t resize) {
    return guarded([&] {
        std::vector<std::vector<gams::Feature>> f(n);
        for (const std::string &ln : split_lines(features)) {
            std::istringstream is(ln);
            std::string ci, id, s0, s1;
            std::getline(is, ci, '\t');
            std::getline(is, id, '\t');
            std::getline(is, s0, '\t');
            std::getline(is, s1, '\t');
            f.at((size_t)std::stoul(ci)).push_back(gams::Feature{id, std::stoi(s0), std::stoi(s1)});
        }
        gams::SwArgs a;
        a.size = size;
        a.max = max;
        a.resize = resize;
        std::string out;
        for (auto &s : gams::sw_proc_ctgs_multi(std::vector<gams_gpu_t *>(handles, handles + n_handles),
                                                make_ctgs(n, ids, chrs, starts, ends),
                                                std::vector<const uint8_t *>(seqs, seqs + n), f, a))
            out += s;
        return out;
    });
}

// the same, also reporting the milliseconds of the operator itself (gams::sw_proc_ctgs_multi: upload, kernels, text;
// without the parsing of `features` in front and the concatenation behind, which belong to this wrapper)
char *gams_host_sw_multi_timed(gams_gpu_t *const *handles, uint32_t n_handles, uint32_t n, const char *const *ids,
                               const char *const *chrs, const int32_t *starts, const int32_t *ends,
                               const uint8_t *const *seqs, const char *features, int32_t size, int32_t max, int32_t resize,
                               double *operator_ms, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        std::vector<std::vector<gams::Feature>> f(n);
        for (const std::string &ln : split_lines(features)) {
            std::istringstream is(ln);
            std::string ci, id, s0, s1;
            std::getline(is, ci, '\t');
            std::getline(is, id, '\t');
            std::getline(is, s0, '\t');
            std::getline(is, s1, '\t');
            f.at((size_t)std::stoul(ci)).push_back(gams::Feature{id, std::stoi(s0), std::stoi(s1)});
        }
        gams::SwArgs a;
        a.size = size;
        a.max = max;
        a.resize = resize;
        const std::vector<gams::Ctg> cv = make_ctgs(n, ids, chrs, starts, ends);
        const std::vector<const uint8_t *> sv(seqs, seqs + n);
        const std::vector<gams_gpu_t *> hv(handles, handles + n_handles);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::string> rows = gams::sw_proc_ctgs_multi(hv, cv, sv, f, a);
        if (operator_ms) *operator_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        size_t total = 0;
        for (auto &s : rows) total += s.size();
        std::string out;
        out.reserve(total);
        for (auto &s : rows) out += s;
        return out;
    });
}

// locate.rs:111-141.  rgs: newline-separated ranges (first TSV column already cut).
// rg_lines (for --count): newline-separated "ctg_id\trange" rows = the rg: records per ctg.
static thread_local double g_operator_ms = 0.0;

char *gams_host_locate(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                       const int32_t *starts, const int32_t *ends, const char *rgs, int is_count,
                       const char *rg_lines) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        if (is_count) {
            std::map<std::string, std::vector<gams::Range>> rg_of;
            for (uint32_t i = 0; i < n; ++i) rg_of[ids[i]];  // build_idx_rg visits every ctg (redis.rs:279-302)
            for (const std::string &ln : split_lines(rg_lines)) {
                size_t tab = ln.find('\t');
                if (tab == std::string::npos) continue;
                gams::Range r = gams::Range::from_str(ln.substr(tab + 1));
                if (r.valid) rg_of[ln.substr(0, tab)].push_back(r);
            }
            loc.set_rg_index(rg_of);
        }
        const std::vector<std::string> lines = split_lines(rgs);
        const auto t0 = std::chrono::steady_clock::now();
        std::string out = loc.locate(lines, is_count != 0);
        g_operator_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return out;
    });
}

// milliseconds the last gams_host_locate / gams_host_anno of this thread spent inside the operator itself
// (Locator::locate, gams::anno: parsing of the range strings, device lookups, row text) -- without this wrapper's
// splitting of its arguments into lines and, for --count, the build of the rg index, which are the caller's
double gams_host_last_operator_ms(void) { return g_operator_ms; }

// locate --seq (locate.rs:124-134).  seq_lines: "ctg_id\tbases" rows.
char *gams_host_locate_seq(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                           const int32_t *starts, const int32_t *ends, const char *rgs, const char *seq_lines) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        std::map<std::string, std::string> seq_of;
        for (const std::string &ln : split_lines(seq_lines)) {
            size_t tab = ln.find('\t');
            if (tab != std::string::npos) seq_of[ln.substr(0, tab)] = ln.substr(tab + 1);
        }
        return loc.locate_seq(split_lines(rgs), seq_of);
    });
}

// utils.rs:39-67 read_range with the drop-first quirk: rows "ctg_id\trange" in bucket order
char *gams_host_read_range(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                           const int32_t *starts, const int32_t *ends, const char *lines) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        std::vector<std::string> first_cols;
        for (const std::string &ln : split_lines(lines)) first_cols.push_back(ln.substr(0, ln.find('\t')));
        std::string out;
        for (auto &kv : gams::read_range(loc, first_cols))
            for (const gams::Range &r : kv.second) out += kv.first + "\t" + r.to_string() + "\n";
        return out;
    });
}

// `gams peak` (peak.rs:24-177) for a set of ctgs: `lines` = rows of a wave TSV.  Output rows, per ctg in
// id order: id, range, length, gc, signal, left_wave_length, left_amplitude, left_signal,
// right_wave_length, right_amplitude, right_signal (the fields of gams::Peak, data.rs:30-43).
char *gams_host_peak(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                     const int32_t *starts, const int32_t *ends, const uint8_t *const *seqs, const char *lines) {
    return guarded([&] {
        std::vector<gams::Ctg> ctgs = make_ctgs(n, ids, chrs, starts, ends);
        gams::Locator loc(h, ctgs);
        auto peaks_of = gams::read_peak(loc, split_lines(lines));
        std::string out;
        // ctgs in id order (the map's), all of their merged ranges through one batched call
        std::vector<gams::Ctg> pc;
        std::vector<const uint8_t *> ps;
        std::vector<std::vector<std::pair<gams::Range, std::string>>> pp;
        for (auto &kv : peaks_of) {
            uint32_t slot = 0;
            while (slot < n && ctgs[slot].id != kv.first) ++slot;
            pc.push_back(ctgs[slot]);
            ps.push_back(seqs[slot]);
            pp.push_back(kv.second);
        }
        for (const std::vector<gams::Peak> &recs : gams::peak_records_batch(h, pc, ps, pp))
            for (const gams::Peak &p : recs)
                out += p.id + "\t" + p.range + "\t" + std::to_string(p.length) + "\t" + gams::fmt_f32(p.gc) + "\t" +
                       p.signal + "\t" + std::to_string(p.left_wave_length) + "\t" + gams::fmt_f32(p.left_amplitude) +
                       "\t" + p.left_signal + "\t" + std::to_string(p.right_wave_length) + "\t" +
                       gams::fmt_f32(p.right_amplitude) + "\t" + p.right_signal + "\n";
        return out;
    });
}

// rg.rs:41-77 / feature.rs:47-95: rows "key\tjson" of the records the loaders SET (tag == NULL: rg)
char *gams_host_loader_records(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                               const int32_t *starts, const int32_t *ends, const char *lines, const char *tag) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        std::vector<std::string> first_cols;
        for (const std::string &ln : split_lines(lines)) first_cols.push_back(ln.substr(0, ln.find('\t')));
        std::string out;
        for (const gams::Record &r : tag ? gams::feature_records(loc, first_cols, tag) : gams::rg_records(loc, first_cols))
            out += r.key + "\t" + r.json + "\n";
        return out;
    });
}

// tsv.rs for "rg:*" (tag == NULL) / "feature:*" records of the loaders above
char *gams_host_loader_tsv(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                           const int32_t *starts, const int32_t *ends, const char *lines, const char *tag) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        std::vector<std::string> first_cols;
        for (const std::string &ln : split_lines(lines)) first_cols.push_back(ln.substr(0, ln.find('\t')));
        return gams::tsv_records(tag ? gams::feature_records(loc, first_cols, tag) : gams::rg_records(loc, first_cols),
                                 tag != nullptr);
    });
}

// gzip framing of seq: values (redis.rs:149-161); *out_len receives the length
char *gams_host_decode_gz(const uint8_t *bytes, uint64_t n, uint64_t *out_len) {
    return guarded([&] {
        std::string s = gams::decode_gz(bytes, n);
        if (out_len) *out_len = s.size();
        return s;
    });
}
char *gams_host_encode_gz(const uint8_t *bytes, uint64_t n, uint64_t *out_len) {
    try {
        std::string s = gams::encode_gz(bytes, n);
        if (out_len) *out_len = s.size();
        char *p = (char *)std::malloc(s.size() + 1);
        if (p) std::memcpy(p, s.data(), s.size());
        return p;
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

// utils.rs:7-22 for many ranges: one ctg id (or empty) per line
char *gams_host_find(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                     const int32_t *starts, const int32_t *ends, const char *rgs) {
    return guarded([&] {
        gams::Locator loc(h, make_ctgs(n, ids, chrs, starts, ends));
        std::vector<gams::Range> v;
        for (const std::string &ln : split_lines(rgs)) {
            gams::Range r = gams::Range::from_str(ln);
            r.strand.clear();
            if (!r.valid) r.chr = "\x01invalid";  // never located
            v.push_back(r);
        }
        std::string out;
        for (const std::string &id : loc.find(v)) out += id + "\n";
        return out;
    });
}

// anno.rs:95-142.  runlists: newline-separated "chr\trunlist" rows of the JSON set.
char *gams_host_anno(gams_gpu_t *h, uint32_t n, const char *const *ids, const char *const *chrs,
                     const int32_t *starts, const int32_t *ends, const char *runlists, const char *lines,
                     int header, const char *prefix, uint32_t idx_id, uint32_t idx_range) {
    return guarded([&] {
        std::map<std::string, gams::Runlist> sets;
        for (const std::string &ln : split_lines(runlists)) {
            size_t tab = ln.find('\t');
            if (tab == std::string::npos) continue;
            gams::Runlist &rl = sets[ln.substr(0, tab)];
            std::istringstream is(ln.substr(tab + 1));
            std::string part;
            while (std::getline(is, part, ',')) {
                if (part.empty() || part == "-") continue;
                size_t dash = part.find('-', 1);
                int32_t lo = std::atoi(part.substr(0, dash).c_str());
                int32_t hi = dash == std::string::npos ? lo : std::atoi(part.substr(dash + 1).c_str());
                rl.lo.push_back(lo);
                rl.hi.push_back(hi);
            }
        }
        const std::vector<gams::Ctg> cv = make_ctgs(n, ids, chrs, starts, ends);
        const std::vector<std::string> lv = split_lines(lines);
        const auto t0 = std::chrono::steady_clock::now();
        std::string out = gams::anno(h, sets, cv, lv, header != 0, prefix ? prefix : "", idx_id, idx_range);
        g_operator_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return out;
    });
}

// gen.rs:81-157 for one chromosome; rows "id\trange\tchr_id\tchr_start\tchr_end\tchr_strand\tlength"
// (the columns of `gams tsv -s "ctg:*"`, tests/S288c/ctg.tsv)
char *gams_host_gen(gams_gpu_t *h, const char *chr_id, const uint8_t *seq, uint64_t len, int32_t piece,
                    int32_t fill, int32_t min_len) {
    return guarded([&] {
        gams::GenArgs a;
        a.piece = piece;
        a.fill = fill;
        a.min = min_len;
        std::string out;
        for (const gams::Ctg &c : gams::gen_ctgs(h, chr_id, seq, len, a))
            out += c.id + "\t" + c.range + "\t" + c.chr_id + "\t" + std::to_string(c.chr_start) + "\t" +
                   std::to_string(c.chr_end) + "\t" + c.chr_strand + "\t" + std::to_string(c.length) + "\n";
        return out;
    });
}

// tsv.rs:31-71 for "ctg:*" (no device work)
char *gams_host_tsv_ctgs(uint32_t n, const char *const *ids, const char *const *chrs, const int32_t *starts,
                         const int32_t *ends) {
    return guarded([&] { return gams::tsv_ctgs(make_ctgs(n, ids, chrs, starts, ends)); });
}

// header lines of `gams wave` (which == 0) and `gams sw` (which == 1)
char *gams_host_header(int which) { return dup(which == 0 ? gams::wave_header() : gams::sw_header()); }

// locate --count / anno coverage over several handles (one per GPU)
int gams_host_count_multi(gams_gpu_t *const *handles, uint32_t n_handles, uint32_t n_groups, const uint64_t *group_off,
                          const uint32_t *starts, const uint32_t *stops, const uint32_t *q_group, const uint32_t *qs,
                          const uint32_t *qe, uint64_t nq, int32_t *count) {
    char *r = guarded([&] {
        gams::count_multi(std::vector<gams_gpu_t *>(handles, handles + n_handles), n_groups, group_off, starts, stops,
                          q_group, qs, qe, nq, count);
        return std::string();
    });
    if (!r) return g_code ? g_code : -1;
    free(r);
    return 0;
}

int gams_host_cover_multi(gams_gpu_t *const *handles, uint32_t n_handles, uint32_t n_groups, const uint64_t *group_off,
                          const int32_t *lo, const int32_t *hi, const uint32_t *q_group, const int32_t *clip_lo,
                          const int32_t *clip_hi, const int32_t *qs, const int32_t *qe, uint64_t nq, float *prop) {
    char *r = guarded([&] {
        gams::cover_multi(std::vector<gams_gpu_t *>(handles, handles + n_handles), n_groups, group_off, lo, hi, q_group,
                          clip_lo, clip_hi, qs, qe, nq, prop);
        return std::string();
    });
    if (!r) return g_code ? g_code : -1;
    free(r);
    return 0;
}

// merge_ints of wave.rs:217-252, for CPU-only tests of the host logic
void gams_host_merge_windows(const uint32_t *w, uint64_t n, int32_t chr_start, int32_t size, int32_t step,
                             float coverage, int64_t *cmin, int64_t *cmax, char *in_graph) {
    gams::merge_windows(w, (size_t)n, chr_start, size, step, coverage, cmin, cmax, in_graph);
}

// serde_json round trip of a ctg: record (redis.rs:127-135); returns the JSON re-serialised from the parse
char *gams_host_ctg_json_roundtrip(const char *json) {
    return guarded([&] { return gams::ctg_json(gams::ctg_from_json(json)); });
}

// formatting helpers exposed for CPU-only tests
char *gams_host_fmt_f32(float v) { return dup(gams::fmt_f32(v)); }
char *gams_host_range_roundtrip(const char *s) {
    gams::Range r = gams::Range::from_str(s);
    return dup(r.valid ? r.to_string() : std::string("<invalid>"));
}


// ---- wire formats (gams_wire.cpp): byte strings out, *out_len = their length ------------------------
char *gams_host_bincode_ctg_bundle(uint32_t n, const char *const *ids, const char *const *chrs, const int32_t *starts,
                                   const int32_t *ends, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] { return gams::wire::bincode_ctg_bundle(make_ctgs(n, ids, chrs, starts, ends)); });
}
// decoded bundle as the ctg.tsv text of `gams tsv` (one line per ctg, key order)
char *gams_host_bincode_ctg_bundle_decode(const uint8_t *bytes, uint64_t n) {
    return guarded([&] { return gams::tsv_ctgs(gams::wire::bincode_ctg_bundle_decode(bytes, n)); });
}
// intervals as parallel arrays; vals: n NUL-terminated strings (NULL = all empty, the idx:rg case)
char *gams_host_bincode_lapper(uint64_t n, const uint32_t *starts, const uint32_t *stops, const char *const *vals,
                               uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        std::vector<gams::wire::LapperIv> ivs(n);
        for (uint64_t i = 0; i < n; ++i) {
            ivs[i].start = starts[i];
            ivs[i].stop = stops[i];
            if (vals && vals[i]) ivs[i].val = vals[i];
        }
        return gams::wire::bincode_lapper(ivs);
    });
}
// decoded Lapper as text: "start\tstop\tval" per interval, then "#starts ...", "#stops ...", "#max_len N cov C merged B"
char *gams_host_bincode_lapper_decode(const uint8_t *bytes, uint64_t n) {
    return guarded([&] {
        const gams::wire::LapperBlob b = gams::wire::bincode_lapper_decode(bytes, n);
        std::string o;
        for (const auto &v : b.intervals) o += std::to_string(v.start) + "\t" + std::to_string(v.stop) + "\t" + v.val + "\n";
        o += "#starts";
        for (uint32_t x : b.starts) o += " " + std::to_string(x);
        o += "\n#stops";
        for (uint32_t x : b.stops) o += " " + std::to_string(x);
        o += "\n#max_len " + std::to_string(b.max_len) + " cov " + (b.has_cov ? std::to_string(b.cov) : std::string("None")) +
             " merged " + (b.overlaps_merged ? "true" : "false") + "\n";
        return o;
    });
}
// args: n byte strings (arg_len[i] bytes each)
char *gams_host_resp_command(uint32_t n, const char *const *args, const uint64_t *arg_len, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        std::vector<std::string> a(n);
        for (uint32_t i = 0; i < n; ++i) a[i].assign(args[i], (size_t)arg_len[i]);
        return gams::wire::resp_command(a);
    });
}
char *gams_host_resp_scan_values(const char *pattern, uint64_t *out_len) {
    return guarded_bytes(out_len, [&] {
        return gams::wire::resp_eval(gams::wire::scan_values_script(), {}, {pattern ? pattern : "", "1000"});
    });
}
// parses one reply; returns its flat text form (resp_dump), *consumed = bytes used (0: incomplete, "" returned)
char *gams_host_resp_parse(const char *bytes, uint64_t n, uint64_t *consumed) {
    return guarded([&] {
        gams::wire::RespValue v;
        const size_t used = gams::wire::resp_parse(bytes, n, v);
        if (consumed) *consumed = used;
        return used ? gams::wire::resp_dump(v) : std::string();
    });
}

// idx: blobs (n of them, blob i = blob_len[i] bytes) -> device index, one group per blob; NULL on error
void *gams_host_index_from_lappers(void *h, uint32_t n, const uint8_t *const *blobs, const uint64_t *blob_len) {
    try {
        g_err.clear();
        g_code = 0;
        std::vector<gams::wire::LapperBlob> dec;
        for (uint32_t i = 0; i < n; ++i) dec.push_back(gams::wire::bincode_lapper_decode(blobs[i], (size_t)blob_len[i]));
        return gams::wire::index_from_lappers(static_cast<gams_gpu_t *>(h), dec);
    } catch (const gams::Error &e) {
        g_err = e.what();
        g_code = e.code;
    } catch (const std::exception &e) {
        g_err = e.what();
        g_code = -1;
    }
    return nullptr;
}
}  // extern "C"
