// gams_host.hpp -- C++17 host side of the hot path, above the C ABI (include/gams_gpu.h).
//
// The reference's host is Rust; no Rust toolchain exists in this image, so the host layer that
// a Rust build would keep is written in C++ against the same extern "C" boundary.  It mirrors
// the reference's seam fn(&Ctg, &ArgMatches) -> String (src/libs/utils.rs:216-220) for
//   wave::proc_ctg   src/cmd_gams/wave.rs:121-215  (+ merge_ints :217-252)
//   sw::proc_ctg     src/cmd_gams/sw.rs:108-194    (+ Sw Display, src/libs/data.rs:58-83)
//   locate loop      src/cmd_gams/locate.rs:111-141
//   anno loop        src/cmd_gams/anno.rs:95-142
// Everything numeric runs on the GPU through libgams_gpu.so; this layer only keeps the
// reference's bookkeeping: range grammar, peak merging, TSV text.
#pragma once

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/gams_gpu.h"

namespace gams {

// src/libs/data.rs:5-14
struct Ctg {
    std::string id;      // "ctg:{chr}:{serial}"
    std::string range;   // "{chr}:{start}-{end}"
    std::string chr_id;
    int32_t chr_start = 0, chr_end = 0;
    std::string chr_strand = "+";
    int32_t length = 0;
};

// src/libs/data.rs:16-22 (only what sw needs)
struct Feature {
    std::string id;      // "feature:{ctg_id}:{serial}"
    int32_t start = 0, end = 0;
};

// intspan::Range (chr(strand):start-end)
struct Range {
    std::string name, chr, strand;
    int32_t start = 0, end = 0;
    bool valid = false;
    static Range from_str(const std::string &s);
    std::string to_string() const;
};

// Rust `{}` for f32: shortest round-trip digits, positional
std::string fmt_f32(float v);
// IntSpan::runlist of one span
std::string runlist(int64_t s, int64_t e);

struct WaveArgs {          // defaults of src/cmd_gams/wave.rs:23-99
    int32_t size = 100, step = 10;
    uint32_t lag = 100;
    float threshold = 3.0f, influence = 1.0f, coverage = 0.2f;
    bool signal = false;
};

struct SwArgs {            // defaults of src/cmd_gams/sw.rs:9-85
    int32_t size = 100, max = 20, resize = 500;
};

class Error : public std::runtime_error {
public:
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// Where the time of one wave_proc_ctgs* call went, in ms (added to, so a caller may sum several calls).  With
// `sync` set the device is drained at every stage boundary, so each interval belongs to the stage just
// queued (a breakdown run: slower than the pipelined call, whose only host waits are in peaks); without it
// the slots hold host time between checkpoints and only total_ms means much.
struct WaveStages {
    bool sync = false;
    unsigned threads = 0;          // inflate workers used (gz form)
    uint64_t peaks = 0;            // signalled windows fetched
    double inflate_upload_ms = 0;  // gz form: seq: values -> bases in a page-locked image -> HBM (overlapped)
    double upload_ms = 0;          // buffer form: host buffers -> HBM (gams_seqset_upload_all)
    double plan_ms = 0, kernel_ms = 0, peaks_ms = 0, format_ms = 0, total_ms = 0;
};

// wave.rs:121-215 for a batch of ctgs in ONE device pass; returns one String per ctg
std::vector<std::string> wave_proc_ctgs(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                        const std::vector<const uint8_t *> &seqs, const WaveArgs &a,
                                        WaveStages *stages = nullptr);
// The same from the `seq:{ctg}` values as the store holds them (gzip members, redis.rs:149-161):
// blobs[i] / blob_len[i] = the value of ctgs[i], which must inflate to exactly ctgs[i].length bases.
// `threads` workers inflate one ctg at a time each (the reference gunzips inside its --parallel workers,
// wave.rs:288-299 -> redis.rs:142-161) straight into a page-locked image of the device buffer, and the
// finished stretches go to the DMA engine while the rest still inflates.  threads == 0: 16.
std::vector<std::string> wave_proc_ctgs_gz(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                           const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len,
                                           const WaveArgs &a, unsigned threads = 0, WaveStages *stages = nullptr);
std::string wave_proc_ctg(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq, const WaveArgs &a);
// The multi-GPU form: ctgs are split over the handles by longest-
// processing-time-first on their window counts, one host thread per handle, each thread walks
// its share in batches of at most `batch_bytes` bases (upload of batch k+1 overlaps the kernel
// of batch k).  No device talks to another; the per-ctg Strings come back in ctg order.
std::vector<std::string> wave_proc_ctgs_multi(const std::vector<gams_gpu_t *> &handles, const std::vector<Ctg> &ctgs,
                                              const std::vector<const uint8_t *> &seqs, const WaveArgs &a,
                                              uint64_t batch_bytes = 1ull << 30);
// longest-processing-time-first assignment (weights -> owner per item); ties by index
std::vector<uint32_t> lpt_assign(const std::vector<uint64_t> &weights, uint32_t n_owners);
// merge_ints (wave.rs:217-252) over the ascending window indices of one sign of one ctg: the merged
// component's chromosome span per window, and whether the window has any edge (is printed as "(+)").
void merge_windows(const uint32_t *w, size_t n, int32_t chr_start, int32_t size, int32_t step, float coverage,
                   int64_t *cmin, int64_t *cmax, char *in_graph);

// `locate --count` / `anno` over several devices: the groups (ctgs, or chromosomes of
// the runlist set) are split over the handles by longest-processing-time-first on their interval
// counts, every query is routed to the device that owns its group (a counting sort on the host:
// the only "exchange step" of the path), one host thread per handle runs its share, and the
// results scatter back to the query order.  Same arguments as gams_gpu_count / gams_gpu_cover with
// the index given as arrays; queries of unknown groups (>= n_groups) get 0.
void count_multi(const std::vector<gams_gpu_t *> &handles, uint32_t n_groups, const uint64_t *group_off,
                 const uint32_t *starts, const uint32_t *stops, const uint32_t *q_group, const uint32_t *qs,
                 const uint32_t *qe, uint64_t nq, int32_t *count);
void cover_multi(const std::vector<gams_gpu_t *> &handles, uint32_t n_groups, const uint64_t *group_off,
                 const int32_t *lo, const int32_t *hi, const uint32_t *q_group, const int32_t *clip_lo,
                 const int32_t *clip_hi, const int32_t *qs, const int32_t *qe, uint64_t nq, float *prop);

// sw.rs:108-194
std::string sw_proc_ctg(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq, const std::vector<Feature> &features,
                        const SwArgs &a);
// several ctgs on one handle: one seqset, one gams_gpu_sw_batch call and one readback per batch of
// <= batch_bytes bases; rows returned per ctg in the order given ("" for a ctg without features)
std::vector<std::string> sw_proc_ctgs(gams_gpu_t *h, const std::vector<Ctg> &ctgs, const std::vector<const uint8_t *> &seqs,
                                      const std::vector<std::vector<Feature>> &features, const SwArgs &a,
                                      uint64_t batch_bytes = 256ull << 20);
// `gams sw --parallel` over several devices: ctgs split over the handles by LPT on their feature
// counts, one host thread per handle, rows returned per ctg in the order given.
std::vector<std::string> sw_proc_ctgs_multi(const std::vector<gams_gpu_t *> &handles, const std::vector<Ctg> &ctgs,
                                            const std::vector<const uint8_t *> &seqs,
                                            const std::vector<std::vector<Feature>> &features, const SwArgs &a);

// idx:ctg: / idx:rg: on the device (redis.rs:236-324) + the locate loop (locate.rs:111-141)
class Locator {
public:
    Locator(gams_gpu_t *h, const std::vector<Ctg> &ctgs);
    ~Locator();
    // idx:rg:{ctg}: ranges already bucketed per ctg (rg loader, utils.rs:39-67)
    void set_rg_index(const std::map<std::string, std::vector<Range>> &rg_of_ctg);
    // find_one_idx for every range; "" when not located (utils.rs:7-22)
    std::vector<std::string> find(const std::vector<Range> &rgs);
    // locate output: "{rg}\t{ctg_id}\n" or with count "{rg}\t{count}\n" (locate.rs:135-140)
    std::string locate(const std::vector<std::string> &rgs, bool is_count);
    // locate --seq (locate.rs:124-134): ">{rg}\n{bases}\n"; seq_of maps ctg id -> gunzipped seq
    std::string locate_seq(const std::vector<std::string> &rgs, const std::map<std::string, std::string> &seq_of);
    const Ctg *ctg(const std::string &id) const;

private:
    gams_gpu_t *h_;
    std::vector<Ctg> ctgs_;                       // in index order
    std::map<std::string, uint32_t> chr_group_;   // chr -> group of the ctg index
    std::map<std::string, uint32_t> ctg_slot_;    // ctg id -> position in ctgs_
    gams_index_t *ctg_ix_ = nullptr;
    gams_index_t *rg_ix_ = nullptr;
    std::map<std::string, uint32_t> rg_group_;    // ctg id -> group of the rg index
};

class Locator;

// src/libs/redis.rs:149-161: the `seq:` values are gzip members (flate2, Compression::fast)
std::string decode_gz(const uint8_t *bytes, size_t n);
// ... into caller memory (at most cap bytes; more is an error); returns the bytes written
size_t decode_gz_into(const uint8_t *bytes, size_t n, uint8_t *dst, size_t cap);
// a batch of values on `threads` host threads (0: 16), one value per worker at a time
std::vector<std::string> decode_gz_many(const std::vector<const uint8_t *> &blobs, const std::vector<uint64_t> &blob_len,
                                        unsigned threads = 0);
std::string encode_gz(const uint8_t *bytes, size_t n);

// src/libs/utils.rs:39-67 read_range: locate every valid range and bucket it by ctg, INCLUDING
// the reference's quirk: `.entry(k).and_modify(push).or_default()` only creates the bucket for
// the first range of a ctg, so that range is dropped (tests/cli.rs:250,301: 79 lines -> 69).
std::map<std::string, std::vector<Range>> read_range(Locator &loc, const std::vector<std::string> &lines);

// src/libs/data.rs:30-43
struct Peak {
    std::string id, range, signal;
    int32_t length = 0;
    float gc = 0.0f;
    int32_t left_wave_length = 0, right_wave_length = 0;
    float left_amplitude = 0.0f, right_amplitude = 0.0f;
    std::string left_signal, right_signal;
};
// src/libs/utils.rs:83-116 read_peak: rows of a wave TSV (range \t gc \t signal) bucketed by ctg,
// strand stripped, with the same drop-first-per-ctg quirk as read_range; header / invalid rows skipped.
std::map<std::string, std::vector<std::pair<Range, std::string>>> read_peak(Locator &loc,
                                                                          const std::vector<std::string> &lines);
// src/cmd_gams/peak.rs:47-158 for one ctg: gc of every peak range (device), then the left /
// right wavelength, amplitude and neighbour signal.  Serials start at 1 (a fresh cnt:peak:).
std::vector<Peak> peak_records(gams_gpu_t *h, const Ctg &ctg, const uint8_t *seq,
                               const std::vector<std::pair<Range, std::string>> &peaks);
// the same for several ctgs on one handle: one seqset and ONE gams_gpu_range_gc_batch call per batch of
// <= batch_bytes bases; out[c] = the Peak records of ctgs[c]
std::vector<std::vector<Peak>> peak_records_batch(gams_gpu_t *h, const std::vector<Ctg> &ctgs,
                                                  const std::vector<const uint8_t *> &seqs,
                                                  const std::vector<std::vector<std::pair<Range, std::string>>> &peaks,
                                                  uint64_t batch_bytes = 256ull << 20);

// src/cmd_gams/rg.rs:41-77 / feature.rs:47-95: the records the loaders SET, as (key, JSON) pairs in
// the reference's order (ctg id order, then file order), serials from 1 per ctg (a fresh cnt:).
// JSON text as serde_json writes the structs of src/libs/data.rs:16-28 (field order, no spaces).
struct Record {
    std::string key, json;
};
// serde_json text of a Ctg as `gen` stores it under its id (redis.rs:127-130, data.rs:5-14) and the
// reverse (redis.rs:132-135); field order and number formats as serde_json writes them.
std::string ctg_json(const Ctg &c);
Ctg ctg_from_json(const std::string &json);

// header lines the commands print in front of the per-ctg rows (wave.rs:263-266, sw.rs:204-216)
inline const char *wave_header() { return "#range\tgc_content\tsignal\n"; }
inline const char *sw_header() { return "id\trange\ttype\tdistance\tgc_content\tgc_mean\tgc_stddev\tgc_cv\trg_count\n"; }

// src/cmd_gams/tsv.rs:31-71: `gams tsv -s "ctg:*"` = a header of the struct's field names
// (data.rs:5-14, serde order) and one tab-separated row per record; ctgs in the order given.
std::string tsv_ctgs(const std::vector<Ctg> &ctgs);
// the same for "feature:*" / "rg:*" records: the JSON objects' values in field order
std::string tsv_records(const std::vector<Record> &records, bool features);
std::vector<Record> rg_records(Locator &loc, const std::vector<std::string> &lines);
std::vector<Record> feature_records(Locator &loc, const std::vector<std::string> &lines, const std::string &tag);

// gen.rs:81-157 for one chromosome: ambiguous-base scan (device), fill, excise, --piece split;
// ctg ids "ctg:{chr}:{serial}" with serial from 1 (gen.rs:133-134).  `seq` is the whole chromosome.
struct GenArgs {            // defaults of src/cmd_gams/gen.rs:26-49
    int32_t piece = 500000, fill = 50, min = 5000;
};
std::vector<Ctg> gen_ctgs(gams_gpu_t *h, const std::string &chr_id, const uint8_t *seq, uint64_t len,
                          const GenArgs &a);

// anno.rs:95-142; `sets` = runlists per chr (sorted disjoint spans)
struct Runlist {
    std::vector<int32_t> lo, hi;
};
std::string anno(gams_gpu_t *h, const std::map<std::string, Runlist> &sets, const std::vector<Ctg> &ctgs,
                 const std::vector<std::string> &lines, bool header, const std::string &prefix, size_t idx_id,
                 size_t idx_range);

// ---- wire formats either side of the path (gams_wire.cpp; SURVEY 8 f-3, parity unpinned) ----------
namespace wire {
// bundle:ctg:{chr}: bincode 1.3.3 of BTreeMap<String, Ctg> (redis.rs:216-233)
std::string bincode_ctg_bundle(const std::vector<Ctg> &ctgs);
std::vector<Ctg> bincode_ctg_bundle_decode(const uint8_t *bytes, size_t n);
// idx:ctg:{chr} / idx:rg:{ctg}: bincode of rust-lapper's Lapper<u32, String> (redis.rs:236-324)
struct LapperIv {
    uint32_t start = 0, stop = 0;   // half-open, stop = end + 1 (redis.rs:245-248, 291-294)
    std::string val;                // ctg id for idx:ctg, "" for idx:rg
};
struct LapperBlob {
    std::vector<LapperIv> intervals;      // sorted by (start, stop)
    std::vector<uint32_t> starts, stops;  // each sorted on its own
    uint32_t max_len = 0, cov = 0;
    bool has_cov = false, overlaps_merged = false;
};
std::string bincode_lapper(const std::vector<LapperIv> &intervals);   // Lapper::new + serialize
LapperBlob bincode_lapper_decode(const uint8_t *bytes, size_t n);
// device index over decoded idx: blobs, one group per blob (caller destroys it with gams_index_destroy)
gams_index_t *index_from_lappers(gams_gpu_t *h, const std::vector<LapperBlob> &blobs);
// RESP2
struct RespValue {
    enum Type { Simple, Error, Integer, Bulk, Null, Array } type = Null;
    std::string str;
    int64_t integer = 0;
    std::vector<RespValue> array;
};
std::string resp_command(const std::vector<std::string> &args);
std::string resp_pipeline_set(const std::vector<std::pair<std::string, std::string>> &kv);
std::string resp_eval(const std::string &script, const std::vector<std::string> &keys,
                      const std::vector<std::string> &argv);
const char *scan_values_script();
// one reply from the front of `bytes`: returns the bytes consumed, 0 if the reply is not complete yet
size_t resp_parse(const char *bytes, size_t n, RespValue &out);
std::string resp_dump(const RespValue &v);
}  // namespace wire

}  // namespace gams
