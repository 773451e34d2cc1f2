// wave.hip -- GC sliding windows + smoothed z-score on gfx950.
//
// Replaces the body of proc_ctg in the reference (src/cmd_gams/wave.rs:138-155):
//   gams::sliding            src/libs/window.rs:78-94
//   bio gc_content per window src/cmd_gams/wave.rs:144-153
//   gams::thresholding_algo  src/libs/stat.rs:16-56
// and the collection of signalled windows (wave.rs:170-187).
//
// One workgroup = one tile of consecutive windows of one ctg, plus the halo of
// lag+1 windows in front of it whose gc values the z-score of the tile's first
// windows needs (z-score state never crosses a ctg: wave.rs:294-297).  Workgroups
// never communicate; every tile writes its peaks into its own fixed slot.
//
// Two tile kernels share that scheme:
//   wave_fast_kernel<W,SIZE,STEP,LAG>  8-bit counts, 32-bit variance math (every BASELINE
//       configuration): 1-bit-per-base stream in LDS, rolling window counts, rolling
//       S1/S2, branch-free integer decision with a rigorous guard band, wave-cooperative
//       exact-order f32 re-evaluation of the windows inside the band.  See its header.
//   wave_tile_kernel<KT,WIDE>          any size/lag up to 65535: chunk prefix PM (prefix<<16 |
//       mask), k[w] = P(w*step+size) - P(w*step), prefix arrays Q1 = sum k, Q2 = sum k^2,
//       same decision + guard band, scalar exact path.
// Signals are bit-identical to thresholding_algo in both (the guard band only decides
// which windows take the exact path).  Parameters whose halo does not fit a tile at all (large
// steps, very large lags) run untiled: wave_direct_count_kernel + wave_direct_signal_kernel.
//
// influence != 1 makes filtered[] (stat.rs:42) a serial recurrence per ctg: those plans guess the signals with the
// influence == 1 kernels and iterate signals -> filtered -> signals to the fixed point, every window in parallel
// (wave_repair.hpp; influence 0 has its own fill-forward filter and a "freeze" guess), then wave_compact_kernel.  The
// round-2 form -- wave_serial_wave_kernel, one wavefront per ctg in exact f32 order (one lane per ctg beyond lag
// 16383) on the counts of a counts-only tile pass -- is what takes over if the sweeps do not settle.
//
// Compile with -ffp-contract=off: the exact paths must not fuse (x-m)*(x-m)+acc.

#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>

#include "wave_kernels.hpp"
#include "wave_repair.hpp"
#include "wave_rows.hpp"

#include <charconv>


// =============================================================================
// host side
// =============================================================================
// gams_wave_rows_*: the plan's TSV rows made on the device (wave_rows.hpp)
struct WaveRows {
    uint32_t dmax = 0, max_name = 0;
    uint8_t *arena = nullptr;                 // RowCtg[n_ctg] | names | gc text table | words[4 + n_ctg + 1]
    size_t arena_bytes = 0;
    RowCtg *d_ctgs = nullptr;
    char *d_names = nullptr;
    uint8_t *d_gctab = nullptr;
    unsigned long long *d_words = nullptr;     // [0] records, [1] text bytes, [2] peaks, [3] fullest tile, [4 ..] ctg_off[n_ctg + 1]
    uint8_t *tmp = nullptr;                   // per-record and per-block tables, room for `cap` records
    size_t tmp_bytes = 0;
    uint64_t cap = 0;
    char *d_text = nullptr;
    size_t d_text_bytes = 0;
    char *h_text = nullptr;                   // page-locked
    size_t h_text_bytes = 0;
    unsigned long long *h_words = nullptr;    // page-locked: [0] records, [1] text bytes, [2] peaks (offsets kernel), [3] fullest tile, [4..] ctg_off
    size_t h_words_bytes = 0;
    uint64_t copy_bytes = 0;                  // size of the speculative text copy (held by the captured graphs)
    bool use_graph = true;
    struct RowGraphKey {
        const void *dense, *tmp, *text, *h_text, *peaks, *tile_cnt, *tile_off;
        uint64_t dense_cap, cap, copy_bytes;
        uint32_t tile_cap, nt;
    };
    struct RowGraph {
        hipGraphExec_t exec = nullptr;
        RowGraphKey key{};
    } graph[gams_gpu::kMaxWays];
    uint64_t copied = 0;                      // text bytes the last begin() already sent to the host
    uint64_t last_bytes = 0;                  // text bytes of the previous pass (sizes the speculative copy)
    hipEvent_t done = nullptr;
    bool begun = false;
};

// gams_wave_signal_text: the `--signal` rows of a pass as text (wave_rows.hpp, sig_*_kernel)
struct WaveSig {
    uint8_t *arena = nullptr;                 // SigTile[nt] | blk_len[nt] | blk_off[nt + 2] | RowCtg[n_ctg] | names | gc table | words[n_ctg]
    size_t arena_bytes = 0;
    uint32_t n_tiles = 0;
    SigTile *d_tiles = nullptr;
    uint32_t *d_blk_len = nullptr;
    unsigned long long *d_blk_off = nullptr, *d_words = nullptr;
    RowCtg *d_ctgs = nullptr;
    char *d_names = nullptr;
    uint8_t *d_gctab = nullptr;
    size_t names_cap = 0;
    char *d_text = nullptr, *h_text = nullptr;      // h_text: page-locked
    size_t d_text_bytes = 0, h_text_bytes = 0;
    unsigned long long *h_words = nullptr;          // page-locked: words[n_ctg], then ctg_off[n_ctg + 1] made on the host
    size_t h_words_bytes = 0;
};

struct Launcher;
struct gams_wave_plan {
    gams_seqset_t *set = nullptr;
    gams_wave_params_t prm{};
    uint32_t flags = 0;
    bool serial = false;          // influence != 1
    bool repair = false;          // ... by guess-and-iterate (wave_repair.hpp): the kernels run as for influence == 1 into
                                  // the dense rows, then filtered[] and the signals are iterated to their fixed point
    JacTile *d_jtiles = nullptr;  // repair: tiles of 256 windows (pooled block)
    size_t d_jtiles_bytes = 0;
    uint32_t n_jtiles = 0;
    float *d_xtab = nullptr;      // repair: xtab[k] = k as f32 / size as f32, [size + 1] (inside arena_fixed)
    bool jac0 = false;            // repair with influence == 0: fill-forward filter + freeze guess (jac0_* kernels)
    bool direct = false;          // halo beyond a tile: one lane per window, no tiling (wave_direct_*_kernel)
    bool wide = false, k16 = false;
    int fast_w = 0;               // W of wave_fast_kernel (0: generic wave_tile_kernel)
    uint32_t tw = 0;              // windows per tile
    uint32_t max_chunks = 0, max_win = 0;
    size_t lds_bytes = 0;
    uint64_t total_windows = 0;
    std::vector<WaveCtgDev> ctgs;
    std::vector<WaveTile> tiles;
    // device
    WaveCtgDev *d_ctgs = nullptr;
    WaveTile *d_tiles = nullptr;
    // Outputs of one pass.  A plan of depth D owns D of these "ways"; consecutive gams_wave_run
    // calls rotate over them, each way on its own HIP stream, so that independent passes overlap
    // on the device (a 12-Mb pass alone is launch-latency bound).
    struct Way {
        gams_peak_t *d_peaks = nullptr;         // one slot of tile_cap records per tile (pooled block)
        size_t d_peaks_bytes = 0;
        uint32_t *d_tile_cnt = nullptr;         // inside arena_geom
        unsigned long long *d_counters = nullptr;   // ring of kCounterRing slots, zeroed once per lap (pooled)
        size_t d_counters_bytes = 0;
        uint64_t runs = 0;                      // passes this way has run
        uint32_t last_ring = 0;                 // ring slot of its last pass
        uint32_t *d_dense_cnt = nullptr;        // --signal rows / input of the serial kernel (pooled)
        int8_t *d_dense_sig = nullptr;
        float *d_filtered = nullptr;
        size_t d_dense_cnt_bytes = 0, d_dense_sig_bytes = 0, d_filtered_bytes = 0;
        uint8_t *d_jac = nullptr;               // repair: filtered[] | dirty blocks | control words (one pooled block)
        size_t d_jac_bytes = 0;
        bool jac0_table = false;                // influence == 0: the freeze table of this way has been filled
        bool jac_serial = false;                // the last pass was handed to the one-wavefront-per-ctg recurrence
        unsigned long long *h_ctl = nullptr;    // page-locked copy of the control words, made behind every batch of sweeps
        size_t h_ctl_bytes = 0;
        uint32_t jac_sweeps = 0;                // sweeps queued for the way's current pass
        bool jac_settled = true;                // the host has seen a sweep without a flip (or ran the fallback)
        hipEvent_t done = nullptr;              // pipelined mode: recorded behind each run's kernels
        hipEvent_t ran_ev = nullptr;            // lets the readback stream queue behind the last run
        uint64_t seen_upload = 0;               // seqset upload generation this way's stream has waited for
        bool seen_ready = false;                // ... and the plan's const table
    };
    Way way[gams_gpu::kMaxWays];
    uint32_t depth = 1;                         // ways in use
    uint32_t lane = 0;                          // way k runs on stream (lane + k) % kMaxWays (gams_wave_plan_set_lane)
    int taper_req = -1;                         // gams_wave_plan_set_taper: -1 auto, 0 off, 1 on
    bool taper = false;                         // the tile table ends in W = 8 and W = 4 tiles (wave_fast_taper_kernel)
    int taper4_pct = 25, taper8_pct = 50;       // size of the two tails, % of a round of workgroup slots (gams_wave_plan_set_taper_shape)
    uint32_t queue_threads = gams_gpu::kMaxWays;   // host threads gams_wave_run_n queues from (gams_wave_plan_set_queue_threads)
    uint32_t last_way = 0;                      // way of the most recent run
    uint32_t sel_age = 0;                       // readers look at the run `sel_age` before the most recent one
    hipEvent_t ready = nullptr;                 // recorded on the compute stream behind the const table
    size_t d_dense_bytes = 0;
    gams_peak_t *h_peaks = nullptr;             // pinned: packed peaks of the last gams_wave_peaks
    size_t h_peaks_bytes = 0;
    uint32_t tile_cap = 0, tile_cap_req = 0;
    uint32_t tw_req = 0;                        // gams_wave_plan_set_tile request (0: the library's choice)
    uint32_t nth = 256;                         // threads per workgroup of the tile kernel (64 / 128: step-1 kernels, W = 28)
    uint32_t nth_req = 0;                       // gams_wave_plan_set_threads request (0: the library's choice)
    gams_peak_t *d_dense = nullptr;             // packed copy made by gams_wave_peaks
    uint64_t dense_cap = 0;
    uint64_t run_idx = 0;
    int8_t *d_const_sig = nullptr;              // [size+1], see wave_const_table_kernel
    unsigned long long *d_stamps = nullptr;     // diagnostics, [tiles][8]
    unsigned long long *d_tile_off = nullptr;
    // two pooled arenas hold the small tables: `fixed` = ctgs | const_sig (life of the plan),
    // `geom` = tiles | tile_off (+2 totals) | tile_cnt per way (replaced when the tiling changes)
    uint8_t *arena_fixed = nullptr, *arena_geom = nullptr;
    size_t arena_fixed_bytes = 0, arena_geom_bytes = 0;
    bool ran = false;
    struct Launcher *launcher[gams_gpu::kMaxWays - 1] = {};   // extra queueing threads of gams_wave_run_n (made on first use)
    bool pipelined = false;       // gams_wave_plan_set_pipelined: an event per run; readers wait on it
    bool attr_set = false;        // dynamic-LDS attribute applied for the current geometry
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0;
    float sq[6] = {0, 0, 0, 0, 0, 0};   // aA, aB, gA0, gA1, gB0, gB1 (wave_squared_band)
    WaveRows *rows = nullptr;           // gams_wave_rows_setup
    WaveSig *sigtext = nullptr;         // gams_wave_signal_text
    float guard_safety = 1.5f;          // gams_wave_plan_set_guard
    bool guard_exact = false;           // every window through the exact path
};
static void wave_launcher_stop(gams_wave_plan_t *p);


namespace {

constexpr uint32_t kCounterRing = 128;       // one counter slot (kShards lines) per run: no per-run memset
constexpr size_t kSlotWords = (size_t)kShards * kShardWords;
constexpr uint32_t kMaxTileBytes = 65520;  // chunk prefix is 16 bits
constexpr uint32_t kMaxTw = 8192;          // 2 bits/iteration in a 64-bit register
constexpr uint32_t kOffOneGroup = 32768;   // tables up to here: wave_offsets_kernel (one workgroup, one launch)

size_t wave_lds_bytes(uint32_t max_chunks, uint32_t max_win, bool wide, bool k16) {
    size_t b = 0;
    const size_t mwp = (max_win + 3u) & ~1u;
    b += mwp * (wide ? 8 : 4);                        // Q2
    b += mwp * 4;                                     // Q1
    b += (size_t)((max_chunks + 4) & ~1u) * 4;        // PM
    b += 8 * 8;                                       // scratch
    b += 132 * 4;                                     // PC
    b += (size_t)(max_win + 8) * (k16 ? 2 : 1);       // K
    return (b + 15) & ~(size_t)15;
}

// Guard band of the integer decision, in units of D = |n*k - S1| (see DESIGN.md
// "z-score guard band" for the derivation).  u = 2^-24.
void wave_guard_band(const gams_wave_params_t &p, double safety, bool all_exact, float g[4]) {
    const double u = std::ldexp(1.0, -24);
    const double n = (double)p.lag, sz = (double)p.size;
    const double thr = std::fabs((double)p.threshold);
    // The squared-domain decision divides by thr*sqrt(n/(n-1)): thresholds outside [1e-6, 1e6]
    // (and non-finite or negative ones) are not worth a fast path, every window is exact.
    if (all_exact || !(std::isfinite(p.threshold)) || p.threshold < 1e-6f || p.threshold > 1e6f || p.lag < 2) {
        g[0] = INFINITY;  // every window takes the exact path
        g[1] = g[2] = g[3] = 0.0f;
        return;
    }
    const double gam = 1.01 * (n + 1.0) * u / (1.0 - (n + 1.0) * u);  // f32 sequential mean
    const double kap = std::sqrt(n / (n - 1.0));
    const double eta = 1.01 * ((n + 3.0) / 2.0 + 2.0) * u;              // sq sum, /, sqrt, *thr
    g[1] = (float)(safety * (gam + 2.0 * u + thr * kap * gam * (1.0 + eta)));
    g[2] = (float)(safety * (eta + 8.0 * u));
    g[3] = (float)(safety * 3.0 * u);
    g[0] = (float)(safety * (2.0 * thr * kap * u * n * sz) + 1e-3);
}

// The same band for z_decide (wave_kernels.hpp): with g = g2 + g3, c = g0 + g1*S1 and
// sb = thr*sqrt(n/(n-1)),   A = D*(1-g)/((1+g) sb) - c/((1+g) sb),   B = D*(1+g)/((1-g) sb) + c/((1-g) sb).
// The kernel's own roundings (one fma for c, one for A or B, 2u on V) are covered by rounding the
// multiplier of A down and everything else up by a few u here, so that a decided window is
// decided under the exact-arithmetic band above: A_f^2 > V_f implies A^2 > V, B_f^2 < V_f implies B^2 < V.
void wave_squared_band(const gams_wave_params_t &p, const float g[4], float sq[6]) {
    for (int i = 0; i < 6; ++i) sq[i] = 0.0f;
    if (!(g[0] < INFINITY)) return;
    const double u = std::ldexp(1.0, -24);
    const double n = (double)p.lag;
    const double sb = std::fabs((double)p.threshold) * std::sqrt(n / (n - 1.0));
    const double gg = (double)g[2] + (double)g[3];
    const double lo = 1.0 - 5.0 * u, hi = 1.0 + 7.0 * u;
    sq[0] = (float)((1.0 - gg) / ((1.0 + gg) * sb) * lo);   // aA
    sq[1] = (float)((1.0 + gg) / ((1.0 - gg) * sb) * hi);   // aB
    sq[2] = (float)((double)g[0] / ((1.0 + gg) * sb) * hi); // gA0
    sq[3] = (float)((double)g[1] / ((1.0 + gg) * sb) * hi); // gA1
    sq[4] = (float)((double)g[0] / ((1.0 - gg) * sb) * hi); // gB0
    sq[5] = (float)((double)g[1] / ((1.0 - gg) * sb) * hi); // gB1
}

// W of the baked instantiations of wave_fast_kernel (parameters in the instruction stream)
// 0: the parameters are arguments; 1: size, step and lag baked into the instruction stream (BASELINE's configurations);
// 2: size and step baked, the lag an argument (`--lag N` next to the default size: the reference's own benchmark
// runs 100 / 5 / 200 and 100 / 20 / 50, doc/benchmark/Atha.md:55,276-280)
int wave_baked_kind(const gams_wave_params_t &q, int w) {
    const bool headline = q.size == 100 && q.step == 10 && q.lag == 100;   // every BASELINE step-10 config
    const bool step1 = q.size == 100 && q.step == 1 && q.lag == 100;       // BASELINE configs[3] (GRCh38, step 1)
    if ((headline && (w == 12 || w == 8 || w == 4)) || (step1 && (w == 28 || w == 20 || w == 12))) return 1;
    if (q.size != 100 || q.lag + 1u > 128u * (uint32_t)w) return 0;        // at least half of the tile's slots are windows
    if ((q.step == 5 || q.step == 10 || q.step == 20) && (w == 12 || w == 8 || w == 4)) return 2;
    if (q.step == 1 && w == 28) return 2;
    if ((q.step == 1 || q.step == 5) && w == 20) return 2;   // 5120 windows x 5 bases: the bytes of a W = 10 tile at step 10
    return 0;
}
bool wave_is_baked(const gams_wave_params_t &q, int w) { return wave_baked_kind(q, w) != 0; }

size_t wave_fast_lds_bytes(uint32_t max_chunks, uint32_t w, uint32_t lag, bool dense, uint32_t nth = 256) {
    size_t b = (size_t)((((max_chunks + 8u) >> 1) + 16u + 3u) & ~3u) * 4;   // BM: 16 mask bits per chunk + pad
    b += 16 * 4;                                         // scratch
    b += (nth * w + lag + 1u + 31u) & ~15u;              // K (threads past the tile's end still read their slots)
    b += (nth + 16) * 8;                                 // PS: block sums of the baked kernels
    b += nth * 2;                                        // RK: ranks of phase 4b
    if (dense) b += ((nth * w + 15u) & ~15u) + 16u;      // SG (+ the dword behind the last group, read with it)
    return (b + 15) & ~(size_t)15;
}

void wave_fill_tiles(gams_wave_plan_t *p) {
    p->tiles.clear();
    for (uint32_t c = 0; c < p->set->n_ctg; ++c) {
        const uint32_t n = p->ctgs[c].n_win;
        for (uint32_t w = 0; w < n; w += p->tw)
            p->tiles.push_back(WaveTile{c, w, n, 0u, p->ctgs[c].seq_off, p->ctgs[c].win_base});
    }
}

// Tapered tile table for the baked W = 12 kernel: the ctgs holding the last windows of the batch are
// cut into W = 4 tiles, the ones before them into W = 8 tiles (see wave_fast_taper_kernel).  The two
// tails are a quarter / half a round of workgroup slots, and at most 8 % / 17 % of the batch.
void wave_fill_tiles_tapered(gams_wave_plan_t *p, uint32_t slots) {
    const gams_wave_params_t &q = p->prm;
    const uint32_t tw12 = 256u * 12u - q.lag - 1u, tw8 = 256u * 8u - q.lag - 1u, tw4 = 256u * 4u - q.lag - 1u;
    const uint64_t T = p->total_windows;
    // % of a round of workgroup slots for the W = 4 / W = 8 tails (defaults from gpurun_out/r2_taper_sweep.log:
    // 384 Mb 71.6 us without tails, 70.0 at 50/50, 68.5 at 25/50, 70.9 at 100/100; the 120-Mb launch 28.4-28.7 for all)
    const int k4 = p->taper4_pct, k8 = p->taper8_pct;                       // gams_wave_plan_set_taper_shape (0..100)
    const uint64_t x4 = std::min<uint64_t>((uint64_t)slots * k4 / 100 * tw4, T * 8 / 100);
    const uint64_t y8 = std::min<uint64_t>((uint64_t)slots * k8 / 100 * tw8, T * 17 / 100);
    p->tiles.clear();
    for (uint32_t c = 0; c < p->set->n_ctg; ++c) {
        const uint32_t n = p->ctgs[c].n_win;
        const uint64_t after = T - p->ctgs[c].win_base;           // windows from this ctg's first to the end of the batch
        const uint32_t w_kind = after <= x4 ? 4u : after <= x4 + y8 ? 8u : 12u;
        const uint32_t tw = w_kind == 4u ? tw4 : w_kind == 8u ? tw8 : tw12;
        for (uint32_t w = 0; w < n; w += tw)
            p->tiles.push_back(WaveTile{c, w, n, w_kind, p->ctgs[c].seq_off, p->ctgs[c].win_base});
    }
}

int wave_build_geometry(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t tw_req) {
    const gams_wave_params_t &q = p->prm;
    const uint64_t halo_bytes = (uint64_t)(q.lag + 1) * q.step + (uint64_t)q.size + 32;
    p->fast_w = 0;
    p->attr_set = false;
    p->direct = false;
    p->taper = false;
    // fast kernel: 8-bit counts, 32-bit variance math with 24-bit multiplies
    const bool fast_ok = (!p->serial || p->repair) && q.size <= 255 && q.step <= 32 && (uint64_t)q.lag * q.size <= 65535 &&
                         (uint64_t)q.lag * q.size * q.size < (1ull << 24) && q.lag >= 2;
    const bool step1_prm = q.size == 100 && q.step == 1 && wave_baked_kind(q, 28) != 0;
    if (fast_ok && (tw_req == 0 || tw_req == 1024 || tw_req == 2048 || tw_req == 3072 || tw_req == 5120 ||
                    (tw_req == 7168 && step1_prm))) {
        static const int cand[5] = {28, 20, 12, 8, 4};
        int pick = 0;
        for (int w : cand) {
            const uint64_t tw = 256ull * w;
            if (halo_bytes + tw * q.step > kMaxTileBytes) continue;
            if (tw_req) {
                if (tw_req == tw) pick = w;
                continue;
            }
            // Default W by the number of tiles it would give (a small genome is launch-latency
            // bound and wants many short workgroups; a saturated chip wants the lower instruction
            // count per window of the bigger tiles).  Measured us per pass, W = 4 / 8 / 12, one pass
            // at a time | four in flight:
            //   1.2 M windows   7.9 /  8.2 /  9.3  |  3.02 / 2.84 / 3.07
            //   1.8 M           9.1 /  9.2 / 10.3  |  4.34 / 3.45 / 3.55
            //   2.4 M          11.3 / 10.4 / 11.0  |  5.76 / 4.35 / 4.07
            //   3.6 M          15.0 / 14.2 / 14.2  |  8.42 / 6.44 / 5.91
            //   38 M (384 Mb)   107 /   82 /   77
            // W = 28 / 20 for the baked step-1 kernel (3.8e8 windows: 320 us at W = 28, 334 at W = 20, 503 at
            // W = 12 in round 1; 1.2e7 windows, less than a round of W = 28 tiles: 23.5 vs 21.0 us), otherwise
            // only on request.
            const uint64_t tiles = p->total_windows / tw;
            const bool step1 = q.size == 100 && q.step == 1 && q.lag == 100;   // baked W = 20 fits 64 VGPRs
            const bool flight = p->depth >= 2;
            // (round 3: W = 28 tiles of ONE wave -- 64 threads, 1,691 windows at lag 100 -- from 4,096 such tiles on:
            // 384 Mb 325 -> 285 us, 120 Mb 112 -> 100 us, 12 Mb 21.9 (W = 20) -> 21.2 us; gpurun_out/r3_ab_threads*.log)
            const bool s1w28 = step1 || (q.size == 100 && q.step == 1 && wave_baked_kind(q, 28) == 2);
            // (peaks only: with the dense rows the stores of a whole workgroup's windows are worth more -- 384 Mb --signal
            // 429 us with four waves per tile, 443 with two, 464 with one; gpurun_out/r3_dense_rate2.txt)
            const bool narrow_ok = !(p->flags & GAMS_WAVE_DENSE) && !p->serial;
            if (pick == 0 && w == 28 && s1w28 && p->nth_req == 0 && narrow_ok && q.lag + 1u <= 32u * 28u &&
                p->total_windows / (64u * 28u) >= 4096)
                pick = w;
            if (pick == 0 && w == 28 && s1w28 && tiles >= 4096)
                pick = w;
            if (pick == 0 && w == 20 && (step1 || (q.size == 100 && q.step == 1 && wave_baked_kind(q, 20) == 2)) && tiles >= 1024)
                pick = w;
            // step 5: twice the windows per byte of step 10, W = 20 amortises the per-thread work (384 Mb: 112 -> 108 us at
            // lag 100, 131 -> 114 us at lag 200, one seqset)
            if (pick == 0 && w == 20 && q.size == 100 && q.step == 5 && wave_baked_kind(q, 20) == 2 && tiles >= 2048) pick = w;
            // (step 20 with size 100: 60 KB of bases per W = 12 tile; W = 8 is 2.5 % faster on 384 Mb, HBM bound either way)
            const bool step20 = q.size == 100 && q.step == 20 && wave_baked_kind(q, 8) == 2;
            if (pick == 0 && w == 12 && !step20 && tiles >= (flight ? 768u : 1536u)) pick = w;
            if (pick == 0 && w == 8 && tiles >= (flight ? 512u : 1024u)) pick = w;
            if (pick == 0 && w == 4) pick = w;
        }
        if (pick) {
            p->fast_w = pick;
            // step-1 W = 28 kernels: a tile per one or two waves instead of four (see wave_fast_tile's NTH), while at
            // least half of the tile's slots stay windows
            p->nth = 256;
            if (pick == 28 && q.step == 1 && wave_is_baked(q, pick)) {
                const bool narrow_ok = !(p->flags & GAMS_WAVE_DENSE) && !p->serial;
                const uint32_t want = p->nth_req == 0 ? (narrow_ok ? 64u : 256u) : p->nth_req;   // the library's choice
                if ((want == 64 || want == 128) && q.lag + 1u <= (want / 2u) * 28u) p->nth = want;
            }
            // (diagnostics: the headline kernel in workgroups of one or two waves, on request only -- see DESIGN 3.1)
            if (pick == 12 && q.size == 100 && q.step == 10 && q.lag == 100 && (p->nth_req == 64 || p->nth_req == 128))
                p->nth = p->nth_req;
            // The W = 20 step-5 kernel (lag as an argument) sits between the two regimes: tiles of TWO waves for peaks-only
            // plans over 4,096 such tiles or more (384 Mb 98.0 -> 89.8 us, one wave 94.8; 120 Mb 38.9 -> 38.3 us;
            // gpurun_out/r3_ab_threads5.log)
            if (pick == 20 && q.size == 100 && q.step == 5 && wave_baked_kind(q, 20) == 2) {
                const bool narrow_ok = !(p->flags & GAMS_WAVE_DENSE) && !p->serial;
                const uint32_t want = p->nth_req ? p->nth_req
                                      : narrow_ok && p->total_windows / (128u * 20u) >= 4096 ? 128u : 256u;
                if ((want == 64 || want == 128) && q.lag + 1u <= (want / 2u) * 20u) p->nth = want;
            }
            // baked kernels: the tile's windows plus the lag+1 in front fill nth*W slots exactly
            p->tw = wave_is_baked(q, pick) ? p->nth * pick - q.lag - 1u : 256u * pick;
            p->max_win = p->tw + q.lag + 1;
            // rounded up to whole rows of one chunk per thread: the baked kernels store every row they load
            p->max_chunks = ((uint32_t)((halo_bytes + (uint64_t)p->tw * q.step + 15) / 16) + 1 + p->nth - 1u) / p->nth * p->nth;
            p->k16 = false;
            p->wide = false;
            p->lds_bytes = wave_fast_lds_bytes(p->max_chunks, (uint32_t)pick, q.lag,
                                               (p->flags & GAMS_WAVE_DENSE) != 0 || p->serial, p->nth);
            // a launch of at least a round and a half of workgroups ends in smaller tiles, unless the host
            // keeps passes in flight (their tails overlap anyway, and the small tiles cost 3-4 % more work)
            const uint32_t slots = 8u * (uint32_t)std::max(h->cus, 1);
            const bool headline = q.size == 100 && q.step == 10 && q.lag == 100;
            const bool want = p->taper_req < 0 ? p->depth == 1 : p->taper_req != 0;
            // (not for influence != 1: the dense rows are compacted by wave_compact_kernel, which knows one tile size)
            p->taper = want && !p->serial && tw_req == 0 && pick == 12 && headline && p->nth == 256 &&
                       p->total_windows / p->tw >= slots + slots / 2;
            if (p->taper)
                wave_fill_tiles_tapered(p, slots);
            else
                wave_fill_tiles(p);
            return GAMS_OK;
        }
    }
    uint32_t tw = tw_req;
    if (tw == 0) {
        // default: ~40 KB of bases per tile, 3 workgroups per CU
        uint64_t budget = 40960 > halo_bytes ? 40960 - halo_bytes : 0;
        tw = (uint32_t)std::min<uint64_t>(budget / (uint64_t)q.step, 4096);
    }
    tw = std::min(tw, kMaxTw) & ~255u;
    if (tw < 256) tw = 256;
    while (tw > 256 && halo_bytes + (uint64_t)tw * q.step > kMaxTileBytes) tw -= 256;
    auto go_direct = [&] {
        // (lag+1)*step + size + 256*step beyond the 64-KB tile, or prefix arrays beyond the LDS:
        // untiled kernels; the tile table only serves the peak compaction
        p->direct = true;
        p->tw = 1024;
        p->max_win = 0;
        p->max_chunks = 0;
        p->k16 = p->wide = false;
        p->lds_bytes = 0;
        wave_fill_tiles(p);
        return GAMS_OK;
    };
    if (halo_bytes + (uint64_t)tw * q.step > kMaxTileBytes) return go_direct();
    p->tw = tw;
    p->max_win = tw + q.lag + 1;
    p->max_chunks = (uint32_t)((halo_bytes + (uint64_t)tw * q.step + 15) / 16) + 1;
    p->k16 = q.size > 255;
    // narrow integer path: V = n*S2 - S1^2 and the tile prefix of k^2 fit 32 bits
    const uint64_t ns = (uint64_t)q.lag * (uint64_t)q.size;
    const uint64_t q2max = (uint64_t)(p->max_win + 1) * (uint64_t)q.size * (uint64_t)q.size;
    p->wide = !(ns <= 65535 && q2max < (1ull << 32));
    p->lds_bytes = wave_lds_bytes(p->max_chunks, p->max_win, p->wide, p->k16);
    p->attr_set = false;
    if (p->lds_bytes > 160 * 1024) return go_direct();
    wave_fill_tiles(p);
    return GAMS_OK;
}

void wave_set_band(gams_wave_plan_t *p) {
    float g[4];
    wave_guard_band(p->prm, (double)p->guard_safety, p->guard_exact, g);
    p->g0 = g[0];
    p->g1 = g[1];
    p->g2 = g[2];
    p->g3 = g[3];
    wave_squared_band(p->prm, g, p->sq);
}

inline size_t wave_align256(size_t b) { return (b + 255) & ~(size_t)255; }

// the stream way k runs on: the handle's compute stream, or one of its auxiliary streams
hipStream_t wave_stream(gams_gpu_t *h, const gams_wave_plan_t *p, uint32_t k) {
    const uint32_t si = (p->lane + k) % (uint32_t)gams_gpu::kMaxWays;
    return si == 0 ? h->compute : h->aux[si - 1];
}

// the way the readers look at
gams_wave_plan::Way &wave_read_way(gams_wave_plan_t *p) {
    return p->way[(p->last_way + p->depth - (p->sel_age % p->depth)) % p->depth];
}
uint32_t wave_read_way_index(const gams_wave_plan_t *p) {
    return (p->last_way + p->depth - (p->sel_age % p->depth)) % p->depth;
}

int wave_sync_ways(gams_gpu_t *h, gams_wave_plan_t *p) {
    for (uint32_t k = 0; k < p->depth; ++k) GAMS_HIP(h, hipStreamSynchronize(wave_stream(h, p, k)));
    return GAMS_OK;
}

int wave_upload_geometry(gams_gpu_t *h, gams_wave_plan_t *p) {
    gams_pool_free(h, false, p->arena_geom, p->arena_geom_bytes);
    p->arena_geom = nullptr;
    p->d_tiles = nullptr;
    p->d_tile_off = nullptr;
    const size_t nt = std::max<size_t>(p->tiles.size(), 1);
    const size_t b_tiles = wave_align256(nt * sizeof(WaveTile));
    const size_t b_off = wave_align256((nt + 2) * sizeof(unsigned long long));   // + the two totals
    const size_t b_cnt = wave_align256(nt * sizeof(uint32_t));
    for (auto &w : p->way) w.d_tile_cnt = nullptr;
    GAMS_HIP(h, gams_pool_alloc(h, false, b_tiles + b_off + b_cnt * p->depth,
                                reinterpret_cast<void **>(&p->arena_geom), &p->arena_geom_bytes));
    p->d_tiles = reinterpret_cast<WaveTile *>(p->arena_geom);
    p->d_tile_off = reinterpret_cast<unsigned long long *>(p->arena_geom + b_tiles);
    for (uint32_t k = 0; k < p->depth; ++k)
        p->way[k].d_tile_cnt = reinterpret_cast<uint32_t *>(p->arena_geom + b_tiles + b_off + b_cnt * k);
    if (p->flags & GAMS_WAVE_PEAKS) {
        // a slot of tw/8 records per tile (typical density is 1-2 % of the windows); a tile that
        // overflows makes gams_wave_peaks() regrow the slots to tw records and run again
        gams_pool_free(h, false, p->d_dense, p->d_dense_bytes);
        p->d_dense = nullptr;
        p->dense_cap = 0;
        p->tile_cap = std::max<uint32_t>(p->tile_cap_req ? p->tile_cap_req : p->tw / 8u, 16u);
        for (auto &w : p->way) {
            gams_pool_free(h, false, w.d_peaks, w.d_peaks_bytes);
            w.d_peaks = nullptr;
        }
        for (uint32_t k = 0; k < p->depth; ++k)
            GAMS_HIP(h, gams_pool_alloc(h, false, nt * (size_t)p->tile_cap * sizeof(gams_peak_t),
                                        reinterpret_cast<void **>(&p->way[k].d_peaks), &p->way[k].d_peaks_bytes));
    }
    if (!p->tiles.empty())
        GAMS_HIP(h, hipMemcpy(p->d_tiles, p->tiles.data(), p->tiles.size() * sizeof(WaveTile),
                              hipMemcpyHostToDevice));
    return GAMS_OK;
}

// The repair path's device tables of one way, carved from one pooled block
struct JacBufs {
    float *f;                        // filtered[], one per row
    uint32_t *fblk;                  // per block of kJacTile rows: 1 + the sweep in which filtered last changed there
    unsigned long long *ctl;         // kJacWords control words
    // influence == 0 only
    int32_t *lastu, *tile_last;
    unsigned long long *freeze;
    int8_t *ftab;
    uint8_t *frow;
    size_t bytes;
};
JacBufs wave_jac_carve(uint8_t *base, uint64_t total_windows, const gams_wave_plan_t *p = nullptr) {
    size_t o = 0;
    auto take = [&](size_t b) {
        uint8_t *q = base ? base + o : nullptr;
        o += wave_align256(b);
        return q;
    };
    JacBufs z{};
    z.f = reinterpret_cast<float *>(take(std::max<uint64_t>(total_windows, 1) * 4));
    z.fblk = reinterpret_cast<uint32_t *>(take((size_t)(total_windows / kJacTile + 2) * 4));
    z.ctl = reinterpret_cast<unsigned long long *>(take(kJacWords * 8));
    if (p && p->jac0) {
        const size_t size1 = (size_t)p->prm.size + 1;
        z.lastu = reinterpret_cast<int32_t *>(take(std::max<uint64_t>(total_windows, 1) * 4));
        z.tile_last = reinterpret_cast<int32_t *>(take(std::max<size_t>(p->n_jtiles, 1) * 4));
        z.freeze = reinterpret_cast<unsigned long long *>(take(std::max<size_t>(p->set->n_ctg, 1) * 2 * 8));
        z.ftab = reinterpret_cast<int8_t *>(take(size1 * size1));
        z.frow = reinterpret_cast<uint8_t *>(take(size1));
    }
    z.bytes = o;
    return z;
}

// per-way buffers that do not depend on the tiling: counter ring, dense rows, filtered[]
int wave_alloc_ways(gams_gpu_t *h, gams_wave_plan_t *p) {
    const uint64_t base = std::max<uint64_t>(p->total_windows, 1);
    const bool need_dense = (p->flags & GAMS_WAVE_DENSE) || p->serial || p->direct;
    for (uint32_t k = 0; k < p->depth; ++k) {
        gams_wave_plan::Way &w = p->way[k];
        if (!w.d_counters) {
            GAMS_HIP(h, gams_pool_alloc(h, false, kCounterRing * kSlotWords * sizeof(unsigned long long),
                                        reinterpret_cast<void **>(&w.d_counters), &w.d_counters_bytes));
            // queued in front of the way's first run on its own stream
            GAMS_HIP(h, hipMemsetAsync(w.d_counters, 0, kCounterRing * kSlotWords * sizeof(unsigned long long),
                                       wave_stream(h, p, k)));
            w.runs = 0;
        }
        if (need_dense && !w.d_dense_cnt) {
            GAMS_HIP(h, gams_pool_alloc(h, false, base * sizeof(uint32_t), reinterpret_cast<void **>(&w.d_dense_cnt),
                                        &w.d_dense_cnt_bytes));
            GAMS_HIP(h, gams_pool_alloc(h, false, base, reinterpret_cast<void **>(&w.d_dense_sig),
                                        &w.d_dense_sig_bytes));
        }
        if (p->serial && !p->repair && !w.d_filtered)
            GAMS_HIP(h, gams_pool_alloc(h, false, base * sizeof(float), reinterpret_cast<void **>(&w.d_filtered),
                                        &w.d_filtered_bytes));
        if (p->repair && !w.d_jac) {
            GAMS_HIP(h, gams_pool_alloc(h, false, wave_jac_carve(nullptr, p->total_windows, p).bytes,
                                        reinterpret_cast<void **>(&w.d_jac), &w.d_jac_bytes));
            GAMS_HIP(h, gams_pool_alloc(h, true, kJacWords * 8, reinterpret_cast<void **>(&w.h_ctl), &w.h_ctl_bytes));
        }
    }
    return GAMS_OK;
}

template <typename KT, bool WIDE>
int wave_launch(gams_gpu_t *h, gams_wave_plan_t *p, const WaveArgs &a, hipStream_t st) {
    auto kern = wave_tile_kernel<KT, WIDE>;
    if (!p->attr_set) {
        GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(kern), p->lds_bytes));
        p->attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p->tiles.size()), dim3(256), p->lds_bytes, st, a);
    GAMS_HIP(h, hipGetLastError());
    return GAMS_OK;
}

template <int W, int SIZE, int STEP, int LAG, bool NT, int NTH = 256>
int wave_launch_fast_nt(gams_gpu_t *h, gams_wave_plan_t *p, const WaveArgs &a, hipStream_t st) {
    auto kern = wave_fast_kernel<W, SIZE, STEP, LAG, NT, NTH>;
    if (!p->attr_set) {
        GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(kern), p->lds_bytes));
        p->attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p->tiles.size()), dim3(NTH), p->lds_bytes, st, a.tiles, a.seq, a);
    GAMS_HIP(h, hipGetLastError());
    return GAMS_OK;
}

// sequence loads with the streaming hint once the batch is too large to live in L2 between passes
// (kStreamBytes: twice the 32 MiB of L2)
constexpr uint64_t kStreamBytes = 64ull << 20;
template <int W, int SIZE, int STEP, int LAG, int NTH = 256>
int wave_launch_fast(gams_gpu_t *h, gams_wave_plan_t *p, const WaveArgs &a, hipStream_t st) {
    return p->set->bytes > kStreamBytes ? wave_launch_fast_nt<W, SIZE, STEP, LAG, true, NTH>(h, p, a, st)
                                        : wave_launch_fast_nt<W, SIZE, STEP, LAG, false, NTH>(h, p, a, st);
}

template <bool NT>
int wave_launch_taper(gams_gpu_t *h, gams_wave_plan_t *p, const WaveArgs &a, hipStream_t st) {
    auto kern = wave_fast_taper_kernel<100, 10, 100, NT>;
    if (!p->attr_set) {
        GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(kern), p->lds_bytes));
        p->attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)p->tiles.size()), dim3(256), p->lds_bytes, st, a.tiles, a.seq, a);
    GAMS_HIP(h, hipGetLastError());
    return GAMS_OK;
}

}  // namespace

extern "C" {

int64_t gams_window_count(int64_t len, int32_t size, int32_t step) {
    if (size <= 0 || step <= 0) return -1;
    if (len < size) return 0;
    return (len - size) / step + 1;  // window.rs:78-94 in closed form
}

int gams_wave_plan_create(gams_gpu_t *h, gams_seqset_t *s, const gams_wave_params_t *params, uint32_t flags,
                          gams_wave_plan_t **out) {
    if (!h || !s || !params || !out) return gams_fail(h, GAMS_EINVAL, "wave_plan_create: null argument");
    if (!(flags & (GAMS_WAVE_PEAKS | GAMS_WAVE_DENSE)))
        return gams_fail(h, GAMS_EINVAL, "wave_plan_create: flags must request PEAKS and/or DENSE");
    if (params->size <= 0 || params->step <= 0)
        return gams_fail(h, GAMS_EINVAL, "wave: size and step must be positive");
    if (params->lag == 0) return gams_fail(h, GAMS_ESHORT, "wave: lag == 0 (the reference panics, stat.rs:30)");
    if (params->size > 65535 || params->lag > 65535)
        return gams_fail(h, GAMS_EUNSUPPORTED, "wave: size and lag are limited to 65535");
    GAMS_HIP(h, hipSetDevice(h->device));
    gams_wave_plan_t *p = new gams_wave_plan_t();
    p->set = s;
    p->prm = *params;
    p->flags = flags;
    p->serial = !(params->influence == 1.0f);
    p->ctgs.resize(s->n_ctg);
    uint64_t base = 0;
    for (uint32_t c = 0; c < s->n_ctg; ++c) {
        const int64_t n = gams_window_count(s->len[c], params->size, params->step);
        if (n < (int64_t)params->lag) {
            delete p;
            return gams_fail(h, GAMS_ESHORT,
                             "wave: ctg " + std::to_string(c) + " has " + std::to_string(n) +
                                 " windows < lag (the reference panics, stat.rs:30)");
        }
        p->ctgs[c] = WaveCtgDev{s->off[c], base, s->len[c], (uint32_t)n};
        base += (uint64_t)n;
    }
    p->total_windows = base;
    // influence != 1: guess-and-iterate while a tile's history (256 + lag + 1 floats) fits a comfortable share of the
    // LDS; the one-wavefront-per-ctg recurrence beyond that
    p->repair = p->serial && params->lag >= 2 && params->lag <= 16000;
    p->jac0 = p->repair && params->influence == 0.0f && params->size <= 400;   // (a (size + 1)^2 table)
    int rc = wave_build_geometry(h, p, 0);
    if (rc != GAMS_OK) {
        delete p;
        return rc;
    }
    wave_set_band(p);
    auto fail = [&](int code) {
        gams_wave_plan_destroy(h, p);
        return code;
    };
#define PLAN_HIP(call)                                                                    \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            (void)hipGetLastError();                                                      \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                   \
            return fail(GAMS_EHIP);                                                       \
        }                                                                                 \
    } while (0)
    {
        const size_t b_ctgs = wave_align256(std::max<size_t>(s->n_ctg, 1) * sizeof(WaveCtgDev));
        const size_t b_const = wave_align256((size_t)params->size + 1);
        const size_t b_xtab = p->repair ? wave_align256(((size_t)params->size + 1) * sizeof(float)) : 0;
        PLAN_HIP(gams_pool_alloc(h, false, b_ctgs + b_const + b_xtab, reinterpret_cast<void **>(&p->arena_fixed),
                                 &p->arena_fixed_bytes));
        p->d_ctgs = reinterpret_cast<WaveCtgDev *>(p->arena_fixed);
        p->d_const_sig = reinterpret_cast<int8_t *>(p->arena_fixed + b_ctgs);
        if (p->repair) {
            // the data value of a window with k G/C bases, as the reference computes it: k as f32 / size as f32
            p->d_xtab = reinterpret_cast<float *>(p->arena_fixed + b_ctgs + b_const);
            std::vector<float> xt((size_t)params->size + 1);
            for (size_t k = 0; k < xt.size(); ++k) xt[k] = (float)k / (float)params->size;
            PLAN_HIP(hipMemcpy(p->d_xtab, xt.data(), xt.size() * sizeof(float), hipMemcpyHostToDevice));
            std::vector<JacTile> jt;
            for (uint32_t c = 0; c < s->n_ctg; ++c)
                for (uint32_t w0 = 0; w0 < p->ctgs[c].n_win; w0 += kJacTile)
                    jt.push_back(JacTile{c, w0, p->ctgs[c].n_win, 0u, p->ctgs[c].win_base});
            p->n_jtiles = (uint32_t)jt.size();
            PLAN_HIP(gams_pool_alloc(h, false, std::max<size_t>(jt.size(), 1) * sizeof(JacTile),
                                     reinterpret_cast<void **>(&p->d_jtiles), &p->d_jtiles_bytes));
            if (!jt.empty())
                PLAN_HIP(hipMemcpy(p->d_jtiles, jt.data(), jt.size() * sizeof(JacTile), hipMemcpyHostToDevice));
        }
    }
    if (s->n_ctg)
        PLAN_HIP(hipMemcpy(p->d_ctgs, p->ctgs.data(), s->n_ctg * sizeof(WaveCtgDev), hipMemcpyHostToDevice));
    rc = wave_upload_geometry(h, p);
    if (rc != GAMS_OK) return fail(rc);
    rc = wave_alloc_ways(h, p);
    if (rc != GAMS_OK) return fail(rc);
    hipLaunchKernelGGL(wave_const_table_kernel, dim3((params->size + 256) / 256), dim3(256), 0, h->compute,
                       p->d_const_sig, (uint32_t)params->size, params->lag, params->threshold);
    PLAN_HIP(hipGetLastError());
    PLAN_HIP(hipEventCreateWithFlags(&p->ready, hipEventDisableTiming));
    PLAN_HIP(hipEventRecord(p->ready, h->compute));
    p->way[0].seen_ready = true;   // same stream
#undef PLAN_HIP
    *out = p;
    return GAMS_OK;
}

void gams_wave_plan_destroy(gams_gpu_t *h, gams_wave_plan_t *p) {
    if (!p) return;
    wave_launcher_stop(p);
    if (h) {
        (void)hipSetDevice(h->device);
        for (uint32_t k = 0; k < p->depth; ++k)
            if (wave_stream(h, p, k)) (void)hipStreamSynchronize(wave_stream(h, p, k));
        if (h->readback) (void)hipStreamSynchronize(h->readback);
    }
    if (p->rows) {
        WaveRows *r = p->rows;
        gams_pool_free(h, false, r->arena, r->arena_bytes);
        gams_pool_free(h, false, r->tmp, r->tmp_bytes);
        gams_pool_free(h, false, r->d_text, r->d_text_bytes);
        gams_pool_free(h, true, r->h_text, r->h_text_bytes);
        gams_pool_free(h, true, r->h_words, r->h_words_bytes);
        if (r->done) (void)hipEventDestroy(r->done);
        for (auto &g : r->graph)
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
        delete r;
        p->rows = nullptr;
    }
    if (p->sigtext) {
        WaveSig *g = p->sigtext;
        gams_pool_free(h, false, g->arena, g->arena_bytes);
        gams_pool_free(h, false, g->d_text, g->d_text_bytes);
        gams_pool_free(h, true, g->h_text, g->h_text_bytes);
        gams_pool_free(h, true, g->h_words, g->h_words_bytes);
        delete g;
        p->sigtext = nullptr;
    }
    gams_pool_free(h, false, p->arena_fixed, p->arena_fixed_bytes);
    gams_pool_free(h, false, p->arena_geom, p->arena_geom_bytes);
    gams_pool_free(h, false, p->d_jtiles, p->d_jtiles_bytes);
    gams_pool_free(h, false, p->d_dense, p->d_dense_bytes);
    gams_pool_free(h, true, p->h_peaks, p->h_peaks_bytes);
    for (auto &w : p->way) {
        gams_pool_free(h, false, w.d_peaks, w.d_peaks_bytes);
        gams_pool_free(h, false, w.d_counters, w.d_counters_bytes);
        gams_pool_free(h, false, w.d_dense_cnt, w.d_dense_cnt_bytes);
        gams_pool_free(h, false, w.d_dense_sig, w.d_dense_sig_bytes);
        gams_pool_free(h, false, w.d_filtered, w.d_filtered_bytes);
        gams_pool_free(h, false, w.d_jac, w.d_jac_bytes);
        gams_pool_free(h, true, w.h_ctl, w.h_ctl_bytes);
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.ran_ev) (void)hipEventDestroy(w.ran_ev);
    }
    if (p->ready) (void)hipEventDestroy(p->ready);
    (void)hipFree(p->d_stamps);
    delete p;
}

uint64_t gams_wave_total_windows(const gams_wave_plan_t *p) { return p ? p->total_windows : 0; }

uint32_t gams_wave_ctg_windows(const gams_wave_plan_t *p, uint32_t i) {
    return (p && i < p->ctgs.size()) ? p->ctgs[i].n_win : 0;
}

int gams_wave_plan_set_tile(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t tile_windows) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_tile: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    int rc = wave_sync_ways(h, p);
    if (rc != GAMS_OK) return rc;
    rc = wave_build_geometry(h, p, tile_windows);
    if (rc != GAMS_OK) return rc;
    p->tw_req = tile_windows;
    p->ran = false;
    (void)hipFree(p->d_stamps);
    p->d_stamps = nullptr;
    rc = wave_upload_geometry(h, p);
    // a requested tile can move the plan from the tiled to the untiled kernels (prefix arrays beyond
    // the LDS), which need the dense rows
    if (rc == GAMS_OK) rc = wave_alloc_ways(h, p);
    return rc;
}

int gams_wave_plan_set_threads(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t threads) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_threads: null argument");
    if (threads != 0 && threads != 64 && threads != 128 && threads != 256)
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_threads: 0 (the library's choice), 64, 128 or 256");
    p->nth_req = threads;
    return gams_wave_plan_set_tile(h, p, p->tw_req);
}

int gams_wave_plan_set_guard(gams_gpu_t *h, gams_wave_plan_t *p, float safety, int all_exact) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_guard: null argument");
    if (!(safety >= 1.0f) || !(safety <= 1e6f))
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_guard: safety must be in [1, 1e6]");
    p->guard_safety = safety;
    p->guard_exact = all_exact != 0;
    wave_set_band(p);
    return GAMS_OK;
}

// ---- influence != 1: guess-and-iterate (wave_repair.hpp) ------------------------------------------------------
constexpr uint32_t kJacFirstBatch = 6;      // sweeps queued with the pass (3-5 settle the usual case), then batches of 8

static JacArgs wave_jac_args(gams_wave_plan_t *p, uint32_t k) {
    gams_wave_plan::Way &w = p->way[k];
    const JacBufs z = wave_jac_carve(w.d_jac, p->total_windows, p);
    JacArgs a{};
    a.tiles = p->d_jtiles;
    a.n_tiles = p->n_jtiles;
    a.cnt = w.d_dense_cnt;
    a.sig = w.d_dense_sig;
    a.xtab = p->d_xtab;
    a.f = z.f;
    a.fblk = z.fblk;
    a.ctl = z.ctl;
    a.lag = p->prm.lag;
    a.sweep = 0;
    a.thr = p->prm.threshold;
    a.influence = p->prm.influence;
    a.lastu = z.lastu;
    a.tile_last = z.tile_last;
    a.freeze = z.freeze;
    a.ftab = z.ftab;
    a.frow = z.frow;
    a.size1 = (uint32_t)p->prm.size + 1u;
    a.n_ctg = p->set->n_ctg;
    return a;
}

// the peaks of the dense rows into the tile slots (what the readers pack)
static int wave_compact_dense(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t k, hipStream_t st) {
    gams_wave_plan::Way &w = p->way[k];
    if (!(p->flags & GAMS_WAVE_PEAKS) || p->tiles.empty()) return GAMS_OK;
    hipLaunchKernelGGL(wave_compact_kernel, dim3((unsigned)p->tiles.size()), dim3(256), 0, st, p->d_ctgs, p->d_tiles, p->tw,
                       w.d_dense_cnt, w.d_dense_sig, w.d_peaks, p->tile_cap, w.d_tile_cnt);
    GAMS_HIP(h, hipGetLastError());
    return GAMS_OK;
}

// queue `n` more sweeps on way k's stream (first: with the initialisation in front), then the compaction of the dense
// rows and the copy of the control words
static int wave_jac_sweeps(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t k, uint32_t n, bool first) {
    gams_wave_plan::Way &w = p->way[k];
    hipStream_t st = wave_stream(h, p, k);
    JacArgs a = wave_jac_args(p, k);
    const unsigned grid = std::max(p->n_jtiles, 1u);
    const size_t lds = ((size_t)kJacTile + a.lag + 1) * sizeof(float);
    GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(jac_eval_kernel), lds));
    if (first) {
        hipLaunchKernelGGL(jac_init_kernel, dim3(grid), dim3(256), 0, st, a);
        if (p->jac0 && !w.jac0_table) {
            // the decisions behind a freeze point, once per way (the parameters belong to the plan)
            GAMS_HIP(h, hipMemsetAsync(const_cast<uint8_t *>(a.frow), 1, a.size1, st));
            hipLaunchKernelGGL(jac0_table_kernel, dim3((a.size1 * a.size1 + 255u) / 256u), dim3(256), 0, st,
                               const_cast<int8_t *>(a.ftab), const_cast<uint8_t *>(a.frow), a.xtab, a.size1, a.lag, a.thr);
            w.jac0_table = true;
        }
    }
    const uint32_t max_sweeps = p->jac0 ? kJacMaxSweeps0 : kJacMaxSweeps;
    if (p->n_jtiles)
        for (uint32_t i = 0; i < n && w.jac_sweeps < max_sweeps; ++i, ++w.jac_sweeps) {
            a.sweep = w.jac_sweeps;
            if (p->jac0) {
                const unsigned g0 = (grid + kJac0Group - 1u) / kJac0Group;
                hipLaunchKernelGGL(jac0_scan_kernel, dim3(g0), dim3(256), 0, st, a);
                hipLaunchKernelGGL(jac0_fill_kernel, dim3(g0), dim3(256), 0, st, a);
            } else {
                hipLaunchKernelGGL(jac_filter_kernel, dim3(grid), dim3(256), 0, st, a);
            }
            hipLaunchKernelGGL(jac_eval_kernel, dim3(grid), dim3(256), lds, st, a);
        }
    GAMS_HIP(h, hipGetLastError());
    int rc = wave_compact_dense(h, p, k, st);
    if (rc != GAMS_OK) return rc;
    GAMS_HIP(h, hipMemcpyAsync(w.h_ctl, a.ctl, kJacWords * 8, hipMemcpyDeviceToHost, st));
    return GAMS_OK;
}

// Has way k's pass reached its fixed point?  Waits for what is queued, looks at the control words, queues more sweeps
// while a sweep still flipped signals, and after kJacMaxSweeps (or at a run of thousands of signalled windows) lets the
// one-wavefront-per-ctg recurrence compute the batch from the counts.  Every reader of a pass calls this first.
static int wave_jac_settle(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t k) {
    if (!p->repair) return GAMS_OK;
    gams_wave_plan::Way &w = p->way[k];
    if (w.jac_settled || p->n_jtiles == 0) return GAMS_OK;
    hipStream_t st = wave_stream(h, p, k);
    for (;;) {
        GAMS_HIP(h, hipStreamSynchronize(st));
        bool fixed = false;
        for (uint32_t i = 0; i < w.jac_sweeps && !fixed; ++i) fixed = w.h_ctl[i] == 0ull;
        const bool abandon = w.h_ctl[kJacAbandon] != 0ull;
        if (fixed && !abandon) break;
        if (abandon || w.jac_sweeps >= (p->jac0 ? kJacMaxSweeps0 : kJacMaxSweeps)) {
            const gams_wave_params_t &q = p->prm;
            const uint32_t n = p->set->n_ctg;
            if (q.lag + 1u <= kSerialRing) {
                const size_t ring = (size_t)((q.lag + 1u + 63u) & ~63u) * sizeof(float);
                GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(wave_serial_wave_kernel), kSerialRing * sizeof(float)));
                hipLaunchKernelGGL(wave_serial_wave_kernel, dim3(n), dim3(64), ring, st, p->d_ctgs, n, w.d_dense_cnt,
                                   w.d_dense_sig, q.lag, q.threshold, q.influence, (float)q.size);
            } else {
                hipLaunchKernelGGL(wave_serial_kernel, dim3((n + 63) / 64), dim3(64), 0, st, p->d_ctgs, n, w.d_dense_cnt,
                                   w.d_dense_sig, wave_jac_carve(w.d_jac, p->total_windows).f, q.lag, q.threshold, q.influence,
                                   (float)q.size);
            }
            GAMS_HIP(h, hipGetLastError());
            int rc = wave_compact_dense(h, p, k, st);
            if (rc != GAMS_OK) return rc;
            GAMS_HIP(h, hipStreamSynchronize(st));
            w.jac_serial = true;
            break;
        }
        // (influence 0: the dense regime takes tens of sweeps, a host round trip per batch: 8, 16, 32, 32, ...)
        int rc = wave_jac_sweeps(h, p, k, !p->jac0 ? 8u : w.jac_sweeps < 14u ? 8u : w.jac_sweeps < 30u ? 16u : 32u, false);
        if (rc != GAMS_OK) return rc;
    }
    w.jac_settled = true;
    return GAMS_OK;
}

int gams_wave_plan_settled(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t *sweeps, int *serial) {
    if (!h || !p || !sweeps || !serial) return gams_fail(h, GAMS_EINVAL, "wave_plan_settled: null argument");
    if (!p->ran) return gams_fail(h, GAMS_ESTATE, "wave_plan_settled: no run to read");
    GAMS_HIP(h, hipSetDevice(h->device));
    const uint32_t k = wave_read_way_index(p);
    const int rc = wave_jac_settle(h, p, k);
    if (rc != GAMS_OK) return rc;
    *sweeps = p->repair ? p->way[k].jac_sweeps : 0u;
    *serial = p->repair ? (p->way[k].jac_serial ? 1 : 0) : (p->serial ? 1 : 0);
    return GAMS_OK;
}

// One pass on way k: touches only that way's state and its stream (run_n drives different ways
// from different host threads); everything else it reads is fixed once the plan exists.
static int wave_pass_on_way(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t k) {
    gams_wave_plan::Way &w = p->way[k];
    hipStream_t st = wave_stream(h, p, k);
    h->reader_epoch.fetch_add(1, std::memory_order_relaxed);   // a reader of the seqset is about to be queued
    // inputs: the way's stream queues behind the uploads and the plan's const table (no host wait)
    if (st == h->compute) {
        int wrc = gams_seqset_wait_uploads(h, p->set);
        if (wrc != GAMS_OK) return wrc;
    } else if (w.seen_upload != p->set->upload_gen && p->set->uploaded) {
        GAMS_HIP(h, hipStreamWaitEvent(st, p->set->uploaded, 0));
    }
    w.seen_upload = p->set->upload_gen;
    if (!w.seen_ready) {
        GAMS_HIP(h, hipStreamWaitEvent(st, p->ready, 0));
        w.seen_ready = true;
    }
    const uint32_t slot = (uint32_t)(w.runs % kCounterRing);
    if (slot == 0 && w.runs > 0)
        GAMS_HIP(h, hipMemsetAsync(w.d_counters, 0, kCounterRing * kSlotWords * sizeof(unsigned long long), st));
    ++w.runs;
    w.last_ring = slot;
    if (p->pipelined && !w.done) GAMS_HIP(h, hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    if (p->tiles.empty()) {
        if (p->pipelined) GAMS_HIP(h, hipEventRecord(w.done, st));
        return GAMS_OK;
    }
    const gams_wave_params_t &q = p->prm;
    WaveArgs a{};
    a.seq = p->set->d_seq;
    a.ctgs = p->d_ctgs;
    a.tiles = p->d_tiles;
    a.size = (uint32_t)q.size;
    a.step = (uint32_t)q.step;
    a.lag = q.lag;
    a.tw = p->tw;
    a.max_chunks = p->max_chunks;
    a.max_win = p->max_win;
    a.flags = p->serial ? GAMS_WAVE_DENSE : p->flags;
    a.no_signal = (q.lag < 2 || (p->serial && !p->repair)) ? 1u : 0u;
    a.thr = q.threshold;
    a.thr_abs = std::fabs(q.threshold);
    a.fsize = (float)q.size;
    a.flag_f = (float)q.lag;
    a.cvar = q.lag > 1 ? (float)((double)q.lag / ((double)q.lag - 1.0)) : 0.0f;
    a.g0 = p->g0;
    a.g1 = p->g1;
    a.g2 = p->g2;
    a.g3 = p->g3;
    a.aA = p->sq[0];
    a.aB = p->sq[1];
    a.gA0 = p->sq[2];
    a.gA1 = p->sq[3];
    a.gB0 = p->sq[4];
    a.gB1 = p->sq[5];
    a.peaks = w.d_peaks;
    a.tile_cap = p->tile_cap;
    a.counters = w.d_counters + kSlotWords * slot;
    a.const_sig = p->d_const_sig;
    a.stamps = p->d_stamps;
    a.tile_cnt = w.d_tile_cnt;
    a.dense_cnt = w.d_dense_cnt;
    a.dense_sig = w.d_dense_sig;
    if (p->direct) {
        const uint64_t total = p->total_windows;
        const unsigned blocks = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(wave_direct_count_kernel, dim3(blocks), dim3(256), 0, st, p->set->d_seq, p->d_ctgs,
                           p->set->n_ctg, total, (uint32_t)q.size, (uint32_t)q.step, w.d_dense_cnt);
        GAMS_HIP(h, hipGetLastError());
        if (!p->serial || p->repair) {
            hipLaunchKernelGGL(wave_direct_signal_kernel, dim3(blocks), dim3(256), 0, st, p->d_ctgs, p->set->n_ctg,
                               total, w.d_dense_cnt, w.d_dense_sig, q.lag, q.threshold, (float)q.size,
                               q.lag < 2 ? 1u : 0u);
            GAMS_HIP(h, hipGetLastError());
        }
    }
    int rc = GAMS_OK;
    const int kind = p->fast_w ? wave_baked_kind(q, p->fast_w) : 0;
    const bool baked = kind == 1;
    const bool step1 = baked && q.step == 1;
    if (p->direct)
        ;   // counted and decided above
    else if (kind == 2) {
        // size and step baked, lag from the arguments
#define GAMS_RL(WW, ST) rc = wave_launch_fast<WW, 100, ST, 0>(h, p, a, st)
        const int key = p->fast_w * 100 + q.step;
        switch (key) {
        case 405: GAMS_RL(4, 5); break;
        case 805: GAMS_RL(8, 5); break;
        case 1205: GAMS_RL(12, 5); break;
        case 410: GAMS_RL(4, 10); break;
        case 810: GAMS_RL(8, 10); break;
        case 1210: GAMS_RL(12, 10); break;
        case 420: GAMS_RL(4, 20); break;
        case 820: GAMS_RL(8, 20); break;
        case 1220: GAMS_RL(12, 20); break;
        case 2001: GAMS_RL(20, 1); break;
        case 2005:
            rc = p->nth == 64    ? wave_launch_fast<20, 100, 5, 0, 64>(h, p, a, st)
                 : p->nth == 128 ? wave_launch_fast<20, 100, 5, 0, 128>(h, p, a, st)
                                 : wave_launch_fast<20, 100, 5, 0>(h, p, a, st);
            break;
        case 2801:
            rc = p->nth == 64    ? wave_launch_fast<28, 100, 1, 0, 64>(h, p, a, st)
                 : p->nth == 128 ? wave_launch_fast<28, 100, 1, 0, 128>(h, p, a, st)
                                 : wave_launch_fast<28, 100, 1, 0>(h, p, a, st);
            break;
        default: rc = gams_fail(h, GAMS_ESTATE, "wave: no kernel for this tile size / step"); break;
        }
#undef GAMS_RL
    } else if (p->fast_w == 28)
        rc = p->nth == 64    ? wave_launch_fast<28, 100, 1, 100, 64>(h, p, a, st)   // baked only (wave_build_geometry)
             : p->nth == 128 ? wave_launch_fast<28, 100, 1, 100, 128>(h, p, a, st)
                             : wave_launch_fast<28, 100, 1, 100>(h, p, a, st);
    else if (p->fast_w == 20)
        rc = baked ? wave_launch_fast<20, 100, 1, 100>(h, p, a, st) : wave_launch_fast<20, 0, 0, 0>(h, p, a, st);
    else if (p->taper)
        rc = p->set->bytes > kStreamBytes ? wave_launch_taper<true>(h, p, a, st) : wave_launch_taper<false>(h, p, a, st);
    else if (p->fast_w == 12)
        rc = !baked ? wave_launch_fast<12, 0, 0, 0>(h, p, a, st)
             : step1 ? wave_launch_fast<12, 100, 1, 100>(h, p, a, st)
             : p->nth == 64 ? wave_launch_fast<12, 100, 10, 100, 64>(h, p, a, st)
             : p->nth == 128 ? wave_launch_fast<12, 100, 10, 100, 128>(h, p, a, st)
                     : wave_launch_fast<12, 100, 10, 100>(h, p, a, st);
    else if (p->fast_w == 8)
        rc = baked ? wave_launch_fast<8, 100, 10, 100>(h, p, a, st) : wave_launch_fast<8, 0, 0, 0>(h, p, a, st);
    else if (p->fast_w == 4)
        rc = baked ? wave_launch_fast<4, 100, 10, 100>(h, p, a, st) : wave_launch_fast<4, 0, 0, 0>(h, p, a, st);
    else if (p->k16)
        rc = p->wide ? wave_launch<uint16_t, true>(h, p, a, st) : wave_launch<uint16_t, false>(h, p, a, st);
    else
        rc = p->wide ? wave_launch<uint8_t, true>(h, p, a, st) : wave_launch<uint8_t, false>(h, p, a, st);
    if (rc != GAMS_OK) return rc;
    if (p->direct && !p->serial && (p->flags & GAMS_WAVE_PEAKS)) {
        hipLaunchKernelGGL(wave_compact_kernel, dim3((unsigned)p->tiles.size()), dim3(256), 0, st, p->d_ctgs,
                           p->d_tiles, p->tw, w.d_dense_cnt, w.d_dense_sig, w.d_peaks, p->tile_cap, w.d_tile_cnt);
        GAMS_HIP(h, hipGetLastError());
    }
    if (p->repair) {
        // influence != 1, guess-and-iterate (wave_repair.hpp): the dense rows hold the counts and the S1 signals; a first
        // batch of sweeps is queued behind them -- no host wait here; whoever reads the pass checks that it settled
        // (wave_jac_settle) and queues more sweeps if it did not
        w.jac_sweeps = 0;
        w.jac_settled = false;
        w.jac_serial = false;
        rc = wave_jac_sweeps(h, p, k, kJacFirstBatch, true);
        if (rc != GAMS_OK) return rc;
    } else if (p->serial) {
        const uint32_t n = p->set->n_ctg;
        if (q.lag + 1u <= kSerialRing) {
            // one wavefront per ctg, the last lag + 1 filtered values in an LDS ring
            const size_t ring = (size_t)((q.lag + 1u + 63u) & ~63u) * sizeof(float);
            GAMS_HIP(h, gams_lds_attr(h, reinterpret_cast<const void *>(wave_serial_wave_kernel), kSerialRing * sizeof(float)));
            hipLaunchKernelGGL(wave_serial_wave_kernel, dim3(n), dim3(64), ring, st, p->d_ctgs, n, w.d_dense_cnt,
                               w.d_dense_sig, q.lag, q.threshold, q.influence, (float)q.size);
        } else {
            hipLaunchKernelGGL(wave_serial_kernel, dim3((n + 63) / 64), dim3(64), 0, st, p->d_ctgs, n, w.d_dense_cnt,
                               w.d_dense_sig, w.d_filtered, q.lag, q.threshold, q.influence, (float)q.size);
        }
        GAMS_HIP(h, hipGetLastError());
        if (p->flags & GAMS_WAVE_PEAKS) {
            hipLaunchKernelGGL(wave_compact_kernel, dim3((unsigned)p->tiles.size()), dim3(256), 0, st, p->d_ctgs,
                               p->d_tiles, p->tw, w.d_dense_cnt, w.d_dense_sig, w.d_peaks, p->tile_cap, w.d_tile_cnt);
            GAMS_HIP(h, hipGetLastError());
        }
    }
    // An event per run costs a marker packet between back-to-back launches (measured: 8.8 ->
    // 11.7 us per 12-Mb pass), so it is only recorded for plans that overlap several runs.
    if (p->pipelined) GAMS_HIP(h, hipEventRecord(w.done, st));
    return GAMS_OK;
}

// wait for the run the readers look at: its own event in pipelined mode, else its whole stream
static int wave_wait_last_run(gams_gpu_t *h, gams_wave_plan_t *p) {
    gams_wave_plan::Way &w = wave_read_way(p);
    if (p->pipelined && w.done)
        GAMS_HIP(h, hipEventSynchronize(w.done));
    else
        GAMS_HIP(h, hipStreamSynchronize(wave_stream(h, p, wave_read_way_index(p))));
    return GAMS_OK;
}

int gams_wave_plan_set_pipelined(gams_gpu_t *h, gams_wave_plan_t *p, int enable) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_pipelined: null argument");
    p->pipelined = enable != 0;
    return GAMS_OK;
}

int gams_wave_run(gams_gpu_t *h, gams_wave_plan_t *p) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_run: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    const uint32_t k = (uint32_t)(p->run_idx % p->depth);
    int rc = wave_pass_on_way(h, p, k);
    if (rc != GAMS_OK) return rc;
    ++p->run_idx;
    p->last_way = k;
    p->sel_age = 0;
    p->ran = true;
    return GAMS_OK;
}

// The second queueing thread of gams_wave_run_n.  It lives as long as the plan (binding a new host
// thread to the device costs ~100 us, more than a short batch), sleeps between batches and, once
// armed, spins until the caller has queued the first round.
// A queueing thread of gams_wave_run_n.  It lives as long as the plan (binding a new host thread
// to the device costs ~100 us, more than a short batch), sleeps between batches and, once armed,
// spins until the caller has queued the first round.
struct Launcher {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool armed = false, quit = false, finished = true;
    std::atomic<int> go{0};            // 0: wait, 1: run the job, 2: skip it
    gams_gpu_t *h = nullptr;
    gams_wave_plan_t *p = nullptr;
    uint64_t first = 0;
    uint32_t rest = 0;
    uint32_t share = 0, shares = 1;    // this thread queues the ways k with k % shares == share
    int rc = GAMS_OK;
};

static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

// passes first .. first+rest-1 whose way belongs to `share` (way order is kept per way)
static int wave_queue_share(gams_gpu_t *h, gams_wave_plan_t *p, uint64_t first, uint32_t rest, uint32_t share,
                            uint32_t shares) {
    for (uint32_t j = 0; j < rest; ++j) {
        const uint32_t k = (uint32_t)((first + j) % p->depth);
        if (k % shares != share) continue;
        const int rc = wave_pass_on_way(h, p, k);
        if (rc != GAMS_OK) return rc;
    }
    return GAMS_OK;
}

static void wave_launcher_main(Launcher *L, int device) {
    const bool bound = hipSetDevice(device) == hipSuccess;
    std::unique_lock<std::mutex> lk(L->mu);
    for (;;) {
        L->cv.wait(lk, [&] { return L->armed || L->quit; });
        if (L->quit) return;
        L->armed = false;
        lk.unlock();
        int g;
        while ((g = L->go.load(std::memory_order_acquire)) == 0) cpu_relax();   // a few microseconds
        int rc = GAMS_OK;
        if (g == 1) rc = bound ? wave_queue_share(L->h, L->p, L->first, L->rest, L->share, L->shares) : GAMS_EHIP;
        lk.lock();
        L->rc = rc;
        L->finished = true;
        L->cv.notify_all();
    }
}

static void wave_launcher_stop(gams_wave_plan_t *p) {
    for (Launcher *&L : p->launcher) {
        if (!L) continue;
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->quit = true;
        }
        L->cv.notify_all();
        L->th.join();
        delete L;
        L = nullptr;
    }
}

int gams_wave_run_n(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t n) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_run_n: null argument");
    // One host thread queues a pass every ~3.3 us (hipLaunchKernelGGL); with three or four passes
    // in flight that is slower than the device drains them (2.85 us per 12-Mb pass), so a batch is
    // queued by several threads: the caller keeps way 0, launcher threads take the other ways
    // (K = 1000 passes at depth 4: 4.4 / 3.1 / 3.0 / 3.0 us per pass with 1 / 2 / 3 / 4 threads;
    // K = 200: 3.8 / 3.8 / 3.6 / 3.5).  The first round goes through gams_wave_run on the caller: it
    // applies the kernel attribute and the first-use stream waits.
    const uint32_t shares = std::min(p->queue_threads, p->depth);   // default: one queueing thread per way
    const bool threaded = p->depth >= 3 && shares >= 2 && n >= 4 * p->depth;
    if (!threaded) {
        for (uint32_t i = 0; i < n; ++i) {
            int rc = gams_wave_run(h, p);
            if (rc != GAMS_OK) return rc;
        }
        return GAMS_OK;
    }
    const uint32_t lead = p->depth;
    const uint32_t rest = n - lead;
    const uint64_t first = p->run_idx + lead;     // index of the first pass the threads share
    for (uint32_t t = 1; t < shares; ++t) {
        Launcher *&L = p->launcher[t - 1];
        if (!L) {
            L = new Launcher();
            L->th = std::thread(wave_launcher_main, L, h->device);
        }
        {
            std::lock_guard<std::mutex> lk(L->mu);
            L->h = h;
            L->p = p;
            L->first = first;
            L->rest = rest;
            L->share = t;
            L->shares = shares;
            L->go.store(0, std::memory_order_relaxed);
            L->finished = false;
            L->armed = true;
        }
        L->cv.notify_all();                        // wakes up while the first round is queued
    }
    int rc = GAMS_OK;
    for (uint32_t i = 0; i < lead && rc == GAMS_OK; ++i) rc = gams_wave_run(h, p);
    for (uint32_t t = 1; t < shares; ++t)
        p->launcher[t - 1]->go.store(rc == GAMS_OK ? 1 : 2, std::memory_order_release);
    if (rc == GAMS_OK) rc = wave_queue_share(h, p, first, rest, 0u, shares);
    for (uint32_t t = 1; t < shares; ++t) {
        Launcher *L = p->launcher[t - 1];
        std::unique_lock<std::mutex> lk(L->mu);
        L->cv.wait(lk, [&] { return L->finished; });
        if (rc == GAMS_OK) rc = L->rc;
    }
    if (rc != GAMS_OK) return rc;
    p->run_idx = first + rest;
    p->last_way = (uint32_t)((p->run_idx - 1) % p->depth);
    p->sel_age = 0;
    p->ran = true;
    return GAMS_OK;
}

int gams_wave_plan_set_depth(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t depth) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_depth: null argument");
    if (depth < 1 || depth > (uint32_t)gams_gpu::kMaxWays)
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_depth: depth must be 1.." + std::to_string(gams_gpu::kMaxWays));
    GAMS_HIP(h, hipSetDevice(h->device));
    int rc = wave_sync_ways(h, p);
    if (rc != GAMS_OK) return rc;
    for (uint32_t k = 1; k < depth; ++k)
        if (!h->aux[k - 1]) GAMS_HIP(h, hipStreamCreateWithFlags(&h->aux[k - 1], hipStreamNonBlocking));
    p->depth = depth;
    p->ran = false;
    p->run_idx = 0;
    p->last_way = 0;
    p->sel_age = 0;
    rc = wave_build_geometry(h, p, p->tw_req);      // the default tile depends on the depth
    if (rc != GAMS_OK) return rc;
    (void)hipFree(p->d_stamps);                     // sized for the old tiling
    p->d_stamps = nullptr;
    rc = wave_upload_geometry(h, p);
    if (rc == GAMS_OK) rc = wave_alloc_ways(h, p);
    return rc;
}

int gams_wave_plan_set_taper(gams_gpu_t *h, gams_wave_plan_t *p, int mode) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_taper: null argument");
    if (mode < -1 || mode > 1) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_taper: mode must be -1 (auto), 0 or 1");
    GAMS_HIP(h, hipSetDevice(h->device));
    int rc = wave_sync_ways(h, p);
    if (rc != GAMS_OK) return rc;
    p->taper_req = mode;
    const bool before = p->taper;
    rc = wave_build_geometry(h, p, p->tw_req);
    if (rc != GAMS_OK) return rc;
    if (p->taper == before) return GAMS_OK;       // same tile table
    p->ran = false;
    (void)hipFree(p->d_stamps);
    p->d_stamps = nullptr;
    rc = wave_upload_geometry(h, p);
    if (rc == GAMS_OK) rc = wave_alloc_ways(h, p);
    return rc;
}

int gams_wave_plan_set_taper_shape(gams_gpu_t *h, gams_wave_plan_t *p, int pct4, int pct8) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_taper_shape: null argument");
    if (pct4 < 0 || pct4 > 100 || pct8 < 0 || pct8 > 100)
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_taper_shape: percentages must be 0..100");
    GAMS_HIP(h, hipSetDevice(h->device));
    int rc = wave_sync_ways(h, p);
    if (rc != GAMS_OK) return rc;
    p->taper4_pct = pct4;
    p->taper8_pct = pct8;
    rc = wave_build_geometry(h, p, p->tw_req);
    if (rc != GAMS_OK) return rc;
    p->ran = false;
    (void)hipFree(p->d_stamps);
    p->d_stamps = nullptr;
    rc = wave_upload_geometry(h, p);
    if (rc == GAMS_OK) rc = wave_alloc_ways(h, p);
    return rc;
}

int gams_wave_plan_set_queue_threads(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t n) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_queue_threads: null argument");
    if (n < 1 || n > (uint32_t)gams_gpu::kMaxWays)
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_queue_threads: 1..4");
    p->queue_threads = n;
    return GAMS_OK;
}

int gams_wave_plan_kernel_name(gams_gpu_t *h, gams_wave_plan_t *p, char *buf, size_t n) {
    if (!h || !p || !buf || n == 0) return gams_fail(h, GAMS_EINVAL, "wave_plan_kernel_name: null argument");
    // the same decisions as wave_pass_on_way, spelled the way rocprofv3 prints the instantiation
    const gams_wave_params_t &q = p->prm;
    const int kind = p->fast_w ? wave_baked_kind(q, p->fast_w) : 0;
    const bool baked = kind != 0;
    const char *nt = p->set->bytes > kStreamBytes ? "true" : "false";
    std::string name;
    if (p->repair)
        name = "jac_eval_kernel";      // of the pass's kernels (counts + S1 signals, sweeps) the one that takes longest
    else if (p->serial)
        name = q.lag + 1u <= kSerialRing ? "wave_serial_wave_kernel" : "wave_serial_kernel";
    else if (p->direct)
        name = "wave_direct_count_kernel + wave_direct_signal_kernel";
    else if (p->taper)
        name = std::string("wave_fast_taper_kernel<100, 10, 100, ") + nt + ">";
    else if (p->fast_w) {
        const std::string prm = baked ? "100, " + std::to_string(q.step) + (kind == 1 ? ", 100, " : ", 0, ") : "0, 0, 0, ";
        name = "wave_fast_kernel<" + std::to_string(p->fast_w) + ", " + prm + nt + ", " + std::to_string(p->nth) + ">";
    } else
        name = std::string("wave_tile_kernel<") + (p->k16 ? "unsigned short, " : "unsigned char, ") + (p->wide ? "true>" : "false>");
    std::snprintf(buf, n, "%s", name.c_str());
    return GAMS_OK;
}

int gams_wave_plan_set_lane(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t lane) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_lane: null argument");
    if (lane >= (uint32_t)gams_gpu::kMaxWays)
        return gams_fail(h, GAMS_EINVAL, "wave_plan_set_lane: lane must be 0.." + std::to_string(gams_gpu::kMaxWays - 1));
    GAMS_HIP(h, hipSetDevice(h->device));
    int rc = wave_sync_ways(h, p);
    if (rc != GAMS_OK) return rc;
    for (uint32_t k = 1; k < (uint32_t)gams_gpu::kMaxWays; ++k)
        if (!h->aux[k - 1]) GAMS_HIP(h, hipStreamCreateWithFlags(&h->aux[k - 1], hipStreamNonBlocking));
    p->lane = lane;
    // the ways now run on other streams: they have to queue behind the uploads and the const table again
    for (auto &w : p->way) {
        w.seen_upload = 0;
        w.seen_ready = false;
    }
    p->set->dirty = true;
    return GAMS_OK;
}

int gams_wave_plan_select(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t age) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_select: null argument");
    if (age >= p->depth || age >= p->run_idx)
        return gams_fail(h, GAMS_ESTATE, "wave_plan_select: that run is no longer (or not yet) held");
    p->sel_age = age;
    return GAMS_OK;
}

int gams_wave_peaks(gams_gpu_t *h, gams_wave_plan_t *p, const gams_peak_t **peaks, uint64_t *n_peaks) {
    if (!h || !p || !peaks || !n_peaks) return gams_fail(h, GAMS_EINVAL, "wave_peaks: null argument");
    if (!(p->flags & GAMS_WAVE_PEAKS)) return gams_fail(h, GAMS_ESTATE, "wave_peaks: plan has no PEAKS output");
    if (!p->ran) return gams_fail(h, GAMS_ESTATE, "wave_peaks: no run to read");
    GAMS_HIP(h, hipSetDevice(h->device));
    const size_t nt = p->tiles.size();
    if (nt == 0) {
        *peaks = nullptr;
        *n_peaks = 0;
        return GAMS_OK;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        {
            const int src = wave_jac_settle(h, p, wave_read_way_index(p));   // influence != 1: the pass has reached its fixed point
            if (src != GAMS_OK) return src;
        }
        // two words behind the offsets (re-read every attempt: a regrow replaces the arena)
        unsigned long long *const d_totals = p->d_tile_off + nt;
        // The readback stream queues behind the run (pipelined: behind this plan's run only, later
        // runs of other plans keep going); the host waits once, for the two totals.
        gams_wave_plan::Way &w = wave_read_way(p);
        if (p->pipelined && w.done) {
            GAMS_HIP(h, hipStreamWaitEvent(h->readback, w.done, 0));
        } else {
            if (!w.ran_ev) GAMS_HIP(h, hipEventCreateWithFlags(&w.ran_ev, hipEventDisableTiming));
            GAMS_HIP(h, hipEventRecord(w.ran_ev, wave_stream(h, p, wave_read_way_index(p))));
            GAMS_HIP(h, hipStreamWaitEvent(h->readback, w.ran_ev, 0));
        }
        if (!p->d_dense) {
            // typical density is 1-3 % of the windows; a fuller result regrows below
            const uint64_t want = p->total_windows / 16 + 4096;
            GAMS_HIP(h, gams_pool_alloc(h, false, want * sizeof(gams_peak_t),
                                        reinterpret_cast<void **>(&p->d_dense), &p->d_dense_bytes));
            p->dense_cap = p->d_dense_bytes / sizeof(gams_peak_t);
        }
        if (nt <= kOffOneGroup) {
            hipLaunchKernelGGL(wave_offsets_kernel, dim3(1), dim3(1024), 0, h->readback, w.d_tile_cnt, (uint32_t)nt,
                               p->d_tile_off, d_totals);
        } else {
            const unsigned spans = (unsigned)((nt + kOffSpan - 1) / kOffSpan);
            hipLaunchKernelGGL(wave_offsets_sum_kernel, dim3(spans), dim3(1024), 0, h->readback, w.d_tile_cnt, (uint32_t)nt,
                               p->d_tile_off);
            hipLaunchKernelGGL(wave_offsets_base_kernel, dim3(1), dim3(1024), 0, h->readback, (uint32_t)nt, p->d_tile_off,
                               d_totals);
            hipLaunchKernelGGL(wave_offsets_scan_kernel, dim3(spans), dim3(1024), 0, h->readback, w.d_tile_cnt, (uint32_t)nt,
                               p->d_tile_off);
        }
        GAMS_HIP(h, hipGetLastError());
        hipLaunchKernelGGL(wave_gather_kernel, dim3((unsigned)nt), dim3(64), 0, h->readback, w.d_peaks,
                           p->tile_cap, w.d_tile_cnt, p->d_tile_off, p->d_dense, (unsigned long long)p->dense_cap);
        GAMS_HIP(h, hipGetLastError());
        GAMS_HIP(h, hipMemcpyAsync(h->pin_scratch, d_totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                                   h->readback));
        GAMS_HIP(h, hipStreamSynchronize(h->readback));
        const uint64_t total = h->pin_scratch[0];
        const uint64_t worst = h->pin_scratch[1];
        // Buffers below are sized from these two device-written words: refuse values no pass over this
        // plan can produce (a tile signals at most its own windows, all tiles together at most the
        // batch's) instead of allocating and copying by them.
        if (worst > p->tw || total > p->total_windows || total > (uint64_t)nt * worst)
            return gams_fail(h, GAMS_EHIP,
                             "wave_peaks: inconsistent peak counts from the device (total " + std::to_string(total) +
                                 ", fullest tile " + std::to_string(worst) + ", " + std::to_string(p->total_windows) +
                                 " windows in " + std::to_string(nt) + " tiles of at most " + std::to_string(p->tw) + ")");
        if (worst > p->tile_cap) {
            // some tile signalled more windows than its slot holds: the device has reported the
            // fullest tile (runs are deterministic, so that is what the slots need -- at GRCh38 step 1 a
            // slot of tw records per tile would be ~50 GB per way) and every pass still held is run
            // again into the new slots: same inputs, same results.  The run counter only steps back by
            // the passes repeated, so the way rotation and the history gams_wave_plan_select sees stay.
            p->tile_cap_req = (uint32_t)std::min<uint64_t>(p->tw, (worst + 15u) & ~(uint64_t)15u);
            const uint32_t age = p->sel_age;
            const uint32_t again = (uint32_t)std::min<uint64_t>(p->depth, p->run_idx);
            int rc = wave_sync_ways(h, p);
            if (rc == GAMS_OK) rc = wave_upload_geometry(h, p);
            if (rc == GAMS_OK) {
                p->run_idx -= again;
                for (uint32_t k = 0; k < again && rc == GAMS_OK; ++k) rc = gams_wave_run(h, p);
                p->sel_age = age;
            }
            if (rc != GAMS_OK) return rc;
            continue;
        }
        if (total) {
            if (total > p->dense_cap) {
                gams_pool_free(h, false, p->d_dense, p->d_dense_bytes);
                p->d_dense = nullptr;
                p->dense_cap = 0;
                const uint64_t want = total + total / 4 + 1024;
                GAMS_HIP(h, gams_pool_alloc(h, false, want * sizeof(gams_peak_t),
                                            reinterpret_cast<void **>(&p->d_dense), &p->d_dense_bytes));
                p->dense_cap = p->d_dense_bytes / sizeof(gams_peak_t);
                hipLaunchKernelGGL(wave_gather_kernel, dim3((unsigned)nt), dim3(64), 0, h->readback, w.d_peaks,
                                   p->tile_cap, w.d_tile_cnt, p->d_tile_off, p->d_dense,
                                   (unsigned long long)p->dense_cap);
                GAMS_HIP(h, hipGetLastError());
            }
            if (total * sizeof(gams_peak_t) > p->h_peaks_bytes) {
                gams_pool_free(h, true, p->h_peaks, p->h_peaks_bytes);
                p->h_peaks = nullptr;
                p->h_peaks_bytes = 0;
                GAMS_HIP(h, gams_pool_alloc(h, true, (total + total / 4 + 1024) * sizeof(gams_peak_t),
                                            reinterpret_cast<void **>(&p->h_peaks), &p->h_peaks_bytes));
            }
                GAMS_HIP(h, hipMemcpyAsync(p->h_peaks, p->d_dense, total * sizeof(gams_peak_t),
                                       hipMemcpyDeviceToHost, h->readback));
                GAMS_HIP(h, hipStreamSynchronize(h->readback));
            }
        *peaks = p->h_peaks;   // NULL when there is none
        *n_peaks = total;
        return GAMS_OK;
    }
    return gams_fail(h, GAMS_EHIP, "wave_peaks: peak buffer overflow persisted");
}

// ---- rows as text ---------------------------------------------------------------------------------
}  // extern "C"
namespace {
// Rust's `{}` of an f32: the shortest digits that round-trip, positional notation (what wave.rs:196 prints)
std::string rows_fmt_f32(float v) {
    if (v == 0.0f) return "0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
    std::string sci(buf, r.ptr), digits, out;             // d[.ddd]e[+-]XX  (v > 0 here)
    size_t i = 0;
    for (; i < sci.size() && sci[i] != 'e'; ++i)
        if (sci[i] != '.') digits += sci[i];
    const int ex = std::atoi(sci.c_str() + i + 1), nd = (int)digits.size();
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

struct RowTables {
    uint8_t *flags;
    int2 *headpos, *blk_head;
    uint32_t *tailwin, *len, *blk_len;
    unsigned long long *blk_off;
    uint32_t nb_cap;
    size_t bytes;
};
RowTables rows_carve(uint8_t *base, uint64_t cap) {
    const uint32_t nb = (uint32_t)((cap + kRowsBlock - 1) / kRowsBlock);
    size_t o = 0;
    auto take = [&](size_t b) {
        uint8_t *q = base ? base + o : nullptr;
        o += wave_align256(b);
        return q;
    };
    RowTables t{};
    t.headpos = reinterpret_cast<int2 *>(take(cap * 8));
    t.tailwin = reinterpret_cast<uint32_t *>(take(cap * 4));
    t.len = reinterpret_cast<uint32_t *>(take(cap * 4));
    t.flags = take(cap);
    t.blk_head = reinterpret_cast<int2 *>(take((size_t)nb * 8));
    t.blk_len = reinterpret_cast<uint32_t *>(take((size_t)nb * 4));
    t.blk_off = reinterpret_cast<unsigned long long *>(take(((size_t)nb + 2) * 8));
    t.nb_cap = nb;
    t.bytes = o;
    return t;
}

// The kernels and copies of one rows pass over way `wi`, in order: launched on `st` as they are (graph == nullptr), or
// added to `graph` as a chain of nodes.  (Built node by node rather than by stream capture: a capture is invalidated
// by what OTHER host threads do meanwhile, and the host layer runs one thread per handle.)
int rows_emit(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t wi, hipStream_t st, uint64_t copy_bytes, hipGraph_t graph) {
    WaveRows *r = p->rows;
    const size_t nt = p->tiles.size();
    unsigned long long *d_totals = p->d_tile_off + nt;
    gams_wave_plan::Way &w = p->way[wi];
    const RowTables t = rows_carve(r->tmp, r->cap);
    const uint32_t n_ctg = p->set->n_ctg;
    RowArgs a{};
    a.rec = p->d_dense;
    a.n_rec = d_totals;
    a.cap = std::min<uint64_t>(r->cap, p->dense_cap);
    a.tile_cap = p->tile_cap;
    a.ctgs = r->d_ctgs;
    a.n_ctg = n_ctg;
    a.names = r->d_names;
    a.gctab = r->d_gctab;
    a.size = (uint32_t)p->prm.size;
    a.step = (uint32_t)p->prm.step;
    a.dmax = r->dmax;
    a.flags = t.flags;
    a.headpos = t.headpos;
    a.blk_head = t.blk_head;
    a.tailwin = t.tailwin;
    a.len = t.len;
    a.blk_len = t.blk_len;
    a.blk_off = t.blk_off;
    a.nb_cap = t.nb_cap;
    a.text = r->d_text;
    a.text_cap = r->d_text_bytes;
    a.words = r->d_words;
    hipGraphNode_t last = nullptr;
    hipError_t err = hipSuccess;
    auto kernel = [&](const void *fn, unsigned grid, unsigned block, void **args) {
        if (err != hipSuccess) return;
        if (!graph) {
            err = hipLaunchKernel(fn, dim3(grid), dim3(block), args, 0, st);
            return;
        }
        hipKernelNodeParams kp{};
        kp.func = const_cast<void *>(fn);
        kp.gridDim = dim3(grid);
        kp.blockDim = dim3(block);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        kp.extra = nullptr;
        hipGraphNode_t node = nullptr;
        err = hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &kp);
        last = node;
    };
    auto copy = [&](void *dst, const void *src, size_t bytes) {
        if (err != hipSuccess || bytes == 0) return;
        if (!graph) {
            err = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
            return;
        }
        hipGraphNode_t node = nullptr;
        err = hipGraphAddMemcpyNode1D(&node, graph, last ? &last : nullptr, last ? 1 : 0, dst, src, bytes, hipMemcpyDeviceToHost);
        last = node;
    };
    // argument blocks (read when the launch / the node is made)
    uint32_t *tile_cnt = w.d_tile_cnt;
    uint32_t nt32 = (uint32_t)nt;
    unsigned long long *tile_off = p->d_tile_off;
    void *args_off1[] = {&tile_cnt, &nt32, &tile_off, &d_totals};
    const gams_peak_t *slots = w.d_peaks;
    uint32_t tile_cap = p->tile_cap;
    const uint32_t *ctile_cnt = w.d_tile_cnt;
    const unsigned long long *ctile_off = p->d_tile_off;
    gams_peak_t *dense = p->d_dense;
    unsigned long long dense_cap = p->dense_cap;
    void *args_gather[] = {&slots, &tile_cap, &ctile_cnt, &ctile_off, &dense, &dense_cap};
    void *args_rows[] = {&a};
    const uint32_t *blk_len = t.blk_len;
    uint32_t nb = t.nb_cap;
    unsigned long long *blk_off = t.blk_off, *blk_tot = t.blk_off + nb;
    void *args_off2[] = {&blk_len, &nb, &blk_off, &blk_tot};
    // (exclusive prefix of n counts: one workgroup, or spans of kOffSpan counts in three launches -- wave_kernels.hpp)
    auto offsets = [&](void **all4, void **cnt_n_off, void **n_off_tot, uint32_t n) {
        if (n <= kOffOneGroup) {
            kernel(reinterpret_cast<const void *>(wave_offsets_kernel), 1, 1024, all4);
            return;
        }
        const unsigned spans = (n + kOffSpan - 1u) / kOffSpan;
        kernel(reinterpret_cast<const void *>(wave_offsets_sum_kernel), spans, 1024, cnt_n_off);
        kernel(reinterpret_cast<const void *>(wave_offsets_base_kernel), 1, 1024, n_off_tot);
        kernel(reinterpret_cast<const void *>(wave_offsets_scan_kernel), spans, 1024, cnt_n_off);
    };
    void *args_off1a[] = {&tile_cnt, &nt32, &tile_off}, *args_off1b[] = {&nt32, &tile_off, &d_totals};
    offsets(args_off1, args_off1a, args_off1b, nt32);
    kernel(reinterpret_cast<const void *>(wave_gather_kernel), (unsigned)nt, 64, args_gather);
    kernel(reinterpret_cast<const void *>(rows_link_kernel), nb, 256, args_rows);
    kernel(reinterpret_cast<const void *>(rows_heads_kernel), 1, 1024, args_rows);
    kernel(reinterpret_cast<const void *>(rows_tail_kernel), (unsigned)((r->cap + 255) / 256), 256, args_rows);
    kernel(reinterpret_cast<const void *>(rows_len_kernel), nb, 256, args_rows);
    void *args_off2a[] = {&blk_len, &nb, &blk_off}, *args_off2b[] = {&nb, &blk_off, &blk_tot};
    offsets(args_off2, args_off2a, args_off2b, nb);
    kernel(reinterpret_cast<const void *>(rows_write_kernel), nb, 256, args_rows);
    // the words (sizes, totals, per-ctg offsets: one block) and -- sized by the previous pass -- the text itself go to the
    // host behind the kernels
    copy(r->h_words, r->d_words, ((size_t)n_ctg + 1 + 4) * 8);
    copy(r->h_text, r->d_text, copy_bytes);
    if (err != hipSuccess) {
        (void)hipGetLastError();
        return gams_fail(h, GAMS_EHIP, std::string("wave_rows: ") + hipGetErrorString(err));
    }
    return GAMS_OK;
}

// queue everything of one rows pass on the readback stream, behind the run the readers look at.  The dozen launches
// and copies are one graph per way (a plan's buffers do not move between passes; the graph is rebuilt when one does): one
// call instead of ten on the host thread that also queues the passes themselves.
int rows_queue(gams_gpu_t *h, gams_wave_plan_t *p) {
    WaveRows *r = p->rows;
    const uint32_t wi = wave_read_way_index(p);
    gams_wave_plan::Way &w = p->way[wi];
    {
        const int src = wave_jac_settle(h, p, wi);      // influence != 1: the pass has reached its fixed point
        if (src != GAMS_OK) return src;
    }
    // The rows go on the stream the pass itself ran on: ordered behind it without an event, and the rows of plans
    // on different lanes run side by side (eight short dependent kernels and a 2-MB copy per batch: on one shared
    // stream they were 117 us per batch for three plans in flight, the passes themselves 22).
    hipStream_t st = wave_stream(h, p, wi);
    if (!p->d_dense) {
        const uint64_t want = p->total_windows / 16 + 4096;
        GAMS_HIP(h, gams_pool_alloc(h, false, want * sizeof(gams_peak_t), reinterpret_cast<void **>(&p->d_dense),
                                    &p->d_dense_bytes));
        p->dense_cap = p->d_dense_bytes / sizeof(gams_peak_t);
    }
    const uint64_t cap = std::min<uint64_t>(p->dense_cap, 0x7FFFFF00ull);
    if (r->cap < cap) {
        gams_pool_free(h, false, r->tmp, r->tmp_bytes);
        r->tmp = nullptr;
        r->cap = 0;
        GAMS_HIP(h, gams_pool_alloc(h, false, rows_carve(nullptr, cap).bytes, reinterpret_cast<void **>(&r->tmp), &r->tmp_bytes));
        r->cap = cap;
    }
    const uint64_t want_text = std::max<uint64_t>(r->cap * ((uint64_t)r->max_name + 44u), 1u << 20);
    if (r->d_text_bytes < want_text) {
        gams_pool_free(h, false, r->d_text, r->d_text_bytes);
        r->d_text = nullptr;
        r->d_text_bytes = 0;
        GAMS_HIP(h, gams_pool_alloc(h, false, want_text, reinterpret_cast<void **>(&r->d_text), &r->d_text_bytes));
    }
    // the text's speculative copy: sized by the previous pass, kept while it still fits (the graph holds the size)
    uint64_t copy_bytes = r->copy_bytes;
    if (r->last_bytes && (copy_bytes < r->last_bytes || copy_bytes > 2 * r->last_bytes + (1u << 20)))
        copy_bytes = std::min<uint64_t>(r->d_text_bytes, (r->last_bytes + r->last_bytes / 8 + 65536) & ~(uint64_t)65535);
    if (copy_bytes > r->h_text_bytes) {
        gams_pool_free(h, true, r->h_text, r->h_text_bytes);
        r->h_text = nullptr;
        r->h_text_bytes = 0;
        GAMS_HIP(h, gams_pool_alloc(h, true, copy_bytes, reinterpret_cast<void **>(&r->h_text), &r->h_text_bytes));
    }
    r->copy_bytes = copy_bytes;
    r->copied = copy_bytes;
    const WaveRows::RowGraphKey key{p->d_dense, r->tmp, r->d_text, r->h_text, w.d_peaks, w.d_tile_cnt, p->d_tile_off, p->dense_cap,
                          r->cap, copy_bytes, p->tile_cap, (uint32_t)p->tiles.size()};
    WaveRows::RowGraph &g = r->graph[wi];
    bool launched = false;
    if (r->use_graph) {
        if (!g.exec || std::memcmp(&g.key, &key, sizeof key) != 0) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
            hipGraph_t graph = nullptr;
            hipError_t e = hipGraphCreate(&graph, 0);
            if (e == hipSuccess && rows_emit(h, p, wi, st, copy_bytes, graph) != GAMS_OK) e = hipErrorUnknown;
            if (e == hipSuccess) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            if (graph) (void)hipGraphDestroy(graph);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                g.exec = nullptr;
                r->use_graph = false;       // this runtime will not take the graph: plain launches from here on
            } else {
                g.key = key;
            }
        }
        if (g.exec) {
            GAMS_HIP(h, hipGraphLaunch(g.exec, st));
            launched = true;
        }
    }
    if (!launched) {
        const int rc = rows_emit(h, p, wi, st, copy_bytes, nullptr);
        if (rc != GAMS_OK) return rc;
    }
    if (!r->done) GAMS_HIP(h, hipEventCreateWithFlags(&r->done, hipEventDisableTiming));
    GAMS_HIP(h, hipEventRecord(r->done, st));
    r->begun = true;
    return GAMS_OK;
}
}  // namespace
extern "C" {

int gams_wave_rows_setup(gams_gpu_t *h, gams_wave_plan_t *p, const char *const *chr, const int32_t *chr_start,
                         float coverage) {
    if (!h || !p || (p->set->n_ctg && (!chr || !chr_start))) return gams_fail(h, GAMS_EINVAL, "wave_rows_setup: null argument");
    if (!(p->flags & GAMS_WAVE_PEAKS)) return gams_fail(h, GAMS_ESTATE, "wave_rows_setup: plan has no PEAKS output");
    const gams_wave_params_t &q = p->prm;
    // merge_ints (wave.rs:217-252) links windows i < j iff they intersect and size / |intersection| >= coverage
    // (both ratios are that one: all windows have one size).  The device makes rows when every intersecting
    // pair links, i.e. when the smallest possible ratio -- at distance 1 -- already passes.
    const int64_t dmax = ((int64_t)q.size + q.step - 1) / q.step - 1;
    if (dmax >= 1) {
        const float inter = (float)(int32_t)(q.size - q.step);
        if (!((float)q.size / inter >= coverage))
            return gams_fail(h, GAMS_EUNSUPPORTED,
                             "wave_rows_setup: this --coverage links only some of the overlapping windows; merge the "
                             "peaks on the host (gams_wave_peaks)");
    }
    const uint32_t n_ctg = p->set->n_ctg;
    for (uint32_t c = 0; c < n_ctg; ++c) {
        if (!chr[c]) return gams_fail(h, GAMS_EINVAL, "wave_rows_setup: null chromosome name");
        if (chr_start[c] < 0 || (int64_t)chr_start[c] + p->ctgs[c].len >= 0x7FFFFFFFll)
            return gams_fail(h, GAMS_EINVAL, "wave_rows_setup: chromosome coordinates must be in [0, 2^31)");
    }
    GAMS_HIP(h, hipSetDevice(h->device));
    if (p->rows) {                       // set up again (other names / coverage): drop the old tables
        GAMS_HIP(h, hipStreamSynchronize(h->readback));
        WaveRows *r = p->rows;
        gams_pool_free(h, false, r->arena, r->arena_bytes);
        gams_pool_free(h, true, r->h_words, r->h_words_bytes);
        r->arena = nullptr;
        r->h_words = nullptr;
        for (auto &g : r->graph) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            g.exec = nullptr;
        }
    } else {
        p->rows = new WaveRows();
    }
    WaveRows *r = p->rows;
    r->dmax = (uint32_t)dmax;
    r->begun = false;
    // names (one copy per distinct pointer is not worth the bookkeeping: a few bytes per ctg), gc text table
    std::vector<RowCtg> rc(std::max<uint32_t>(n_ctg, 1));
    std::string blob;
    r->max_name = 0;
    for (uint32_t c = 0; c < n_ctg; ++c) {
        const size_t len = std::strlen(chr[c]);
        size_t at = blob.find(chr[c]);   // ctgs of one chromosome share its name
        if (at == std::string::npos || (len == 0)) {
            at = blob.size();
            blob += chr[c];
        }
        rc[c] = RowCtg{(uint32_t)at, (uint32_t)len, chr_start[c], 0u};
        r->max_name = std::max<uint32_t>(r->max_name, (uint32_t)len);
    }
    std::vector<uint8_t> gct(((size_t)q.size + 1) * kGcStride, 0);
    for (int32_t k = 0; k <= q.size; ++k) {
        const std::string t = rows_fmt_f32((float)k / (float)q.size);     // gc_content as wave.rs prints it
        if (t.size() > kGcStride - 1) return gams_fail(h, GAMS_EUNSUPPORTED, "wave_rows_setup: gc_content text too long");
        gct[(size_t)k * kGcStride] = (uint8_t)t.size();
        std::memcpy(&gct[(size_t)k * kGcStride + 1], t.data(), t.size());
    }
    const size_t b_ctgs = wave_align256(rc.size() * sizeof(RowCtg)), b_names = wave_align256(std::max<size_t>(blob.size(), 1)),
                 b_gc = wave_align256(gct.size()), b_words = wave_align256(((size_t)n_ctg + 1 + 4) * 8);
    GAMS_HIP(h, gams_pool_alloc(h, false, b_ctgs + b_names + b_gc + b_words, reinterpret_cast<void **>(&r->arena),
                                &r->arena_bytes));
    r->d_ctgs = reinterpret_cast<RowCtg *>(r->arena);
    r->d_names = reinterpret_cast<char *>(r->arena + b_ctgs);
    r->d_gctab = r->arena + b_ctgs + b_names;
    r->d_words = reinterpret_cast<unsigned long long *>(r->arena + b_ctgs + b_names + b_gc);
    GAMS_HIP(h, hipMemcpy(r->d_ctgs, rc.data(), rc.size() * sizeof(RowCtg), hipMemcpyHostToDevice));
    if (!blob.empty()) GAMS_HIP(h, hipMemcpy(r->d_names, blob.data(), blob.size(), hipMemcpyHostToDevice));
    GAMS_HIP(h, hipMemcpy(r->d_gctab, gct.data(), gct.size(), hipMemcpyHostToDevice));
    GAMS_HIP(h, gams_pool_alloc(h, true, ((size_t)n_ctg + 1 + 4) * 8, reinterpret_cast<void **>(&r->h_words), &r->h_words_bytes));
    return GAMS_OK;
}

int gams_wave_rows_begin(gams_gpu_t *h, gams_wave_plan_t *p) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_rows_begin: null argument");
    if (!p->rows) return gams_fail(h, GAMS_ESTATE, "wave_rows_begin: call gams_wave_rows_setup first");
    if (!p->ran) return gams_fail(h, GAMS_ESTATE, "wave_rows_begin: no run to read");
    GAMS_HIP(h, hipSetDevice(h->device));
    if (p->tiles.empty()) {
        p->rows->begun = true;
        return GAMS_OK;
    }
    return rows_queue(h, p);
}

int gams_wave_rows_end(gams_gpu_t *h, gams_wave_plan_t *p, const char **text, uint64_t *text_bytes,
                       const uint64_t **ctg_off) {
    if (!h || !p || !text || !text_bytes) return gams_fail(h, GAMS_EINVAL, "wave_rows_end: null argument");
    if (!p->rows || !p->rows->begun) return gams_fail(h, GAMS_ESTATE, "wave_rows_end: no gams_wave_rows_begin to finish");
    GAMS_HIP(h, hipSetDevice(h->device));
    WaveRows *r = p->rows;
    const uint32_t n_ctg = p->set->n_ctg;
    r->begun = false;
    if (p->tiles.empty()) {
        for (uint32_t c = 0; c <= n_ctg; ++c) r->h_words[4 + c] = 0;
        *text = nullptr;
        *text_bytes = 0;
        if (ctg_off) *ctg_off = reinterpret_cast<const uint64_t *>(r->h_words + 4);
        return GAMS_OK;
    }
    const size_t nt = p->tiles.size();
    for (int attempt = 0; attempt < 4; ++attempt) {
        GAMS_HIP(h, hipEventSynchronize(r->done));
        const uint64_t n_rec = r->h_words[0], bytes = r->h_words[1], total = r->h_words[2], worst = r->h_words[3];
        if (worst > p->tw || total > p->total_windows || total > (uint64_t)nt * worst)
            return gams_fail(h, GAMS_EHIP, "wave_rows: inconsistent peak counts from the device");
        int rc = GAMS_OK;
        if (worst > p->tile_cap) {
            // a tile signalled more windows than its slot holds: larger slots, the passes still held run again
            // (see gams_wave_peaks), then the rows once more
            p->tile_cap_req = (uint32_t)std::min<uint64_t>(p->tw, (worst + 15u) & ~(uint64_t)15u);
            const uint32_t age = p->sel_age;
            const uint32_t again = (uint32_t)std::min<uint64_t>(p->depth, p->run_idx);
            rc = wave_sync_ways(h, p);
            if (rc == GAMS_OK) rc = wave_upload_geometry(h, p);
            if (rc == GAMS_OK) {
                p->run_idx -= again;
                for (uint32_t k = 0; k < again && rc == GAMS_OK; ++k) rc = gams_wave_run(h, p);
                p->sel_age = age;
            }
        } else if (total > p->dense_cap || total > r->cap) {
            if (total > 0x7FFFFF00ull) return gams_fail(h, GAMS_EUNSUPPORTED, "wave_rows: more than 2^31 peaks in one pass");
            gams_pool_free(h, false, p->d_dense, p->d_dense_bytes);
            p->d_dense = nullptr;
            p->dense_cap = 0;
            GAMS_HIP(h, gams_pool_alloc(h, false, (total + total / 4 + 1024) * sizeof(gams_peak_t),
                                        reinterpret_cast<void **>(&p->d_dense), &p->d_dense_bytes));
            p->dense_cap = p->d_dense_bytes / sizeof(gams_peak_t);
        } else if (bytes > r->d_text_bytes) {
            return gams_fail(h, GAMS_EHIP, "wave_rows: text beyond its bound");
        } else {
            (void)n_rec;
            if (bytes > r->copied) {
                // the speculative copy fell short (the first pass, or more text than last time): the rest now
                if (r->h_text_bytes < bytes) {
                    char *nt_ = nullptr;
                    size_t nb_ = 0;
                    GAMS_HIP(h, gams_pool_alloc(h, true, bytes + bytes / 4 + 4096, reinterpret_cast<void **>(&nt_), &nb_));
                    if (r->copied) std::memcpy(nt_, r->h_text, r->copied);
                    gams_pool_free(h, true, r->h_text, r->h_text_bytes);
                    r->h_text = nt_;
                    r->h_text_bytes = nb_;
                }
                GAMS_HIP(h, hipMemcpyAsync(r->h_text + r->copied, r->d_text + r->copied, bytes - r->copied,
                                           hipMemcpyDeviceToHost, h->readback));
                GAMS_HIP(h, hipStreamSynchronize(h->readback));
            }
            r->last_bytes = bytes;
            // ctgs without a row begin where the next ctg's rows begin
            unsigned long long *off = r->h_words + 4;
            off[n_ctg] = bytes;
            for (uint32_t c = n_ctg; c-- > 0;)
                if (off[c] == ~0ull) off[c] = off[c + 1];
            *text = bytes ? r->h_text : nullptr;
            *text_bytes = bytes;
            if (ctg_off) *ctg_off = reinterpret_cast<const uint64_t *>(off);
            return GAMS_OK;
        }
        if (rc != GAMS_OK) return rc;
        rc = rows_queue(h, p);
        if (rc != GAMS_OK) return rc;
    }
    return gams_fail(h, GAMS_EHIP, "wave_rows: buffers kept overflowing");
}

int gams_wave_signal_text(gams_gpu_t *h, gams_wave_plan_t *p, const char *const *chr, const int32_t *chr_start,
                          const char **text, uint64_t *text_bytes, const uint64_t **ctg_off) {
    if (!h || !p || !text || !text_bytes || !ctg_off || (p->set->n_ctg && (!chr || !chr_start)))
        return gams_fail(h, GAMS_EINVAL, "wave_signal_text: null argument");
    if (!(p->flags & GAMS_WAVE_DENSE)) return gams_fail(h, GAMS_ESTATE, "wave_signal_text: plan has no DENSE output");
    if (!p->ran) return gams_fail(h, GAMS_ESTATE, "wave_signal_text: no run to read");
    const gams_wave_params_t &q = p->prm;
    const uint32_t n_ctg = p->set->n_ctg;
    for (uint32_t c = 0; c < n_ctg; ++c) {
        if (!chr[c]) return gams_fail(h, GAMS_EINVAL, "wave_signal_text: null chromosome name");
        if (chr_start[c] < 0 || (int64_t)chr_start[c] + p->ctgs[c].len >= 0x7FFFFFFFll)
            return gams_fail(h, GAMS_EINVAL, "wave_signal_text: chromosome coordinates must be in [0, 2^31)");
    }
    GAMS_HIP(h, hipSetDevice(h->device));
    {
        int wrc = wave_jac_settle(h, p, wave_read_way_index(p));
        if (wrc == GAMS_OK) wrc = wave_wait_last_run(h, p);
        if (wrc != GAMS_OK) return wrc;
    }
    // names, gc text table (as gams_wave_rows_setup), tiles of kSigRows windows
    std::vector<RowCtg> rc(std::max<uint32_t>(n_ctg, 1));
    std::string blob;
    for (uint32_t c = 0; c < n_ctg; ++c) {
        const size_t len = std::strlen(chr[c]);
        size_t at = blob.find(chr[c]);
        if (at == std::string::npos || len == 0) {
            at = blob.size();
            blob += chr[c];
        }
        rc[c] = RowCtg{(uint32_t)at, (uint32_t)len, chr_start[c], 0u};
    }
    std::vector<uint8_t> gct(((size_t)q.size + 1) * kGcStride, 0);
    for (int32_t k = 0; k <= q.size; ++k) {
        const std::string t = rows_fmt_f32((float)k / (float)q.size);
        if (t.size() > kGcStride - 1) return gams_fail(h, GAMS_EUNSUPPORTED, "wave_signal_text: gc_content text too long");
        gct[(size_t)k * kGcStride] = (uint8_t)t.size();
        std::memcpy(&gct[(size_t)k * kGcStride + 1], t.data(), t.size());
    }
    if (!p->sigtext) p->sigtext = new WaveSig();
    WaveSig *g = p->sigtext;
    if (!g->arena || g->names_cap < blob.size()) {
        gams_pool_free(h, false, g->arena, g->arena_bytes);
        g->arena = nullptr;
        std::vector<SigTile> st;
        for (uint32_t c = 0; c < n_ctg; ++c)
            for (uint32_t w0 = 0; w0 < p->ctgs[c].n_win; w0 += kSigRows)
                st.push_back(SigTile{c, w0, p->ctgs[c].n_win, 0u, p->ctgs[c].win_base});
        g->n_tiles = (uint32_t)st.size();
        g->names_cap = std::max<size_t>(blob.size(), 64) * 2;
        const size_t nt = std::max<size_t>(st.size(), 1);
        const size_t b_tiles = wave_align256(nt * sizeof(SigTile)), b_len = wave_align256(nt * 4), b_off = wave_align256((nt + 2) * 8),
                     b_ctgs = wave_align256(rc.size() * sizeof(RowCtg)), b_names = wave_align256(g->names_cap),
                     b_gc = wave_align256(gct.size()), b_words = wave_align256(std::max<size_t>(n_ctg, 1) * 8);
        GAMS_HIP(h, gams_pool_alloc(h, false, b_tiles + b_len + b_off + b_ctgs + b_names + b_gc + b_words,
                                    reinterpret_cast<void **>(&g->arena), &g->arena_bytes));
        uint8_t *o = g->arena;
        g->d_tiles = reinterpret_cast<SigTile *>(o), o += b_tiles;
        g->d_blk_len = reinterpret_cast<uint32_t *>(o), o += b_len;
        g->d_blk_off = reinterpret_cast<unsigned long long *>(o), o += b_off;
        g->d_ctgs = reinterpret_cast<RowCtg *>(o), o += b_ctgs;
        g->d_names = reinterpret_cast<char *>(o), o += b_names;
        g->d_gctab = o, o += b_gc;
        g->d_words = reinterpret_cast<unsigned long long *>(o);
        if (!st.empty()) GAMS_HIP(h, hipMemcpy(g->d_tiles, st.data(), st.size() * sizeof(SigTile), hipMemcpyHostToDevice));
        gams_pool_free(h, true, g->h_words, g->h_words_bytes);
        g->h_words = nullptr;
        GAMS_HIP(h, gams_pool_alloc(h, true, ((size_t)2 * n_ctg + 2) * 8, reinterpret_cast<void **>(&g->h_words), &g->h_words_bytes));
    }
    *text = nullptr;
    *text_bytes = 0;
    unsigned long long *off = g->h_words + n_ctg;          // ctg_off[n_ctg + 1]
    if (g->n_tiles == 0) {
        for (uint32_t c = 0; c <= n_ctg; ++c) off[c] = 0;
        *ctg_off = reinterpret_cast<const uint64_t *>(off);
        return GAMS_OK;
    }
    hipStream_t st = h->readback;
    GAMS_HIP(h, hipMemcpyAsync(g->d_ctgs, rc.data(), rc.size() * sizeof(RowCtg), hipMemcpyHostToDevice, st));
    if (!blob.empty()) GAMS_HIP(h, hipMemcpyAsync(g->d_names, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    GAMS_HIP(h, hipMemcpyAsync(g->d_gctab, gct.data(), gct.size(), hipMemcpyHostToDevice, st));
    GAMS_HIP(h, hipMemsetAsync(g->d_words, 0xFF, std::max<size_t>(n_ctg, 1) * 8, st));
    gams_wave_plan::Way &w = wave_read_way(p);
    SigArgs a{};
    a.tiles = g->d_tiles;
    a.n_tiles = g->n_tiles;
    a.cnt = w.d_dense_cnt;
    a.sig = w.d_dense_sig;
    a.ctgs = g->d_ctgs;
    a.names = g->d_names;
    a.gctab = g->d_gctab;
    a.size = (uint32_t)q.size;
    a.step = (uint32_t)q.step;
    a.blk_len = g->d_blk_len;
    a.blk_off = g->d_blk_off;
    a.words = g->d_words;
    hipLaunchKernelGGL(sig_len_kernel, dim3(g->n_tiles), dim3(256), 0, st, a);
    unsigned long long *const d_totals = g->d_blk_off + g->n_tiles;
    if (g->n_tiles <= kOffOneGroup) {
        hipLaunchKernelGGL(wave_offsets_kernel, dim3(1), dim3(1024), 0, st, g->d_blk_len, g->n_tiles, g->d_blk_off, d_totals);
    } else {
        const unsigned spans = (g->n_tiles + kOffSpan - 1u) / kOffSpan;
        hipLaunchKernelGGL(wave_offsets_sum_kernel, dim3(spans), dim3(1024), 0, st, g->d_blk_len, g->n_tiles, g->d_blk_off);
        hipLaunchKernelGGL(wave_offsets_base_kernel, dim3(1), dim3(1024), 0, st, g->n_tiles, g->d_blk_off, d_totals);
        hipLaunchKernelGGL(wave_offsets_scan_kernel, dim3(spans), dim3(1024), 0, st, g->d_blk_len, g->n_tiles, g->d_blk_off);
    }
    GAMS_HIP(h, hipGetLastError());
    GAMS_HIP(h, hipMemcpyAsync(h->pin_scratch, d_totals, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    GAMS_HIP(h, hipStreamSynchronize(st));       // (the staged tables above live on this call's stack until here)
    const uint64_t total = h->pin_scratch[0];
    if (total > g->d_text_bytes) {
        gams_pool_free(h, false, g->d_text, g->d_text_bytes);
        g->d_text = nullptr;
        g->d_text_bytes = 0;
        GAMS_HIP(h, gams_pool_alloc(h, false, total + total / 16 + 4096, reinterpret_cast<void **>(&g->d_text), &g->d_text_bytes));
    }
    if (total > g->h_text_bytes) {
        gams_pool_free(h, true, g->h_text, g->h_text_bytes);
        g->h_text = nullptr;
        g->h_text_bytes = 0;
        GAMS_HIP(h, gams_pool_alloc(h, true, total + total / 16 + 4096, reinterpret_cast<void **>(&g->h_text), &g->h_text_bytes));
    }
    a.text = g->d_text;
    a.text_cap = total;
    hipLaunchKernelGGL(sig_write_kernel, dim3(g->n_tiles), dim3(256), 0, st, a);
    GAMS_HIP(h, hipGetLastError());
    GAMS_HIP(h, hipMemcpyAsync(g->h_words, g->d_words, std::max<size_t>(n_ctg, 1) * 8, hipMemcpyDeviceToHost, st));
    if (total) GAMS_HIP(h, hipMemcpyAsync(g->h_text, g->d_text, total, hipMemcpyDeviceToHost, st));
    GAMS_HIP(h, hipStreamSynchronize(st));
    // where each ctg's rows begin: ctgs without a window begin where the next one does
    off[n_ctg] = total;
    for (uint32_t c = n_ctg; c-- > 0;) off[c] = p->ctgs[c].n_win ? g->h_words[c] : off[c + 1];
    *text = g->h_text;
    *text_bytes = total;
    *ctg_off = reinterpret_cast<const uint64_t *>(off);
    return GAMS_OK;
}

int gams_wave_dense(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t i, uint32_t *gc_count, int8_t *signal) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_dense: null argument");
    if (!(p->flags & GAMS_WAVE_DENSE)) return gams_fail(h, GAMS_ESTATE, "wave_dense: plan has no DENSE output");
    if (!p->ran) return gams_fail(h, GAMS_ESTATE, "wave_dense: no run to read");
    if (i >= p->ctgs.size()) return gams_fail(h, GAMS_EINVAL, "wave_dense: ctg index out of range");
    GAMS_HIP(h, hipSetDevice(h->device));
    {
        int wrc = wave_jac_settle(h, p, wave_read_way_index(p));
        if (wrc == GAMS_OK) wrc = wave_wait_last_run(h, p);
        if (wrc != GAMS_OK) return wrc;
    }
    const WaveCtgDev &c = p->ctgs[i];
    if (c.n_win == 0) return GAMS_OK;
    gams_wave_plan::Way &w = wave_read_way(p);
    if (gc_count)
        GAMS_HIP(h, hipMemcpy(gc_count, w.d_dense_cnt + c.win_base, (size_t)c.n_win * sizeof(uint32_t),
                              hipMemcpyDeviceToHost));
    if (signal)
        GAMS_HIP(h, hipMemcpy(signal, w.d_dense_sig + c.win_base, (size_t)c.n_win, hipMemcpyDeviceToHost));
    return GAMS_OK;
}

int gams_wave_plan_set_stamps(gams_gpu_t *h, gams_wave_plan_t *p, int enable) {
    if (!h || !p) return gams_fail(h, GAMS_EINVAL, "wave_plan_set_stamps: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    {
        int src = wave_sync_ways(h, p);
        if (src != GAMS_OK) return src;
    }
    (void)hipFree(p->d_stamps);
    p->d_stamps = nullptr;
    if (enable) {
        const size_t n = std::max<size_t>(p->tiles.size(), 1) * 16;
        GAMS_HIP(h, hipMalloc(&p->d_stamps, n * sizeof(unsigned long long)));
        GAMS_HIP(h, hipMemsetAsync(p->d_stamps, 0, n * sizeof(unsigned long long), h->compute));
        GAMS_HIP(h, hipStreamSynchronize(h->compute));
    }
    return GAMS_OK;
}

int gams_wave_stamps(gams_gpu_t *h, gams_wave_plan_t *p, double *mean_cycles, uint64_t *span_cycles) {
    if (!h || !p || !mean_cycles || !span_cycles) return gams_fail(h, GAMS_EINVAL, "wave_stamps: null argument");
    if (!p->d_stamps) return gams_fail(h, GAMS_ESTATE, "wave_stamps: stamps are off");
    GAMS_HIP(h, hipSetDevice(h->device));
    {
        int src = wave_sync_ways(h, p);
        if (src != GAMS_OK) return src;
    }
    const size_t nt = p->tiles.size();
    std::vector<unsigned long long> st(std::max<size_t>(nt, 1) * 16);
    GAMS_HIP(h, hipMemcpy(st.data(), p->d_stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int k = 0; k < 8; ++k) mean_cycles[k] = 0.0;
    size_t used = 0;
    unsigned long long r_lo = ~0ull, r_hi = 0, r_sum = 0;
    double cyc_sum = 0.0;
    for (size_t t = 0; t < nt; ++t) {
        const unsigned long long *w = &st[t * 16];
        bool ok = w[8] != 0 && w[9] >= w[8];
        for (int k = 0; k < 6; ++k) ok &= w[k] != 0 && w[k + 1] >= w[k] && w[k + 1] - w[k] < (1ull << 28);
        if (!ok) continue;
        ++used;
        for (int k = 0; k < 6; ++k) mean_cycles[k] += (double)(w[k + 1] - w[k]);
        mean_cycles[6] += (double)(w[6] - w[0]);
        r_lo = std::min(r_lo, w[8]);
        r_hi = std::max(r_hi, w[9]);
        r_sum += w[9] - w[8];
        cyc_sum += (double)(w[6] - w[0]);
    }
    if (used)
        for (int k = 0; k < 7; ++k) mean_cycles[k] /= (double)used;
    // shader clock in GHz = cycles / (ticks * 10 ns)
    mean_cycles[7] = r_sum ? cyc_sum / ((double)r_sum * 10.0) : 0.0;
    // high half: launch span in 10-ns ticks; low half: sum of workgroup lifetimes in ticks / 16
    *span_cycles = used ? (((r_hi - r_lo) & 0xffffffffull) << 32) | ((r_sum >> 4) & 0xffffffffull) : 0;
    return GAMS_OK;
}

int gams_wave_stamps_raw(gams_gpu_t *h, gams_wave_plan_t *p, uint64_t *out, uint64_t n_words) {
    if (!h || !p || !out) return gams_fail(h, GAMS_EINVAL, "wave_stamps_raw: null argument");
    if (!p->d_stamps) return gams_fail(h, GAMS_ESTATE, "wave_stamps_raw: stamps are off");
    GAMS_HIP(h, hipSetDevice(h->device));
    {
        int src = wave_sync_ways(h, p);
        if (src != GAMS_OK) return src;
    }
    const uint64_t n = std::min<uint64_t>(n_words, (uint64_t)p->tiles.size() * 16);
    GAMS_HIP(h, hipMemcpy(out, p->d_stamps, n * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return GAMS_OK;
}

int gams_wave_exact_count(gams_gpu_t *h, gams_wave_plan_t *p, uint64_t *n_exact) {
    if (!h || !p || !n_exact) return gams_fail(h, GAMS_EINVAL, "wave_exact_count: null argument");
    GAMS_HIP(h, hipSetDevice(h->device));
    gams_wave_plan::Way &w = wave_read_way(p);
    GAMS_HIP(h, hipStreamSynchronize(wave_stream(h, p, wave_read_way_index(p))));
    std::vector<unsigned long long> cnt(kSlotWords);
    GAMS_HIP(h, hipMemcpy(cnt.data(), w.d_counters + kSlotWords * w.last_ring,
                          kSlotWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    uint64_t tot = 0;
    for (uint32_t sh = 0; sh < kShards; ++sh) tot += cnt[sh * kShardWords + 1];
    *n_exact = tot;
    return GAMS_OK;
}

int gams_gpu_wave(gams_gpu_t *h, const uint8_t *seq, uint32_t len, const gams_wave_params_t *params,
                  uint32_t *gc_count, int8_t *signal, uint32_t *n_windows) {
    if (!h || !seq || !params) return gams_fail(h, GAMS_EINVAL, "gpu_wave: null argument");
    gams_seqset_t *s = nullptr;
    int rc = gams_seqset_create(h, 1, &len, &s);
    if (rc != GAMS_OK) return rc;
    rc = gams_seqset_upload(h, s, 0, seq);
    gams_wave_plan_t *p = nullptr;
    if (rc == GAMS_OK) rc = gams_wave_plan_create(h, s, params, GAMS_WAVE_DENSE, &p);
    if (rc == GAMS_OK) rc = gams_wave_run(h, p);
    if (rc == GAMS_OK) rc = gams_wave_dense(h, p, 0, gc_count, signal);
    if (rc == GAMS_OK && n_windows) *n_windows = gams_wave_ctg_windows(p, 0);
    if (p) gams_wave_plan_destroy(h, p);
    gams_seqset_destroy(h, s);
    return rc;
}

}  // extern "C"
