/*
 * gams_gpu_diag.h -- measurement and tuning entries of libgams_gpu.so.
 *
 * Nothing here is needed to run the path (gams_gpu.h is the binding surface); these are what
 * bench.py, tools/ and the parity tests use to time kernels, name them for rocprofv3, look inside
 * a launch and move the knobs whose defaults were chosen from such measurements.  Results never
 * depend on any of them.
 */
#ifndef GAMS_GPU_DIAG_H
#define GAMS_GPU_DIAG_H

#include "gams_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* HIP-event stopwatch on the handle's compute stream (the stream the kernels of this library
 * are launched on; plans of depth > 1 also use auxiliary streams, which stop() queues the
 * compute stream behind before it records the closing event).  stop() synchronises and returns ms. */
int gams_gpu_timer_start(gams_gpu_t *h);
int gams_gpu_timer_stop(gams_gpu_t *h, float *ms);
/* Device time (HIP events around the kernel, excluding the host<->device copies) of the last
 * gams_gpu_sw / gams_gpu_count / gams_gpu_locate / gams_gpu_cover call on this handle. */
int gams_gpu_last_kernel_ms(gams_gpu_t *h, float *ms);

/* Tapered launches.  A launch of the headline parameters (size 100, step 10, lag 100) over at least a
 * round and a half of workgroups ends in smaller tiles (the ctgs holding the last 17 % / 8 % of the
 * windows are cut into tiles of 2/3 and 1/3 the size), so that its last workgroups are short-lived and
 * the chip drains in ~3 instead of ~10 us; the small tiles cost 3-4 % more work.  mode -1 (default):
 * on for plans of depth 1; 0: off -- what a host wants that keeps several passes in flight (plans on
 * lanes, or depth > 1): their tails overlap anyway; 1: on.  Results are identical either way. */
int gams_wave_plan_set_taper(gams_gpu_t *h, gams_wave_plan_t *plan, int mode);
/* Size of the two tails of a tapered launch, in % of a round of workgroup slots (8 per CU): the ctgs
 * holding the last pct4 % of a round are cut into the smallest tiles, the pct8 % before them into the
 * middle ones; at most 8 % / 17 % of the batch.  0..100, default 25 / 50 (profiles/r02_taper_sweep.txt). */
int gams_wave_plan_set_taper_shape(gams_gpu_t *h, gams_wave_plan_t *plan, int pct4, int pct8);
/* Host threads gams_wave_run_n queues a long batch of passes from (1..4, default one per way). */
int gams_wave_plan_set_queue_threads(gams_gpu_t *h, gams_wave_plan_t *plan, uint32_t n);
/* Name of the kernel that does the plan's work, spelled as rocprofv3 --kernel-trace prints the
 * instantiation (without the namespace), e.g. "wave_fast_taper_kernel<100, 10, 100, true>": lets a
 * benchmark line name the row of the profile its launch duration must agree with.  Follows the plan's
 * current settings (set_tile / set_taper / the seqset's size).  NUL-terminated, truncated to n. */
int gams_wave_plan_kernel_name(gams_gpu_t *h, gams_wave_plan_t *plan, char *buf, size_t n);
/* influence != 1: how the selected pass reached the reference's answer -- the sweeps of guess-and-iterate it took
 * (waits for them like every reader of the pass), and whether it was handed to the one-wavefront-per-ctg recurrence
 * instead (serial = 1).  influence == 1: 0 sweeps, serial = 0. */
int gams_wave_plan_settled(gams_gpu_t *h, gams_wave_plan_t *plan, uint32_t *sweeps, int *serial);

/* Tuning/diagnostics: windows per tile (0 = library default), and how many
 * windows of the last run took the exact-order f32 re-evaluation. */
int gams_wave_plan_set_tile(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t tile_windows);
/* Threads per workgroup of the step-1 tile kernels (size 100, step 1, W = 28): 64, 128 or 256, 0 = the library's
 * choice.  A tile is then 64 / 128 / 256 threads x 28 windows; other plans ignore the request.  Rebuilds the tiling
 * like gams_wave_plan_set_tile. */
int gams_wave_plan_set_threads(gams_gpu_t *h, gams_wave_plan_t *p, uint32_t threads);
int gams_wave_exact_count(gams_gpu_t *h, gams_wave_plan_t *p, uint64_t *n_exact);
/* Diagnostics of the integer decision's guard band (stat.rs:36-38 is an f32 comparison; windows
 * whose integer margin is inside the band are re-evaluated in the reference's exact f32 order).
 * safety (default 1.5, >= 1) multiplies the derived error bound; all_exact != 0 sends every window
 * down the exact path.  Results are identical for every setting -- only the share of windows that
 * take the exact path changes -- which is what the tests check.  Applies to the next run. */
int gams_wave_plan_set_guard(gams_gpu_t *h, gams_wave_plan_t *p, float safety, int all_exact);
/* Diagnostics: with stamps on, thread 0 of every workgroup records the shader clock at the
 * kernel's phase boundaries.  mean_cycles[0..5] = mean duration of load+classify, chunk
 * prefix, window counts, z-score, exact re-evaluation, outputs; [6] = whole workgroup;
 * span_cycles = first workgroup start to last workgroup end. */
int gams_wave_plan_set_stamps(gams_gpu_t *h, gams_wave_plan_t *p, int enable);
int gams_wave_stamps(gams_gpu_t *h, gams_wave_plan_t *p, double *mean_cycles /* [8] */,
                     uint64_t *span_cycles);
/* the raw stamp words: 16 per workgroup (tile), see wave_stamp() in gams_amd/csrc/wave.hip */
int gams_wave_stamps_raw(gams_gpu_t *h, gams_wave_plan_t *p, uint64_t *out, uint64_t n_words);

#ifdef __cplusplus
}
#endif
#endif
