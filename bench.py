#!/usr/bin/env python3
"""bench.py -- GC windows/s of the `wave` hot path on MI355X (driver contract in the task brief).

A step = one pass of the hot path over one batch: every ctg of a synthetic genome (resident in
HBM) -> gc counts -> smoothed z-score signals -> compacted peak records, one kernel launch.
Default workload (N = 1): BASELINE.json configs[1], an S288c-sized genome (12,157,105 bp, 17
chromosomes, --piece 500000), size 100 / step 10 / lag 100 / threshold 3 / influence 1.
With N > 1 every rank owns its own genome of that shape (ctgs shard with no exchange: weak
scaling), launched one process per GPU by torch.distributed.run.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def profiled_traffic(workload):
    """HBM bytes per launch of the wave kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json, made by tools/prof.sh): FETCH_SIZE and WRITE_SIZE are in KiB and were
    collected in separate passes; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced
    stream, so it is doubled (MI355X_MICROARCH.md, HBM section).  None if no profile matches."""
    path = os.path.join(ROOT, "profiles", f"r01_{workload}_wave_pmc.json")
    try:
        with open(path) as fh:
            p = json.load(fh)
        return (2.0 * p["FETCH_SIZE"]["mean"] + p["WRITE_SIZE"]["mean"]) * 1024.0
    except (OSError, KeyError, ValueError):
        return None

WORKLOADS = {
    # name: (chromosome lengths, piece, size, step, lag)
    "S288c": ("S288C_LENGTHS", 500000, 100, 10, 100),
    "Atha": ("ATHA_LENGTHS", 500000, 100, 10, 100),
    "synth384": ("SYNTH384_LENGTHS", 1000000, 100, 10, 100),
    "GRCh38-step10": ("GRCH38_LENGTHS", 1000000, 100, 10, 100),
    "GRCh38-step1": ("GRCH38_LENGTHS", 1000000, 100, 1, 100),
}


def build_workload(name, rank, scale=1.0):
    from gams_amd import synth

    lengths_name, piece, size, step, lag = WORKLOADS[name]
    lengths = getattr(synth, lengths_name)
    if scale != 1.0:
        lengths = [max(20000, int(x * scale)) for x in lengths]
    # each rank gets its own genome (different chromosome seeds)
    ctgs = synth.genome_ctgs(lengths, piece, first_chr_index=1 + 1000 * rank)
    return ctgs, dict(size=size, step=step, lag=lag, threshold=3.0, influence=1.0), sum(lengths)


def cpu_baseline(ctgs, prm, budget_windows=6_000_000, min_seconds=10.0):
    """The oracle (CPU restatement of the reference algorithm) timed on this box, one thread,
    on a bounded sample of the same workload: whole ctgs until the window budget is reached, and
    that sample again until about `min_seconds` of CPU work have been timed."""
    from oracle import oracle as ora

    ora.lib()
    done, t0, used, passes = 0, time.perf_counter(), 0, 0
    results = []
    while True:
        n_here = 0
        for i, c in enumerate(ctgs):
            cnt, _, sig = ora.wave_windows(c["seq"], prm["size"], prm["step"], prm["lag"], prm["threshold"],
                                           prm["influence"])
            if passes == 0:
                results.append((cnt, sig))
                used += 1
            done += cnt.size
            n_here += cnt.size
            if n_here >= budget_windows:
                break
        passes += 1
        if time.perf_counter() - t0 >= min_seconds:
            break
    dt = time.perf_counter() - t0
    return done / dt, used, done, results, passes, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=5000,
                    help="untimed passes before the timed region (the clocks take a few ms of load to settle)")
    ap.add_argument("--workload", default="S288c", choices=sorted(WORKLOADS))
    ap.add_argument("--scale", type=float, default=1.0, help="shrink/grow chromosome lengths (testing)")
    ap.add_argument("--tile", type=int, default=0, help="windows per tile (0 = library default)")
    ap.add_argument("--depth", type=int, default=4,
                    help="passes in flight (gams_wave_plan_set_depth): consecutive steps rotate over this many HIP "
                         "streams and output sets; 1 = one pass at a time")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the beyond-L3 extra measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    import torch

    dist = None
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)
    backend = os.environ.get("GAMS_BENCH_BACKEND", "nccl")   # "gloo": rehearsal with several ranks on one GPU
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)

    from gams_amd import _lib, engine

    eng = engine.Engine(device)
    arch, cus, hbm = eng.device_info()

    ctgs, prm, genome_bp = build_workload(args.workload, rank, args.scale)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, tile_windows=args.tile, **prm)
    n_windows = plan.total_windows

    def barrier():
        eng.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # The timed region: K full passes, `depth` of them in flight (each on its own stream, into its
    # own output buffers); every pass reads the whole batch and leaves its compacted peaks in HBM.
    # gams_wave_run_n(K) is the host's step loop in C (K x gams_wave_run; beyond two passes in flight
    # it queues from two host threads, because one thread queues a launch every 3.3 us and the
    # device finishes a 12-Mb pass every 2.9 us).
    plan.set_depth(args.depth)
    plan.run_n(args.warmup)
    barrier()
    t0 = time.perf_counter()
    plan.run_n(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        w = torch.tensor([float(n_windows)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        total_windows = float(w.item())
    else:
        total_windows = float(n_windows)
    # every pass still held (the last `depth`) must carry the same peaks
    held = []
    for age in range(min(args.depth, args.steps)):
        plan.select(age)
        held.append(plan.peaks())
    held_equal = all(np.array_equal(held[0], x) for x in held[1:])

    # Roofline leg: the dominant kernel's own launch duration, one pass at a time, HIP events on
    # the stream it runs on (this is what rocprofv3 --kernel-trace reports per launch).
    plan.set_depth(1)
    serial_steps = max(1, min(args.steps, 200))
    for _ in range(min(args.warmup, 20) + 1):
        plan.run()
    eng.sync()
    eng.timer_start()
    for _ in range(serial_steps):
        plan.run()
    kernel_ms = eng.timer_stop()      # HIP events on the library's compute stream
    eng.sync()

    peaks = plan.peaks()
    n_exact = plan.exact_count()
    # beside `value`: the same pass with the compacted peaks packed and copied to the host every step
    rb_steps = max(1, min(args.steps, 50))
    eng.sync()
    t0r = time.perf_counter()
    for _ in range(rb_steps):
        plan.run()
        plan.peaks()
    readback_wps = n_windows * rb_steps / (time.perf_counter() - t0r)

    out = None
    if rank == 0:
        step_bytes = prm["step"]                       # `step` bytes read per window
        launch_ms = kernel_ms / serial_steps           # one wave_fast_kernel launch per pass
        achieved = n_windows * step_bytes / (launch_ms * 1e-3) / 1e9
        out = {
            "metric": "GC windows/s (size 100, step 10)",
            "value": total_windows * args.steps / dt,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-shaped synthetic genome per GPU ({genome_bp} bp, {len(ctgs)} ctgs, "
                            f"piece {WORKLOADS[args.workload][1]}): wave size {prm['size']} step {prm['step']} "
                            f"lag {prm['lag']} threshold 3 influence 1, peaks compacted on device",
                "windows_per_gpu_per_step": int(n_windows),
                "peaks_per_step": int(peaks.size),
                "exact_path_windows": int(n_exact),
                "passes_in_flight": args.depth,
                "held_passes_identical": bool(held_equal),
                "windows_per_s_one_pass_at_a_time": n_windows / (launch_ms * 1e-3),
                "windows_per_s_with_host_readback": readback_wps,
                "device": arch,
                "parallelism": f"ctg-sharded x{world}, no collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "wave_fast_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": profiled_traffic(args.workload) if args.scale == 1.0 and args.tile == 0 else None,
                "algorithmic_bytes": int(n_windows) * step_bytes,
                "bytes_per_window": step_bytes,
                "launch_ms": launch_ms,
                "achieved_with_passes_in_flight": int(n_windows) * step_bytes / (dt / args.steps) / 1e9,
            },
        }
        if not args.no_cpu and world == 1:             # the CPU legs run on rank 0 at N=1 only
            cpu_wps, used, done, res, cpu_passes, cpu_dt = cpu_baseline(ctgs, prm)
            # parity in the same run: peaks of the sampled ctgs against the oracle
            ok = True
            for c, (cnt, sig) in enumerate(res):
                idx = np.flatnonzero(sig)
                mine = peaks[peaks["ctg"] == c]
                ok &= bool(np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], sig[idx])
                           and np.array_equal(mine["gc_count"], cnt[idx]))
            out["cpu_baseline"] = {
                "value": cpu_wps, "unit": "windows/s", "cores": 1, "kind": "port",
                "sample": f"first {used} ctgs of the same workload, {cpu_passes} times over ({done} windows, "
                          f"{cpu_dt:.1f} s), oracle/gams_oracle.c single thread; host has {os.cpu_count()} cpus",
            }
            # the reference's --parallel T model (one ctg per worker thread, wave.rs:288-299), for context
            from concurrent.futures import ThreadPoolExecutor
            from oracle import oracle as ora

            T = min(16, os.cpu_count() or 1, len(ctgs))
            t0p = time.perf_counter()
            with ThreadPoolExecutor(T) as ex:
                sizes = list(ex.map(lambda c: ora.wave_windows(c["seq"], prm["size"], prm["step"], prm["lag"],
                                                               prm["threshold"], prm["influence"])[0].size, ctgs))
            out["cpu_baseline_parallel"] = {"value": sum(sizes) / (time.perf_counter() - t0p), "unit": "windows/s",
                                            "cores": T, "kind": "port",
                                            "sample": f"all {len(ctgs)} ctgs, one ctg per worker thread"}
            out["parity_vs_oracle"] = ok
    if rank == 0 and world == 1 and not args.no_extra and args.workload == "S288c" and args.scale == 1.0:
        # The 12 Mb workload lives in L2/MALL and one launch lasts microseconds.  Also measure a
        # genome larger than the 256 MiB Infinity Cache so that the HBM roofline fraction of the
        # kernel itself can be read (reported beside, never as `value`).
        plan.close()
        ss.close()
        from gams_amd import synth

        big = synth.genome_ctgs(synth.SYNTH384_LENGTHS, 1000000, first_chr_index=500)
        ss = engine.SeqSet(eng, [c["seq"] for c in big])
        plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, tile_windows=args.tile, **prm)
        for _ in range(3):
            plan.run()
        eng.sync()
        eng.timer_start()
        reps = 20
        for _ in range(reps):
            plan.run()
        ms = eng.timer_stop() / reps
        nw = plan.total_windows
        gbps = nw * prm["step"] / (ms * 1e-3) / 1e9
        out["extra"] = {
            "workload": f"synthetic {sum(len(c['seq']) for c in big)} bp ({len(big)} ctgs, piece 1000000), "
                        f"beyond the 256 MiB L3",
            "windows_per_s": nw / (ms * 1e-3), "launch_ms": ms, "achieved_GBps": gbps,
            "frac_of_8TBps": gbps / HBM_PEAK_GBPS,
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    plan.close()
    ss.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
