#!/usr/bin/env python3
"""bench.py -- GC windows/s of the `wave` hot path on MI355X (driver contract in the task brief).

A step = one pass of the hot path over one batch: every ctg of one synthetic genome (resident in
HBM) -> gc counts -> smoothed z-score signals -> compacted peak records, one kernel launch.

Default workload (N = 1): BASELINE.json configs[2], an A. thaliana-sized genome (119,667,750 bp,
7 chromosomes, --piece 500000), size 100 / step 10 / lag 100 / threshold 3 / influence 1 -- the
largest step-10 configuration that BASELINE names for one GPU.  Three distinct genomes of that
shape stay resident (360 MB, more than the 256 MiB Infinity Cache) and the steps rotate over them,
so every pass streams its bytes from HBM the way batch after batch of ctgs does
(src/cmd_gams/wave.rs:288-299); the three plans sit on three HIP streams of the handle, i.e. up to
three passes are in flight.  Before the W warm-up steps the clocks are ramped with at least 50 ms
of passes, whatever W is.

N > 1 (one rank per GPU, torch.distributed.run):
  default      weak scaling: N x 3 genomes of the same shape, assigned to the ranks by
               gams_amd.shard.lpt_assign on their window counts (three each); no exchange on the
               data path, the barrier and the max-over-ranks timing are the only collectives.
  --workload GRCh38-step10 | GRCh38-step1
               strong scaling (BASELINE configs[3]): ONE 3.09-Gb genome whose 2,937 ctgs are
               LPT-sharded over the ranks by window count; total windows/s = all windows / the
               slowest rank's time.

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
# kernel arguments in device memory (gams_gpu_create sets it too, but torch may initialise HIP first)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
ROUND = "r03"

WORKLOADS = {
    # name: (chromosome lengths, piece, size, step, lag, scaling for N > 1)
    "S288c": ("S288C_LENGTHS", 500000, 100, 10, 100, "weak"),
    "Atha": ("ATHA_LENGTHS", 500000, 100, 10, 100, "weak"),
    "synth384": ("SYNTH384_LENGTHS", 1000000, 100, 10, 100, "weak"),
    "synth384-step1": ("SYNTH384_LENGTHS", 1000000, 100, 1, 100, "weak"),     # configs[3]'s kernel on a genome that is quick to make
    "GRCh38-step10": ("GRCH38_LENGTHS", 1000000, 100, 10, 100, "strong"),
    "GRCh38-step1": ("GRCH38_LENGTHS", 1000000, 100, 1, 100, "strong"),
}


def profiled_traffic(workload):
    """HBM bytes per launch of the wave kernel from the committed rocprofv3 PMC passes
    (profiles/*_pmc.json, made by tools/prof.sh): FETCH_SIZE and WRITE_SIZE are in KiB and were
    collected in separate passes; on gfx950 FETCH_SIZE counts half the bytes of a wide coalesced
    stream, so it is doubled (MI355X_MICROARCH.md, HBM section).  -> (bytes, the file they come from);
    (None, None) if no profile matches.  PMC passes cannot run inside the timed process (they need
    rocprofv3 around it and serialise dispatches), so this is a committed measurement of the same
    command, not a live one -- `traffic_source` in the JSON line says which."""
    for rnd in (ROUND, "r02", "r01"):
        rel = os.path.join("profiles", f"{rnd}_{workload}_wave_pmc.json")
        try:
            with open(os.path.join(ROOT, rel)) as fh:
                p = json.load(fh)
            return (2.0 * p["FETCH_SIZE"]["mean"] + p["WRITE_SIZE"]["mean"]) * 1024.0, rel
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def window_count(n, size, step):
    return max(0, (n - size) // step + 1)


def build_batches(name, rank, world, copies, scale):
    """-> (batches, params, genome_bp, scaling, shard_note); a batch = list of ctg dicts."""
    from gams_amd import shard, synth

    lengths_name, piece, size, step, lag, scaling = WORKLOADS[name]
    lengths = getattr(synth, lengths_name)
    if scale != 1.0:
        lengths = [max(20000, int(x * scale)) for x in lengths]
    prm = dict(size=size, step=step, lag=lag, threshold=3.0, influence=1.0)
    if scaling == "strong":
        # ONE genome.  Its layout is closed form, ownership (LPT by window count; wave.rs:288-299: the ctg
        # is the unit of work) is decided on the layout, and a rank generates only the ctgs it owns
        mine, loads, n_ctgs = synth.sharded_genome_ctgs(lengths, piece, rank, world, size, step)
        note = (f"{n_ctgs} ctgs LPT-sharded x{world} by windows; rank loads max/mean "
                f"{max(loads) / (sum(loads) / world):.4f}; rank {rank} generated its own {len(mine)} ctgs only")
        return [mine], prm, sum(lengths), "strong", note
    # weak: world * copies genomes of the same shape; the unit handed to a rank is a genome (its
    # chromosomes are seeded by the genome's index, so a rank generates only what it owns)
    n_genomes = world * copies
    per_genome = sum(window_count(x, size, step) for x in lengths)   # equal weights: LPT deals them round
    owner = shard.lpt_assign([per_genome] * n_genomes, world)
    mine = [g for g, o in enumerate(owner) if o == rank]
    batches = [synth.genome_ctgs(lengths, piece, first_chr_index=1 + 1000 * g) for g in mine]
    note = f"{n_genomes} genomes dealt to {world} rank(s) by lpt_assign; rank {rank} holds genomes {mine}"
    return batches, prm, sum(lengths), "weak", note


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
        for c in ctgs:
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


def end_to_end(eng, ctgs, prm, reps=4):
    """PCIe-inclusive truth beside `value` (never `value`): one batch = every ctg of one genome of the workload,
    through the host operator gams::wave_proc_ctgs(_gz) -- the seam of wave.rs:121-215 -- from (a) the gzip'd
    `seq:` values as the store holds them (redis.rs:142-161) and (b) gunzipped host buffers, to the TSV text.
    Per form: the best whole call of `reps` (host clock inside the C++ layer), and the stages of one call run
    with the device drained at every stage boundary."""
    from concurrent.futures import ThreadPoolExecutor

    from gams_amd import host

    T = max(1, min(16, os.cpu_count() or 1))        # inflate workers: the box's CPU share for one GPU
    hc = [dict(id=c["id"], chr_id=c["chr_id"], chr_start=c["chr_start"], chr_end=c["chr_end"], seq=c["seq"]) for c in ctgs]
    with ThreadPoolExecutor(T) as ex:               # flate2 Compression::fast = zlib level 1 (redis.rs:149-154)
        for c, gz in zip(hc, ex.map(lambda c: host.encode_gz(bytes(c["seq"])), hc)):
            c["gz"] = gz
    windows = sum(window_count(len(c["seq"]), prm["size"], prm["step"]) for c in hc)
    bases = sum(len(c["seq"]) for c in hc)
    kw = dict(size=prm["size"], step=prm["step"], lag=prm["lag"], threshold=prm["threshold"], influence=prm["influence"])
    out = {"workload": f"{len(hc)} ctgs, {bases} bases, {windows} windows: one batch, text of every ctg returned",
           "gz_bytes": int(sum(len(c["gz"]) for c in hc))}
    texts = []
    for tag, fn in (("from_gz_seq_values", lambda sync: host.wave_gz(eng, hc, threads=T, sync=sync, **kw)),
                    ("from_gunzipped_buffers", lambda sync: host.wave_timed(eng, hc, sync=sync, **kw))):
        fn(False)                                   # first call: page-locks the image / staging blocks (pooled after)
        best, text = None, None
        for _ in range(reps):
            text, st = fn(False)
            if best is None or st["total_ms"] < best["total_ms"]:
                best = st
        _, staged = fn(True)
        texts.append(text)
        keep = ("inflate_upload_ms", "upload_ms", "plan_ms", "kernel_ms", "peaks_ms", "format_ms", "total_ms")
        out[tag] = {"total_ms": best["total_ms"], "windows_per_s": windows / (best["total_ms"] * 1e-3),
                    "sequence_GBps": bases / (best["total_ms"] * 1e-3) / 1e9,
                    "rows": text.count(b"\n"), "text_bytes": len(text), "peaks": best["peaks"],
                    "stages_ms_device_drained_per_stage": {k: staged[k] for k in keep if staged[k] > 0 or k == "total_ms"}}
        if tag == "from_gz_seq_values":
            out[tag]["inflate_threads"] = best["inflate_threads"]
            out[tag]["inflate"] = "libdeflate.so.0 (dlopen) or zlib, one ctg per worker at a time, into a page-locked image; DMA follows the workers"
    out["same_text_both_forms"] = texts[0] == texts[1]
    return out


def secondary_metrics(eng):
    """sw rows/s and interval queries/s on one GPU's share of BASELINE configs[2] /
    configs[4]: device time of the kernel (HIP events around it), inputs resident, with the
    algorithmic bytes per unit and the fraction of the HBM roofline that gives."""
    import ctypes as C

    from gams_amd import _lib, engine, synth

    lib = eng.lib

    def kernel_ms():
        ms = C.c_float()
        eng.check(lib.gams_gpu_last_kernel_ms(eng.h, C.byref(ms)))
        return ms.value

    def entry(units, ms, bytes_per_unit, unit, what):
        rate = units / (ms * 1e-3)
        gbps = rate * bytes_per_unit / 1e9
        return {"value": rate, "unit": unit, "kernel_ms": ms, "bytes_per_unit": bytes_per_unit,
                "achieved_GBps": gbps, "frac_of_8TBps": gbps / HBM_PEAK_GBPS, "workload": what}

    out = {}
    # ---- sw: 1e5 point features over a 30-Mb chromosome (configs[2] shape, one chromosome) ----
    chrom = synth.chromosome(30_427_671, 1)
    ctgs = synth.gen_ctgs("1", chrom, piece=500000)
    ss = engine.SeqSet(eng, [c["seq"] for c in ctgs])
    feats = synth.point_features(ctgs, 100000)
    # every ctg's features through ONE gams_gpu_sw_batch call (one launch); the second repetition is the
    # measured one (gc index built, pools warm)
    sel = np.array([i for i, f in enumerate(feats) if f.size], np.uint32)
    cst = np.array([ctgs[i]["chr_start"] for i in sel], np.int32)
    foff = np.concatenate([[0], np.cumsum([feats[i].size for i in sel])]).astype(np.uint64)
    fall = np.ascontiguousarray(np.concatenate([feats[i] for i in sel]), np.int32)
    n = C.c_uint64()
    eng.check(lib.gams_gpu_sw_batch(eng.h, ss.p, sel.size, sel.ctypes.data, cst.ctypes.data, foff.ctypes.data,
                                    fall.ctypes.data, fall.ctypes.data, 100, 20, 500, None, 0, None, C.byref(n)))
    rows = np.zeros(n.value, _lib.SW_ROW_DTYPE)
    for rep in range(2):
        eng.check(lib.gams_gpu_sw_batch(eng.h, ss.p, sel.size, sel.ctypes.data, cst.ctypes.data, foff.ctypes.data,
                                        fall.ctypes.data, fall.ctypes.data, 100, 20, 500, rows.ctypes.data, rows.size,
                                        None, C.byref(n)))
    rows_total, ms_total = n.value, kernel_ms()
    out["sw_rows_per_s"] = entry(rows_total, ms_total, 84, "rows/s",
                                 f"{int(fall.size)} point features, {len(ctgs)} ctgs of a 30.4-Mb chromosome, size 100 "
                                 f"max 20 resize 500 ({rows_total} rows), every ctg in one gams_gpu_sw_batch launch")
    ss.close()
    # ---- intervals: configs[4] cut to one of 8 GPUs ----
    w = synth.c5_workload(share=8)
    nq = int(w["q_start"].size)
    ix = C.c_void_p()
    t0 = time.perf_counter()
    eng.check(lib.gams_index_create(eng.h, w["n_ctg"], w["rg_off"].ctypes.data, w["rg_start"].ctypes.data,
                                    w["rg_stop"].ctypes.data, C.byref(ix)))
    build_s = time.perf_counter() - t0
    qe_excl = w["q_end"]                      # locate/count pass (rg.start, rg.end): the end is exclusive (utils.rs:35)
    cnt = np.zeros(nq, np.int32)
    for _ in range(2):
        eng.check(lib.gams_gpu_count(eng.h, ix, w["q_ctg"].ctypes.data, w["q_start"].ctypes.data,
                                     qe_excl.ctypes.data, nq, cnt.ctypes.data))
    out["count_queries_per_s"] = entry(nq, kernel_ms(), 16, "queries/s",
                                       f"{nq} unsorted queries (length 1-2000) against {int(w['rg_start'].size)} stored "
                                       f"point ranges in {w['n_ctg']} ctgs")
    out["count_queries_per_s"]["index_build_s"] = build_s
    t0 = time.perf_counter()
    eng.check(lib.gams_gpu_count(eng.h, ix, w["q_ctg"].ctypes.data, w["q_start"].ctypes.data,
                                 qe_excl.ctypes.data, nq, cnt.ctypes.data))
    call_s = time.perf_counter() - t0
    out["count_queries_per_s"]["whole_call_from_host_arrays_per_s"] = nq / call_s
    lib.gams_index_destroy(eng.h, ix)
    # locate: which ctg holds a range (idx:ctg:{chr}: one group per chromosome)
    ixc = C.c_void_p()
    eng.check(lib.gams_index_create(eng.h, w["n_chr"], w["ctg_off"].ctypes.data, w["ctg_start"].ctypes.data,
                                    w["ctg_stop"].ctypes.data, C.byref(ixc)))
    hit = np.zeros(nq, np.int64)
    for _ in range(2):
        eng.check(lib.gams_gpu_locate(eng.h, ixc, w["q_chr"].ctypes.data, w["q_start"].ctypes.data,
                                      qe_excl.ctypes.data, nq, hit.ctypes.data))
    out["locate_queries_per_s"] = entry(nq, kernel_ms(), 16, "queries/s",
                                        f"{nq} ranges against {w['n_ctg']} ctgs of {w['n_chr']} chromosomes")
    lib.gams_index_destroy(eng.h, ixc)
    # anno: covered proportion of each range inside its ctg
    sp = C.c_void_p()
    eng.check(lib.gams_spans_create(eng.h, w["n_chr"], w["sp_off"].ctypes.data, w["sp_lo"].ctypes.data,
                                    w["sp_hi"].ctypes.data, C.byref(sp)))
    s = w["q_start"].astype(np.int32)
    e = w["q_end"].astype(np.int32)
    cl = (((s - 1) // w["piece"]) * w["piece"] + 1).astype(np.int32)
    ch = (cl + (w["piece"] - 1)).astype(np.int32)
    prop = np.zeros(nq, np.float32)
    for _ in range(2):
        eng.check(lib.gams_gpu_cover(eng.h, sp, w["q_chr"].ctypes.data, cl.ctypes.data, ch.ctypes.data,
                                     s.ctypes.data, e.ctypes.data, nq, prop.ctypes.data))
    out["anno_lines_per_s"] = entry(nq, kernel_ms(), 16, "lines/s",
                                    f"{nq} lines against {int(w['sp_lo'].size)} spans in {w['n_chr']} chromosomes")
    lib.gams_spans_destroy(eng.h, sp)
    return out


def self_launch(n):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks as
    children -- `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` -- BEFORE
    this process has imported torch or loaded the library (a process that has touched the GPU must never
    exec or re-launch), relay the children's output (rank 0 prints the one JSON line) and return their
    exit code.  The rendezvous is on 127.0.0.1 at a port the kernel just handed out."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(port)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is relayed line by line: JSON lines (rank 0's result) to stdout, anything else
    # (gloo's connection chatter in a rehearsal) to stderr, so stdout stays the ONE line the contract asks for
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in proc.stdout:
            dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            dst.write(line)
            dst.flush()
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return 130


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="Atha", choices=sorted(WORKLOADS))
    ap.add_argument("--copies", type=int, default=3,
                    help="weak workloads: distinct genomes resident per GPU, the steps rotate over them")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink/grow chromosome lengths (testing)")
    ap.add_argument("--tile", type=int, default=0, help="windows per tile (0 = library default)")
    ap.add_argument("--depth", type=int, default=0,
                    help="passes in flight of a single-batch workload (gams_wave_plan_set_depth); 0 = 2. "
                         "Multi-batch workloads put batch j on HIP stream j % 4 instead")
    ap.add_argument("--one-at-a-time", action="store_true", help="every batch on one stream, depth 1 (profiling)")
    ap.add_argument("--ramp-ms", type=float, default=50.0, help="time-based clock ramp before the warm-up steps")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the S288c / beyond-L3 extra measurements")
    ap.add_argument("--no-secondary", action="store_true", help="skip the sw / interval metrics")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (gz / host buffers -> TSV text) block")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="join the ranks, run the bench's three collectives once, print a JSON line and stop "
                         "(no GPU work: checks the launch path)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started as plain `python bench.py --gpus N`: be our own launcher (see self_launch)
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    # stdout carries ONE line, rank 0's JSON: everything else this process or its libraries print there
    # (RCCL's version banner at the first collective, gloo's connection chatter) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch

    dist = None
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)
    # RCCL needs one device per rank; with fewer devices than ranks (a rehearsal of the N > 1 path on a
    # one-GPU box) the barrier and the two scalar reductions go over gloo instead -- nothing else changes,
    # there is no collective on the data path.  GAMS_BENCH_BACKEND overrides.
    backend = os.environ.get("GAMS_BENCH_BACKEND") or ("nccl" if n_dev >= world else "gloo")
    force_dist = os.environ.get("GAMS_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ   # N = 1 through RCCL (test)
    if world > 1 or force_dist:
        import torch.distributed as dist

        if n_dev:
            torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    if args.rendezvous_only:
        # plumbing check of the N > 1 entry path (runs without a GPU too): every rank joins, one barrier,
        # one MAX and one SUM reduction -- the only collectives of the whole bench -- then rank 0 reports
        got = None
        if dist is not None:
            red_dev = "cuda" if backend == "nccl" else "cpu"
            dist.barrier()
            t = torch.tensor([float(rank)], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = torch.tensor([1.0], dtype=torch.float64, device=red_dev)
            dist.all_reduce(w, op=dist.ReduceOp.SUM)
            got = [float(t.item()), float(w.item())]
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit({"rendezvous": "ok", "n_gpus": world, "backend": backend if dist is not None else None,
                  "max_rank_sum_ranks": got, "devices_visible": n_dev})
        return

    from gams_amd import _lib, engine

    eng = engine.Engine(device)
    arch, cus, hbm = eng.device_info()

    batches, prm, genome_bp, scaling, shard_note = build_batches(args.workload, rank, world, args.copies, args.scale)
    sets = [engine.SeqSet(eng, [c["seq"] for c in b]) for b in batches]
    plans = [engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, tile_windows=args.tile, **prm) for ss in sets]
    nb = len(plans)
    win_per_batch = [int(p.total_windows) for p in plans]

    def set_flight(on):
        """on: batch j on stream j % 4 (or, with one batch, `depth` ways); off: everything serial on one stream"""
        for j, p in enumerate(plans):
            if nb == 1:
                p.set_depth((args.depth or 2) if on else 1)
            p.set_lane(j % 4 if on and nb > 1 else 0)
            # tapered launches (smaller tiles at the end of the tile table) shorten the tail of a launch
            # that runs alone; with passes in flight the tails overlap and the small tiles are only extra
            # work, so a host that keeps batches in flight turns them off (gams_wave_plan_set_taper)
            p.set_taper(0 if on else -1)

    def barrier():
        eng.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def run_steps(n, start=0):
        for i in range(n):
            plans[(start + i) % nb].run()
        return sum(win_per_batch[(start + i) % nb] for i in range(n))

    in_flight = not args.one_at_a_time
    set_flight(in_flight)
    # time-based clock ramp (independent of --warmup): the chip needs tens of ms of load before its
    # clocks and the driver's queues are in steady state
    t_ramp = time.perf_counter()
    ramp_steps = 0
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms or ramp_steps < 2 * nb:
        run_steps(nb * 8, ramp_steps)
        ramp_steps += nb * 8
        eng.sync()
    run_steps(args.warmup, ramp_steps)
    barrier()
    t0 = time.perf_counter()
    my_windows = run_steps(args.steps, ramp_steps + args.warmup)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        red_dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        w = torch.tensor([float(my_windows)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        total_windows = float(w.item())
    else:
        total_windows = float(my_windows)
    peaks = [p.peaks() for p in plans]
    n_exact = sum(p.exact_count() for p in plans)

    # Roofline leg: the dominant kernel's own launch duration, one pass at a time on the library's
    # compute stream, HIP events on that stream (what rocprofv3 --kernel-trace reports per launch);
    # the batches still rotate, so every launch streams from HBM.
    set_flight(False)
    serial_steps = max(nb * 4, min(max(args.steps, 30), 300))
    run_steps(nb * 3)
    eng.sync()
    eng.timer_start()
    run_steps(serial_steps)
    kernel_ms = eng.timer_stop()      # HIP events on the library's compute stream
    eng.sync()
    launch_ms = kernel_ms / serial_steps
    kernel_name = plans[0].kernel_name()   # the instantiation this leg launched (rocprofv3's row of the same name)
    serial_windows = sum(win_per_batch[i % nb] for i in range(serial_steps)) / serial_steps

    # run + peaks, pipelined: what a host that consumes the peaks pays per batch (wave.rs:157-214 reads them).
    # Batch j runs on lane j; while its kernel is in flight the oldest batch's peaks are packed on the device
    # (wave_offsets_kernel + wave_gather_kernel on the readback stream) and fetched into page-locked memory.
    pass_with_peaks_ms, pass_with_rows_ms, rows_text_bytes, resident_to_tsv_ms = None, None, None, None
    if world == 1 and nb > 1:
        set_flight(True)
        for p in plans:
            p.set_pipelined(True)
        for i in range(2 * nb):
            plans[i % nb].run()
            plans[(i + 1) % nb].peaks_count() if i >= nb - 1 else None
        eng.sync()
        k_steps = max(args.steps, 60)
        t0 = time.perf_counter()
        for i in range(k_steps):
            plans[i % nb].run()
            plans[(i + 1) % nb].peaks_count()          # the batch queued nb - 1 steps ago
        eng.sync()
        pass_with_peaks_ms = (time.perf_counter() - t0) / k_steps * 1e3
        # ... and with the rows: merged and formatted on the device (gams_wave_rows_*), the TSV text of the batch
        # lands in page-locked memory: what wave.rs:157-214 produces, with no host merge / formatting left
        pass_with_rows_ms, rows_text_bytes = None, None
        try:
            for p, b in zip(plans, batches):
                p.rows_setup([c["chr_id"] for c in b], [c["chr_start"] for c in b], 0.2)
            for i in range(2 * nb):
                plans[i % nb].run()
                plans[i % nb].rows_begin()
                if i >= nb - 1:
                    plans[(i + 1) % nb].rows_end(copy=False)
            for i in range(nb - 1):
                plans[(2 * nb + i + 1) % nb].rows_end(copy=False)
            eng.sync()
            t0 = time.perf_counter()
            for i in range(k_steps):
                if i >= nb:
                    plans[i % nb].rows_end(copy=False)      # the rows queued nb steps ago, before the slot is reused
                plans[i % nb].run()
                plans[i % nb].rows_begin()
            rows_text_bytes = [plans[(k_steps + j) % nb].rows_end(copy=False) for j in range(nb)]
            eng.sync()
            pass_with_rows_ms = (time.perf_counter() - t0) / k_steps * 1e3
            # one batch alone, nothing in flight: resident sequence -> the batch's TSV text in host memory
            t0 = time.perf_counter()
            for i in range(30):
                plans[i % nb].run()
                plans[i % nb].rows_begin()
                plans[i % nb].rows_end(copy=False)
            resident_to_tsv_ms = (time.perf_counter() - t0) / 30 * 1e3
        except _lib.GamsError as e:
            pass_with_rows_ms = f"unavailable: {e}"
        for p in plans:
            p.set_pipelined(False)
        set_flight(False)

    out = None
    traffic, traffic_src = profiled_traffic(args.workload) if args.scale == 1.0 and args.tile == 0 else (None, None)
    if rank == 0:
        step_bytes = prm["step"]                       # `step` bytes read per window
        achieved = serial_windows * step_bytes / (launch_ms * 1e-3) / 1e9
        lengths_name, piece = WORKLOADS[args.workload][0], WORKLOADS[args.workload][1]
        out = {
            "metric": f"GC windows/s (size {prm['size']}, step {prm['step']})",
            "value": total_windows / dt,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "pass_with_peaks_ms": pass_with_peaks_ms,
            "pass_with_rows_ms": pass_with_rows_ms,
            "resident_to_tsv_ms": resident_to_tsv_ms,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}-shaped synthetic genome ({genome_bp} bp, piece {piece}): wave size "
                            f"{prm['size']} step {prm['step']} lag {prm['lag']} threshold 3 influence 1, peaks "
                            f"compacted on device; {nb} batch(es) resident per GPU, steps rotate over them",
                "batches_per_gpu": nb,
                "ctgs_per_batch": [len(b) for b in batches],
                "windows_per_step": win_per_batch,
                "resident_bytes_per_gpu": int(sum(sum(len(c["seq"]) for c in b) for b in batches)),
                "peaks_per_step": [int(p.size) for p in peaks],
                "rows_text_bytes_per_step": rows_text_bytes,
                "exact_path_windows_per_rotation": int(n_exact),
                "passes_in_flight": (min(nb, 4) if nb > 1 else (args.depth or 2)) if in_flight else 1,
                "tapered_launches": "off in the timed region (passes in flight), on in the roofline leg (one pass at a time)"
                                    if in_flight else "on",
                "clock_ramp_ms": args.ramp_ms,
                "clock_ramp_steps": ramp_steps,
                "windows_per_s_one_pass_at_a_time": serial_windows / (launch_ms * 1e-3),
                "sharding": shard_note,
                "device": arch,
                "parallelism": f"ctg-sharded x{world}, no collective on the data path",
                "collectives": (f"{backend}: barrier + MAX(time) + SUM(windows) around the timed region"
                                if dist is not None else "none (one rank)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": (f"{traffic_src}: 2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over "
                                   f"this command (committed profile, not measured in this run)") if traffic_src else None,
                "algorithmic_bytes": int(serial_windows * step_bytes),
                "bytes_per_window": step_bytes,
                "launch_ms": launch_ms,
                "launches_timed": serial_steps,
                "achieved_with_passes_in_flight": total_windows / world * step_bytes / dt / 1e9,
            },
        }
        if not args.no_cpu and world == 1:             # the CPU legs run on rank 0 at N=1 only
            cpu_wps, used, done, res, cpu_passes, cpu_dt = cpu_baseline(batches[0], prm)
            # parity in the same run: peaks of the sampled ctgs against the oracle
            ok = True
            for c, (cnt, sig) in enumerate(res):
                idx = np.flatnonzero(sig)
                mine = peaks[0][peaks[0]["ctg"] == c]
                ok &= bool(np.array_equal(mine["window"], idx) and np.array_equal(mine["signal"], sig[idx])
                           and np.array_equal(mine["gc_count"], cnt[idx]))
            out["cpu_baseline"] = {
                "value": cpu_wps, "unit": "windows/s", "cores": 1, "kind": "port",
                "sample": f"first {used} ctgs of the first batch, {cpu_passes} time(s) over ({done} windows, "
                          f"{cpu_dt:.1f} s), oracle/gams_oracle.c single thread; host has {os.cpu_count()} cpus",
            }
            # the reference's --parallel T model (one ctg per worker thread, wave.rs:288-299) at T = all
            # host cores, bounded to the first batch
            from concurrent.futures import ThreadPoolExecutor
            from oracle import oracle as ora

            T = max(1, min(os.cpu_count() or 1, len(batches[0])))
            t0p = time.perf_counter()
            with ThreadPoolExecutor(T) as ex:
                sizes = list(ex.map(lambda c: ora.wave_windows(c["seq"], prm["size"], prm["step"], prm["lag"],
                                                               prm["threshold"], prm["influence"])[0].size,
                                    batches[0]))
            out["cpu_baseline_parallel"] = {"value": sum(sizes) / (time.perf_counter() - t0p), "unit": "windows/s",
                                            "cores": T, "kind": "port",
                                            "sample": f"all {len(batches[0])} ctgs of the first batch, one ctg per "
                                                      f"worker thread, {T} threads on {os.cpu_count()} cpus"}
            out["parity_vs_oracle"] = ok
        if world == 1 and not args.no_e2e:
            try:                                    # (an optional block must not take the line with it)
                out["end_to_end"] = end_to_end(eng, batches[0], prm)
            except Exception as e:  # noqa: BLE001
                out["end_to_end"] = {"error": f"{type(e).__name__}: {e}"}
    for p in plans:
        p.close()
    for ss in sets:
        ss.close()
    del batches
    if rank == 0 and world == 1 and not args.no_extra and args.scale == 1.0:
        from gams_amd import synth

        extra = {}
        try:
            # (a) BASELINE configs[1]: the 12-Mb genome lives in L2/MALL and one launch lasts microseconds
            #     (launch-latency bound); (b) a 384-Mb genome, one launch beyond the Infinity Cache.
            for tag, lengths, piece, depth in (("S288c", synth.S288C_LENGTHS, 500000, 4),
                                               ("synth384", synth.SYNTH384_LENGTHS, 1000000, 1)):
                if tag == args.workload:
                    continue
                g = synth.genome_ctgs(lengths, piece, first_chr_index=500 if tag == "synth384" else 1)
                ss = engine.SeqSet(eng, [c["seq"] for c in g])
                plan = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, tile_windows=args.tile, **prm)
                t_r = time.perf_counter()                      # time-based ramp here too: 20 launches are 0.15-1.4 ms
                while (time.perf_counter() - t_r) * 1e3 < 30.0:
                    for _ in range(50):
                        plan.run()
                    eng.sync()
                reps = 200 if tag == "S288c" else 40
                eng.timer_start()
                for _ in range(reps):
                    plan.run()
                ms = eng.timer_stop() / reps
                nw = plan.total_windows
                e = {"workload": f"{sum(len(c['seq']) for c in g)} bp, {len(g)} ctgs, piece {piece}",
                     "windows_per_launch": int(nw), "launch_ms": ms, "windows_per_s": nw / (ms * 1e-3),
                     "achieved_GBps": nw * prm["step"] / (ms * 1e-3) / 1e9,
                     "frac_of_8TBps": nw * prm["step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS}
                if depth > 1:
                    plan.set_depth(depth)
                    plan.run_n(2000)
                    eng.sync()
                    t0 = time.perf_counter()
                    plan.run_n(1000)
                    eng.sync()
                    e["windows_per_s_4_in_flight"] = nw * 1000 / (time.perf_counter() - t0)
                extra[tag] = e
                plan.close()
                if tag == "synth384" and prm["step"] != 1:
                    # SURVEY 8(d)'s secondary figure, configs[3]'s geometry (size 100 / step 1) on the same resident bytes:
                    # compute bound (1 B per window), a tile per wave
                    p1 = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, **dict(prm, step=1))
                    for _ in range(20):
                        p1.run()
                    eng.sync()
                    eng.timer_start()
                    for _ in range(20):
                        p1.run()
                    ms1 = eng.timer_stop() / 20
                    extra["synth384_step1"] = {"workload": e["workload"] + ", size 100 step 1", "kernel": p1.kernel_name(),
                                               "windows_per_launch": int(p1.total_windows), "launch_ms": ms1,
                                               "windows_per_s": p1.total_windows / (ms1 * 1e-3),
                                               "bytes_per_window": 1, "peaks": int(p1.peaks_count())}
                    p1.close()
                ss.close()
            # --influence other than 1 (stat.rs:42: a recurrence per ctg; guess-and-iterate on the device): ms per pass over the
            # 59 ctgs of one 30-Mb chromosome, a reader waiting for the fixed point, and how the pass settled
            if prm["step"] == 10:
                c30 = synth.gen_ctgs("1", synth.chromosome(30_427_671, 1), piece=500000)
                ss = engine.SeqSet(eng, [c["seq"] for c in c30])
                infl = {}
                for tag, influence, thr in (("influence_0.5", 0.5, 3.0), ("influence_0", 0.0, 3.0), ("influence_0_threshold_2", 0.0, 2.0)):
                    pl = engine.WavePlan(eng, ss, flags=_lib.WAVE_PEAKS, **dict(prm, influence=influence, threshold=thr))
                    for _ in range(3):
                        pl.run()
                        pl.peaks_count()
                    eng.sync()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        pl.run()
                        n_pk = pl.peaks_count()
                    eng.sync()
                    ms_i = (time.perf_counter() - t0) / 10 * 1e3
                    sweeps, serial = pl.settled()
                    infl[tag] = {"ms_per_pass": ms_i, "windows_per_s": pl.total_windows / (ms_i * 1e-3), "peaks": int(n_pk),
                                 "sweeps_queued": sweeps, "serial_fallback": bool(serial)}
                    pl.close()
                infl["workload"] = f"{sum(len(c['seq']) for c in c30)} bp, {len(c30)} ctgs: pass + packed peaks in host memory"
                extra["influence_30Mb"] = infl
                ss.close()
        except Exception as e:  # noqa: BLE001
            extra["error"] = f"{type(e).__name__}: {e}"
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            out["secondary"] = secondary_metrics(eng)
        except Exception as e:  # noqa: BLE001
            out["secondary"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        emit(out)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
