"""N > 1 path on CPU: two gloo ranks shard the ctgs of a genome (LPT), each computes its ctgs'
rows, rank 0 gathers them in ctg order and must reproduce the serial output.  The compute leg
uses the CPU oracle here (no GPU in this container); the sharding and gather are the code under
test and are what bench.py / a multi-GPU host use unchanged."""
import os
import subprocess
import sys

import pytest

from gams_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["GAMS_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GAMS_ROOT"], "tests"))
import torch.distributed as dist
import helpers
from gams_amd import shard
from oracle import oracle as ora

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seqs = helpers.load_s288c()
ctgs = []
for chr_id in ("I", "Mito"):
    ctgs += helpers.gen_ctgs(chr_id, seqs[chr_id], piece=30000)
owned = shard.shard_ctgs(ctgs, world, rank)
rows = [ora.wave_proc_ctg(ctgs[i]["chr_id"], ctgs[i]["chr_start"], ctgs[i]["chr_end"], ctgs[i]["seq"]) for i in owned]
all_rows = shard.gather_rows(rows, owned, len(ctgs), dist)
if rank == 0:
    serial = [ora.wave_proc_ctg(c["chr_id"], c["chr_start"], c["chr_end"], c["seq"]) for c in ctgs]
    assert all_rows == serial, "gathered rows differ from the serial run"
    n_owned = [len(shard.shard_ctgs(ctgs, world, r)) for r in range(world)]
    assert sum(n_owned) == len(ctgs) and min(n_owned) > 0
    print("SHARD_OK", len(ctgs), n_owned)
dist.barrier()
dist.destroy_process_group()
'''


def test_lpt_is_a_partition_and_balanced():
    w = [500, 30, 30, 400, 1, 260, 240, 10]
    owner = shard.lpt_assign(w, 2)
    loads = [sum(x for x, o in zip(w, owner) if o == r) for r in range(2)]
    assert sorted(set(owner)) == [0, 1]
    assert abs(loads[0] - loads[1]) <= max(w) // 4
    assert shard.lpt_assign(w, 1) == [0] * len(w)
    assert shard.lpt_assign([], 4) == []


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GAMS_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "SHARD_OK" in res.stdout


def test_grch38_ctgs_lpt_partition_is_balanced():
    """BASELINE configs[3]: the 2,937 ctgs of the GRCh38-shaped genome (piece 1000000), sharded by
    window count (src/cmd_gams/wave.rs:288-299: the ctg is the unit of work).  Every ctg has exactly
    one owner and the heaviest rank carries at most 2 % more than the mean at N = 2, 4, 8."""
    from gams_amd import synth

    lens = synth.layout_ctg_lengths(synth.GRCH38_LENGTHS, 1000000)
    assert len(lens) == 2937
    for step in (1, 10):
        w = [(x - 100) // step + 1 for x in lens]
        for n in (2, 4, 8):
            owner = shard.lpt_assign(w, n)
            assert len(owner) == len(w) and set(owner) == set(range(n))
            loads = [sum(x for x, o in zip(w, owner) if o == r) for r in range(n)]
            assert sum(loads) == sum(w)
            assert max(loads) <= 1.02 * sum(loads) / n, (step, n, loads)


def test_layout_matches_generated_ctgs():
    from gams_amd import synth

    lengths = [45_000_000, 2_000_000, 30_000]
    real = [len(c["seq"]) for c in synth.genome_ctgs(lengths, 1000000, first_chr_index=3)]
    assert real == synth.layout_ctg_lengths(lengths, 1000000)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_strong_shards_are_a_partition(world):
    """bench.py --workload GRCh38-step10 at a small scale: the ranks' batches are disjoint, cover the
    genome's layout exactly, and every rank generated only the ctgs it owns."""
    import bench
    from gams_amd import synth

    lengths = [max(20000, int(x * 0.004)) for x in synth.GRCH38_LENGTHS]
    lay = synth.layout_ctgs(lengths, 1000000)
    seen = {}
    for rank in range(world):
        batches, prm, genome_bp, scaling, note = bench.build_batches("GRCh38-step10", rank, world, 3, 0.004)
        assert scaling == "strong" and len(batches) == 1 and prm["step"] == 10 and genome_bp == sum(lengths)
        assert f"generated its own {len(batches[0])} ctgs only" in note
        for c in batches[0]:
            key = (int(c["chr_id"]), c["chr_start"], c["chr_end"])
            assert key not in seen
            seen[key] = rank
            assert len(c["seq"]) == c["chr_end"] - c["chr_start"] + 1
    assert sorted(seen) == sorted((k, s, e) for k, _, s, e in lay)
    w = [(e - s + 1 - 100) // 10 + 1 for _, _, s, e in lay]
    owner = shard.lpt_assign(w, world)
    assert [seen[(k, s, e)] for k, _, s, e in lay] == owner


def test_per_ctg_generation_is_rank_independent_and_gen_gives_back_the_layout():
    """synth.ctg_bases is a function of (seed, chromosome, start): the ctgs a rank of a 3-rank job
    generates equal the same ctgs of the 1-rank job; and `gen` (gen.rs:81-126 as restated in
    synth.gen_ctgs) over the assembled chromosomes -- ctgs + the planted 10-kb N runs -- cuts exactly
    the closed-form layout ownership was decided on."""
    import numpy as np

    from gams_amd import synth

    lengths = [45_000_000, 2_000_000, 30_000]
    every, loads1, n = synth.sharded_genome_ctgs(lengths, 1000000, 0, 1)
    assert n == len(every) == 44 and len(loads1) == 1
    part, loads3, _ = synth.sharded_genome_ctgs(lengths, 1000000, 1, 3)
    assert sum(loads3) == loads1[0] and 0 < len(part) < n
    by_key = {(c["chr_id"], c["chr_start"]): c for c in every}
    for c in part:
        assert np.array_equal(c["seq"], by_key[(c["chr_id"], c["chr_start"])]["seq"])
    lay = synth.layout_ctgs(lengths, 1000000)
    for k, length in enumerate(lengths):
        chrom = np.full(length, 0x4E, np.uint8)
        for c in every:
            if c["chr_id"] == str(k + 1):
                chrom[c["chr_start"] - 1:c["chr_end"]] = c["seq"]
        got = [(c["chr_start"], c["chr_end"]) for c in synth.gen_ctgs(str(k + 1), chrom, piece=1000000)]
        assert got == [(s, e) for kk, _, s, e in lay if kk == k + 1]
    # composition: GC share and soft-masked share like synth.chromosome
    a = every[3]["seq"]
    up = a & 0xDF
    assert 0.34 < float(((up == 0x47) | (up == 0x43)).mean()) < 0.42
    assert 0.1 < float(((a & 0x20) != 0).mean()) < 0.3
    assert 0 < int((a == 0x4E).sum()) < 2000


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py starts the two ranks itself (children of a
    parent that has not imported torch), they join over gloo (no GPU here), run the bench's barrier and its
    MAX / SUM reductions, and stdout is exactly rank 0's one JSON line."""
    import json

    res = _bench(["--gpus", "2", "--rendezvous-only"])
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out == {"rendezvous": "ok", "n_gpus": 2, "backend": "gloo", "max_rank_sum_ranks": [1.0, 2.0],
                   "devices_visible": out["devices_visible"]}


def test_bench_reports_a_failed_rank():
    res = _bench(["--gpus", "2", "--rendezvous-only"], {"GAMS_BENCH_BACKEND": "no-such-backend"})
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    res = _bench(["--gpus", "2", "--rendezvous-only"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr


def test_bench_weak_deals_whole_genomes():
    import bench

    b0, prm, bp, scaling, note = bench.build_batches("S288c", 1, 2, 2, 0.02)
    assert scaling == "weak" and len(b0) == 2
    ids = {c["chr_id"] for b in b0 for c in b}
    b1, *_ = bench.build_batches("S288c", 0, 2, 2, 0.02)
    assert ids.isdisjoint({c["chr_id"] for b in b1 for c in b})     # different genomes (chromosome seeds)
