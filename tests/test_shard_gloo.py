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
    genome, and rank loads are what lpt_assign gives."""
    import bench

    seen, total_bp, n_ctgs = set(), None, 0
    for rank in range(world):
        batches, prm, genome_bp, scaling, note = bench.build_batches("GRCh38-step10", rank, world, 3, 0.004)
        assert scaling == "strong" and len(batches) == 1 and prm["step"] == 10
        for c in batches[0]:
            key = (c["chr_id"], c["chr_start"], c["chr_end"])
            assert key not in seen
            seen.add(key)
        n_ctgs += len(batches[0])
        total_bp = genome_bp
    assert sum(e - s + 1 for _, s, e in seen) <= total_bp
    # every ctg of the genome is somewhere
    from gams_amd import synth
    lengths = [max(20000, int(x * 0.004)) for x in synth.GRCH38_LENGTHS]
    assert n_ctgs == len(synth.genome_ctgs(lengths, 1000000))


def test_bench_weak_deals_whole_genomes():
    import bench

    b0, prm, bp, scaling, note = bench.build_batches("S288c", 1, 2, 2, 0.02)
    assert scaling == "weak" and len(b0) == 2
    ids = {c["chr_id"] for b in b0 for c in b}
    b1, *_ = bench.build_batches("S288c", 0, 2, 2, 0.02)
    assert ids.isdisjoint({c["chr_id"] for b in b1 for c in b})     # different genomes (chromosome seeds)
