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
