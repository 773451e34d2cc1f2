"""The N > 1 entry paths of bench.py on the one-GPU box: the 8-GPU scaling run is the
driver's, so what can be shown here is that every way of starting it reaches the timed region and
prints one JSON line -- bench.py launching its own ranks, the driver's torch.distributed.run form, both
scaling modes, and the RCCL (`nccl`) branch executed once with a single rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "6", "--warmup", "2", "--ramp-ms", "5", "--no-cpu", "--no-extra", "--no-secondary", "--no-e2e"]


def _env(extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GAMS_BENCH_BACKEND",
              "GAMS_BENCH_FORCE_DIST"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(extra or {})
    return env


def _one_json_line(res):
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def _bench(args, extra_env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(extra_env),
                          capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("workload,scale,scaling", [("Atha", "0.05", "weak"), ("GRCh38-step10", "0.01", "strong")])
def test_bench_self_launch_two_ranks_on_one_gpu(workload, scale, scaling):
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE): two ranks share the one device, so the
    barrier and the reductions go over gloo (chosen by bench.py because devices < ranks)."""
    out = _one_json_line(_bench(["--gpus", "2", "--workload", workload, "--scale", scale] + QUICK))
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["steps"] == 6
    assert out["value"] > 0 and out["unit"] == "windows/s"
    per_step = out["config"]["windows_per_step"]
    assert all(w > 0 for w in per_step)
    if scaling == "strong":
        assert "LPT-sharded x2" in out["config"]["sharding"]


def test_bench_one_rank_through_rccl():
    """The nccl branch -- init_process_group('nccl', device_id=...), barrier, all_reduce MAX / SUM on
    cuda tensors -- executed on an MI355X with one rank, started the way the driver starts N ranks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--workload", "Atha", "--scale", "0.05"] + QUICK
    res = subprocess.run(cmd, env=_env({"GAMS_BENCH_FORCE_DIST": "1", "GAMS_BENCH_BACKEND": "nccl"}),
                         capture_output=True, text=True, timeout=900)
    out = _one_json_line(res)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["collectives"].startswith("nccl")


def test_bench_n1_launcher_path_equals_direct_path():
    """N = 1 through torch.distributed.run does the same work as `python bench.py`: same windows per step,
    same peaks, and a rate in the same range (both time 40 steps of a 6-Mb genome)."""
    args = ["--gpus", "1", "--workload", "Atha", "--scale", "0.05", "--steps", "40", "--warmup", "5",
            "--no-cpu", "--no-extra", "--no-secondary"]
    direct = _one_json_line(_bench(args))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29548", os.path.join(ROOT, "bench.py")] + args
    launched = _one_json_line(subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900))
    assert launched["config"]["windows_per_step"] == direct["config"]["windows_per_step"]
    assert launched["config"]["peaks_per_step"] == direct["config"]["peaks_per_step"]
    assert 0.5 < launched["value"] / direct["value"] < 2.0
