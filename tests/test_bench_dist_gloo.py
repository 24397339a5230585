"""bench.py's N>1 control flow under 2-process gloo on CPU (stub engine): shards per rank, barrier, MAX of the step
time over ranks, all-gather of the ParamNet scalars, rank-0-only JSON line -- so that the driver's first real multi-GPU
launch (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) cannot fail on plumbing."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_bench_rank_logic_under_gloo(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
           "--batch", "5", "--dry-run-cpu"]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 5 * world
    assert d["config"]["gathered_param_rows"] == 5 * world  # all-gather of the (B_local, 8) scalars
    # whole-job throughput over the MAX of the ranks' times: the slowest stub rank sleeps 2 ms x world per forward
    assert d["ms_per_step"] >= 2.0 * world * 0.9
    assert abs(d["value"] - 5 * world * 3 / (d["ms_per_step"] * 3e-3)) <= 0.02 * d["value"]
    assert "dry run" in d["data"] and "roofline" not in d and "cpu_baseline" not in d


def test_bench_single_process_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", "2", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["global_batch"] == 4 and d["config"]["gathered_param_rows"] == 4


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torchrun must run two ranks (the driver's SCALE command is `bench.py --gpus N`): the line says n_gpus == 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1", "--batch", "4"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 8 and d["config"]["gathered_param_rows"] == 8


def test_bench_refuses_mismatched_world():
    """--gpus 8 inside a 1-rank environment (WORLD_SIZE=1 set by a launcher) must not print an n_gpus = 1 line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-cpu", "--steps", "1", "--warmup", "0", "--batch", "2"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "does not match WORLD_SIZE" in r.stderr


@pytest.mark.parametrize("world", [1, 3])
def test_bench_mixed_workload_shards_by_bucket(world):
    """configs[4]: the mixed-resolution stream is sharded round-robin within each (H, W) bucket; ragged shards gather to the full stream; per-bucket figures are reported."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--dry-run-cpu", "--workload", "mixed", "--steps", "2", "--warmup", "1", "--batch", "6"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    n = 6 * world
    assert d["n_gpus"] == world and d["config"]["global_batch"] == n and d["config"]["gathered_param_rows"] == n
    pb = d["per_bucket"]
    assert set(pb) == {"384x512", "640x640", "1024x1365"}
    assert sum(v["images_per_step_all_ranks"] for v in pb.values()) == n
    assert pb["640x640"]["images_per_step_all_ranks"] == sum(1 for i in range(n) if i % 4 in (1, 2))
    assert abs(d["value"] - n * 2 / (d["ms_per_step"] * 2e-3)) <= 0.02 * d["value"]
