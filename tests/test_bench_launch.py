"""bench.py --gpus N must run N ranks by itself: without a launcher around it the script re-runs itself
under torch.distributed.run, and it refuses to print a line for fewer GPUs than it was asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_plan_decisions():
    argv = ["--gpus", "4", "--steps", "3"]
    assert bench.launch_plan(1, {}, []) is None                                  # N = 1: run in place
    assert bench.launch_plan(4, {"WORLD_SIZE": "4"}, argv) is None              # already a rank of a launcher
    plan = bench.launch_plan(4, {}, argv, device_count=8)
    assert plan[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert plan[plan.index("--nproc-per-node") + 1] == "4" and plan[plan.index("--master-addr") + 1] == "127.0.0.1"
    assert plan[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and plan[-len(argv):] == argv
    with pytest.raises(SystemExit) as e:                                         # fewer GPUs than asked for: no line at all
        bench.launch_plan(8, {}, argv, device_count=1)
    assert "only 1 GPU(s) visible" in str(e.value)
    assert bench.launch_plan(2, {}, argv, share_gpu=True, device_count=1) is not None   # one-GPU test mode
    with pytest.raises(SystemExit):
        bench.launch_plan(0, {}, argv)


def test_gpus_without_devices_exits_nonzero_and_prints_no_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""   # (no GPU here anyway; on a GPU box this hides them)
    env["CUDA_VISIBLE_DEVICES"] = ""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "n_gpus" not in p.stdout and "GPU(s) visible" in p.stderr


def test_world_size_must_match_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and "n_gpus" not in p.stdout


@pytest.mark.gpu
def test_gpus_2_self_launches_two_ranks_on_one_gpu():
    """No torchrun around it: the line must say n_gpus 2 (gloo rendezvous, both ranks on cuda:0 — the one-GPU stand-in
    for the driver's `python bench.py --gpus N`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu",
                        "--workload", "mid", "--steps", "2", "--warmup", "1", "--configs", "none", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and len(line["rows_per_rank"]) == 2 and line["value"] > 0
