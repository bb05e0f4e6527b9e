"""bench.py as the driver runs it: a subprocess, one JSON line, n_gpus equal to --gpus, roofline object present."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + list(flags), capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_single_rank_line():
    out = _run("--gpus", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline")
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["unit"] == "pairs/s" and out["scaling"] == "weak"
    assert out["value"] > 0 and out["ms_per_step"] > 0
    assert "from the raw minibatch" in out["config"]["workload"] and "32256" in out["config"]["workload"]
    roof = out["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["ms_per_launch"] < out["ms_per_step"]
    assert out["loss"] == out["loss"]          # finite, not NaN
    # whole-step efficiency on executed flops, and the box-size sensitivity sweep of the default invocation
    assert 0 < roof["step_frac"] < roof["frac"] + 0.2 and roof["step_frac"] < 1
    assert out["ranks_seen"] == 1 and len(out["rank_ms_per_step"]) == 1 and out["peak_memory_gb"] > 1
    sens = out["sensitivity"]
    assert [s["boxes"] for s in sens] == ["boxes x1.5", "boxes x2.5", "every box = full image"]
    fr = [out["shared_windows"]["fraction"]] + [s["pair_specific_fraction"] for s in sens]
    assert fr == sorted(fr) and fr[-1] == 1.0 and sens[-1]["path"].startswith("per-pair")
    assert all(s["ms_per_step"] > 0 and s["peak_memory_gb"] > 1 for s in sens)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)


def test_bench_launcher_refuses_more_ranks_than_gpus():
    """``--gpus N`` with fewer than N devices: the launcher says so instead of starting ranks that would share a GPU."""
    import torch
    n = torch.cuda.device_count()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


def test_bench_two_ranks_end_to_end_on_one_gpu_over_gloo():
    """The whole multi-rank bench path on hardware - launcher, rendezvous, per-rank synthetic shards, early fc1 reduction from the
    side stream + flat bucket, barrier, MAX-reduced time, one JSON line with n_gpus = 2 - with both ranks sharing the one GPU of
    the test box over gloo (test hook SGC_BENCH_SHARE_GPU; RCCL itself needs one GPU per rank and is the driver's to run)."""
    env = dict(os.environ, SGC_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--objects", "20",
                        "--images", "4", "--backend", "gloo", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert out["loss"] == out["loss"] and out["scaling"] == "weak"
