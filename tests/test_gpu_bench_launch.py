"""``python bench.py --gpus N`` the way the driver calls it (no launcher around it): the script starts its own N ranks
(the reference's Trainer spawns its own too, train_script.py:215-218).  On a 1-GPU box the ranks share cuda:0 over gloo
(--dist-backend gloo; the numbers of such a run mean nothing, the code path is what is checked)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, timeout=900):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv, "--dist-backend", "gloo"], env=env, capture_output=True, text=True,
                       cwd=ROOT, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly ONE JSON line expected (rank 0), got {len(lines)}: {r.stdout[-2000:]}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_gpus_2_self_launches_sampling():
    line = _bench("--gpus", "2", "--steps", "5", "--warmup", "2", "--puzzles", "2", "--replays", "2", "--no-cpu-baseline")
    assert line["n_gpus"] == 2 and line["distributed"]["world_size"] == 2 and line["distributed"]["backend"] == "gloo"
    assert len(line["distributed"]["per_rank_ms_per_step"]) == 2
    assert line["config"]["global_puzzles"] == 4 and line["steps"] == 5 and line["value"] > 0
    assert line["metric"].startswith("denoising steps/sec (900-piece dense graph, T=100)")
    # value = whole-job puzzles x steps / median pass time
    assert abs(line["value"] - 4 * 5 / (line["ms_per_step"] * 5e-3)) / line["value"] < 1e-6


@pytest.mark.gpu
def test_bench_gpus_2_self_launches_training():
    line = _bench("--gpus", "2", "--config", "5", "--steps", "3", "--warmup", "1", "--train-puzzles", "4")
    assert line["n_gpus"] == 2 and line["distributed"]["world_size"] == 2
    assert line["config"]["global_puzzles"] == 8 and line["value"] > 0


def test_bench_gpus_n_without_enough_gpus_says_so():
    """No GPU (this container) or fewer than N: a clear message, not an assertion about WORLD_SIZE."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("enough GPUs here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and "AssertionError" not in r.stderr
