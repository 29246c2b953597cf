"""RCCL on the one GPU the test box has: a ONE-rank "nccl" process group executes the collectives of the data-parallel
training path (BASELINE config 5; the reference gets them from Lightning's NCCL DDP, train_script.py:215-218) -- the branch
every multi-rank gloo test of this suite cannot reach (ranks sharing a GPU cannot use RCCL).  Own processes: a default
process group must not leak into the other tests."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "tools", "rccl_worker.py")


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _worker(mode, tmp_path):
    dump = str(tmp_path / f"{mode}.pt")
    r = subprocess.run([sys.executable, WORKER, mode, str(_port()), dump], env=_env(), capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line), torch.load(dump)


@pytest.mark.gpu
def test_one_rank_rccl_group_runs_the_gradient_exchange(tmp_path):
    """Three optimizer steps of two accumulated micro-batches each under a one-rank RCCL group -- bucketed / overlapped exchange
    and serial exchange -- against the same steps with no process group: an all-reduce over one rank must change nothing."""
    nccl, dn = _worker("nccl", tmp_path)
    ref, dr = _worker("nodist", tmp_path)
    assert nccl["backend"] == "nccl" and nccl["exchange_active"] and not ref["exchange_active"]
    assert nccl["allreduce_identity"]
    # the early bucket really went through the side stream in the overlapped runs, and never in the serial ones
    assert all(nccl["early_pending_seen_overlap_1"]) and not any(nccl["early_pending_seen_overlap_0"])
    assert 0 < nccl["bucket_split"][0] < nccl["bucket_split"][1]
    # equal up to the summation order of the time_emb gradient's atomics (k_time_scatter): 1e-6 of the buffer's max-abs
    assert nccl["overlap_vs_serial_grad"] < 1e-6 and nccl["overlap_vs_serial_params"] < 1e-5
    assert nccl["grad_abs_sum_overlap_1"] > 0

    def reldiff(a, b):
        return float((a - b).abs().max() / b.abs().max())
    assert reldiff(dn["grad"][True], dr["grad"][False]) < 1e-6       # exchanged over one rank == not exchanged
    assert reldiff(dn["flat"][True], dr["flat"][False]) < 1e-5       # ... and so are the parameters after three fused Adafactor steps
    assert max(abs(a - b) for a, b in zip(nccl["losses_overlap_1"], ref["losses_overlap_0"])) < 1e-5


@pytest.mark.gpu
def test_bench_config_5_under_a_launcher_uses_rccl():
    """`torchrun --nproc-per-node 1 bench.py --config 5 --gpus 1`: bench.py's init_process_group("nccl", device_id=...) and the
    training line's exchange accounting, on RCCL."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--config", "5", "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--train-puzzles", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["distributed"] == {"world_size": 1, "backend": "nccl", "per_rank_ms_per_step": None}
    ex = line["gradient_exchange"]
    assert ex["overlapped"] and ex["exposed_ms"] >= 0 and ex["serial_ms"] > 0
    assert sum(ex["bucket_bytes"].values()) > 12e6        # the 12.9 MB flat gradient buffer, in two buckets
    assert line["value"] > 0
