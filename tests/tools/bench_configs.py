#!/usr/bin/env python
"""Per-configuration numbers of SURVEY 8(d) beside the headline bench: configs 1, 2, 3, 3', 4 on one GPU
(hipGraph DDIM loop, puzzle-steps/s, per-kernel-class times), the sparse path's achieved bandwidth
against the HBM roofline, and the CPU oracle timed on the host cores for the configurations the survey
lists (bounded samples).  One JSON line per configuration; the committed copy is
profiles/r01/configs_v2.jsonl.

  python tests/tools/bench_configs.py [--no-cpu] [--only NAME]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import diffusion as ODF  # noqa: E402
from oracle import weights as W  # noqa: E402

F_NODE_2D, F_NODE_3D = 6_432_128, 6_164_704
HBM_PEAK = 8000.0       # GB/s, MI355X_MICROARCH.md


def graphs(kind, n, G, seed):
    rng = np.random.default_rng(seed)
    if kind == "dense":
        eis = [W.dense_edge_index(n, True)] * G
    elif kind == "dense_noloop":
        eis = [W.dense_edge_index(n, False)] * G
    else:
        d = int(kind[len("regular"):])
        eis = [W.random_regular_edge_index(n, d, rng) for _ in range(G)]
    return W.collate(eis, [n] * G)


CONFIGS = [
    # name, variant, arch, V, n, graph, c, T, ratio, mean, Gs, precision
    dict(name="1_6x6_trans_T50", variant="2d", arch="transformer", V=0, n=36, graph="dense_noloop", c=2, T=50, ratio=1,
         mean="EPSILON", Gs=[1], prec="fp32", cpu_steps=25),
    dict(name="2_12x12_rot_dense", variant="2d", arch="transformer", V=0, n=144, graph="dense", c=4, T=100, ratio=1,
         mean="START_X", Gs=[1, 64, 512], prec="bf16", cpu_steps=8),
    dict(name="3_30x30_exphander_d90", variant="2d", arch="exophormer", V=8, n=900, graph="regular90", c=4, T=100, ratio=1,
         mean="START_X", Gs=[1, 32], prec="bf16", cpu_steps=3),
    dict(name="3_30x30_exphander_d539", variant="2d", arch="exophormer", V=8, n=900, graph="regular539", c=4, T=100, ratio=1,
         mean="START_X", Gs=[1, 32], prec="bf16", cpu_steps=1),
    dict(name="3p_30x30_dense", variant="2d", arch="transformer", V=0, n=900, graph="dense", c=4, T=100, ratio=1,
         mean="START_X", Gs=[1, 8, 32], prec="bf16", cpu_steps=0),          # CPU number: bench.py's cpu_baseline
    dict(name="4_3d_P20", variant="3d", arch="transformer", V=0, n=20, graph="dense", c=7, T=300, ratio=10,
         mean="START_X", Gs=[1, 256], prec="bf16", cpu_steps=10),
]


def make_state(cfg):
    if cfg["variant"] == "3d":
        return W.make_denoiser_state(cfg["T"], 7, 6, D=832, hidden=256, variant="3d", arch="transformer", virt_nodes=0, seed=0)
    return W.make_denoiser_state(cfg["T"], cfg["c"], cfg["c"], D=1152, hidden=128, variant="2d", arch=cfg["arch"],
                                 virt_nodes=cfg["V"], seed=0)


def init_pose(cfg, N, gen, dev):
    if cfg["variant"] == "3d":
        x = torch.zeros((N, 7), device=dev)
        x[:, 0] = 1.0
        x[:, 4:] = torch.randn((N, 3), generator=gen, device=dev)
        return x
    return torch.randn((N, cfg["c"]), generator=gen, device=dev)


def run_gpu(cfg, G, dev):
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    sd = make_state(cfg)
    eng = DenoiserEngine(sd, variant=cfg["variant"], arch=cfg["arch"], virt_nodes=cfg["V"], precision=cfg["prec"], device=dev)
    ei, batch = graphs(cfg["graph"], cfg["n"], G, 3)
    N = cfg["n"] * G
    F = 768 if cfg["variant"] == "3d" else 1088
    gen = torch.Generator(device=dev).manual_seed(11)
    feats = torch.randn((N, F), generator=gen, device=dev)
    x = init_pose(cfg, N, gen, dev)
    plan = eng.plan(ei.to(dev), batch.to(dev))
    E = plan.n_edges
    sch = Schedule(ODF.make_schedule(cfg["T"]), dev)
    mt = _lib.MEAN_START_X if cfg["mean"] == "START_X" else _lib.MEAN_EPSILON
    iters = (cfg["T"] + cfg["ratio"] - 1) // cfg["ratio"]

    def loop(graph):
        return eng.sample_loop(plan, sch, x, feats, ratio=cfg["ratio"], mean_type=mt, keep_trajectory=False,
                               use_graph=graph, restage=False)
    eng.set_features(plan, feats)
    loop(True)
    torch.cuda.synchronize()
    reps = max(1, int(2000 // iters))
    t0 = time.perf_counter()
    for _ in range(reps):
        _, xf = loop(True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (reps * iters)
    assert torch.isfinite(xf).all()
    eng.profile(True)
    loop(False)
    prof = eng.profile_read()
    eng.profile(False)
    kern = {k: round(ms / n * 1e3, 2) for k, (ms, n) in prof.items() if n}
    rec = {"config": cfg["name"], "G": G, "N": N, "E": int(E), "precision": cfg["prec"], "dense_path": bool(plan.dense),
           "puzzle_steps_per_s": G / dt, "ms_per_batch_step": dt * 1e3, "kernels_us": kern}
    f_node = F_NODE_3D if cfg["variant"] == "3d" else F_NODE_2D
    hcs = [256, 256, 256, 832 if cfg["variant"] == "3d" else 1152]
    f_edge = 4 * sum(hcs)
    rec["algorithmic_tflops"] = (N * f_node + E * f_edge) / dt / 1e12
    if not plan.dense:
        # sparse path roofline: bytes per edge per layer = 2 * H*C * s (K row + V row) + 4 (index),
        # per node 2 * H*C * s (q + out); the last layer (H*C = D) carries 60 % of it
        s = 2 if cfg["prec"] == "bf16" else 4
        nn = plan.n_nodes
        b_last = E * (2 * hcs[3] * s + 4) + nn * 2 * hcs[3] * s
        b_hid = E * (2 * 256 * s + 4) + nn * 2 * 256 * s
        rec["sparse_roofline"] = {
            "attn_last": {"bytes": b_last, "us": kern.get("attn_last"), "GBps": b_last / (kern["attn_last"] * 1e-6) / 1e9,
                          "frac_of_hbm_peak": b_last / (kern["attn_last"] * 1e-6) / 1e9 / HBM_PEAK},
            "attn_hidden": {"bytes": b_hid, "us": kern.get("attn_hidden"), "GBps": b_hid / (kern["attn_hidden"] * 1e-6) / 1e9,
                            "frac_of_hbm_peak": b_hid / (kern["attn_hidden"] * 1e-6) / 1e9 / HBM_PEAK},
            "note": "algorithmic gather bytes / kernel time; at G=1 K/V sit in L2 / Infinity Cache, so this is not HBM traffic there",
        }
    return rec


def run_cpu(cfg, threads):
    """The oracle on the host cores, G = 1, `cpu_steps` DDIM iterations.  Small puzzles use 16 threads:
    torch with one thread per core of a 256-core host is pathologically slow on 36 x 1152 tensors
    (measured 0.09 steps/s with 256 threads)."""
    if not cfg["cpu_steps"]:
        return None
    threads = threads if cfg["n"] >= 900 else min(threads, 16)
    torch.set_num_threads(threads)
    sd = make_state(cfg)
    ei, batch = graphs(cfg["graph"], cfg["n"], 1, 3)
    N = cfg["n"]
    F = 768 if cfg["variant"] == "3d" else 1088
    g = torch.Generator().manual_seed(11)
    feats = torch.randn((N, F), generator=g)
    sch = ODF.make_schedule(cfg["T"])
    k = cfg["cpu_steps"]
    t0 = time.perf_counter()
    if cfg["variant"] == "3d":
        x = torch.zeros((N, 7))
        x[:, 0] = 1.0
        x[:, 4:] = torch.randn((N, 3), generator=g)
        ODF.p_sample_loop_3d(sd, sch, x, ei, feats, batch, cfg["T"], cfg["ratio"], cfg["mean"], max_iters=k)
    else:
        x = torch.randn((N, cfg["c"]), generator=g)
        ODF.p_sample_loop(sd, sch, x, ei, feats, batch, cfg["T"], cfg["ratio"], cfg["mean"], cfg["arch"], cfg["V"], max_iters=k)
    dt = time.perf_counter() - t0
    return {"puzzle_steps_per_s": k / dt, "cores": threads, "kind": "port", "sample": f"{k} DDIM steps, G=1, oracle fp32, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--cpu-only", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for cfg in CONFIGS:
        if args.only and args.only not in cfg["name"]:
            continue
        cpu = None if args.no_cpu else run_cpu(cfg, os.cpu_count() or 1)
        if args.cpu_only:
            print(json.dumps({"config": cfg["name"], "cpu_baseline": cpu}), flush=True)
            continue
        for G in cfg["Gs"]:
            rec = run_gpu(cfg, G, dev)
            if G == 1:
                rec["cpu_baseline"] = cpu
            print(json.dumps(rec), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
