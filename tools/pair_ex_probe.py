"""Whole-loop A/B of the two-branch loop for the reference's other samplers (da_sample_loop_pair_ex): 64 puzzles of 900
pieces, T = 100, bf16; DA_TWO_BRANCH=0 / 1 in one process, three rounds.  Synthetic weights / inputs (oracle/weights.py is
only used as the generator of random tensors here, as in bench.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import weights as W, diffusion as ODF
from diffassemble_amd import DenoiserEngine, Schedule, _lib

dev = torch.device("cuda:0")
G, n = 64, 900
sd = W.make_denoiser_state(100, 4, 4, seed=1)
x, feats = W.make_inputs(G * n, 4, 1088, 1)
ei, batch = W.collate([W.dense_edge_index(n, True)] * G, [n] * G)
eng = DenoiserEngine(sd, precision="bf16", device=dev)
plan = eng.plan(ei.to(dev), batch.to(dev))
sch = Schedule(ODF.make_schedule(100), dev)
xd, fd = x.to(dev), feats.to(dev)
for name, kw in (("cfg w=1.5", dict(cfg_w=1.5)), ("ddim eta=0.5", dict(eta=0.5)), ("ddpm", dict(sampler="DDPM", mean_type=_lib.MEAN_EPSILON))):
    kw = dict(dict(ratio=1, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True), **kw)
    for rnd in range(3):
        for tb in ("0", "1"):
            os.environ["DA_TWO_BRANCH"] = tb
            eng.sample_loop(plan, sch, xd, fd, **kw)              # capture / warm
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.sample_loop(plan, sch, xd, fd, restage=False, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            print(f"{name:14s} two_branch={tb} loop {ms:8.2f} ms  = {ms / 100:.4f} ms per step", flush=True)
