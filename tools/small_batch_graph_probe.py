#!/usr/bin/env python3
"""Small Batches sit at the launch floor: is the hipGraph replay of the loop (the default) or a plain stream of launches the cheaper way to issue
~11 dependent kernels per step?  The scripted ragged exophormer Batch of bench.py --config scripted, 30 DDIM steps, both ways, interleaved.
    python tools/small_batch_graph_probe.py [puzzles=8]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from diffassemble_amd import _lib, expander

dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(20)
sides = [int(v) for v in rng.choice(np.arange(6, 21, 2), size=G)]
cfg = dict(name="", variant="2d", arch="exophormer", V=8, n=None, graph="ragged_regular", rotation=True, T=300, ratio=10, mean="START_X", G=G, prec="bf16",
           N_total=sum(v * v for v in sides), pairs_total=sum(v ** 4 for v in sides))
model = bench.build_module(cfg, dev, "bf16")
eng = model.model.engine(dev)
ei, batch, degs = expander.ragged_regular_batch(sides, 60, rng, dev)
N = cfg["N_total"]
gen = torch.Generator(device=dev).manual_seed(77)
feats = torch.randn((N, 1088), generator=gen, device=dev)
x_T = torch.randn((N, 4), generator=gen, device=dev)
plan = eng.plan(ei, batch)
sch = model._schedule()
eng.set_features(plan, feats)
K = 30


def run(graph):
    return eng.sample_loop(plan, sch, x_T, feats, ratio=10, mean_type=_lib.MEAN_START_X, max_iters=K, keep_trajectory=False, use_graph=graph, restage=False)


def timed(graph, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        run(graph)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * K) * 1e3


a = run(True)[1]; b = run(False)[1]
print("same poses:", bool(torch.equal(a, b)))
tg, te = [], []
for i in range(8):
    for g in ((True, False) if i % 2 == 0 else (False, True)):
        (tg if g else te).append(timed(g))
print(f"{G} puzzles, {N} pieces: hipGraph replay {statistics.median(tg):.4f} ms per step, stream of launches {statistics.median(te):.4f} ms per step")
