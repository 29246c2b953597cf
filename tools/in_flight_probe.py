#!/usr/bin/env python3
"""How many independent Batches in flight fill the chip?  Dense 900-piece puzzles (the headline's kernels), G puzzles per Batch, N Batches in
flight through DenoiserEngine.sample_loop_batches, against the default pair loop on N * G puzzles.   python tools/in_flight_probe.py [G=32]"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from diffassemble_amd import _lib

dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = bench.CONFIGS["3p"]
n, K = cfg["n"], 20
model = bench.build_module(cfg, dev, "bf16")
eng = model.model.engine(dev)
sch = model._schedule()
gen = torch.Generator(device=dev).manual_seed(1)
mt = _lib.MEAN_START_X


def timed(fn, reps=15):
    fn(); fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


for nb in (2, 3, 4, 6):
    items = []
    for j in range(nb):
        ei, batch = bench.dense_batch(G, n, dev, loops=True)
        items.append((eng.plan(ei, batch), torch.randn((G * n, 4), generator=gen, device=dev), torch.randn((G * n, 1088), generator=gen, device=dev)))
        del ei, batch
    ps, xs, fs = [i[0] for i in items], [i[1] for i in items], [i[2] for i in items]
    eng.sample_loop_batches(ps, sch, xs, fs, ratio=1, mean_type=mt, max_iters=K, restage=True)
    dt = timed(lambda: eng.sample_loop_batches(ps, sch, xs, fs, ratio=1, mean_type=mt, max_iters=K, restage=False))
    print(f"{nb} Batches of {G} puzzles in flight: {dt / K * 1e3:.4f} ms per step of all = {nb * G * K / dt:,.0f} puzzle-steps/s", flush=True)
    del items, ps, xs, fs
    eng._batches_state = None
    torch.cuda.empty_cache()
# the default pair loop on 2 G puzzles
ei, batch = bench.dense_batch(2 * G, n, dev, loops=True)
plan = eng.plan(ei, batch)
x = torch.randn((2 * G * n, 4), generator=gen, device=dev); f = torch.randn((2 * G * n, 1088), generator=gen, device=dev)
eng.sample_loop(plan, sch, x, f, ratio=1, mean_type=mt, max_iters=K, keep_trajectory=False, use_graph=True, restage=True)
dt = timed(lambda: eng.sample_loop(plan, sch, x, f, ratio=1, mean_type=mt, max_iters=K, keep_trajectory=False, use_graph=True, restage=False))
print(f"default pair loop, {2 * G} puzzles: {dt / K * 1e3:.4f} ms per step = {2 * G * K / dt:,.0f} puzzle-steps/s")
