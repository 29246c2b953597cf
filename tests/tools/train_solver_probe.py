#!/usr/bin/env python
"""Exploration helper for tests/test_gpu_benched_mode.py::test_end_metric_*: how long / how the GPU training path has to
run before the denoiser solves synthetic puzzles (features carry the true pose).  Prints accuracy per setting."""
import math
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import test_gpu_benched_mode as T  # noqa: E402


def evaluate(m, dev, side, G, gen, prec="fp32"):
    from diffassemble_amd.engine import greedy_assign
    n = side * side
    x_gt, feats, ei, batch = T._puzzle_batch(side, G, gen, T.FEAT_NOISE)
    y = torch.linspace(-1, 1, side)
    grid = torch.stack(torch.meshgrid(y, y, indexing="xy"), -1).reshape(-1, 2).repeat(G, 1).to(dev)
    ptr = torch.arange(0, (G + 1) * n, n, dtype=torch.int32, device=dev)
    m.model.precision = prec
    imgs, _ = m.p_sample_loop(tuple(x_gt.shape), None, ei.to(dev), batch.to(dev), patch_feats=feats.to(dev))
    img = imgs[-1]
    off = torch.arange(G, device=dev).repeat_interleave(n) * n
    ass = greedy_assign(img[:, :2].contiguous(), grid, ptr, ptr)
    cells = torch.empty(G * n, dtype=torch.int64, device=dev)
    cells[ass[:, 0] + off] = ass[:, 1]
    gt = greedy_assign(x_gt[:, :2].contiguous().to(dev), grid, ptr, ptr)
    gtc = torch.empty(G * n, dtype=torch.int64, device=dev)
    gtc[gt[:, 0] + off] = gt[:, 1]
    rot = torch.cosine_similarity(img[:, 2:], x_gt[:, 2:].to(dev)) > math.cos(math.pi / 4)
    err = float((img[:, :2] - x_gt[:, :2].to(dev)).abs().max())
    return float(((cells == gtc) & rot).float().mean()), err


dev = torch.device("cuda:0")
for noise in (0.1,):
    for steps, lr in ((1500, 2e-3), (2500, 2e-3)):
        T.FEAT_NOISE = noise
        t0 = time.time()
        m, loss = T._train_solver(dev, [6, 12, 12, 16], steps=steps, lr=lr)
        torch.cuda.synchronize()
        dt = time.time() - t0
        gen = torch.Generator().manual_seed(7)
        a12, e12 = evaluate(m, dev, 12, 4, gen)
        a30, e30 = evaluate(m, dev, 30, 2, gen)
        print(f"feat_noise {noise} steps {steps} lr {lr}: train {dt:.1f}s loss {loss:.2e}  acc12 {a12:.3f} (max err {e12:.3f})  "
              f"acc30 {a30:.3f} (max err {e30:.3f}; half cell 12: {1/11:.3f} 30: {1/29:.3f})", flush=True)
