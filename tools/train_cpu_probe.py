#!/usr/bin/env python
"""Is the training step host-bound?  BASELINE config 5 shapes: the time the host needs to ENQUEUE 200 steps (no synchronisation
inside) against the time until the GPU has finished them.   python tools/train_cpu_probe.py [puzzles] [side]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 64
side = int(sys.argv[2]) if len(sys.argv) > 2 else 12
n = side * side
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = GNN_Diffusion(steps=300, sampling="DDIM", rotation=True, visual_pretrained=False, model_mean_type=ModelMeanType.EPSILON).to(dev).train()
opt = m.configure_optimizers()
gen = torch.Generator(device=dev).manual_seed(99)
feats = torch.randn((G * n, 1088), generator=gen, device=dev)
x0 = torch.randn((G * n, 4), generator=gen, device=dev)
idx = torch.arange(n, device=dev)
src, dst = torch.meshgrid(idx, idx, indexing="ij")
ei = torch.cat([torch.stack([src.reshape(-1), dst.reshape(-1)]) + g * n for g in range(G)], 1)
batch = torch.arange(G, device=dev).repeat_interleave(n)
te = m.model.train_engine(dev)
te.precision = os.environ.get("PREC", "bf16")


def step():
    t = torch.randint(0, 300, (G,), generator=gen, device=dev)[batch]
    opt.zero_grad()
    loss = m.p_losses(x0, t, loss_type="huber", cond=None, edge_index=ei, batch=batch, patch_feats=feats)
    loss.backward()
    m.sync_gradients()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3 * (t1 - t0) / K:.4f} ms per step; until the GPU is done {1e3 * (t2 - t0) / K:.4f} ms per step; "
      f"the GPU still had {1e3 * (t2 - t1):.2f} ms of work queued when the host finished")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(50):
    step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
