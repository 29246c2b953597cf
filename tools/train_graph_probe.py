#!/usr/bin/env python
"""What does a hipGraph of the training step's forward + backward buy?  BASELINE config 5 shapes (64 puzzles of 12x12, bf16-operand
mode): the eager step (p_losses + backward, as bench.py --config 5 runs it) against ONE captured graph of the same calls replayed,
and the same with the optimizer step outside the graph.   python tools/train_graph_probe.py [puzzles] [side]"""
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
t_static = torch.randint(0, 300, (G,), generator=gen, device=dev)[batch]


def fb():
    opt.zero_grad(set_to_none=False)
    loss = m.p_losses(x0, t_static, loss_type="huber", cond=None, edge_index=ei, batch=batch, patch_feats=feats)
    loss.backward()
    return loss


def timeit(f, k=100):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def eager_step():
    t_static.copy_(torch.randint(0, 300, (G,), generator=gen, device=dev)[batch])
    fb()
    opt.step()


print(f"eager forward+backward            {timeit(fb):.4f} ms")
print(f"eager step (t draw, f+b, Adafactor) {timeit(eager_step):.4f} ms")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss_static = fb()
torch.cuda.synchronize()
print(f"graph forward+backward            {timeit(graph.replay):.4f} ms")


def graph_step():
    t_static.copy_(torch.randint(0, 300, (G,), generator=gen, device=dev)[batch])
    graph.replay()
    opt.step()


print(f"graph step (t draw, replay, Adafactor) {timeit(graph_step):.4f} ms   loss {float(loss_static):.5f}")
