"""Determinism stress of the hybrid path (masked MFMA attention + virtual-row kernel on a side stream, fork / join
captured into the loop's hipGraph): repeated forwards and repeated graph replays must be bit-identical, and equal to
the single-stream run (DA_DISABLE_HYBRID_OVERLAP=1 in a second process is compared through a checksum)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffassemble_amd import DenoiserEngine, Schedule, _lib
from oracle import diffusion as ODF
from oracle import weights as W

dev = torch.device("cuda:0")
G, n, V = int(os.environ.get("G", 8)), 900, 8
sd = W.make_denoiser_state(100, 4, 4, arch="exophormer", virt_nodes=V, seed=0)
eng = DenoiserEngine(sd, arch="exophormer", virt_nodes=V, precision="bf16", device=dev)
e1 = W.random_regular_edge_index(n, 90, np.random.default_rng(0))
ei = torch.cat([e1 + g * n for g in range(G)], 1).to(dev)
batch = torch.arange(G, device=dev).repeat_interleave(n)
gen = torch.Generator(device=dev).manual_seed(7)
feats = torch.randn((G * n, 1088), generator=gen, device=dev)
x = torch.randn((G * n, 4), generator=gen, device=dev)
plan = eng.plan(ei, batch)
assert plan.hybrid == 1
sch = Schedule(ODF.make_schedule(100), dev)
ref = eng.forward(plan, x, 50, feats).clone()
bad = sum(int(not torch.equal(eng.forward(plan, x, 50), ref)) for _ in range(100))
print(f"forward: {bad}/100 differ")
_, xf = eng.sample_loop(plan, sch, x, feats, keep_trajectory=False, use_graph=True)
rf = xf.clone()
bad2 = 0
for _ in range(20):
    _, xf = eng.sample_loop(plan, sch, x, feats, keep_trajectory=False, use_graph=True)
    bad2 += int(not torch.equal(xf, rf))
print(f"graph loop: {bad2}/20 differ; checksum {float(rf.double().sum()):.10f} finite {bool(torch.isfinite(rf).all())}")
sys.exit(1 if bad or bad2 else 0)
