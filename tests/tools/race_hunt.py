"""Repeat one forward many times on identical inputs; count runs whose output differs / is non-finite."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffassemble_amd import DenoiserEngine
from oracle import weights as W
dev = torch.device("cuda:0")
G = int(os.environ.get("G", 32)); n = 900; reps = int(os.environ.get("REPS", 100))
prec = os.environ.get("PREC", "bf16")
sd = W.make_denoiser_state(100, 4, 4, seed=0)
eng = DenoiserEngine(sd, precision=prec, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn((G * n, 1088), generator=gen, device=dev)
x = torch.randn((G * n, 4), generator=gen, device=dev)
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1)
batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = eng.plan(ei, batch); del ei
ref = eng.forward(plan, x, 57, feats).clone()
nbad = ndiff = 0; first = None
for k in range(reps):
    out = eng.forward(plan, x, 57, None)
    if not torch.isfinite(out).all(): nbad += 1
    d = (out != ref).any(1)
    if d.any():
        ndiff += 1
        if first is None: first = (k, d.nonzero().flatten()[:8].tolist(), int(d.sum()))
print(f"G={G} prec={prec} MFMA_off={os.environ.get('DA_DISABLE_MFMA','0')}: {reps} reps, non-finite {nbad}, differing {ndiff}, first {first}")
