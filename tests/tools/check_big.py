"""Large-batch self-consistency: dense MFMA path vs CSR path on G puzzles of 900 pieces."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from diffassemble_amd import DenoiserEngine
from oracle import weights as W
dev = torch.device("cuda:0")
G = int(os.environ.get("G", 32)); n = 900
prec = os.environ.get("PREC", "bf16")
sd = W.make_denoiser_state(100, 4, 4, seed=0)
eng = DenoiserEngine(sd, precision=prec, device=dev)
gen = torch.Generator(device=dev).manual_seed(1234)
feats = torch.randn((G * n, 1088), generator=gen, device=dev)
x = torch.randn((G * n, 4), generator=gen, device=dev)
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1)
batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = eng.plan(ei, batch)
for rep in range(3):
    out = eng.forward(plan, x, 57, feats)
    torch.cuda.synchronize()
    bad = ~torch.isfinite(out)
    print("rep", rep, "dense finite:", not bool(bad.any()), "bad rows:", bad.any(1).nonzero().flatten()[:10].tolist(), "count", int(bad.any(1).sum()))
os.environ["DA_DISABLE_MFMA"] = "0"
out_csr, _ = eng.forward(plan, x, 57, feats, return_alpha=True) if G <= 4 else (None, None)
if out_csr is not None:
    print("max |dense - csr|", float((out - out_csr).abs().max()))
# per-graph comparison: graph 0 alone
p1 = eng.plan(ei[:, : n * n], batch[:n])
o1 = eng.forward(p1, x[:n], 57, feats[:n])
print("graph0 alone vs in-batch:", float((o1 - out[:n]).abs().max()))
# sampling loop: eager vs hipGraph, finite-ness per step
from diffassemble_amd import Schedule, _lib
from oracle import diffusion as ODF
sch = Schedule(ODF.make_schedule(100), dev)
for use_graph in (False, True, True):
    traj, xf = eng.sample_loop(plan, sch, x, feats, ratio=1, mean_type=_lib.MEAN_START_X, max_iters=20, keep_trajectory=True, use_graph=use_graph)
    torch.cuda.synchronize()
    fin = [bool(torch.isfinite(traj[k]).all()) for k in range(20)]
    print("graph" if use_graph else "eager", "finite per step:", fin, "absmax", [round(float(traj[k].abs().max()), 2) for k in (0, 5, 10, 19)])
