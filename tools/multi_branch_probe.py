"""Would MORE than two branches pay?  B independent one-branch loops (DA_TWO_BRANCH=0: one hipGraph each, own engine, own workspace)
of 64 / B puzzles of 900 pieces on B torch streams, launched back to back and timed together -- B = 1, 2 (what the pair loop does
inside one graph), 4, 8 -- against the pair loop at 64 puzzles.  usage: python tools/multi_branch_probe.py"""
import os, sys, time
os.environ["DA_TWO_BRANCH"] = os.environ.get("DA_TWO_BRANCH", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as B
from diffassemble_amd import _lib
dev = torch.device("cuda:0")
cfg = B.CONFIGS["3p"]
n, T = cfg["n"], cfg["T"]
total = int(os.environ.get("TOTAL", 64))
for nb in (1, 2, 4, 8):
    G = total // nb
    loops = []
    for b in range(nb):
        model = B.build_module(cfg, dev, "bf16")
        eng = model.model.engine(dev)
        gen = torch.Generator(device=dev).manual_seed(1234 + b)
        feats = torch.randn((G * n, 1088), generator=gen, device=dev)
        x_T = torch.randn((G * n, 4), generator=gen, device=dev)
        ei, batch = B.dense_batch(G, n, dev, loops=True)
        plan = eng.plan(ei, batch)
        sch = model._schedule()
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            eng.set_features(plan, feats)
            eng.sample_loop(plan, sch, x_T, feats, ratio=1, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True, restage=False)
        loops.append((model, eng, plan, sch, x_T, feats, st))
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for (model, eng, plan, sch, x_T, feats, st) in loops:
            with torch.cuda.stream(st):
                eng.sample_loop(plan, sch, x_T, feats, ratio=1, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True, restage=False)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"{nb} branch(es) x {G} puzzles: {best / T * 1e3:.4f} ms per step of {total} puzzles = {total * T / best:,.0f} puzzle-steps/s", flush=True)
    del loops
    torch.cuda.empty_cache()
