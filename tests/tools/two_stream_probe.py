"""Does co-scheduling two half Batches on two HIP streams (the projection of one under the attention of the other) beat one
Batch of twice the size?  Two DenoiserEngines with the same weights, G puzzles each, hipGraph replays launched back to
back on two streams, against one engine with 2 G puzzles."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench as B
from diffassemble_amd import _lib
dev = torch.device("cuda:0")
cfg = B.CONFIGS["3p"]
G = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = B.build_module(cfg, dev, "bf16")
sd = model.model._denoiser_state()
from diffassemble_amd.engine import DenoiserEngine
def mk(Gx, seed):
    eng = model.model.engine(dev) if seed == 0 else DenoiserEngine({k: v.clone() for k, v in sd.items()}, variant="2d", arch="transformer", precision="bf16", device=dev) 
    ei, batch = B.dense_batch(Gx, 900, dev)
    plan = eng.plan(ei, batch)
    gen = torch.Generator(device=dev).manual_seed(seed + 7)
    feats = torch.randn((Gx * 900, 1088), generator=gen, device=dev)
    x = torch.randn((Gx * 900, 4), generator=gen, device=dev)
    return eng, plan, feats, x
sch = model._schedule()
def run(eng, plan, feats, x):
    return eng.sample_loop(plan, sch, x, feats, ratio=1, mean_type=_lib.MEAN_START_X, max_iters=100, keep_trajectory=False, use_graph=True, restage=False)
big = mk(2 * G, 0)
big[0].set_features(big[1], big[2]); run(*big); run(*big)
torch.cuda.synchronize(); t = time.time()
for _ in range(5): run(*big)
torch.cuda.synchronize(); t_big = (time.time() - t) / 5
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
engs = [mk(G, 1 + i) for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
for e, s in zip(engs, streams):
    with torch.cuda.stream(s):
        e[0].set_features(e[1], e[2]); run(*e); run(*e)
torch.cuda.synchronize(); t = time.time()
for _ in range(5):
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s): run(*e)
torch.cuda.synchronize(); t_two = (time.time() - t) / 5
print(f"one Batch of {2*G}: {t_big*1e3:.2f} ms per 100 steps ({2*G*100/t_big:.0f} puzzle-steps/s); {NS} x {G} on {NS} streams: {t_two*1e3:.2f} ms ({NS*G*100/t_two:.0f})")
