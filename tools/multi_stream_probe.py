import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
from oracle import weights as W, diffusion as ODF
from diffassemble_amd import DenoiserEngine, Schedule, _lib
dev = torch.device("cuda:0")
n, G = 900, int(sys.argv[1]); NE = int(sys.argv[2])     # puzzles per engine, engines
sd = W.make_denoiser_state(100, 4, 4, seed=5)
sch = Schedule(ODF.make_schedule(100), dev)
engs, plans, xs, fs, streams = [], [], [], [], []
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n); one = torch.stack([r, c])
for e in range(NE):
    eng = DenoiserEngine(sd, precision="bf16", device=dev)
    ei = torch.cat([one + g * n for g in range(G)], 1); batch = torch.arange(G, device=dev).repeat_interleave(n)
    plan = eng.plan(ei, batch); del ei
    gen = torch.Generator(device=dev).manual_seed(7 + e)
    f = torch.randn((G * n, 1088), generator=gen, device=dev); x = torch.randn((G * n, 4), generator=gen, device=dev)
    engs.append(eng); plans.append(plan); xs.append(x); fs.append(f); streams.append(torch.cuda.Stream(device=dev))
def run():
    for e in range(NE):
        with torch.cuda.stream(streams[e]):
            engs[e].sample_loop(plans[e], sch, xs[e], fs[e], ratio=1, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True, restage=False)
for e in range(NE):
    engs[e].set_features(plans[e], fs[e])
for _ in range(2): run()
torch.cuda.synchronize()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
ts.sort(); t = ts[len(ts) // 2]
print(f"{NE} engines x {G} puzzles (two_branch={os.environ.get('DA_TWO_BRANCH','auto')}): {NE * G * 100 / t:.0f} puzzle-steps/s, {t * 10:.4f} ms per step")
