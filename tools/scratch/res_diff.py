import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_attn_resident as T
res = {}
for tag, val in (("ring", "0"), ("resident", "1")):
    f = f"/tmp/{tag}.pt"
    env = dict(os.environ, DA_ATTN_RES=val)
    r = subprocess.run([sys.executable, "-c", T._DUMP.format(root=ROOT, tests=os.path.join(ROOT, "tests")), f], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res[tag] = torch.load(f)
for (a, la), (b, lb) in zip(res["ring"], res["resident"]):
    a = a.float(); b = b.float()
    d = (a - b).abs()
    nz = (d > 0).nonzero()
    print("launches", la, lb, "shape", tuple(a.shape), "differing", int((d > 0).sum()), "max", float(d.max()), "max|a|", float(a.abs().max()))
    if len(nz):
        rows = nz[:, 0].unique()
        print("  rows differing:", len(rows), "first", rows[:20].tolist(), "cols of first", nz[nz[:, 0] == rows[0]][:, 1][:16].tolist())
