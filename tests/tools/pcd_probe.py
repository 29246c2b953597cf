"""Per-case error of the HIP point-cloud encoder against the oracle (and timing at the Breaking Bad shape)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "golden")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np, torch
import cases as C
from oracle import vn_dgcnn as OV, weights as W
from diffassemble_amd.pcd_encoder import PcdEncoderEngine, knn
dev = torch.device("cuda:0")
G = C.load_golden3()
for spec in C.PCD_ENC + [dict(name="p13_n200", P=13, N=200, seed=9, wseed=5, inv=False), dict(name="p4_n500", P=4, N=500, seed=19, wseed=5, inv=False)]:
    sd, pts = C.pcd_encoder_case(spec)
    out = PcdEncoderEngine(sd, inv=spec["inv"], device=dev).forward(pts.to(dev)).cpu().numpy()
    want, mid = OV.forward(sd, pts.numpy(), inv=spec["inv"], return_intermediates=True)
    idx = knn(pts.to(dev)).cpu().numpy()
    flips1 = int((np.sort(idx, -1) != np.sort(mid["idx1"], -1)).any(-1).sum())
    x1 = torch.from_numpy(mid["x1"].reshape(spec["P"], spec["N"], 63)).to(dev)
    idx2 = knn(x1).cpu().numpy()
    flips2 = int((np.sort(idx2, -1) != np.sort(mid["idx2"], -1)).any(-1).sum())
    e = np.abs(out - want).max(1) / np.abs(want).max()
    print(spec["name"], "rel per cloud", np.array2string(e, precision=2), "flipped lists stage1", flips1, "stage2 (oracle x1)", flips2)
if len(sys.argv) > 1:
    P, N = int(sys.argv[1]), 1000
    sd = W.make_vn_dgcnn_state(128, 1)
    eng = PcdEncoderEngine(sd, device=dev)
    pts = W.make_point_clouds(P, N, 2).to(dev)
    for _ in range(2): eng.forward(pts)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(5): eng.forward(pts)
    torch.cuda.synchronize(); print(f"P={P} N={N}: {(time.time()-t)/5*1e3:.2f} ms per call")
