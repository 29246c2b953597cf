"""Timing of the HIP point-cloud encoder at the Breaking Bad shape (P fragments x 1000 points); for rocprofv3."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from oracle import weights as W
from diffassemble_amd.pcd_encoder import PcdEncoderEngine
dev = torch.device("cuda:0")
P, N = int(sys.argv[1]) if len(sys.argv) > 1 else 640, int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = PcdEncoderEngine(W.make_vn_dgcnn_state(128, 1), device=dev)
pts = W.make_point_clouds(P, N, 2).to(dev)
for _ in range(2):
    eng.forward(pts)
torch.cuda.synchronize(); t = time.time()
for _ in range(reps):
    eng.forward(pts)
torch.cuda.synchronize()
print(f"P={P} N={N}: {(time.time() - t) / reps * 1e3:.2f} ms per call, {P * reps / (time.time() - t):.0f} fragments/s")
