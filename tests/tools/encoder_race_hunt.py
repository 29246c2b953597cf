"""Stress the persistent encoder kernels (manual vmcnt accounting, LDS ring reuse): repeated passes over the same
input must be bit-identical, for several batch sizes / chunkings and both precisions."""
import sys

import torch

sys.path.insert(0, ".")
from diffassemble_amd.encoder import EncoderEngine
from oracle import weights as W

dev = torch.device("cuda")
sd = W.make_encoder_state(1)
bad = 0
for prec in ("bf16", "fp32"):
    for n, chunk, reps in ((28800, None, 12), (4100, 1024, 60), (1032, 1032, 120), (999, 200, 120), (37, None, 200)):
        if prec == "fp32" and n > 5000:
            continue
        x = torch.rand((n, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(n), device=dev)
        eng = EncoderEngine(sd, precision=prec, device=dev, chunk=chunk)
        ref = eng.forward(x).clone()
        diff = 0
        for _ in range(reps):
            out = eng.forward(x)
            if not torch.equal(out, ref):
                diff += 1
        bad += diff
        print(f"{prec} n={n} chunk={chunk}: {diff}/{reps} passes differ", flush=True)
print("RACE HUNT", "FAILED" if bad else "clean")
sys.exit(1 if bad else 0)
