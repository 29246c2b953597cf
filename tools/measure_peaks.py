#!/usr/bin/env python
"""Measured ceilings of the box, to sit beside the vendor peaks in the roofline figures:
the library bf16 GEMM rate (torch.matmul -> hipBLASLt, 8192^3) and the device-to-device copy bandwidth.
Prints one JSON line; committed copy: profiles/r01/measured_peaks.json."""
import json
import time

import torch

dev = torch.device("cuda:0")
out = {}
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        (a @ b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        (a @ b)
    torch.cuda.synchronize()
    out[f"hipblaslt_bf16_gemm_{n}_tflops"] = 2 * n ** 3 * reps / (time.perf_counter() - t0) / 1e12
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)           # 1 GiB, read + write
y = torch.empty_like(x)
for _ in range(3):
    y.copy_(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y.copy_(x)
torch.cuda.synchronize()
out["d2d_copy_GBps_read_plus_write"] = 2 * x.numel() * 20 / (time.perf_counter() - t0) / 1e9
out["device"] = torch.cuda.get_device_name(0)
print(json.dumps(out))
