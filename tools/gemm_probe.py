"""Times da_linear (bf16) on the projection shapes and checks it against a float32 product of the same bf16 operands.
DA_WREG_DIRECT / DA_WREG2 / DA_GEMM_DEBUG select the kernel variants (da_gemm_wreg.hip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffassemble_amd import _lib
dev = torch.device('cuda:0')
def bench(M, K, N, prec='bf16', iters=20):
    dt = torch.bfloat16 if prec == 'bf16' else torch.float32
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    lib = _lib.lib(); P = _lib.PREC_BF16 if prec == 'bf16' else _lib.PREC_F32
    nb = lib.da_linear_packed_bytes(P, K, N) if os.environ.get("PACKED", "1") == "1" else 0
    if nb:
        wp = torch.empty(nb, dtype=torch.uint8, device=dev)
        _lib.check(lib.da_linear_pack(P, K, N, _lib.ptr(w), K, _lib.ptr(wp), _lib.stream_ptr(dev)))
    def run():
        if nb:
            _lib.check(lib.da_linear_packed(P, M, K, N, _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(wp), _lib.ptr(b), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
        else:
            _lib.check(lib.da_linear(P, M, K, N, _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(b), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    ref = x[:4096].float() @ w.float().t() + b
    err = (out[:4096].float() - ref).abs().max().item()
    ref2 = x[-64:].float() @ w.float().t() + b
    err = max(err, (out[-64:].float() - ref2).abs().max().item())
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    fl = 2.0 * M * K * N
    return us, fl / us / 1e6, err
if len(sys.argv) >= 4:
    shapes = [tuple(int(a) for a in sys.argv[1:4])]
else:
  shapes = [(57600, 256, 2560), (57600, 256, 1024), (57600, 128, 1024), (57613, 256, 1024), (28800, 256, 3456), (28800, 256, 2560), (115200, 256, 1024)]
for (M, K, N) in shapes:
    us, tf, err = bench(M, K, N)
    print(f"packed={os.environ.get('PACKED','1')} direct={os.environ.get('DA_WREG_DIRECT','-')} v2={os.environ.get('DA_WREG2','-')} dbg={os.environ.get('DA_GEMM_DEBUG','0')} M={M} K={K} N={N}: {us:8.1f} us {tf:7.1f} TF/s  "
          f"out {M*N*2/1e6:.0f} MB -> {M*N*2/us/1e6:.2f} TB/s  max err {err:.3e}")
