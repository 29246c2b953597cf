import os, sys, torch, time
sys.path.insert(0, '/root/repo')
from diffassemble_amd import engine as E, _lib
dev = torch.device('cuda:0')
def bench(M, K, N, prec='bf16', iters=20):
    dt = torch.bfloat16 if prec == 'bf16' else torch.float32
    x = torch.randn(M, K, device=dev).to(dt); w = torch.randn(N, K, device=dev).to(dt); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    lib = _lib.lib(); P = _lib.PREC_BF16 if prec == 'bf16' else _lib.PREC_F32
    def run():
        _lib.check(lib.da_linear(P, M, K, N, _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(b), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    fl = 2.0 * M * K * N
    return us, fl / us / 1e6
for (M, K, N) in [(28800, 256, 3456), (28800, 256, 1024), (28800, 1152, 1024), (28800, 1152, 128), (28800, 128, 1152), (7200, 256, 3456), (28800, 256, 4608)]:
    us, tf = bench(M, K, N)
    print(f"dbg={os.environ.get('DA_GEMM_DEBUG','0')} M={M} K={K} N={N}: {us:8.1f} us {tf:7.1f} TF/s  out {M*N*2/1e6:.0f} MB -> {M*N*2/us/1e6:.2f} TB/s")
