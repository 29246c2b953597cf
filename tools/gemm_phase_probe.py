"""Per-workgroup cycle breakdown of k_gemm_astat (needs a -DDA_GEMM_PROBE build)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
dev = torch.device('cuda:0')
prof = torch.zeros(4 * 4096, dtype=torch.int64, device=dev)
os.environ["DA_GEMM_PROF_PTR"] = str(prof.data_ptr())
from diffassemble_amd import _lib
lib = _lib.lib(); P = _lib.PREC_BF16
for (M, K, N) in [(28800, 256, 4608), (28800, 256, 1024), (28800, 128, 1152)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        prof.zero_()
        _lib.check(lib.da_linear(P, M, K, N, _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(b), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
        torch.cuda.synchronize()
    pr = prof.view(-1, 4).cpu(); pr = pr[pr[:, 0] > 0].double()
    print(f"M={M} K={K} N={N}: {pr.shape[0]} workgroups; mean cycles total {pr[:,0].mean():.0f} wait {pr[:,1].mean():.0f} mma {pr[:,2].mean():.0f} epilogue {pr[:,3].mean():.0f}; max total {pr[:,0].max():.0f} (cycles are 100 MHz ticks if constant counter)")
