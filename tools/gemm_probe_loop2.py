"""da_linear_packed (row-panel kernel) in a tight loop (power sampling): M K N iters."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffassemble_amd import engine as E
M, K, N, iters = [int(a) for a in sys.argv[1:5]]
dev = torch.device('cuda:0')
x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16(); b = torch.randn(N, device=dev)
lin = E.PackedLinear(w, b)
from diffassemble_amd import _lib
lib = _lib.lib()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    _lib.check(lib.da_linear_packed(_lib.PREC_BF16, M, K, N, _lib.ptr(x), K, _lib.ptr(lin.weight), _lib.ptr(lin.packed), _lib.ptr(lin.bias), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
e.record(); torch.cuda.synchronize()
print(f"packed M={M} K={K} N={N}: {s.elapsed_time(e) / iters * 1e3:.1f} us per launch over {iters} launches")
