"""Timeline of one workgroup of k_gemm_wreg (needs a -DDA_WREG_PROBE build of da_gemm_wreg.hip): consumer waves 0 and 4 and the producer."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
dev = torch.device('cuda:0')
prof = torch.zeros(3 * 64 * 4, dtype=torch.int64, device=dev)
os.environ["DA_GEMM_PROF_PTR"] = str(prof.data_ptr())
from diffassemble_amd import _lib
lib = _lib.lib(); P = _lib.PREC_BF16
M, K, N = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else (57600, 256, 2560)
x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    prof.zero_()
    _lib.check(lib.da_linear(P, M, K, N, _lib.ptr(x), K, _lib.ptr(w), _lib.ptr(b), 0, None, _lib.ptr(out), N, _lib.stream_ptr(dev)))
    torch.cuda.synchronize()
pr = prof.view(3, 64, 4).cpu()
t0 = int(pr[pr > 0].min())
print("tile | wave0: barrier-exit  mfma-done  epi-done | wave4: ... | producer: pre-wait  landed  barrier-exit  issued      (cycles since first stamp)")
for t in range(8, 28):
    r = [int(v) - t0 if v > 0 else -1 for w_ in range(3) for v in pr[w_, t]]
    print(f"{t:3d} | {r[0]:7d} {r[1]:7d} {r[2]:7d} | {r[4]:7d} {r[5]:7d} {r[6]:7d} | {r[8]:7d} {r[9]:7d} {r[10]:7d} {r[11]:7d}")
d = pr[0, 9:60, 0] - pr[0, 8:59, 0]
print("wave 0 tile period: mean %.0f min %d max %d" % (d.double().mean(), d.min(), d.max()))
print("wave 0: barrier-exit -> mfma issued %.0f, -> epilogue done %.0f; producer wait for landing %.0f" % (
    (pr[0, 8:60, 1] - pr[0, 8:60, 0]).double().mean(), (pr[0, 8:60, 2] - pr[0, 8:60, 0]).double().mean(), (pr[2, 8:60, 1] - pr[2, 8:60, 0]).double().mean()))
