"""Per-workgroup cycle breakdown of k_attn_dense wave 0 (needs a -DDA_ATTN_PROBE build)."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
dev = torch.device('cuda:0')
prof = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
os.environ["DA_ATTN_PROF_PTR"] = str(prof.data_ptr())
from diffassemble_amd import _lib
if os.environ.get("PROBE_LIB"): _lib.LIB_PATH = os.environ["PROBE_LIB"]
from diffassemble_amd.graph_plan import build_plan
G = int(os.environ.get("G", 32)); n = 900; H = 8
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1); batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = build_plan(ei, batch, 0); del ei
lib = _lib.lib(); P = _lib.PREC_BF16; dt = torch.bfloat16
for Ch in (144, 32):
    Din = 256; HC = H * Ch
    x = torch.randn(G * n, Din, device=dev).to(dt); w = (torch.randn(4 * HC, Din, device=dev) / 16).to(dt); b = torch.randn(4 * HC, device=dev)
    g = plan.c_struct(); nb = int(lib.da_attn_dense_scratch_bytes(P, C.byref(g), H, Ch))
    scratch = torch.zeros(nb, dtype=torch.uint8, device=dev); out = torch.empty(G * n, HC, device=dev, dtype=dt)
    for _ in range(3):
        prof.zero_()
        _lib.check(lib.da_conv_dense(P, C.byref(g), H, Ch, Din, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, 0, _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr(dev)))
        torch.cuda.synchronize()
    pr = prof.view(-1, 8).cpu(); pr = pr[pr[:, 7] > 0].double()
    full = pr[pr[:, 3] > 0.9 * pr[:, 3].max()]          # workgroups whose wave 0 ran the whole key range
    m = full.mean(0)
    print(f"C={Ch}: {pr.shape[0]} workgroups; wave-0 mean cycles: total {m[0]:.0f} = barrier+wait {m[1]:.0f} + dma issue {m[2]:.0f} + qk {m[3]:.0f} + softmax {m[4]:.0f} + pv {m[5]:.0f} + epilogue {m[6]:.0f} (+ prologue {m[0]-m[1:7].sum():.0f})")
