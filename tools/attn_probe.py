"""Time one dense TransformerConv layer (projection + attention) through da_conv_dense."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffassemble_amd import engine as E, _lib
if os.environ.get("PROBE_LIB"): _lib.LIB_PATH = os.environ["PROBE_LIB"]
from diffassemble_amd.graph_plan import build_plan
import ctypes as C
dev = torch.device('cuda:0')
G = int(os.environ.get("G", 32)); n = int(os.environ.get("N", 900)); H = 8
iters = int(os.environ.get("ITERS", 10))
prec = os.environ.get("PREC", "bf16")
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1)
batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = build_plan(ei, batch, 0); del ei
lib = _lib.lib(); P = _lib.PREC_BF16 if prec == "bf16" else _lib.PREC_F32
dt = torch.bfloat16 if prec == "bf16" else torch.float32
for Ch in [int(x) for x in os.environ.get("CS", "144,32").split(",")]:
    Din = 256; HC = H * Ch
    x = torch.randn(G * n, Din, device=dev).to(dt); w = (torch.randn(4 * HC, Din, device=dev) / 16).to(dt); b = torch.randn(4 * HC, device=dev)
    g = plan.c_struct()
    nb = int(lib.da_attn_dense_scratch_bytes(P, C.byref(g), H, Ch))
    scratch = torch.zeros(nb, dtype=torch.uint8, device=dev); out = torch.empty(G * n, HC, device=dev, dtype=dt)
    def run():
        _lib.check(lib.da_conv_dense(P, C.byref(g), H, Ch, Din, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, 0, _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr(dev)))
    for _ in range(2): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    print(f"C={Ch} G={G} n={n} {prec}: conv (gemm+attn) {s.elapsed_time(e)/iters*1e3:.1f} us; attn flops {G*n*n*4*HC/1e9:.1f} GF")
