"""Ablation timings of k_attn_dense (needs a -DDA_ATTN_PROBE build of da_attn_dense.hip; PROBE_LIB = path of that library).
DA_ATTN_DEBUG bits: 1 no DMA after the prologue, 2 no softmax, 4 no PV, 8 no QK^T, 32 one key tile only."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffassemble_amd import _lib
if os.environ.get("PROBE_LIB"):
    _lib.LIB_PATH = os.environ["PROBE_LIB"]
from diffassemble_amd.graph_plan import build_plan
dev = torch.device('cuda:0')
G = int(os.environ.get("G", 64)); n = int(os.environ.get("N", 900)); H = 8
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1); batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = build_plan(ei, batch, 0); del ei
lib = _lib.lib(); P = _lib.PREC_BF16; dt = torch.bfloat16
for Ch in [int(x) for x in os.environ.get("CS", "32").split(",")]:
    Din = 256; HC = H * Ch
    x = torch.randn(G * n, Din, device=dev).to(dt); w = (torch.randn(4 * HC, Din, device=dev) / 16).to(dt); b = torch.randn(4 * HC, device=dev)
    g = plan.c_struct(); nb = int(lib.da_attn_dense_scratch_bytes(P, C.byref(g), H, Ch))
    scratch = torch.zeros(nb, dtype=torch.uint8, device=dev); out = torch.empty(G * n, HC, device=dev, dtype=dt)
    def run():
        _lib.check(lib.da_conv_dense(P, C.byref(g), H, Ch, Din, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, 0, _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr(dev)))
    res = {}
    dbgs = [int(v) for v in os.environ.get("DBGS", "0,32,1,2,4,8,6,10,12,14,15").split(",")]
    for dbg in dbgs + dbgs:                      # two passes, the second one is reported (the first runs of a process are slower)
        os.environ["DA_ATTN_DEBUG"] = str(dbg)
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): run()
        e.record(); torch.cuda.synchronize()
        res[dbg] = s.elapsed_time(e) / 20 * 1e3
    base = res.get(32, 0.0)
    print(f"C={Ch} G={G} n={n}: " + "  ".join(f"dbg{k}={v:.1f}us" for k, v in res.items()))
    print("   minus the one-tile run (projection + prologue + epilogue): " + "  ".join(f"dbg{k}={v - base:.1f}" for k, v in res.items() if k != 32))
