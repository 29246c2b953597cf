"""Which half of da_conv_dense is non-deterministic: the QKV-scatter GEMM (scratch buffers) or the attention (out)?"""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffassemble_amd import _lib
from diffassemble_amd.graph_plan import build_plan
dev = torch.device('cuda:0'); G = int(os.environ.get("G", 32)); n = 900; H = 8
r = torch.arange(n, device=dev).repeat_interleave(n); c = torch.arange(n, device=dev).repeat(n)
ei = torch.cat([torch.stack([r, c]) + g * n for g in range(G)], 1); batch = torch.arange(G, device=dev).repeat_interleave(n)
plan = build_plan(ei, batch, 0); del ei
lib = _lib.lib(); P = _lib.PREC_BF16; dt = torch.bfloat16
for Ch in (144, 32):
    Din = 256; HC = H * Ch
    x = torch.randn(G * n, Din, device=dev).to(dt); w = (torch.randn(4 * HC, Din, device=dev) / 16).to(dt); b = torch.randn(4 * HC, device=dev)
    g = plan.c_struct(); nb = int(lib.da_attn_dense_scratch_bytes(P, C.byref(g), H, Ch))
    outs, scr = [], []
    for rep in range(6):
        scratch = torch.zeros(nb, dtype=torch.uint8, device=dev); out = torch.empty(G * n, HC, device=dev, dtype=dt)
        _lib.check(lib.da_conv_dense(P, C.byref(g), H, Ch, Din, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, 0, _lib.ptr(out), _lib.ptr(scratch), _lib.stream_ptr(dev)))
        torch.cuda.synchronize(); outs.append(out); scr.append(scratch)
    ds = [int((scr[k] != scr[0]).sum()) for k in range(1, 6)]
    do = [int((outs[k] != outs[0]).any(1).sum()) for k in range(1, 6)]
    print(f"C={Ch} G={G}: scratch bytes differing vs run0 {ds}; out rows differing {do}; finite {bool(torch.isfinite(outs[0].float()).all())}")
    if do[0]:
        rows = (outs[1] != outs[0]).any(1).nonzero().flatten()
        print("   first differing rows", rows[:12].tolist(), " cols of first row", (outs[1][rows[0]] != outs[0][rows[0]]).nonzero().flatten()[:12].tolist())
    if do[0]:
        d = (outs[1].float() - outs[0].float()).abs()
        print("   max abs diff", float(d.max()), "mean |out|", float(outs[0].float().abs().mean()), "num elems differing", int((d > 0).sum()),
              "rows per q-tile-local index histogram:", torch.bincount(((rows % 900) % 128) // 8, minlength=16).tolist())
    if do[0]:
        bad = (outs[1] != outs[0])
        cols = bad.nonzero()[:, 1]
        print("   c histogram (16-wide bins of col % C):", torch.bincount((cols % Ch) // 16, minlength=9).tolist(), " head histogram:", torch.bincount(cols // Ch, minlength=8).tolist())
        rws = bad.any(1).nonzero().flatten()
        print("   graphs with bad rows:", torch.unique(rws // 900).tolist()[:40])
