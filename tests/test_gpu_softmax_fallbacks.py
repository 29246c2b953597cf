"""The shift-free softmax paths of the dense attention kernels under logits they were NOT tuned for (round-3 verdict,
"weak" 1 + 2): PyG's softmax is `exp(a - max) / (sum + 1e-16)` (torch_geometric.utils.softmax as called by
TransformerConv, reference call sites backbones/Transformer_GNN.py:32,38), shift invariant up to an epsilon that is
invisible next to a sum >= 1.  The kernels that skip the shift (k_attn_opt: optimistic, verified on the final row sums;
k_attn_dense FAST mode: range test per key block; k_attn_dual: range test per region) must therefore

  * give the SAME answer whatever constant sits on a row's logits -- down to -45 nat (2^-65: below every accepted
    window) and up to +80 nat (2^115: above it) -- and
  * really take their running-max fallbacks when the un-shifted sums leave the exponent window (da_debug_counters).

Every case goes through the C ABI (da_conv_dense_ex: the layer with the Q rows pre-scaled by log2(e) / sqrt(C), as
da_denoiser_create packs it) and is compared with an fp64 dense evaluation of the PyG formula on the operands the kernel
sees.  The logit offsets are produced inside the projection: a reserved head channel carries a per-query value r_i on the Q
side and the constant 8 on the K side, so row i's logits move by exactly 8 r_i log2-units -- integers here, which makes
exp2(s + offset) = exp2(s) 2^offset exact and the optimistic path's output independent of the offset bit for bit.
"""
import math
import os
import subprocess
import sys

import pytest
import torch

from oracle import weights as W

pytestmark = pytest.mark.gpu
H = 8


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def build_layer(sizes, C, folded, seed, row_offset_log2, key_outlier_log2=None, bf16=False):
    """x [N, 256] and the UNSCALED-by-anyone projection in the kernel's own units: the Q rows already carry
    log2(e) / sqrt(C) (prescale_q="done"), so logits = q'.k are in log2 units.  Channel 0 of every head is reserved for the
    per-query offset (q'[i, h, 0] = r_i = row_offset_log2[i] / 8, k[j, h, 0] = 8), channel 1 for a per-key term
    (q'[i, h, 1] = 8, k[j, h, 1] = u_j = key_outlier_log2[j] / 8); input features 254 / 255 carry r and u."""
    g = torch.Generator().manual_seed(seed)
    N, Din = sum(sizes), 256
    HC = H * C
    x = torch.randn(N, Din, generator=g)
    x[:, 254] = torch.as_tensor(row_offset_log2, dtype=torch.float32) / 8.0
    x[:, 255] = 0.0 if key_outlier_log2 is None else torch.as_tensor(key_outlier_log2, dtype=torch.float32) / 8.0
    sc = math.log2(math.e) / math.sqrt(C)
    wq = torch.randn(HC, Din, generator=g) / Din ** 0.5 * 2.0 * sc
    wk = torch.randn(HC, Din, generator=g) / Din ** 0.5 * 2.0
    nv = H * 32 if folded else HC
    wv = torch.randn(nv, Din, generator=g) / Din ** 0.5
    ws = torch.randn(HC, Din, generator=g) / Din ** 0.5
    bq, bk = torch.randn(HC, generator=g) * 0.1 * sc, torch.randn(HC, generator=g) * 0.1
    bv, bs = torch.randn(nv, generator=g) * 0.1, torch.randn(HC, generator=g) * 0.1
    for w_ in (wq, wk, wv, ws):
        w_[:, 254:] = 0.0                                   # the two carrier features feed the reserved channels only
    for h in range(H):
        for ch, (wq_col, bq_v, wk_col, bk_v) in enumerate([(254, 0.0, None, 8.0), (None, 8.0, 255, 0.0)]):
            r = h * C + ch
            wq[r], wk[r] = 0.0, 0.0
            bq[r], bk[r] = bq_v, bk_v
            if wq_col is not None:
                wq[r, wq_col] = 1.0
            if wk_col is not None:
                wk[r, wk_col] = 1.0
    ws_all = [wq, wk, wv] + ([] if folded else [ws])
    bs_all = [bq, bk, bv] + ([] if folded else [bs])
    if bf16:
        x = x.bfloat16().float()
        ws_all = [w_.bfloat16().float() for w_ in ws_all]
    return x, ws_all, bs_all


def reference(x, ws, bs, sizes, loops, C, folded, bf16):
    """fp64 PyG semantics on complete graphs: alpha = exp(a - max) / (sum + 1e-16) per (target, head) over the graph's
    nodes (the diagonal left out for loop-free graphs; a target without incoming edge gets 0), out = alpha V (+ skip)."""
    xd = x.double()
    q, k, v = [xd @ w_.double().T + b_.double() for w_, b_ in zip(ws[:3], bs[:3])]
    if bf16:                                                # the projection's outputs are stored in bf16
        q, k, v = [t.float().bfloat16().double() for t in (q, k, v)]
    cv = 32 if folded else C
    out = torch.zeros(H, x.shape[0], cv, dtype=torch.float64)
    o = 0
    for n in sizes:
        qg = q[o:o + n].view(n, H, C).permute(1, 0, 2)
        kg = k[o:o + n].view(n, H, C).permute(1, 0, 2)
        vg = v[o:o + n].view(n, H, cv).permute(1, 0, 2)
        a = (qg @ kg.transpose(1, 2)) * math.log(2.0)       # log2 units -> nat
        if not loops:
            a = a.masked_fill(torch.eye(n, dtype=torch.bool), float("-inf"))
        m = a.max(-1, keepdim=True).values
        m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
        e = torch.exp(a - m)
        out[:, o:o + n] = (e / (e.sum(-1, keepdim=True) + 1e-16)) @ vg
        o += n
    if folded:
        return out
    skip = xd @ ws[3].double().T + bs[3].double()
    if bf16:
        skip = skip.float().bfloat16().double()
    return out.permute(1, 0, 2).reshape(x.shape[0], H * C) + skip


def run_layer(dev, sizes, loops, C, folded, prec, x, ws, bs):
    from diffassemble_amd import engine as E
    from diffassemble_amd.graph_plan import build_plan
    ei, batch = W.collate([W.dense_edge_index(n, loops) for n in sizes], sizes)
    plan = build_plan(ei.to(dev), batch.to(dev), 0)
    assert plan.dense == (1 if loops else 2)
    E.debug_counters(reset=True)
    out = E.conv_dense_ex(plan, x.to(dev), torch.cat(ws).to(dev), torch.cat(bs).to(dev), H, C, None, 0, prec,
                          prescale_q="done", folded=folded)
    torch.cuda.synchronize()
    return out.float().cpu(), E.debug_counters(reset=True)


def offsets_for(kind, sizes):
    """(row offsets [N], key outliers [N] or None, who must fall back) in log2 units (integers: multiples of 1/8 after
    the / 8 of the carrier features).  "all": every shift-free kernel; "fast": only the kernels with a per-block test
    (k_attn_dense FAST, k_attn_dual) -- the optimistic kernels judge the FINAL row sums, which stay inside their window
    there; "none": nobody."""
    N = sum(sizes)
    row, key, who = torch.zeros(N), None, "all"
    if kind == "m44":                      # -30.5 nat: inside every window
        row[:] = -44.0
        who = "none"
    elif kind == "m55":                    # -38.1 nat: inside the windows, where + 1e-16 on the raw sum was 2e-3 off
        row[:] = -55.0
        who = "none"
    elif kind == "m80":                    # -55 nat: sums far below 2^-60 -> fallback everywhere
        row[:] = -80.0
    elif kind == "p110":                   # +76 nat on every logit: sums above 2^100 -> fallback everywhere
        row[:] = 110.0
    elif kind == "one_wave":               # queries 32..63 of every graph (one wave of the first query tile) at -80
        o = 0
        for n in sizes:
            row[o + 32:o + min(n, 64)] = -80.0
            o += n
    elif kind == "mixed":                  # neighbouring queries at -80 / 0 / +110: both failures inside one wave
        row[0::3] = -80.0
        row[1::3] = 110.0
    elif kind == "late_outlier":           # one key near the END of every graph at +115 (~ +80 nat): the overflow arrives
        key = torch.zeros(N)               # after FAST mode accumulated a healthy state (hand-off with l > 0)
        o = 0
        for n in sizes:
            key[o + (n * 7) // 8] = 115.0
            o += n
    elif kind == "late_outlier_after_small_sums":   # rows at -55 (tiny but accepted sums) whose LAST key sits at +115: a
        row[:] = -55.0                              # per-block test hands a state with a very negative reference over;
        key = torch.zeros(N)                        # the final sums (~2^60) are inside the optimistic window
        who = "fast"
        o = 0
        for n in sizes:
            key[o + n - 1] = 115.0
            o += n
    else:
        raise ValueError(kind)
    return row, key, who


KINDS = ["m44", "m55", "m80", "p110", "one_wave", "mixed", "late_outlier", "late_outlier_after_small_sums"]
SHAPES = [
    # C, folded
    pytest.param(32, False, id="c32"),
    pytest.param(144, False, id="c144"),
    pytest.param(144, True, id="c144_folded"),
]


def optimistic_kernel(C, folded, prec):
    """Which family takes the layer by default: the optimistic kernels (da_attn_opt.hip: bf16, C = 32 or the folded
    C = 144) or k_attn_dense's FAST mode (fp32, un-folded C = 144); DA_ATTN_DUAL=1 moves the folded bf16 layer to k_attn_dual."""
    return prec == "bf16" and (C == 32 or folded) and not (folded and os.environ.get("DA_ATTN_DUAL") == "1")


def _case(dev, sizes, loops, C, folded, prec, kind, seed=0):
    bf16 = prec == "bf16"
    row, key, who = offsets_for(kind, sizes)
    x, ws, bs = build_layer(sizes, C, folded, seed, row, key, bf16)
    ref = reference(x, ws, bs, sizes, loops, C, folded, bf16)
    out, cnt = run_layer(dev, sizes, loops, C, folded, prec, x, ws, bs)
    assert torch.isfinite(out).all(), kind
    tol = 1e-4 if prec == "fp32" else 1e-2       # bf16: P and the outputs are rounded to 8 bits; the epsilon bug was 2e-3 .. 0.7
    err = rel(out, ref)
    assert err < tol, (kind, prec, err)
    fell = sum(cnt.values())
    opt = optimistic_kernel(C, folded, prec)
    if who == "all" or (who == "fast" and not opt):
        assert fell > 0, (kind, cnt)              # the branch under test really ran ...
        assert cnt["opt_gen_workgroups" if opt else ("dual_gen_slabs" if os.environ.get("DA_ATTN_DUAL") == "1" and folded and bf16
                                                     else "dense_fast_exits")] > 0, (kind, cnt)      # ... in the kernel it was aimed at
    else:
        # ... and the common path does not take it.  (k_attn_dense's per-block test looks at each LANE's 16 keys of a block:
        # in graphs of <= 16 pieces the upper half-lanes see only masked keys, a partial sum of 0, and the wave -- one per
        # head -- leaves FAST mode; harmless, and not the benched shape.)
        assert fell <= 8 * sum(1 for n in sizes if n <= 16), (kind, cnt)
    return out, cnt


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("C,folded", SHAPES)
def test_logit_offsets_ragged_batch(dev, C, folded, prec, kind):
    """Ragged complete graphs with self loops (tails in every tile shape, a 1-piece graph)."""
    _case(dev, [130, 37, 64, 1, 200], True, C, folded, prec, kind)


@pytest.mark.parametrize("kind", ["m55", "m80", "one_wave", "late_outlier"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("C,folded", SHAPES)
def test_logit_offsets_without_self_loops(dev, C, folded, prec, kind):
    """Loop-free graphs (diagonal masked), incl. a 1-piece graph whose only row has NO incoming edge (PyG: attention term 0,
    the row is its skip projection) and a 2-piece graph (one key per row)."""
    _case(dev, [70, 1, 2, 129], False, C, folded, prec, kind)


@pytest.mark.parametrize("kind", ["m55", "m80", "one_wave", "late_outlier", "p110"])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("C,folded", [pytest.param(32, False, id="c32"), pytest.param(144, True, id="c144_folded")])
def test_logit_offsets_900_pieces(dev, C, folded, prec, kind):
    """The benched size: one 900-piece puzzle (8 query tiles, the last with 4 queries; 15 key tiles, the last with 4 keys)."""
    _case(dev, [900], True, C, folded, prec, kind)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("C,folded", SHAPES)
def test_shift_free_paths_are_offset_invariant(dev, C, folded, prec):
    """Softmax is shift invariant: inside the exponent window the shift-free kernels must give the offset-free answer at
    -44 and -55 log2-units too, and without any fallback.  (Before the epsilon fix the -55 rows came out 2e-3 LOW: `+ 1e-16`
    on an un-shifted sum of ~2^-50.)  fp32: 2e-5 norm-wise (the fp32 accumulation of s + offset rounds at a coarser ulp).
    bf16: single outputs may differ by one bf16 ulp (the same rounding of s reaches P), so the test is on the MEAN signed
    relative deviation over the significant outputs -- a systematic 2e-3 shows there, rounding flips average out."""
    sizes = [130, 37, 64, 1, 200]
    outs = []
    for off in (0.0, -44.0, -55.0):
        row = torch.full((sum(sizes),), off)
        x, ws, bs = build_layer(sizes, C, folded, 3, row, None, prec == "bf16")
        out, cnt = run_layer(dev, sizes, True, C, folded, prec, x, ws, bs)
        assert sum(cnt.values()) <= 8, cnt          # (the 1-piece graph: see _case)
        outs.append(out.double())
    big = outs[0].abs() > 0.1 * outs[0].abs().max()
    for o in outs[1:]:
        dev_rel = ((o - outs[0]) / outs[0])[big]
        if prec == "fp32":
            assert rel(o, outs[0]) < 2e-5
        else:
            assert rel(o, outs[0]) < 8e-3                                   # one bf16 ulp of the largest output
            assert abs(float(dev_rel.mean())) < 2e-4, float(dev_rel.mean())


def hybrid_graph(sizes, seed):
    """Random in-graph edges (each ordered pair with probability 1/2, self loops included), a node without any incoming
    edge per graph, and duplicated pairs (both copies leave the adjacency mask and travel as remainder edges)."""
    g = torch.Generator().manual_seed(seed)
    eis, o = [], 0
    for n in sizes:
        a = torch.rand(n, n, generator=g) < 0.5                       # a[i, j]: edge j -> i
        a[n // 2, :] = False                                          # this target has no regular edge at all
        if n > 3:
            a[1, :] = False                                           # ... and this one only gets duplicated (remainder) edges
        dst, src = a.nonzero(as_tuple=True)
        ei = torch.stack([src, dst])
        dup = ei[:, torch.randperm(ei.shape[1], generator=g)[: max(1, n // 4)]]
        extra = torch.tensor([[0, min(2, n - 1)], [min(1, n - 1), min(1, n - 1)]])   # edges 0 -> 1 and 2 -> 1, each twice
        eis.append(torch.cat([ei, dup, dup, extra, extra], 1) + o)
        o += n
    return torch.cat(eis, 1), torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))


def reference_edges(x, ws, bs, ei, C, folded, bf16):
    """fp64 PyG TransformerConv on an edge list (multi-edges count twice): alpha_e = exp(a_e - max_i) / (sum_i + 1e-16)."""
    xd = x.double()
    N = x.shape[0]
    q, k, v = [xd @ w_.double().T + b_.double() for w_, b_ in zip(ws[:3], bs[:3])]
    if bf16:
        q, k, v = [t.float().bfloat16().double() for t in (q, k, v)]
    cv = 32 if folded else C
    src, dst = ei[0], ei[1]
    a = (q.view(N, H, C)[dst] * k.view(N, H, C)[src]).sum(-1) * math.log(2.0)          # [E, H]
    m = torch.full((N, H), float("-inf"), dtype=torch.float64).scatter_reduce(0, dst[:, None].expand(-1, H), a, "amax")
    e = torch.exp(a - m[dst])
    den = torch.zeros(N, H, dtype=torch.float64).index_add_(0, dst, e) + 1e-16
    out = torch.zeros(N, H, cv, dtype=torch.float64).index_add_(0, dst, (e / den[dst])[:, :, None] * v.view(N, H, cv)[src])
    if folded:
        return out.permute(1, 0, 2).contiguous()
    skip = xd @ ws[3].double().T + bs[3].double()
    if bf16:
        skip = skip.float().bfloat16().double()
    return out.reshape(N, H * C) + skip


@pytest.mark.parametrize("kind", ["m44", "m55", "m80", "p110", "mixed", "one_wave"])
@pytest.mark.parametrize("C,folded", [pytest.param(32, False, id="c32"), pytest.param(144, True, id="c144_folded")])
def test_logit_offsets_hybrid_graphs_masked_optimistic_kernel(dev, C, folded, kind):
    """k_attn_optt<MASKED> (bf16 hybrid graphs: adjacency-masked regular edges + remainder edges in the epilogue, the path of
    the Exphander / exophormer configuration): rows without any regular edge (sum 0 is legitimate there), rows fed by
    duplicated edges only, all under the logit offsets; fallback = a workgroup re-run with the running-max recurrence."""
    from diffassemble_amd import engine as E
    from diffassemble_amd.graph_plan import build_plan
    sizes = [130, 70, 200]
    row, key, who = offsets_for(kind, sizes)
    x, ws, bs = build_layer(sizes, C, folded, 5, row, key, True)
    ei, batch = hybrid_graph(sizes, 7)
    ref = reference_edges(x, ws, bs, ei, C, folded, True)
    plan = build_plan(ei.to(dev), batch.to(dev), 0, hybrid="force")
    assert plan.hybrid == 1
    E.debug_counters(reset=True)
    out = E.conv_dense_ex(plan, x.to(dev), torch.cat(ws).to(dev), torch.cat(bs).to(dev), H, C, None, 0, "bf16",
                          prescale_q="done", folded=folded)
    torch.cuda.synchronize()
    cnt = E.debug_counters(reset=True)
    out = out.float().cpu()
    assert torch.isfinite(out).all()
    assert rel(out, ref) < 1e-2, (kind, rel(out, ref))
    if who == "all":
        assert cnt["opt_masked_gen_workgroups"] > 0, cnt
    else:
        assert sum(cnt.values()) == 0, cnt


def test_force_gen_switch_runs_the_fixture_suite_through_the_fallbacks_subprocess():
    """DA_ATTN_FORCE_GEN=1 starts every shift-free kernel in its running-max mode: the reference-fixture forwards (bf16 and
    fp32, 900 pieces included) must still match."""
    from conftest import exp_env
    env = exp_env(DA_ATTN_FORCE_GEN="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(root, "tests", "test_gpu_parity.py"),
                        os.path.join(root, "tests", "test_gpu_benched_mode.py"), "-k",
                        "test_forward_2d_bf16 or test_forward_2d_fp32_vs_oracle_and_golden or test_rot900_dense_forward or "
                        "test_rot900_ddim_trajectory or test_da_conv_dense_matches"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_dual_slab_kernel_fallbacks_subprocess():
    """k_attn_dual (opt-in, DA_ATTN_DUAL=1) takes the folded bf16 last layer: the same offset cases through ITS per-region
    range test and GEN hand-off."""
    from conftest import exp_env
    env = exp_env(DA_ATTN_DUAL="1", DA_TEST_EXPECT_DUAL="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.abspath(__file__), "-k",
                        "bf16 and c144_folded and (ragged_batch or 900_pieces or without_self_loops)"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
