"""The next step's embedding + mlp.0 inside the tail kernel of a plain DDIM loop (k_tail_fused<true>, DdimFuse::nx_*,
da_basic.hip / da_api.hip enqueue_loop): per-row work of efficient_gat.py:131-135 that used to be two launches per step.
DA_TAIL_NEXT=1 / 0 forces it; unset (da_config.tail_next = -1) the large-graph step rule takes it for Batches of >= 512-piece graphs (last test of this file).  The loop with the fusion must reproduce the loop without it
(subprocesses: the switch is read once) far inside the bf16 mode's own distance to the fp32 engine, and the first step (whose h
still comes from the two launches) bit for bit."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RUN = r"""
import sys, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r}); sys.path.insert(0, {golden!r})
import cases as C
from oracle import diffusion as ODF
from diffassemble_amd import DenoiserEngine, Schedule, _lib
dev = torch.device("cuda:0")
out = {{}}
for name, T, ratio, mean in (("rot144_g2_sharp", 100, 1, "START_X"), ("rot144_g1", 300, 10, "EPSILON")):
    spec = C.by_name(name); case = C.build_case(spec)
    sch = Schedule(ODF.make_schedule(T), dev)
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(case["x"].shape, generator=g).to(dev)
    mt = _lib.MEAN_START_X if mean == "START_X" else _lib.MEAN_EPSILON
    for prec in ("bf16", "fp32"):
        eng = DenoiserEngine(case["sd"], variant="2d", arch=spec["arch"], virt_nodes=spec["V"], precision=prec, device=dev)
        plan = eng.plan(case["edge_index"], case["batch"])
        for use_graph in (False, True):
            traj, xf = eng.sample_loop(plan, sch, x0, case["feats"].to(dev), ratio=ratio, mean_type=mt, use_graph=use_graph)
            out[(name, prec, use_graph)] = traj.float().cpu().clone()
torch.save(out, sys.argv[1])
"""


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def test_loop_with_the_next_embedding_in_the_tail_kernel_subprocess(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    res = {}
    for tag, val in (("two_launches", "0"), ("fused", "1")):
        f = tmp_path / f"{tag}.pt"
        env = dict(os.environ, DA_TAIL_NEXT=val)
        r = subprocess.run([sys.executable, "-c", _RUN.format(root=ROOT, tests=os.path.join(ROOT, "tests"), golden=os.path.join(ROOT, "tests", "golden")),
                            str(f)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = torch.load(f)
    for name in ("rot144_g2_sharp", "rot144_g1"):
        for use_graph in (False, True):
            a, b = res["two_launches"][(name, "bf16", use_graph)], res["fused"][(name, "bf16", use_graph)]
            f32 = res["fused"][(name, "fp32", use_graph)]
            assert torch.equal(res["two_launches"][(name, "fp32", use_graph)], f32)        # fp32 engines never take the fused tail
            assert torch.isfinite(b).all()
            assert torch.equal(a[0], b[0]), name                                           # step 1: same kernels
            assert not torch.equal(a, b), "the switch selected nothing"                     # (different MFMA shapes: the last bits differ)
            d_fuse, d_prec = _rel(b, a), _rel(a, f32)
            assert d_fuse < 0.25 * d_prec + 1e-3, (name, use_graph, d_fuse, d_prec)
            assert _rel(b, f32) < 1.25 * d_prec + 1e-3, (name, use_graph, _rel(b, f32), d_prec)
        assert torch.equal(res["fused"][(name, "bf16", False)], res["fused"][(name, "bf16", True)])          # eager loop == captured loop


_RUN_AUTO = r"""
import sys, torch
sys.path.insert(0, {root!r})
from oracle import diffusion as ODF, weights as W
from diffassemble_amd import DenoiserEngine, Schedule, _lib
dev = torch.device("cuda:0")
out = {{}}
sch = Schedule(ODF.make_schedule(100), dev)
for n, G in ((900, 2), (144, 3)):
    sd = W.make_denoiser_state(100, 4, 4, seed=5)
    eng = DenoiserEngine(sd, precision="bf16", device=dev)
    ei, batch = W.collate([W.dense_edge_index(n, True) for _ in range(G)], [n] * G)
    g = torch.Generator().manual_seed(9)
    x0 = torch.randn((G * n, 4), generator=g).to(dev)
    feats = torch.randn((G * n, 1088), generator=g).to(dev)
    plan = eng.plan(ei.to(dev), batch.to(dev))
    traj, _ = eng.sample_loop(plan, sch, x0, feats, ratio=1, mean_type=_lib.MEAN_START_X, max_iters=6, use_graph=True)
    out[n] = traj.float().cpu().clone()
torch.save(out, sys.argv[1])
"""


def test_step_auto_rule_selects_by_graph_size_subprocess(tmp_path):
    """The large-graph step rule (da_config.xpanel = tail_next = -1; da_gemm_xpanel.hip xpanel_mode, da_api.hip enqueue_loop / forward_impl): with neither switch set, Batches
    whose largest graph has >= 512 pieces take the row-panel projections AND the tail kernel's next-step embedding -- bit for bit what
    DA_ENABLE_XPANEL=1 DA_TAIL_NEXT=1 gives -- and smaller graphs take neither (bit for bit both switches 0).  The 900-piece loops with and
    without the rule agree in their first step exactly and afterwards far inside bf16 resolution of the poses."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    res = {}
    for tag, extra in (("default", {}), ("both_on", {"DA_ENABLE_XPANEL": "1", "DA_TAIL_NEXT": "1"}), ("rule_off", {"DA_ENABLE_XPANEL": "0", "DA_TAIL_NEXT": "0"})):
        f = tmp_path / f"{tag}.pt"
        env = {k: v for k, v in os.environ.items() if k not in ("DA_ENABLE_XPANEL", "DA_TAIL_NEXT")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", _RUN_AUTO.format(root=ROOT), str(f)], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        res[tag] = torch.load(f)
    assert torch.equal(res["default"][900], res["both_on"][900])
    assert torch.equal(res["default"][144], res["rule_off"][144])
    a, b = res["default"][900], res["rule_off"][900]
    assert torch.isfinite(a).all() and torch.equal(a[0], b[0])
    assert not torch.equal(a, b), "the rule selected nothing"
    assert _rel(a, b) < 2e-2, _rel(a, b)
