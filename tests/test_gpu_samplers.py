"""The reference's OTHER samplers inside the captured loop (da_sample_loop_ex; VERDICT r02 missing 4 / next 8):
classifier-free guidance (spatial_diffusion.py:568-589), DDIM with eta > 0 (:620-627) and DDPM (:485-510), each compared,
step for step, with the per-step path whose arithmetic the reference fixtures pin (tests/test_gpu_parity.py:
`cfg_ddim/out_t30`, `ddpm_direct/*`, the DDIM step grid) and with the CPU oracle's update formulas."""
import pytest
import torch

import cases as C
from oracle import denoiser as OD
from oracle import diffusion as ODF

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


def _setup(dev, T=50):
    from diffassemble_amd import DenoiserEngine, Schedule
    spec = C.by_name("k36_loop_sharp")
    case = C.build_case(spec)
    eng = DenoiserEngine(case["sd"], precision="fp32", device=dev)
    plan = eng.plan(case["edge_index"], case["batch"])
    sch_cpu = ODF.make_schedule(T)
    return spec, case, eng, plan, sch_cpu, Schedule(sch_cpu, dev)


@pytest.mark.parametrize("graph", [True, False])
def test_cfg_loop_matches_the_per_step_path_and_the_oracle(dev, graph):
    """8 guided DDIM steps: captured loop == forward(cond), forward(zero features), combine, update, step by step; the
    first step also against the oracle's forward + update (the reference formula, :585-589)."""
    from diffassemble_amd import _lib
    spec, case, eng, plan, sch_cpu, sch = _setup(dev)
    w = 0.5
    x0 = case["x"].to(dev)
    feats = case["feats"].to(dev)
    traj, _ = eng.sample_loop(plan, sch, x0, feats, ratio=1, mean_type=_lib.MEAN_START_X, max_iters=8, use_graph=graph, cfg_w=w)
    traj = traj.clone()
    cur = x0
    zeros = torch.zeros_like(feats)
    for k, i in enumerate(range(49, 41, -1)):
        t = torch.full((36,), i, dtype=torch.long, device=dev)
        c = eng.forward(plan, cur, t, feats).clone()
        u = eng.forward(plan, cur, t, zeros).clone()
        cur = eng.ddim_step(sch, cur, (1 + w) * c - w * u, t, 1, _lib.MEAN_START_X).clone()
        assert rel(traj[k], cur) < 2e-6, (k, rel(traj[k], cur))
    # oracle, first step
    t = torch.full((36,), 49, dtype=torch.long)
    oc, _ = OD.eff_gat_forward_with_feats(case["sd"], case["x"], t, case["edge_index"], case["feats"], case["batch"])
    ou, _ = OD.eff_gat_forward_with_feats(case["sd"], case["x"], t, case["edge_index"], torch.zeros_like(case["feats"]), case["batch"])
    ref = ODF.ddim_update(sch_cpu, case["x"], t, (1 + w) * oc - w * ou, 1, "START_X")
    assert rel(traj[0], ref) < 1e-4
    eng.set_features(plan, feats)          # (leave the staged features as the next test expects them)


@pytest.mark.parametrize("sampler,eta", [("DDIM", 0.7), ("DDPM", 0.0)])
def test_stochastic_loops_match_the_oracle_updates_with_injected_noise(dev, sampler, eta):
    """eta > 0 and DDPM inside the captured loop, with the draws injected (what the reference gets from torch.randn_like per
    step): every step against the per-step kernels and the oracle's update on the HIP forward's output; t_index == 0 adds
    no noise in DDPM; a replay of the cached graph with NEW draws gives new poses, with the same draws the same poses."""
    from diffassemble_amd import _lib
    spec, case, eng, plan, sch_cpu, sch = _setup(dev)
    x0 = case["x"].to(dev)
    feats = case["feats"].to(dev)
    g = torch.Generator().manual_seed(11)
    c = case["x"].shape[1]
    noise = torch.randn((50, 36, c), generator=g)
    mt = _lib.MEAN_START_X
    traj, xf = eng.sample_loop(plan, sch, x0, feats, ratio=1, mean_type=mt, use_graph=True, sampler=sampler, eta=eta, noise=noise.to(dev))
    traj, xf = traj.clone(), xf.clone()
    assert torch.isfinite(traj).all() and traj.shape == (50, 36, c)
    cur = x0
    for k, i in enumerate(range(49, -1, -1)):
        t = torch.full((36,), i, dtype=torch.long, device=dev)
        out = eng.forward(plan, cur, t, feats).clone()
        if sampler == "DDPM":
            ref = ODF.ddpm_update(sch_cpu, cur.cpu(), t.cpu(), i, out.cpu(), noise[k])
            nxt = eng.ddpm_step(sch, cur, out, t, None if i == 0 else noise[k].to(dev))
        else:
            ref = ODF.ddim_update(sch_cpu, cur.cpu(), t.cpu(), out.cpu(), 1, "START_X", eta=eta, noise=noise[k])
            nxt = eng.ddim_step(sch, cur, out, t, 1, mt, eta, noise[k].to(dev))
        assert rel(traj[k], nxt) < 2e-6, (k, rel(traj[k], nxt))
        assert rel(traj[k], ref) < 2e-5, (k, rel(traj[k], ref))
        cur = nxt.clone()
    # replays of the cached graph
    _, xf2 = eng.sample_loop(plan, sch, x0, feats, ratio=1, mean_type=mt, use_graph=True, sampler=sampler, eta=eta, noise=noise.to(dev))
    assert torch.equal(xf2, xf)
    _, xf3 = eng.sample_loop(plan, sch, x0, feats, ratio=1, mean_type=mt, use_graph=True, sampler=sampler, eta=eta,
                             generator=torch.Generator(device=dev).manual_seed(3))
    assert not torch.equal(xf3, xf) and torch.isfinite(xf3).all()


def test_module_runs_every_sampler_through_the_captured_loop(dev, monkeypatch):
    """GNN_Diffusion.p_sample_loop: DDPM (the reference's own loop raises on it, SURVEY header), DDIM + guidance and
    DDIM with eta take ONE engine.sample_loop call each -- no per-step Python loop."""
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    from diffassemble_amd import engine as E
    spec = C.by_name("k36_loop_sharp")
    case = C.build_case(spec)
    calls = []
    orig = E.DenoiserEngine.sample_loop

    def spy(self, *a, **k):
        calls.append({kk: k.get(kk) for kk in ("sampler", "eta", "cfg_w")})
        return orig(self, *a, **k)
    monkeypatch.setattr(E.DenoiserEngine, "sample_loop", spy)
    for kw, want in ((dict(sampling="DDPM"), dict(sampler="DDPM", eta=0.0, cfg_w=None)),
                     (dict(sampling="DDIM", classifier_free_w=0.3, classifier_free_prob=0.1), dict(sampler="DDIM", eta=0.0, cfg_w=0.3))):
        m = GNN_Diffusion(steps=50, model_mean_type=ModelMeanType.START_X, visual_pretrained=False, noise_weight=1.0, **kw)
        m.model.load_state_dict(case["sd"], strict=False)
        m = m.to(dev)
        m.model.precision = "fp32"
        calls.clear()
        imgs, atts = m.p_sample_loop(tuple(case["x"].shape), None, case["edge_index"].to(dev), case["batch"].to(dev), patch_feats=case["feats"].to(dev))
        assert len(imgs) == 50 and all(torch.isfinite(i).all() for i in imgs)
        assert len(calls) == 1 and calls[0]["sampler"] == want["sampler"]
        if want["cfg_w"] is None:
            assert calls[0]["cfg_w"] is None
        else:
            assert calls[0]["cfg_w"] == pytest.approx(want["cfg_w"])


def test_guided_module_loop_without_the_mfma_hoist_subprocess():
    """Round-3 advisor finding: `GNN_Diffusion.p_sample_loop` sends every sampler through the captured loop, whose unconditional
    (zero-feature) pass exists on the hoisted mlp.0 path only; with DA_DISABLE_MFMA=1 the library reports "unconditional pass
    ... not available" and the module must fall back to its per-step path (which runs the second pass through
    forward_with_feats on zero features) instead of failing.  Same poses as the default configuration to fp32 rounding."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, "tests/golden")
import cases as C
from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
spec = C.by_name("k36_loop_sharp"); case = C.build_case(spec); dev = torch.device("cuda:0")
m = GNN_Diffusion(steps=50, sampling="DDIM", rotation=False, visual_pretrained=False, model_mean_type=ModelMeanType.START_X,
                  classifier_free_prob=0.1, classifier_free_w=0.5, inference_ratio=10)
m.model.load_state_dict(case["sd"], strict=False); m = m.to(dev).eval(); m.model.precision = "fp32"; m.noise_weight = 0.0
imgs, _ = m.p_sample_loop(tuple(case["x"].shape), None, case["edge_index"].to(dev), case["batch"].to(dev), patch_feats=case["feats"].to(dev))
torch.save(torch.stack(imgs).cpu(), sys.argv[1])
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for k, extra in enumerate(({}, {"DA_DISABLE_MFMA": "1"})):
        path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"da_cfg_fallback_{os.getpid()}_{k}.pt")
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **extra), capture_output=True, text=True, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs.append(torch.load(path))
        os.remove(path)
    assert outs[0].shape == (5, 36, 2) and torch.isfinite(outs[1]).all()
    assert rel(outs[1], outs[0]) < 1e-4


@pytest.mark.parametrize("pair", [False, True], ids=["one_branch", "pair_loop"])
def test_sampling_loops_inside_a_callers_graph_capture(dev, monkeypatch, pair):
    """VERDICT r05 missing 5: a Lightning `predict_step` wrapped in a user graph (torch.cuda.graph) calls p_sample_loop while ITS stream is being
    captured.  The library then must not launch graphs of its own: the loop's kernels -- for the pair loop both branches, between a fork and a join
    event on the library's pair stream -- are enqueued into the caller's capture (da_api.hip stream_is_capturing), and replaying the CALLER's graph
    gives the poses of a plain call bit for bit."""
    import cases as C2
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    from oracle import weights as W
    sd = W.make_denoiser_state(50, 4, 4, seed=11)
    eng = DenoiserEngine(sd, precision="bf16", device=dev)
    n, G = 144, 4
    ei, batch = W.collate([W.dense_edge_index(n, True) for _ in range(G)], [n] * G)
    gen = torch.Generator().manual_seed(3)
    x0 = torch.randn((G * n, 4), generator=gen).to(dev)
    feats = torch.randn((G * n, 1088), generator=gen).to(dev)
    plan = eng.plan(ei.to(dev), batch.to(dev))
    sch = Schedule(ODF.make_schedule(50), dev)
    monkeypatch.setenv("DA_TWO_BRANCH", "1" if pair else "0")
    monkeypatch.setattr(eng, "two_branch_min_graphs", 2, raising=False)
    assert eng._two_branch(plan, False, True) == pair
    kw = dict(ratio=5, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True)
    _, ref = eng.sample_loop(plan, sch, x0, feats, **kw)           # plain call (also creates the library's streams / buffers outside any capture)
    ref = ref.clone()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            _, xf = eng.sample_loop(plan, sch, x0, feats, restage=False, **kw)
    xf.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(xf).all() and torch.equal(xf, ref)
    g.replay()                                                      # (a second replay: the captured loop is self-contained)
    torch.cuda.synchronize()
    assert torch.equal(xf, ref)
