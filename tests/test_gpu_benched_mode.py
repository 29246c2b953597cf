"""Parity of the BENCHED configurations (VERDICT r01 "weak" 1, 2, 4, 5):

* the headline workload -- one 900-piece puzzle on the complete graph -- and the scripted Exphander degree d = 539
  against outputs of the reference's own code (tests/golden/golden_v2.npz, make_golden_v2.py), in fp32 AND in the
  benched bf16 mode, through every dispatch the library has for them (dense + folds, edge-list, hybrid, folds off);
* bf16 over a WHOLE T = 100 sampling loop against the fp32 HIP trajectory (12x12 and 30x30): pose drift bound;
* the end metric: a denoiser TRAINED on the GPU (HIP training path) until it solves synthetic puzzles, then sampled
  in fp32 and bf16 -- identical greedy assignments / rotation decisions and identical accuracy;
* two different same-shaped Batches back to back (plan / feature caches must not go stale);
* packed inference weights follow the fused optimizer;
* greedy assignment against the reference's TorchScript function.

Tolerances:  RTOL32 = 1e-4 norm-wise (max-abs error / max-abs of the reference tensor) and ELEM32 = 1e-3 element-wise
relative on every element whose magnitude is at least 5 % of the tensor's max-abs (fp32 parity mode); RTOLBF = 8e-3
norm-wise for single bf16 forwards (2.6x the largest measured error, 3.1e-3); loop drift bounds are stated in the tests.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases as C
from oracle import denoiser as OD
from oracle import diffusion as ODF
from oracle import weights as W

pytestmark = pytest.mark.gpu
RTOL32, ELEM32, TRAJ32 = 1e-4, 1e-3, 5e-4
RTOLBF = float(os.environ.get('DA_TEST_RTOLBF', 8e-3))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_elem(a, b, floor=0.05):
    """Largest element-wise relative error over the elements of ``b`` that are not near zero
    (|b| >= floor * max|b|)."""
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    big = b.abs() >= floor * b.abs().max()
    return float(((a - b).abs()[big] / b.abs()[big]).max())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def golden2():
    return C.load_golden2()


def make_engine(case, spec, prec, dev):
    from diffassemble_amd import DenoiserEngine
    return DenoiserEngine(case["sd"], variant="2d", arch=spec["arch"], virt_nodes=spec["V"], precision=prec, device=dev)


# ---------------------------------------------------------------------------- 900-piece dense (headline)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_rot900_dense_forward_vs_reference_fixture(dev, golden2, prec):
    """The 64-query / 64-key tail tiles of n = 900 (900 = 7 * 128 + 4 = 14 * 64 + 4), the folded last layer and
    the C = 32 kernel at 15 key tiles, against the reference's forward on the same seeded inputs."""
    spec = C.by_name("rot900_g1")
    case = C.build_case(spec)
    ref = golden2["rot900_g1/out"]
    eng = make_engine(case, spec, prec, dev)
    plan = eng.plan(case["edge_index"], case["batch"])
    assert plan.dense == 1 and plan.n_edges == 810000
    out = eng.forward(plan, case["x"].to(dev), case["t"].to(dev), case["feats"].to(dev))      # dense MFMA path + folds
    tol = RTOL32 if prec == "fp32" else RTOLBF
    assert rel(out, ref) < tol
    if prec == "fp32":
        assert rel_elem(out, ref) < ELEM32
    # the edge-list path (alpha requested): same poses, and the reference's own attention weights
    out2, alpha = eng.forward(plan, case["x"].to(dev), case["t"].to(dev), None, return_alpha=True, alpha_all_layers=True)
    assert rel(out2, ref) < tol
    a = alpha[-1]
    assert rel(a[:256], golden2["rot900_g1/alpha_last_head"]) < tol
    assert rel(a[-256:], golden2["rot900_g1/alpha_last_tail"]) < tol
    st = torch.stack([a.double().sum(), a.double().abs().sum(), (a.double() ** 2).sum()]).cpu()
    assert rel(st, golden2["rot900_g1/alpha_last_stats"]) < (1e-4 if prec == "fp32" else 2e-2)


def test_rot900_ddim_trajectory_vs_reference_fixture(dev, golden2):
    from diffassemble_amd import Schedule, _lib
    lp = C.LOOPS2D_BIG[0]
    spec = C.by_name(lp["base"])
    case = C.build_case(spec)
    sch = Schedule(ODF.make_schedule(lp["T"]), dev)
    x0 = torch.from_numpy(golden2[f"{lp['name']}/x_init"]).to(dev)
    ref = golden2[f"{lp['name']}/imgs"]
    for prec, tol in (("fp32", TRAJ32), ("bf16", RTOLBF)):
        eng = make_engine(case, spec, prec, dev)
        plan = eng.plan(case["edge_index"], case["batch"])
        for use_graph in (True, False):
            traj, _ = eng.sample_loop(plan, sch, x0, case["feats"].to(dev), ratio=lp["ratio"], mean_type=_lib.MEAN_START_X,
                                      max_iters=lp["max_iters"], use_graph=use_graph)
            assert tuple(traj.shape) == ref.shape
            assert rel(traj, ref) < tol, (prec, use_graph)


def test_rot900_with_the_dual_slab_last_layer_kernel_subprocess(dev):
    """DA_ATTN_DUAL=1 (read once per process): the folded last layer through k_attn_dual (two query slabs per wave, generated
    asm regions, persistent workgroups; opt-in since the end of round 3) on the 900-piece fixture, its DDIM trajectory and
    the 64-puzzle determinism check."""
    from conftest import exp_env
    env = exp_env(DA_ATTN_DUAL="1")
    root = os.path.dirname(os.path.dirname(__file__))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k",
                        "test_rot900_dense_forward_vs_reference_fixture or test_rot900_ddim_trajectory_vs_reference_fixture"],
                       env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(root, "tests", "test_gpu_parity.py"), "-k",
                        "test_dense_path_is_deterministic_under_load"], env=env, capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and "2 passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_rot900_with_folds_off_subprocess(dev):
    """The layer-by-layer path (no algebraic folds) on the 900-piece fixture."""
    env = dict(os.environ, DA_DISABLE_FOLDS="3")          # da_config.disable_folds: 1 mlp.2 composition + 2 folded last layer
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", __file__, "-k", "test_rot900_dense_forward_vs_reference_fixture"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------- Exphander d = 539 (the scripted degree)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["auto", "off"])
def test_exo900_d539_forward_vs_reference_fixture(dev, golden2, monkeypatch, prec, mode):
    monkeypatch.setenv("DA_HYBRID", mode)
    spec = C.by_name("exo900_d539_v8")
    case = C.build_case(spec)
    ref = golden2["exo900_d539_v8/out"]
    eng = make_engine(case, spec, prec, dev)
    plan = eng.plan(case["edge_index"], case["batch"])
    assert plan.n_edges == 900 * 539 + 900 + 8 * 908 == 493264
    assert plan.hybrid == (1 if mode == "auto" else 0)
    tol = RTOL32 if prec == "fp32" else RTOLBF
    out = eng.forward(plan, case["x"].to(dev), case["t"].to(dev), case["feats"].to(dev))
    assert rel(out, ref) < tol
    if prec == "fp32":
        assert rel_elem(out, ref) < ELEM32
    out2, alpha = eng.forward(plan, case["x"].to(dev), case["t"].to(dev), None, return_alpha=True)
    assert rel(out2, ref) < tol
    assert rel(alpha[:256], golden2["exo900_d539_v8/alpha_last_head"]) < tol
    assert rel(alpha[-256:], golden2["exo900_d539_v8/alpha_last_tail"]) < tol


# ---------------------------------------------------------------------------- bf16 over a whole loop
@pytest.mark.parametrize("n,G", [(144, 2), (900, 2)])
def test_bf16_full_loop_drift_vs_fp32(dev, n, G):
    """100 DDIM steps (T = 100, ratio 1, START_X, noise_weight 1) in the benched bf16 mode against the fp32 HIP
    trajectory (which the fixtures pin to the reference at 5e-4): the drift of EVERY step's poses, relative to that
    step's max-abs pose, stays below RTOLBF = 8e-3 (measured 2.3 - 2.6e-3) and does not grow with the step count (the update is a contraction toward
    the predicted x0: rounding errors do not compound), and the final poses agree to 2 % of the final max-abs pose."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    side = int(round(n ** 0.5))
    sd = W.make_denoiser_state(100, 4, 4, seed=41, qk_gain=3.0)
    x, feats = W.make_inputs(G * n, 4, 1088, 41)
    ei, batch = W.collate([W.dense_edge_index(n, True)] * G, [n] * G)
    sch = Schedule(ODF.make_schedule(100), dev)
    trajs = {}
    for prec in ("fp32", "bf16"):
        eng = DenoiserEngine(sd, precision=prec, device=dev)
        plan = eng.plan(ei, batch)
        traj, _ = eng.sample_loop(plan, sch, x.to(dev), feats.to(dev), ratio=1, mean_type=_lib.MEAN_START_X, use_graph=True)
        trajs[prec] = traj.clone()
    a, b = trajs["bf16"].double(), trajs["fp32"].double()
    assert torch.isfinite(a).all()
    per_step = (a - b).abs().amax((1, 2)) / b.abs().amax((1, 2))
    print(f"bf16 loop drift {side}x{side}: max over steps {float(per_step.max()):.3e}, first {float(per_step[0]):.3e}, "
          f"last {float(per_step[-1]):.3e}, final max-abs drift {float((a[-1] - b[-1]).abs().max()):.3e}")
    assert float(per_step.max()) < RTOLBF
    assert float(per_step[-1]) < 2e-2


def test_bf16_full_loop_drift_config3_exophormer_hybrid(dev):
    """BASELINE config 3 in its benched mode: exophormer arch, V = 8 virtual nodes, two 900-piece
    Exphander graphs of the scripted degree d = 539 (hybrid path: adjacency-masked matrix-core attention + CSR remainder),
    T = 100 DDIM steps, bf16 against the fp32 HIP trajectory (which `test_exo900_d539_forward_vs_reference_fixture` ties to
    the reference): per-step drift relative to the step's max-abs pose, and the decisions a solved puzzle is judged by --
    greedy grid assignment and rotation quadrant -- compared piece by piece."""
    import numpy as np
    from diffassemble_amd import DenoiserEngine, Schedule, _lib, expander
    n, G, d, V = 900, 2, 539, 8
    sd = W.make_denoiser_state(100, 4, 4, arch="exophormer", virt_nodes=V, seed=43, qk_gain=3.0)
    x, feats = W.make_inputs(G * n, 4, 1088, 43)
    perms = expander.draw_permutations(n, G, np.random.default_rng(11)).to(dev)
    sch = Schedule(ODF.make_schedule(100), dev)
    trajs = {}
    for prec in ("fp32", "bf16"):
        eng = DenoiserEngine(sd, arch="exophormer", virt_nodes=V, precision=prec, device=dev)
        plan = eng.plan_expander(perms, d)
        assert plan.hybrid, "config 3 is benched on the hybrid path"
        traj, _ = eng.sample_loop(plan, sch, x.to(dev), feats.to(dev), ratio=1, mean_type=_lib.MEAN_START_X, use_graph=True)
        trajs[prec] = traj.clone()
    a, b = trajs["bf16"].double(), trajs["fp32"].double()
    assert torch.isfinite(a).all()
    per_step = (a - b).abs().amax((1, 2)) / b.abs().amax((1, 2))
    print(f"bf16 loop drift config 3 (exophormer d=539 V=8, hybrid): max over steps {float(per_step.max()):.3e}, first "
          f"{float(per_step[0]):.3e}, last {float(per_step[-1]):.3e}, final max-abs drift {float((a[-1] - b[-1]).abs().max()):.3e}")
    assert float(per_step.max()) < 8e-3            # measured 2.6e-3: 3x head-room, not 10x
    # rotation quadrant of every piece (cos, sin -> nearest of the four turns) identical
    qa = torch.atan2(a[-1][:, 3], a[-1][:, 2]).div(np.pi / 2).round().remainder(4)
    qb = torch.atan2(b[-1][:, 3], b[-1][:, 2]).div(np.pi / 2).round().remainder(4)
    margin = (torch.atan2(b[-1][:, 3], b[-1][:, 2]).div(np.pi / 2) - qb).abs()         # distance to the quadrant centre, in quarter turns
    decided = margin < 0.4                                                               # (an untrained model sits near boundaries)
    assert bool((qa[decided] == qb[decided]).all())


def test_bf16_full_loop_drift_config4_3d(dev):
    """BASELINE config 4 in its benched mode: 3D fragments (P = 20 per object, D = 832, SE(3) head), T = 300 / ratio 10
    = 30 DDIM steps, 16 objects, bf16 against the fp32 HIP trajectory (tied to the reference by test_ddim_3d_loop): the
    translations by max-abs drift, the rotations by GEODESIC ANGLE between the two unit quaternions (q and -q are one
    rotation) -- the quantity the quaternion / so3_scale algebra of k_ddim3d could amplify."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    P, G = 20, 16
    sd = W.make_denoiser_state(300, 7, None, D=832, hidden=256, variant="3d", seed=44, qk_gain=2.0)
    x, feats = W.make_inputs(G * P, 7, 768, 44)
    x[:, :4] = 0.0
    x[:, 0] = 1.0                                       # identity rotations + random translations (...double_diffusion.py:697-710)
    ei, batch = W.collate([W.dense_edge_index(P, True)] * G, [P] * G)
    sch = Schedule(ODF.make_schedule(300), dev)
    trajs = {}
    for prec in ("fp32", "bf16"):
        eng = DenoiserEngine(sd, variant="3d", precision=prec, device=dev)
        plan = eng.plan(ei, batch)
        traj, _ = eng.sample_loop(plan, sch, x.to(dev), feats.to(dev), ratio=10, mean_type=_lib.MEAN_START_X, use_graph=True)
        trajs[prec] = traj.clone()
    a, b = trajs["bf16"].double(), trajs["fp32"].double()
    assert torch.isfinite(a).all() and a.shape[0] == 30
    dt = (a[..., 4:] - b[..., 4:]).abs().amax((1, 2)) / b[..., 4:].abs().amax((1, 2))
    qa, qb = F.normalize(a[..., :4], dim=-1), F.normalize(b[..., :4], dim=-1)
    ang = 2 * torch.acos((qa * qb).sum(-1).abs().clamp(max=1.0))                     # geodesic angle, radians, per step per fragment
    print(f"bf16 loop drift config 4 (3D, T=300/10): translation max over steps {float(dt.max()):.3e} (last {float(dt[-1]):.3e}); "
          f"rotation geodesic max {float(ang.max()):.3e} rad (last step max {float(ang[-1].max()):.3e}, mean {float(ang[-1].mean()):.3e})")
    assert float(dt.max()) < 1.5e-2                     # measured 4.8e-3
    assert float(ang.max()) < 2e-3                      # measured 5.8e-4 rad = 0.03 degrees at every step of every fragment
    assert float((a[..., :4].norm(dim=-1) - 1).abs().max()) < 1e-3      # quaternions stay unit


FEAT_NOISE = 0.1          # std of the non-pose feature columns (probe: tests/tools/train_solver_probe.py)


def _train_solver(dev, sizes_train, steps=400, seed=0, lr=2e-3, G=8, train_precision=None, history=None):
    """Train the 2D denoiser with the HIP training path (da_train_forward/backward, START_X objective, Huber, as
    training_step does) on synthetic puzzles whose piece features carry the piece's true pose: a few hundred
    steps make it a solver, so that the sampling loop's end metric means something."""
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    torch.manual_seed(seed)
    m = GNN_Diffusion(steps=100, sampling="DDIM", inference_ratio=1, noise_weight=1.0, rotation=True,
                      model_mean_type=ModelMeanType.START_X, visual_pretrained=False, architecture="transformer")
    m = m.to(dev).train()
    if train_precision is not None:
        m.model.train_engine(dev).precision = train_precision       # "fp32" (exact products) | "bf16" (bf16 MFMA operands)
    opt = torch.optim.Adam([p for p in m.model.parameters()], lr=lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=steps)
    gen = torch.Generator(device=dev).manual_seed(seed)     # training batches are drawn ON the device (the host generator
    graphs = {}                                             # made this loop CPU-bound: 137 ms per step)
    for it in range(steps):                           # the training plan is cached on one device copy of each size's graph
        side = sizes_train[it % len(sizes_train)]
        x0, feats, ei, batch = _puzzle_batch(side, G, gen, FEAT_NOISE, with_graph=side not in graphs, device=dev)
        if side not in graphs:
            graphs[side] = (ei.to(dev), batch.to(dev))
        ei_d, batch_d = graphs[side]
        t = torch.randint(0, 100, (G,), generator=gen, device=dev)[batch_d]
        loss = m.p_losses(x0, t, loss_type="huber", cond=None, edge_index=ei_d, batch=batch_d, patch_feats=feats)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        sched.step()
        if history is not None:
            history.append(loss.detach())
    return m.eval(), float(loss.detach())


def _puzzle_batch(side, G, gen, feat_noise=1.0, with_graph=True, device="cpu"):
    """G puzzles of side x side pieces: ground-truth poses (grid xy in [-1, 1] + a random quarter-turn as (cos, sin)),
    features = N(0, 1) with the pose written (scaled) into the first four columns.  ``gen`` lives on ``device``."""
    n = side * side
    y = torch.linspace(-1, 1, side, device=device)
    grid = torch.stack(torch.meshgrid(y, y, indexing="xy"), -1).reshape(-1, 2)
    xs, fs = [], []
    for _ in range(G):
        perm = torch.randperm(n, generator=gen, device=device)
        k = torch.randint(0, 4, (n,), generator=gen, device=device).float() * (np.pi / 2)
        pose = torch.cat([grid[perm], torch.stack([torch.cos(k), torch.sin(k)], 1).round()], 1)
        f = torch.randn(n, 1088, generator=gen, device=device) * feat_noise
        f[:, :4] = pose * 4.0
        xs.append(pose)
        fs.append(f)
    ei, batch = W.collate([W.dense_edge_index(n, True)] * G, [n] * G) if with_graph else (None, None)
    return torch.cat(xs), torch.cat(fs), ei, batch


def test_end_metric_bf16_equals_fp32_on_a_trained_solver(dev):
    """End-metric parity of the perf mode: train a solver on the GPU, then run validation-style sampling
    (T = 100 DDIM loop -> greedy assignment -> piece accuracy + rotation test) in fp32 and bf16 on fresh 12x12 and
    30x30 puzzles.  The decisions -- assigned cell and rotation test of every piece -- must be IDENTICAL and the
    fp32 accuracy must be high enough for the comparison to mean something."""
    import math
    from diffassemble_amd.engine import greedy_assign
    m, last_loss = _train_solver(dev, [6, 12, 12, 16], steps=1500)
    gen = torch.Generator().manual_seed(1234)
    for side, G in ((12, 4), (30, 2)):
        n = side * side
        x_gt, feats, ei, batch = _puzzle_batch(side, G, gen, FEAT_NOISE)
        y = torch.linspace(-1, 1, side)
        grid = torch.stack(torch.meshgrid(y, y, indexing="xy"), -1).reshape(-1, 2).repeat(G, 1).to(dev)
        ptr = torch.arange(0, (G + 1) * n, n, dtype=torch.int32, device=dev)
        res = {}
        x_init = torch.randn(x_gt.shape, generator=gen)
        for prec in ("fp32", "bf16"):
            m.model.precision = prec
            _orig = torch.randn
            torch.randn = lambda *a, **k: x_init.to(dev)
            try:
                imgs, _ = m.p_sample_loop(tuple(x_gt.shape), None, ei.to(dev), batch.to(dev), patch_feats=feats.to(dev))
            finally:
                torch.randn = _orig
            img = imgs[-1]
            ass = greedy_assign(img[:, :2].contiguous(), grid, ptr, ptr)
            cells = torch.empty(G * n, dtype=torch.int64, device=dev)
            rows = ass[:, 0] + torch.arange(G, device=dev).repeat_interleave(n) * n
            cells[rows] = ass[:, 1]
            rot_ok = torch.cosine_similarity(img[:, 2:], x_gt[:, 2:].to(dev)) > math.cos(math.pi / 4)
            res[prec] = (img, cells, rot_ok)
        gt_ass = greedy_assign(x_gt[:, :2].contiguous().to(dev), grid, ptr, ptr)
        gt_cells = torch.empty(G * n, dtype=torch.int64, device=dev)
        gt_cells[gt_ass[:, 0] + torch.arange(G, device=dev).repeat_interleave(n) * n] = gt_ass[:, 1]
        acc = {p: float(((res[p][1] == gt_cells) & res[p][2]).float().mean()) for p in res}
        drift = float((res["bf16"][0] - res["fp32"][0]).abs().max())
        print(f"end metric {side}x{side}: piece accuracy fp32 {acc['fp32']:.4f} bf16 {acc['bf16']:.4f}, "
              f"max final-pose drift {drift:.3e} (half a cell = {1.0 / (side - 1):.3e}), train loss {last_loss:.3e}")
        assert acc["fp32"] > 0.9, "the solver did not train: the comparison would be about chance-level assignments"
        assert torch.equal(res["fp32"][1], res["bf16"][1]), "bf16 changed a piece's assigned cell"
        assert torch.equal(res["fp32"][2], res["bf16"][2]), "bf16 changed a rotation decision"
        assert drift < 0.25 / (side - 1)                     # a quarter of the half-cell margin


def test_training_curve_bf16_operand_mode_tracks_fp32(dev):
    """The bf16-operand TRAINING mode (the benched default of `bench.py --config 5`) against the reference's fp32 arithmetic over
    a whole run, not one step: two solvers trained from the same seed on the same 1000 Batches -- exact fp32
    products vs bf16 matrix-core operands -- must reach the same loss level (mean of the last 50 steps within 5 %; single-Batch
    losses are noisier than that) and the same end metric (piece accuracy of a T = 100 sampling run on fresh 12x12 puzzles, both
    above 0.9 and within 0.02 of each other)."""
    import math
    from diffassemble_amd.engine import greedy_assign
    res = {}
    for prec in ("fp32", "bf16"):
        hist = []
        m, _ = _train_solver(dev, [6, 12, 12, 16], steps=1000, train_precision=prec, history=hist)
        losses = torch.stack(hist).float().cpu()
        gen = torch.Generator().manual_seed(99)
        side, G = 12, 4
        n = side * side
        x_gt, feats, ei, batch = _puzzle_batch(side, G, gen, FEAT_NOISE)
        y = torch.linspace(-1, 1, side)
        grid = torch.stack(torch.meshgrid(y, y, indexing="xy"), -1).reshape(-1, 2).repeat(G, 1).to(dev)
        ptr = torch.arange(0, (G + 1) * n, n, dtype=torch.int32, device=dev)
        x_init = torch.randn(x_gt.shape, generator=gen)
        m.model.precision = "fp32"
        _orig = torch.randn
        torch.randn = lambda *a, **k: x_init.to(dev)
        try:
            imgs, _ = m.p_sample_loop(tuple(x_gt.shape), None, ei.to(dev), batch.to(dev), patch_feats=feats.to(dev))
        finally:
            torch.randn = _orig
        img = imgs[-1]

        def cells_of(pos):
            ass = greedy_assign(pos[:, :2].contiguous(), grid, ptr, ptr)
            c = torch.empty(G * n, dtype=torch.int64, device=dev)
            c[ass[:, 0] + torch.arange(G, device=dev).repeat_interleave(n) * n] = ass[:, 1]
            return c
        rot_ok = torch.cosine_similarity(img[:, 2:], x_gt[:, 2:].to(dev)) > math.cos(math.pi / 4)
        acc = float(((cells_of(img) == cells_of(x_gt.to(dev))) & rot_ok).float().mean())
        res[prec] = (float(losses[-50:].mean()), float(losses[:50].mean()), acc)
        print(f"training curve {prec}: mean loss first 50 steps {res[prec][1]:.4e}, last 50 {res[prec][0]:.4e}, piece accuracy {acc:.4f}")
    assert res["fp32"][0] < 0.2 * res["fp32"][1], "the run did not train"
    assert abs(res["bf16"][0] - res["fp32"][0]) < 0.05 * res["fp32"][0], res
    assert res["fp32"][2] > 0.9 and res["bf16"][2] > 0.9 and abs(res["fp32"][2] - res["bf16"][2]) <= 0.02, res


# ---------------------------------------------------------------------------- caches must not go stale
def _module_for(spec, case, dev, prec="fp32"):
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    m = GNN_Diffusion(steps=spec["steps"], sampling="DDIM", inference_ratio=10, noise_weight=1.0, rotation=True,
                      model_mean_type=ModelMeanType.START_X, architecture=spec["arch"], virt_nodes=spec["V"] or 4,
                      visual_pretrained=False)
    m.model.load_state_dict(case["sd"], strict=False)
    m = m.to(dev).eval()
    m.model.precision = prec
    return m


def test_two_same_shaped_batches_back_to_back_do_not_share_plan_or_features(dev):
    """A Lightning loop frees Batch i before Batch i+1 reaches the device, so the caching allocator hands the new
    same-shaped tensors the SAME addresses (``_version`` 0): a fresh random expander per sample
    (puzzle_dataset.py:194-212) and fresh piece features must be planned / staged again.  Checks
    forward_with_feats, p_sample_ddim and the training forward against the oracle for both Batches."""
    spec = dict(name="stale", sizes=[64, 36], c=4, graph="regular6", arch="exophormer", V=4, steps=300, seed=5, qk_gain=3.0)
    case = C.build_case(spec)
    m = _module_for(spec, case, dev)
    sizes = spec["sizes"]
    N = sum(sizes)
    t = case["t"]
    outs, ptrs = [], []
    for trial in range(2):
        rng = np.random.default_rng(100 + trial)
        ei_cpu, batch_cpu = W.collate([W.random_regular_edge_index(n, 6, rng) for n in sizes], sizes)
        x_cpu, feats_cpu = W.make_inputs(N, 4, 1088, 200 + trial)
        ref, _ = OD.eff_gat_forward_with_feats(case["sd"], x_cpu, t, ei_cpu, feats_cpu, batch_cpu, "exophormer", 4)
        ei, batch, x, feats = ei_cpu.to(dev), batch_cpu.to(dev), x_cpu.to(dev), feats_cpu.to(dev)
        ptrs.append((ei.data_ptr(), feats.data_ptr()))
        out = m.forward_with_feats(x, t.to(dev), None, ei, feats, batch)
        assert rel(out, ref) < RTOL32, f"Batch {trial}: forward_with_feats used a stale plan or stale features"
        # p_sample_ddim on the same Batch (cache hit is fine here), attention path
        m.return_attentions = True
        prev, att = m.p_sample_ddim(x, t.to(dev), 290, None, ei, feats, batch)
        m.return_attentions = False
        assert torch.equal(att[0][0][:, : ei.shape[1]].cpu(), ei_cpu)
        # the training forward has its own plan cache
        m.train()
        with torch.enable_grad():
            pred, _ = m.model.forward_with_feats(x, t.to(dev), None, ei, feats, batch)
        m.eval()
        assert rel(pred, ref) < RTOL32, f"Batch {trial}: training forward used a stale plan"
        outs.append(out.clone())
        del ei, batch, x, feats, out, prev, att, pred                  # Batch i dies before Batch i+1 is moved
    # (with the fix the caches HOLD the keyed tensors, so the allocator can no longer hand Batch 1 the addresses of Batch 0;
    # tests/test_host.py::test_cache_key_holds_its_tensors_and_sees_in_place_edits pins that property)
    assert rel(outs[0], outs[1]) > 1e-2                                # the two Batches really differ


def test_new_patch_feats_same_address_are_restaged_in_the_step_by_step_path(dev):
    spec = C.by_name("rot144_g1")
    case = C.build_case(spec)
    m = _module_for(spec, case, dev)
    ei, batch = case["edge_index"].to(dev), case["batch"].to(dev)
    t = torch.full((144,), 50, dtype=torch.long)
    ptrs = []
    for trial in range(2):
        x_cpu, feats_cpu = W.make_inputs(144, 4, 1088, 300 + trial)
        ref, _ = OD.eff_gat_forward_with_feats(case["sd"], x_cpu, t, case["edge_index"], feats_cpu, case["batch"])
        feats = feats_cpu.to(dev)
        ptrs.append(feats.data_ptr())
        m.sampling, m.eta = "DDIM", 0
        m.return_attentions = True                                     # the non-fast path: no set_features of its own
        out, _ = m.forward_with_feats(x_cpu.to(dev), t.to(dev), None, ei, feats, batch, return_attentions=True)
        assert rel(out, ref) < RTOL32, trial
        del feats
    # in-place edit of a live tensor bumps _version: also restaged
    feats = W.make_inputs(144, 4, 1088, 300)[1].to(dev)
    a = m.forward_with_feats(case["x"].to(dev), t.to(dev), None, ei, feats, batch)
    feats.mul_(0.5)
    ref, _ = OD.eff_gat_forward_with_feats(case["sd"], case["x"], t, case["edge_index"], feats.cpu(), case["batch"])
    b = m.forward_with_feats(case["x"].to(dev), t.to(dev), None, ei, feats, batch)
    assert rel(b, ref) < RTOL32 and rel(a, b) > 1e-3


@pytest.mark.parametrize("fused", [True, False])
def test_packed_inference_weights_follow_the_optimizer(dev, monkeypatch, fused):
    """engine() -> optimizer step -> engine(): the second inference must run on the UPDATED weights (the fused
    Adafactor writes the flat parameter buffer through a raw pointer, which bumps no tensor version)."""
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion as _G
    monkeypatch.setattr(_G, "fused_optimizer", fused, raising=False)
    spec = C.by_name("k36_loop_sharp")
    case = C.build_case(spec)
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    m = GNN_Diffusion(steps=50, sampling="DDIM", model_mean_type=ModelMeanType.START_X, visual_pretrained=False)
    m.model.load_state_dict(case["sd"], strict=False)
    m.model.visual_backbone = None
    m = m.to(dev)
    m.model.precision = "fp32"
    args = (case["x"].to(dev), case["t"].to(dev), None, case["edge_index"].to(dev), case["feats"].to(dev), case["batch"].to(dev))
    m.train()
    te = m.model.train_engine(dev)                                       # binds the parameters to the flat buffer
    opt = m.configure_optimizers()
    m.eval()
    before = m.forward_with_feats(*args).clone()                         # packs the inference engine NOW
    m.train()
    for _ in range(3):
        loss = m.p_losses(args[0], args[1], loss_type="huber", cond=None, edge_index=args[3], batch=args[5], patch_feats=args[4])
        opt.zero_grad()
        loss.backward()
        opt.step()
    m.eval()
    after = m.forward_with_feats(*args)
    sd_now = {k: v.detach().cpu() for k, v in m.model.state_dict().items() if k in case["sd"]}
    ref, _ = OD.eff_gat_forward_with_feats(sd_now, case["x"], case["t"], case["edge_index"], case["feats"], case["batch"])
    assert rel(after, ref) < RTOL32, "inference ran on weights from before the optimizer steps"
    assert rel(after, before) > 1e-4
    assert te is m.model.train_engine(dev)


# ---------------------------------------------------------------------------- greedy assignment vs the reference function
@pytest.mark.parametrize("g", C.GREEDY, ids=lambda s: s["name"])
def test_greedy_assignment_vs_reference_torchscript(dev, golden2, g):
    """da_greedy_assign against OUTPUTS of the reference's own TorchScript greedy_cost_assignment
    (spatial_diffusion.py:179-216, run by make_golden_v2.py): index-exact, in assignment order, distance column too."""
    from diffassemble_amd.model.spatial_diffusion import greedy_cost_assignment
    pos1, pos2 = C.greedy_inputs(g)
    exp = torch.from_numpy(golden2[f"{g['name']}/assignment"])
    got = greedy_cost_assignment(pos1.to(dev), pos2.to(dev)).cpu()
    assert got.shape == exp.shape
    assert torch.equal(got, exp)


def test_greedy_assignment_survives_non_finite_poses(dev):
    """A diverged sample (NaN / Inf poses) must not fault: every row still gets a distinct cell."""
    from diffassemble_amd.engine import greedy_assign
    y = torch.linspace(-1, 1, 6)
    grid = torch.stack(torch.meshgrid(y, y, indexing="xy"), -1).reshape(-1, 2).to(dev)
    pos = grid.flip(0).clone()
    pos[3] = float("nan")
    pos[7, 0] = float("inf")
    out = greedy_assign(pos, grid).cpu()
    torch.cuda.synchronize()
    assert sorted(out[:, 0].tolist()) == list(range(36)) and sorted(out[:, 1].tolist()) == list(range(36))
    allnan = torch.full((36, 2), float("nan"), device=dev)
    out = greedy_assign(allnan, grid).cpu()
    assert sorted(out[:, 0].tolist()) == list(range(36)) and sorted(out[:, 1].tolist()) == list(range(36))


# ---------------------------------------------------------------------------- 3D module surface (train_3d.py's callers)
def test_3d_module_p_sample_loop_and_eval_hooks(dev):
    """The reference-shaped 3D module (spatial_diffusion_3d_test_double_diffusion.GNN_Diffusion): ``p_sample_loop`` driven
    the way ``test_step`` drives it reproduces the reference's own 10-step SE(3) trajectory (fixture ddim3d_t300, modulo
    q == -q), and ``validation_step`` scores the final poses with the metrics of utils_3d.py (checked against the
    oracle restatement, which the reference's functions pin)."""
    from types import SimpleNamespace
    from diffassemble_amd.model.spatial_diffusion_3d_test_double_diffusion import GNN_Diffusion, ModelMeanType
    from oracle import metrics3d as OM
    golden = C.load_golden()
    lp = C.LOOPS3D[0]
    spec = C.by_name(lp["base"])
    case = C.build_case(spec, "3d")
    m = GNN_Diffusion(steps=lp["T"], sampling="DDIM", inference_ratio=lp["ratio"], noise_weight=lp["noise_weight"],
                      model_mean_type=ModelMeanType.START_X, backbone="vn_dgcnn", architecture=spec["arch"])
    missing, unexpected = m.model.load_state_dict(case["sd"], strict=False)
    assert not unexpected and all(k.startswith("pcd_backbone.") for k in missing)      # the fixture feeds pcd_feats
    m = m.to(dev).eval()
    m.model.precision = "fp32"
    x0 = torch.from_numpy(golden[f"{lp['name']}/x_init"])
    ref = torch.from_numpy(golden[f"{lp['name']}/imgs"])
    P = x0.shape[0]
    _orig = torch.randn
    torch.randn = lambda *a, **k: x0[:, 4:].to(dev)                     # the loop's own translation noise draw
    try:
        imgs, atts = m.p_sample_loop((P, 7), None, case["edge_index"].to(dev), case["batch"].to(dev), pcd_feats=case["feats"].to(dev))
    finally:
        torch.randn = _orig
    assert len(imgs) == 30 and len(atts) == 30
    got = torch.stack(imgs[: ref.shape[0]]).cpu()
    assert rel(got[..., 4:], ref[..., 4:]) < TRAJ32
    dq = torch.minimum((got[..., :4] - ref[..., :4]).abs().amax(-1), (got[..., :4] + ref[..., :4]).abs().amax(-1))
    assert float(dq.max()) < TRAJ32
    # validation_step: sampling loop + metrics per object (three objects of 20 / 7 / 13 parts)
    rng = np.random.default_rng(3)
    pcds = torch.from_numpy(rng.standard_normal((P, 200, 3)).astype(np.float32)) * 0.3
    gt = case["x"].clone()
    batch = SimpleNamespace(x=gt.to(dev), pcds=pcds.to(dev), edge_index=case["edge_index"].to(dev), batch=case["batch"].to(dev),
                            pcd_feats=case["feats"].to(dev), category=["everyday", "artifact", "everyday"])
    m.initialize_torchmetrics(["everyday", "artifact"])
    torch.manual_seed(0)
    final = m.validation_step(batch, 0).cpu()
    assert final.shape == (P, 7) and torch.isfinite(final).all()
    exp = {"rmse_t": [], "part_acc": []}
    for g, cat in enumerate(batch.category):
        idx = (case["batch"] == g).nonzero().flatten()
        if cat == "everyday":
            exp["rmse_t"].append(float(OM.trans_rmse(final[idx, 4:], gt[idx, 4:])))
            exp["part_acc"].append(float(OM.part_accuracy(pcds[idx], final[idx, 4:], gt[idx, 4:], final[idx, :4], gt[idx, :4])))
    assert abs(float(m.metrics["rmse_t_everyday"].compute()) - np.mean(exp["rmse_t"])) < 1e-4
    assert abs(float(m.metrics["part_acc_everyday"].compute()) - np.mean(exp["part_acc"])) < 1e-6
    m.validation_epoch_end([])
    out = m.predict_step(batch, 0)
    assert len(out[0]) == 30


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_exo900_d539_banded_expander_plan_vs_reference_fixture(dev, golden2, prec):
    """The SAME 900-piece Exphander d = 539 / exophormer V = 8 case as above, planned from the generator's permutation
    (graph_plan.expander_plan, banded layout: slots in position order, shared slot-space adjacency bits, block classes --
    the masked optimistic kernel skips the empty 32 x 32 blocks and runs the full ones un-masked) instead of from the edge
    list: the reference's own output must come out (the permutation is the first n senders of generate_random_regular_graph's
    edge list, puzzle_dataset.py:136-147), in the exact mode through k_attn_dense<MASKED> with the slot -> node map."""
    spec = C.by_name("exo900_d539_v8")
    case = C.build_case(spec)
    ref = golden2["exo900_d539_v8/out"]
    eng = make_engine(case, spec, prec, dev)
    perm = case["edge_index"][0, :900].clone()
    assert sorted(perm.tolist()) == list(range(900))
    plan = eng.plan_expander(perm[None].to(dev), 539)
    assert plan.hybrid == 1 and plan.slot_node is not None and plan.blk_class is not None and plan.n_edges == 493264
    out = eng.forward(plan, case["x"].to(dev), case["t"].to(dev), case["feats"].to(dev))
    assert rel(out, ref) < (RTOL32 if prec == "fp32" else RTOLBF)
    if prec == "fp32":
        assert rel_elem(out, ref) < ELEM32
    # ... and equals the plan built from the edge list (natural slot order) to rounding
    out_n = eng.forward(eng.plan(case["edge_index"], case["batch"]), case["x"].to(dev), case["t"].to(dev), case["feats"].to(dev))
    assert rel(out, out_n) < (2e-5 if prec == "fp32" else RTOLBF)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("n,d,V,G,arch", [(900, 90, 8, 3, "exophormer"), (900, 539, 8, 2, "exophormer"), (320, 81, 4, 3, "exophormer"),
                                           (256, 201, 0, 4, "transformer"), (1000, 31, 0, 1, "transformer")])
def test_banded_expander_plans_equal_natural_plans(dev, monkeypatch, n, d, V, G, arch, prec):
    """expander_plan in its two layouts (DA_EXPANDER_LAYOUT natural / banded) on several Batches: forwards and a 4-step
    DDIM loop agree (fp32: 2e-5; bf16: within the bf16 bound), sparse degree (most key tiles skipped), odd degree (the
    antipodal matching: a second diagonal of partial blocks), sizes that are not multiples of 32 / 64, no virtual nodes."""
    monkeypatch.setenv("DA_HYBRID", "force")
    from diffassemble_amd import DenoiserEngine, Schedule, _lib, expander
    from oracle import diffusion as ODF
    from oracle import weights as OW
    sd = OW.make_denoiser_state(100, 4, 4, D=1152, hidden=128, variant="2d", arch=arch, virt_nodes=V, seed=n + d, qk_gain=3.0)
    eng = DenoiserEngine(sd, variant="2d", arch=arch, virt_nodes=V, precision=prec, device=dev)
    perms = expander.draw_permutations(n, G, np.random.default_rng(d)).to(dev)
    g = torch.Generator().manual_seed(n)
    x = torch.randn(G * n, 4, generator=g).to(dev)
    feats = torch.randn(G * n, 1088, generator=g).to(dev)
    t = torch.randint(0, 100, (G,), generator=g).repeat_interleave(n).to(dev)
    outs, trajs = [], []
    for layout in ("natural", "banded"):
        monkeypatch.setenv("DA_EXPANDER_LAYOUT", layout)
        plan = eng.plan_expander(perms, d)
        assert plan.hybrid == 1 and (plan.slot_node is not None) == (layout == "banded")
        outs.append(eng.forward(plan, x, t, feats).clone())
        sch = Schedule(ODF.make_schedule(100), dev)
        traj, _ = eng.sample_loop(plan, sch, x, feats, ratio=25, mean_type=_lib.MEAN_START_X, use_graph=True)
        trajs.append(traj.clone())
    tol = 2e-5 if prec == "fp32" else RTOLBF
    assert torch.isfinite(outs[1]).all()
    assert rel(outs[1], outs[0]) < tol
    assert rel(trajs[1], trajs[0]) < (1e-4 if prec == "fp32" else 2 * RTOLBF)


@pytest.mark.parametrize("layout", ["natural", "banded"])
@pytest.mark.parametrize("n,d,V,G", [(900, 539, 8, 3), (900, 90, 8, 2), (320, 81, 4, 2), (256, 200, 0, 4)])
def test_expander_mask_kernel_equals_host_closed_form(dev, n, d, V, G, layout, monkeypatch):
    """graph_plan.expander_plan on the device (da_expander_mask: inverse permutations + bit rows, two launches; the
    shape-only parts cached) against the same function on host tensors (torch closed form, itself checked against the plan
    built from the reference generator's edge list in tests/test_host.py): every plan array bit for bit, odd degrees and
    a second Batch of the same shape included."""
    monkeypatch.setenv("DIFFASSEMBLE_HYBRID", "force")
    monkeypatch.setenv("DA_EXPANDER_LAYOUT", layout)         # natural: da_expander_mask builds the bit rows; banded: index maps only
    from diffassemble_amd import expander, graph_plan as GP
    for seed in (0, 1):
        perms = expander.draw_permutations(n, G, np.random.default_rng(seed))
        a = GP.expander_plan(perms.to(dev), d, virt_nodes=V)
        b = GP.expander_plan(perms, d, virt_nodes=V)
        assert a.hybrid == b.hybrid == 1 and a.n_edges == b.n_edges and a.n_pad == b.n_pad
        for f in ("mask", "mask_ptr", "row_map", "pad_ptr", "graph_ptr", "irr_row_ptr", "irr_col_src") + (("slot_node", "blk_class", "blk_class_ptr") if layout == "banded" else ()):
            assert torch.equal(getattr(a, f).cpu(), getattr(b, f)), (f, seed)


@pytest.mark.parametrize("n,d,G", [(257, 100, 3), (320, 319, 2), (1000, 31, 1)])
def test_expander_mask_kernel_odd_sizes(dev, n, d, G, monkeypatch):
    """Node counts that are not multiples of 8 / 64 (ragged last mask byte), the densest degree (n - 1), odd degrees (the
    antipodal matching needs an even n) and a sparse one under the forced hybrid mode: device plan == host closed form."""
    monkeypatch.setenv("DIFFASSEMBLE_HYBRID", "force")
    monkeypatch.setenv("DA_EXPANDER_LAYOUT", "natural")
    from diffassemble_amd import expander, graph_plan as GP
    perms = expander.draw_permutations(n, G, np.random.default_rng(n + d))
    a, b = GP.expander_plan(perms.to(dev), d, virt_nodes=4), GP.expander_plan(perms, d, virt_nodes=4)
    assert a.hybrid == b.hybrid
    if a.hybrid:
        assert torch.equal(a.mask.cpu(), b.mask) and torch.equal(a.mask_ptr.cpu(), b.mask_ptr)
    else:
        assert torch.equal(a.row_ptr.cpu(), b.row_ptr) and torch.equal(a.col_src.cpu(), b.col_src)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("sizes,loops", [([144] * 4, True), ([36, 144, 100, 64, 9], False), ([900, 900], True)],
                         ids=["4x144", "ragged5_noloops", "2x900"])
def test_two_branch_loop_equals_one_branch(dev, monkeypatch, prec, sizes, loops):
    """da_sample_loop_pair (two half Batches as parallel branches of one hipGraph; the default from 40 000 nodes up): the
    final poses of every puzzle are BIT-identical to the one-branch loop over the whole Batch -- same kernels, same
    reduction orders, the halves only share the weights -- for even and odd splits, ragged sizes, graphs without
    self loops, restaged and re-used features, and on replay of the cached graph."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    N = sum(sizes)
    sd = W.make_denoiser_state(100, 4, 4, seed=43, qk_gain=3.0)
    x, feats = W.make_inputs(N, 4, 1088, 43)
    ei, batch = W.collate([W.dense_edge_index(n, loops) for n in sizes], sizes)
    sch = Schedule(ODF.make_schedule(100), dev)
    eng = DenoiserEngine(sd, precision=prec, device=dev)
    plan = eng.plan(ei, batch)
    xd, fd = x.to(dev), feats.to(dev)
    kw = dict(ratio=10, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True)
    monkeypatch.setenv("DA_TWO_BRANCH", "0")
    _, one = eng.sample_loop(plan, sch, xd, fd, **kw)
    one = one.clone()
    monkeypatch.setenv("DA_TWO_BRANCH", "1")
    monkeypatch.setattr(eng, "two_branch_min_graphs", 2, raising=False)
    assert eng._two_branch(plan, False, True)
    _, two = eng.sample_loop(plan, sch, xd, fd, **kw)
    assert torch.equal(two, one)
    two.zero_()
    _, again = eng.sample_loop(plan, sch, xd, fd, restage=False, **kw)          # cached graph, staged features
    assert torch.equal(again, one)
    # new features at the same address are restaged even with restage=False (the version counter moved)
    fd.mul_(0.5)
    _, half = eng.sample_loop(plan, sch, xd, fd, restage=False, **kw)
    half = half.clone()
    monkeypatch.setenv("DA_TWO_BRANCH", "0")
    _, ref = eng.sample_loop(plan, sch, xd, fd, **kw)
    assert torch.equal(half, ref) and not torch.equal(half, one)
    # kept trajectories go through the pair loop too (each half writes its rows of every iteration): bit-identical again
    monkeypatch.setenv("DA_TWO_BRANCH", "0")
    t1, f1 = eng.sample_loop(plan, sch, xd, fd, ratio=10, mean_type=_lib.MEAN_START_X, keep_trajectory=True, use_graph=True)
    t1, f1 = t1.clone(), f1.clone()
    monkeypatch.setenv("DA_TWO_BRANCH", "1")
    assert eng._two_branch(plan, True, True)
    t2, f2 = eng.sample_loop(plan, sch, xd, fd, ratio=10, mean_type=_lib.MEAN_START_X, keep_trajectory=True, use_graph=True)
    assert t2.shape == t1.shape and torch.equal(t2, t1) and torch.equal(f2, f1) and torch.equal(t2[-1], f2)
    # eager loops and small Batches keep the one-branch path
    assert not eng._two_branch(plan, False, False)
    monkeypatch.setattr(eng, "two_branch_min_graphs", 64, raising=False)
    assert not eng._two_branch(plan, False, True)
    # default ("auto"): by node count
    monkeypatch.delenv("DA_TWO_BRANCH")
    assert not eng._two_branch(plan, False, True)
    monkeypatch.setattr(eng, "two_branch_min_nodes", 10, raising=False)
    assert eng._two_branch(plan, False, True) and eng._two_branch(plan, True, True)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("kind", ["cfg", "eta", "ddpm", "cfg_eta"])
def test_two_branch_loop_with_the_other_samplers_equals_one_branch(dev, monkeypatch, prec, kind):
    """da_sample_loop_pair_ex (ABI 17): classifier-free guidance, DDIM eta > 0 and DDPM on the two-branch loop -- both halves read
    their row ranges of ONE [n_iters, N, c] noise draw, the unconditional pass uses each half's own zero-feature projection --
    give the one-branch da_sample_loop_ex poses and trajectory BIT for bit (uneven split, ragged sizes), also on replay of
    the cached graph with a fresh draw."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    sizes = [144, 100, 64, 36, 144]
    N = sum(sizes)
    sd = W.make_denoiser_state(100, 4, 4, seed=47, qk_gain=3.0)
    x, feats = W.make_inputs(N, 4, 1088, 47)
    ei, batch = W.collate([W.dense_edge_index(n, True) for n in sizes], sizes)
    sch = Schedule(ODF.make_schedule(100), dev)
    eng = DenoiserEngine(sd, precision=prec, device=dev)
    plan = eng.plan(ei, batch)
    xd, fd = x.to(dev), feats.to(dev)
    n_it = 10
    gen = torch.Generator(device="cpu").manual_seed(5)
    kw = dict(ratio=10, mean_type=_lib.MEAN_EPSILON if kind == "ddpm" else _lib.MEAN_START_X, keep_trajectory=True, use_graph=True)
    if kind in ("cfg", "cfg_eta"):
        kw["cfg_w"] = 1.5
    if kind in ("eta", "cfg_eta"):
        kw["eta"] = 0.7
    if kind == "ddpm":
        kw["sampler"] = "DDPM"
    for rep in range(2):
        noise = None if kind == "cfg" else torch.randn((n_it, N, 4), generator=gen).to(dev)
        monkeypatch.setenv("DA_TWO_BRANCH", "0")
        t1, f1 = eng.sample_loop(plan, sch, xd, fd, noise=noise, **kw)
        t1, f1 = t1.clone(), f1.clone()
        monkeypatch.setenv("DA_TWO_BRANCH", "1")
        monkeypatch.setattr(eng, "two_branch_min_graphs", 2, raising=False)
        assert eng._two_branch(plan, True, True)
        t2, f2 = eng.sample_loop(plan, sch, xd, fd, noise=noise, **kw)
        assert torch.isfinite(f2).all() and torch.equal(t2, t1) and torch.equal(f2, f1)
    if kind != "cfg":       # a different draw moves the poses (the buffer really is read)
        t3, f3 = eng.sample_loop(plan, sch, xd, fd, noise=torch.randn((n_it, N, 4), generator=gen).to(dev), **kw)
        assert not torch.equal(f3, f1)


_TAIL_SCRIPT = r"""
import sys, os, torch
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
from oracle import weights as W, diffusion as ODF
from diffassemble_amd import DenoiserEngine, Schedule, _lib
dev = torch.device("cuda:0")
sizes = [int(v) for v in sys.argv[3].split(",")]
N = sum(sizes)
sd = W.make_denoiser_state(100, 4, 4, seed=47, qk_gain=2.0)
x, feats = W.make_inputs(N, 4, 1088, 47)
ei, batch = W.collate([W.dense_edge_index(n, True) for n in sizes], sizes)
eng = DenoiserEngine(sd, precision="bf16", device=dev)
plan = eng.plan(ei, batch)
out = eng.forward(plan, x.to(dev), torch.full((N,), 37, dtype=torch.int64, device=dev), feats.to(dev))
sch = Schedule(ODF.make_schedule(100), dev)
_, xf = eng.sample_loop(plan, sch, x.to(dev), feats.to(dev), ratio=10, mean_type=_lib.MEAN_START_X, keep_trajectory=False, use_graph=True)
torch.save({"out": out.cpu(), "xf": xf.cpu()}, sys.argv[2])
"""


@pytest.mark.parametrize("sizes", ["900", "33,64,31,100,1", ",".join(["144"] * 255 + ["150"])], ids=["900", "ragged", "36870_rows"])
def test_tail_fused_kernel_vs_three_kernel_tail(dev, tmp_path, sizes):
    """k_tail_fused (the folded tail as one MFMA kernel, the bf16 default) against the three-kernel tail it replaces
    (DA_DISABLE_FOLDS=16, read once per process -> two subprocesses): one forward and a 10-step DDIM loop whose update the
    kernel applies itself; row counts that are not multiples of the 32-row slab, and more than 32 768 rows (two slabs per
    wave, the shape of the benched Batches).  The two paths differ only in where the
    32-wide pre-activation is rounded to bf16 (the fused kernel keeps it in fp32), so they agree far inside the bf16
    tolerance of the parity tests."""
    root = os.path.dirname(os.path.dirname(__file__))
    res = {}
    for flag in ("1", "0"):
        path = str(tmp_path / f"tail_{flag}.pt")
        env = dict(os.environ, DA_DISABLE_FOLDS="0" if flag == "1" else "16")
        r = subprocess.run([sys.executable, "-c", _TAIL_SCRIPT, root, path, sizes], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        res[flag] = torch.load(path)
    assert not torch.equal(res["1"]["out"], res["0"]["out"])          # the switch really selects another path
    assert rel(res["1"]["out"], res["0"]["out"]) < 5e-3
    assert rel(res["1"]["xf"], res["0"]["xf"]) < 5e-3


def test_two_branch_loop_as_one_graph_subprocess():
    """DA_PAIR_SPLIT=0: the two branches as parallel branches of ONE hipGraph (the form up to round 5; the default launches two graphs on
    two streams between a fork and a join event): the same bit-identity suite."""
    import subprocess
    env = dict(os.environ, DA_PAIR_SPLIT="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k", "two_branch_loop and not subprocess"],
                       env=env, capture_output=True, text=True, timeout=1200, cwd=root)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
