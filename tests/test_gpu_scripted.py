"""The reference's SCRIPTED workload (singularity/gianscarpe/train_celeba_rot.sh:4-15: ragged Batches of 8 puzzles, sides
6 .. 20, exophormer with 8 virtual nodes, Exphander degree 60 %, T = 300 / inference_ratio 10) -- what `bench.py --config scripted`
times -- against the CPU oracle: forward (fp32 at the 1e-4 bound, bf16 at the bf16 bound), the first DDIM steps of the loop, and one
training step's loss and gradients.  Three Batches: a mixed one and one of small puzzles only at the scripted degree (both hybrid since round 5: adjacency-masked
matrix-core attention over every graph + CSR remainder -- graph_plan._hybrid_worth_it), and a sparse one (degree 0.5 %: the
edge-list kernel k_attn_csr is the product path -- the `bench.py --config csr` regime)."""
import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import diffusion as ODF
from oracle import weights as W

pytestmark = pytest.mark.gpu
RTOL32, RTOLBF = 1e-4, 8e-3


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


def _batch(sides, seed, pct=60):
    from diffassemble_amd import expander
    rng = np.random.default_rng(seed)
    ei, batch, degs = expander.ragged_regular_batch(sides, pct, rng)
    n = int(batch.numel())
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 4, generator=g)
    feats = torch.randn(n, 1088, generator=g)
    t = torch.randint(0, 300, (len(sides),), generator=g)[batch]
    return ei, batch, degs, x, feats, t


CASES = {"mixed_hybrid": ([6, 16, 10, 20, 8, 18, 12, 14], True, 60), "small_hybrid": ([6, 8, 10, 12, 14, 12, 10, 6], True, 60),
         "sparse_csr": ([16, 20, 18], False, 0.5)}


def test_percent_degree_matches_the_dataset_rule():
    from diffassemble_amd import expander
    for side in range(6, 21, 2):
        n = side * side
        d = expander.percent_degree(n, 60)
        assert (n * d) % 2 == 0 and abs(d - 0.6 * (n - 1)) <= 1.5 and 2 <= d < n


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", list(CASES))
def test_scripted_batch_forward_and_loop_vs_oracle(dev, name, prec):
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    sides, hybrid, pct = CASES[name]
    ei, batch, degs, x, feats, t = _batch(sides, 5, pct)
    V = 8
    sd = W.make_denoiser_state(300, 4, 4, arch="exophormer", virt_nodes=V, seed=71, qk_gain=3.0)
    ref, _ = OD.eff_gat_forward_with_feats(sd, x, t, ei, feats, batch, arch="exophormer", virt_nodes=V)
    eng = DenoiserEngine(sd, variant="2d", arch="exophormer", virt_nodes=V, precision=prec, device=dev)
    plan = eng.plan(ei.to(dev), batch.to(dev))
    assert bool(plan.hybrid) == hybrid and not plan.dense
    n = int(batch.numel())
    assert plan.n_edges == sum(s * s * d for s, d in zip(sides, degs)) + n + sum(V * (s * s + V) for s in sides)
    from diffassemble_amd import engine as E
    E.launch_counters(reset=True)
    out = eng.forward(plan, x.to(dev), t.to(dev), feats.to(dev))
    torch.cuda.synchronize()
    cnt = E.launch_counters(reset=True)
    tol = RTOL32 if prec == "fp32" else RTOLBF
    assert rel(out, ref) < tol, rel(out, ref)
    import os
    if hybrid and prec == "bf16" and "DA_DISABLE_FOLDS" not in os.environ and "DA_ATTN_LEVEL" not in os.environ:
        # small hybrid Batches: the exophormer's virtual rows ride the masked attention's launch in the three hidden layers (virt_rows_block, da_attn_opt.hip)
        assert cnt["virtual_rows_in_launch"] == 3, cnt
    else:
        assert cnt["virtual_rows_in_launch"] == 0, cnt
    # the first three DDIM steps of the scripted loop (T = 300, ratio 10, START_X, noise_weight 0 -> x_T = 0 in the script;
    # a random start here exercises more) through the captured graph
    sch_c = ODF.make_schedule(300)
    imgs, _ = ODF.p_sample_loop(sd, sch_c, x, ei, feats, batch, 300, 10, "START_X", "exophormer", V, max_iters=3)
    traj, _ = eng.sample_loop(plan, Schedule(sch_c, dev), x.to(dev), feats.to(dev), ratio=10, mean_type=_lib.MEAN_START_X, max_iters=3,
                              use_graph=True)
    assert rel(traj, torch.stack(imgs)) < (5e-4 if prec == "fp32" else 2e-2)


@pytest.mark.parametrize("name", list(CASES))
def test_scripted_batch_training_step_vs_oracle_autograd(dev, name):
    """p_losses on the ragged Batch: loss and every live parameter's gradient against the oracle's autograd (fp32 mode)."""
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    sides, hybrid, pct = CASES[name]
    ei, batch, degs, x, feats, t = _batch(sides, 9, pct)
    V = 8
    sd = W.make_denoiser_state(300, 4, 4, arch="exophormer", virt_nodes=V, seed=73, qk_gain=3.0)
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    sdc = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_c = ODF.p_losses(sdc, ODF.make_schedule(300), x, t, noise, ei, feats, batch, "START_X", arch="exophormer", virt_nodes=V)
    loss_c.backward()
    m = GNN_Diffusion(steps=300, sampling="DDIM", inference_ratio=10, rotation=True, visual_pretrained=False,
                      model_mean_type=ModelMeanType.START_X, architecture="exophormer", virt_nodes=V)
    m.model.load_state_dict(sd, strict=False)
    m = m.to(dev).train()
    loss = m.p_losses(x.to(dev), t.to(dev), noise=noise.to(dev), loss_type="huber", cond=None, edge_index=ei.to(dev), batch=batch.to(dev),
                      patch_feats=feats.to(dev))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_c)) < 1e-4 * abs(float(loss_c)) + 1e-7
    params = dict(m.model.named_parameters())
    worst = 0.0
    for k, v in sdc.items():
        if v.grad is None or k not in params or params[k].grad is None:
            continue
        gmax = float(v.grad.abs().max())
        if gmax < 1e-9:
            continue
        worst = max(worst, rel(params[k].grad, v.grad))
    assert 0 < worst < 2e-4, worst


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_two_exophormer_batches_in_flight_equal_each_alone(dev, prec):
    """`sample_loop_batches`: two independent hybrid (Exphander + exophormer) Batches as two hipGraphs on two streams -- the way Batches whose
    virtual-node edges forbid a SPLIT (exophormer_gnn.py:183-200 couples a Batch's puzzles) get the pair loop's overlap.  Each Batch's poses are
    bit for bit what `sample_loop` gives it alone, and the second call (cached graphs) reproduces the first."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    V = 8
    sd = W.make_denoiser_state(300, 4, 4, arch="exophormer", virt_nodes=V, seed=72, qk_gain=3.0)
    eng = DenoiserEngine(sd, variant="2d", arch="exophormer", virt_nodes=V, precision=prec, device=dev)
    sch = Schedule(ODF.make_schedule(300), dev)
    items = []
    for sides, seed in (([6, 16, 10, 20, 8, 18, 12, 14], 5), ([12, 8, 14, 6, 10, 16], 9)):
        ei, batch, degs, x, feats, t = _batch(sides, seed, 60)
        plan = eng.plan(ei.to(dev), batch.to(dev))
        assert plan.hybrid
        items.append((plan, x.to(dev), feats.to(dev)))
    alone = []
    for plan, x, f in items:
        _, xf = eng.sample_loop(plan, sch, x, f, ratio=10, mean_type=_lib.MEAN_START_X, max_iters=4, keep_trajectory=False, use_graph=True)
        alone.append(xf.clone())
    for _ in range(2):
        fa, fb = eng.sample_loop_batches([items[0][0], items[1][0]], sch, [items[0][1], items[1][1]], [items[0][2], items[1][2]], ratio=10,
                                         mean_type=_lib.MEAN_START_X, max_iters=4)
        torch.cuda.synchronize()
        assert torch.isfinite(fa).all() and torch.equal(fa, alone[0]) and torch.equal(fb, alone[1])


def test_four_small_batches_in_flight_equal_each_alone(dev):
    """N = 4 Batches in flight (each on a stream of the engine's, da_sample_loop): bit for bit what every Batch computes alone."""
    from diffassemble_amd import DenoiserEngine, Schedule, _lib
    V = 8
    sd = W.make_denoiser_state(300, 4, 4, arch="exophormer", virt_nodes=V, seed=73, qk_gain=3.0)
    eng = DenoiserEngine(sd, variant="2d", arch="exophormer", virt_nodes=V, precision="bf16", device=dev)
    sch = Schedule(ODF.make_schedule(300), dev)
    items = []
    for sides, seed in (([6, 16, 10, 8], 5), ([12, 8, 14, 6], 9), ([10, 10, 12], 11), ([8, 6, 16, 12, 6], 13)):
        ei, batch, degs, x, feats, t = _batch(sides, seed, 60)
        items.append((eng.plan(ei.to(dev), batch.to(dev)), x.to(dev), feats.to(dev)))
    alone = []
    for plan, x, f in items:
        _, xf = eng.sample_loop(plan, sch, x, f, ratio=10, mean_type=_lib.MEAN_START_X, max_iters=4, keep_trajectory=False, use_graph=True)
        alone.append(xf.clone())
    for _ in range(2):
        outs = eng.sample_loop_batches([it[0] for it in items], sch, [it[1] for it in items], [it[2] for it in items], ratio=10,
                                       mean_type=_lib.MEAN_START_X, max_iters=4)
        torch.cuda.synchronize()
        assert len(outs) == 4 and all(torch.equal(o, a) for o, a in zip(outs, alone))
