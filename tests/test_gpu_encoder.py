"""GPU parity tests of the piece encoder: the HIP path through the C ABI
(da_encoder_forward) against the CPU oracle (oracle/encoder.py) and against the fixtures generated from
the reference's own resnet_equivariant / groupy code (tests/golden/make_encoder_golden.py).

Tolerances: fp32 parity mode 1e-4 (max-abs error / max-abs of the reference tensor; the BatchNorm fold
and the MFMA summation order are the only differences); bf16 perf mode 5e-2 (17 bf16-stored layers)."""
import os

import numpy as np
import pytest
import torch

from oracle import denoiser as OD
from oracle import encoder as OE
from oracle import weights as W

pytestmark = pytest.mark.gpu
RTOL32, RTOLBF = 1e-4, 5e-2
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_v1.npz"))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


@pytest.mark.parametrize("name,seed,n", [("enc_s0", 0, 3), ("enc_s1", 1, 5)])
def test_encoder_fp32_matches_reference_fixture(dev, name, seed, n):
    from diffassemble_amd.encoder import EncoderEngine
    eng = EncoderEngine(W.make_encoder_state(seed), precision="fp32", device=dev)
    out = eng.forward(W.make_patches(n, seed + 100).to(dev))
    assert out.shape == (n, 1088) and out.dtype == torch.float32
    assert rel(out, GOLD[f"{name}/feats"]) < RTOL32


@pytest.mark.parametrize("n,chunk", [(1, None), (20, 8), (64, 64), (150, 64)])
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_matches_oracle(dev, n, chunk, prec):
    """Ragged chunking: n not a multiple of the chunk, last chunk smaller than one 128-pixel tile at 4x4."""
    from diffassemble_amd.encoder import EncoderEngine
    sd = W.make_encoder_state(3)
    x = W.make_patches(n, 11 + n)
    ref = OE.visual_features(sd, x)
    eng = EncoderEngine(sd, precision=prec, device=dev, chunk=chunk)
    out = eng.forward(x.to(dev)).float()
    assert rel(out, ref) < (RTOL32 if prec == "fp32" else RTOLBF)
    # second call re-uses the workspace without re-zeroing the halos; a different input must not see stale data
    x2 = W.make_patches(n, 500 + n)
    out2 = eng.forward(x2.to(dev)).float()
    assert rel(out2, OE.visual_features(sd, x2)) < (RTOL32 if prec == "fp32" else RTOLBF)


def test_encoder_is_deterministic_and_chunk_invariant(dev):
    from diffassemble_amd.encoder import EncoderEngine
    sd = W.make_encoder_state(4)
    x = W.make_patches(96, 5).to(dev)
    a = EncoderEngine(sd, precision="bf16", device=dev, chunk=32).forward(x)
    b = EncoderEngine(sd, precision="bf16", device=dev, chunk=96).forward(x)
    c = EncoderEngine(sd, precision="bf16", device=dev, chunk=96).forward(x)
    assert torch.equal(b, c)
    assert torch.equal(a, b)          # a piece's features do not depend on which pieces share its chunk


def test_eff_gat_forward_from_pixels(dev):
    """Eff_GAT(model='resnet18equiv').forward(xy, t, patch_rgb, edge_index, batch): pixels -> encoder ->
    denoiser, all in HIP, against oracle encoder + oracle denoiser (efficient_gat.py:114-119)."""
    from diffassemble_amd.model.backbones import Eff_GAT
    import os as _os
    _os.environ["DIFFASSEMBLE_PRECISION"] = "fp32"
    try:
        steps, n = 50, 36
        m = Eff_GAT(steps=steps, input_channels=4, output_channels=4, model="resnet18equiv",
                    visual_pretrained=False, architecture="transformer")
        dsd = W.make_denoiser_state(steps, 4, 4, seed=9)
        esd = W.make_encoder_state(9)
        missing, unexpected = m.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
        assert not unexpected and all(k.startswith(("linear1.", "linear2.", "mean", "std")) for k in missing), (missing, unexpected)
        m = m.to(dev).eval()
        rng = np.random.default_rng(1)
        xy = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
        t = torch.from_numpy(rng.integers(0, steps, size=n))
        patches = W.make_patches(n, 77)
        ei = W.dense_edge_index(n, True)
        batch = torch.zeros(n, dtype=torch.int64)
        out, att = m(xy.to(dev), t.to(dev), patches.to(dev), ei.to(dev), batch.to(dev))
        feats = OE.visual_features(esd, patches)
        ref, _ = OD.eff_gat_forward_with_feats(dsd, xy, t, ei, feats, batch, arch="transformer")
        assert rel(out, ref) < 2e-4
        m.train()                                   # BatchNorm on batch statistics (tests/test_gpu_encoder_train.py)
        ft = m.visual_features(patches.to(dev))
        assert rel(ft, OE.visual_features(esd, patches, stats={})) < 1e-4 and ft.requires_grad
    finally:
        _os.environ.pop("DIFFASSEMBLE_PRECISION", None)


def test_sampling_loop_from_pixels(dev):
    """GNN_Diffusion(backbone='resnet18equiv').p_sample_loop(shape, cond = piece crops, ...): the encoder runs
    once (spatial_diffusion.py:653), then the hipGraph DDIM loop; against oracle encoder + oracle loop."""
    from oracle import diffusion as ODF
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    T, ratio, n = 100, 10, 64
    m = GNN_Diffusion(steps=T, sampling="DDIM", inference_ratio=ratio, noise_weight=1.0, rotation=True,
                      model_mean_type=ModelMeanType.START_X, visual_pretrained=False, backbone="resnet18equiv")
    dsd, esd = W.make_denoiser_state(T, 4, 4, seed=21), W.make_encoder_state(21)
    missing, unexpected = m.model.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
    assert not unexpected, unexpected
    m = m.to(dev).eval()
    m.model.precision = "fp32"
    patches = W.make_patches(n, 3)
    ei, batch = W.dense_edge_index(n, True), torch.zeros(n, dtype=torch.int64)
    x0 = torch.from_numpy(np.random.default_rng(8).standard_normal((n, 4)).astype(np.float32))
    _orig = torch.randn
    torch.randn = lambda *a, **k: x0.to(dev)
    try:
        imgs, _ = m.p_sample_loop((n, 4), patches.to(dev), ei.to(dev), batch.to(dev))
    finally:
        torch.randn = _orig
    feats = OE.visual_features(esd, patches)
    ref, _ = ODF.p_sample_loop(dsd, ODF.make_schedule(T), x0, ei, feats, batch, T, inference_ratio=ratio)
    assert len(imgs) == len(ref) == 10
    assert rel(torch.stack(imgs), torch.stack(ref)) < 5e-4


def test_all_equivariant_averages_the_four_views(dev):
    """Eff_GAT(all_equivariant=True): patch_rgb [N, 4, 3, 32, 32] -> mean over the four views of the encoder
    output (efficient_gat.py:156-158)."""
    from diffassemble_amd.model.backbones import Eff_GAT
    m = Eff_GAT(steps=10, input_channels=4, output_channels=4, model="resnet18equiv", visual_pretrained=False,
                all_equivariant=True)
    esd = W.make_encoder_state(5)
    m.visual_backbone.load_state_dict(esd)
    m = m.to(dev).eval()
    m.precision = "fp32"
    x = W.make_patches(4 * 6, 9).view(6, 4, 3, 32, 32)
    out = m.visual_features(x.to(dev))
    ref = torch.stack([OE.visual_features(esd, x[:, i]) for i in range(4)]).mean(0)
    assert out.shape == (6, 1088) and rel(out, ref) < RTOL32


def test_training_step_from_pixels_with_frozen_encoder(dev):
    """p_losses(cond = crops): HIP encoder on its running statistics (frozen_eval_stats) feeding the HIP training
    forward / backward of the denoiser; loss and a gradient against oracle encoder + oracle p_losses."""
    from oracle import diffusion as ODF
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    T, n = 100, 36
    m = GNN_Diffusion(steps=T, sampling="DDIM", rotation=True, model_mean_type=ModelMeanType.EPSILON,
                      visual_pretrained=False, backbone="resnet18equiv")
    dsd, esd = W.make_denoiser_state(T, 4, 4, seed=31), W.make_encoder_state(31)
    m.model.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
    m = m.to(dev).train()
    m.model.visual_backbone.frozen_eval_stats = True
    m.model.precision = "fp32"
    rng = np.random.default_rng(4)
    x0 = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    t = torch.full((n,), 37, dtype=torch.int64)
    patches = W.make_patches(n, 6)
    ei, batch = W.dense_edge_index(n, True), torch.zeros(n, dtype=torch.int64)
    loss = m.p_losses(x0.to(dev), t.to(dev), noise=noise.to(dev), loss_type="huber", cond=patches.to(dev),
                      edge_index=ei.to(dev), batch=batch.to(dev))
    loss.backward()
    sd_ref = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    feats = OE.visual_features(esd, patches)
    ref = ODF.p_losses(sd_ref, ODF.make_schedule(T), x0, t, noise, ei, feats, batch, mean_type="EPSILON")
    ref.backward()
    assert rel(loss, ref) < 1e-4
    g = dict(m.model.named_parameters())["final_mlp.0.weight"].grad
    assert rel(g, sd_ref["final_mlp.0.weight"].grad) < 1e-3


def test_training_step_from_pixels_trains_the_encoder(dev):
    """The scripted configuration (--backbone resnet18equiv, freeze_backbone False): p_losses(cond = crops) in train()
    mode runs the encoder on BATCH statistics, and loss.backward() reaches the encoder's parameters through the denoiser's
    d_feats.  Loss, a denoiser gradient and encoder gradients (last block: tight; stem: the chaotic-backward tolerance of
    tests/test_gpu_encoder_train.py) against torch autograd through oracle encoder (training mode) + oracle p_losses."""
    from oracle import diffusion as ODF
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    T, n = 100, 36
    m = GNN_Diffusion(steps=T, sampling="DDIM", rotation=True, model_mean_type=ModelMeanType.EPSILON,
                      visual_pretrained=False, backbone="resnet18equiv", freeze_backbone=False)
    dsd, esd = W.make_denoiser_state(T, 4, 4, seed=32), W.make_encoder_state(32)
    m.model.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
    m = m.to(dev).train()
    m.model.precision = "fp32"
    rng = np.random.default_rng(5)
    x0 = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    t = torch.full((n,), 41, dtype=torch.int64)
    patches = W.make_patches(n, 7)
    ei, batch = W.dense_edge_index(n, True), torch.zeros(n, dtype=torch.int64)
    loss = m.p_losses(x0.to(dev), t.to(dev), noise=noise.to(dev), loss_type="huber", cond=patches.to(dev),
                      edge_index=ei.to(dev), batch=batch.to(dev))
    loss.backward()
    sd_ref = {k: v.double().clone().requires_grad_(True) for k, v in dsd.items()}
    e_ref = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v))
             for k, v in esd.items()}
    OE.MEAN, OE.STD = OE.MEAN.double(), OE.STD.double()
    try:
        feats = OE.visual_features(e_ref, patches.double(), stats={})
    finally:
        OE.MEAN, OE.STD = OE.MEAN.float(), OE.STD.float()
    ref = ODF.p_losses(sd_ref, ODF.make_schedule(T), x0.double(), t, noise.double(), ei, feats, batch, mean_type="EPSILON")
    ref.backward()
    assert rel(loss, ref) < 1e-4
    P = dict(m.model.named_parameters())
    assert rel(P["final_mlp.0.weight"].grad, sd_ref["final_mlp.0.weight"].grad) < 1e-3
    assert rel(P["visual_backbone.linear2.weight"].grad, e_ref["linear2.weight"].grad) < 1e-3
    assert rel(P["visual_backbone.layer4.1.bn2.weight"].grad, e_ref["layer4.1.bn2.weight"].grad) < 2e-2
    for k in ("conv1.weight", "layer1.0.conv1.weight", "layer2.0.shortcut.0.weight"):
        g, r = P["visual_backbone." + k].grad, e_ref[k].grad
        assert abs(float(g.abs().sum()) - float(r.abs().sum())) <= 5e-3 * float(r.abs().sum()) and rel(g, r) < 5e-2, k
    assert int(m.model.visual_backbone.bn1.num_batches_tracked) == 101


def test_optimizer_steps_from_pixels_update_encoder_and_denoiser(dev):
    """configure_optimizers() with a trainable encoder = HybridAdafactor (fused denoiser update + transformers' Adafactor for
    the encoder): after one step from pixels every trained tensor matches the reference's optimizer applied to the oracle's
    autograd gradients; a few more steps on the same Batch reduce the loss; eval() afterwards runs on the updated weights
    and running statistics (packed inference weights are rebuilt)."""
    from oracle import diffusion as ODF
    from transformers.optimization import Adafactor
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    T, n = 100, 36
    m = GNN_Diffusion(steps=T, sampling="DDIM", rotation=True, model_mean_type=ModelMeanType.EPSILON,
                      visual_pretrained=False, backbone="resnet18equiv", freeze_backbone=False)
    dsd, esd = W.make_denoiser_state(T, 4, 4, seed=33), W.make_encoder_state(33)
    m.model.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
    m = m.to(dev).train()
    m.model.precision = "fp32"
    opt = m.configure_optimizers()
    assert type(opt).__name__ == "HybridAdafactor"
    rng = np.random.default_rng(6)
    x0 = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    noise = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32))
    t = torch.full((n,), 23, dtype=torch.int64)
    patches = W.make_patches(n, 8)
    ei, batch = W.dense_edge_index(n, True), torch.zeros(n, dtype=torch.int64)

    def step():
        opt.zero_grad()
        loss = m.p_losses(x0.to(dev), t.to(dev), noise=noise.to(dev), loss_type="huber", cond=patches.to(dev),
                          edge_index=ei.to(dev), batch=batch.to(dev))
        loss.backward()
        opt.step()
        return float(loss.detach())

    l0 = step()
    # reference: oracle forward (training-mode encoder) + autograd + transformers' Adafactor on all live tensors
    ref = {k: v.clone().requires_grad_(True) for k, v in dsd.items()}
    eref = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in esd.items()}
    live = [v for v in list(ref.values()) + list(eref.values()) if torch.is_tensor(v) and v.requires_grad]
    ropt = Adafactor(live)
    lr = ODF.p_losses(ref, ODF.make_schedule(T), x0, t, noise, ei, OE.visual_features(eref, patches, stats={}), batch, mean_type="EPSILON")
    lr.backward()
    ropt.step()
    assert abs(l0 - float(lr)) < 1e-4 * abs(float(lr))
    P = dict(m.model.named_parameters())
    for k in ("mlp.0.weight", "final_mlp.2.bias", "gnn_backbone.module_list.3.lin_value.weight"):
        assert rel(P[k], ref[k]) < 1e-4, k
    for k in ("linear2.weight", "linear2.bias", "linear1.weight"):
        assert rel(P["visual_backbone." + k], eref[k]) < 1e-4, k
    # Adafactor normalises every [3, 3] filter slice of a 5-D weight by that slice's OWN second moments, so slices whose
    # gradient is all noise (the chaotic part of the backward, tests/test_gpu_encoder_train.py) get O(1) normalised updates
    # made of that noise: compare the STEP as a whole (cosine against the reference's step), not entry by entry
    for k in ("layer4.1.conv2.weight", "layer4.1.bn2.bias", "layer2.0.shortcut.0.weight", "conv1.weight", "bn1.weight"):
        d_got = (P["visual_backbone." + k].detach().cpu() - esd[k]).double().flatten()
        d_ref = (eref[k].detach() - esd[k]).double().flatten()
        cos = float((d_got @ d_ref) / (d_got.norm() * d_ref.norm() + 1e-30))
        assert float(d_got.norm()) > 0 and cos > 0.98, (k, cos)
    losses = [l0] + [step() for _ in range(4)]
    assert losses[-1] < losses[0]
    m.eval()
    f_eval = m.model.visual_features(patches[:5].to(dev))
    sd_now = {k: v.detach().cpu() for k, v in m.model.visual_backbone.state_dict().items()}
    assert rel(f_eval, OE.visual_features(sd_now, patches[:5])) < 1e-4


def test_full_size_properties(dev):
    """BASELINE-size batch (the 28 800 crops of 32 900-piece puzzles) through size-independent properties: a
    piece's features depend on nothing but the piece -- permuting the batch permutes the rows bit for bit, a
    different chunking changes nothing, and a prefix of the batch gives the prefix of the result."""
    from diffassemble_amd.encoder import EncoderEngine
    sd = W.make_encoder_state(6)
    n = 32 * 900
    x = torch.rand((n, 3, 32, 32), generator=torch.Generator(device=dev).manual_seed(3), device=dev)
    eng = EncoderEngine(sd, precision="bf16", device=dev)
    a = eng.forward(x).clone()
    assert torch.isfinite(a.float()).all()
    perm = torch.randperm(n, generator=torch.Generator(device=dev).manual_seed(4), device=dev)
    assert torch.equal(eng.forward(x[perm].contiguous()), a[perm])
    assert torch.equal(EncoderEngine(sd, precision="bf16", device=dev, chunk=640).forward(x), a)
    assert torch.equal(eng.forward(x[:1000].contiguous()), a[:1000])
    # and the oracle on a sample of the rows
    idx = torch.arange(0, n, 1901, device=dev)
    ref = OE.visual_features(sd, x[idx].cpu())
    assert rel(a[idx].float(), ref) < RTOLBF


def test_hybrid_adafactor_resumes_from_a_plain_adafactor_checkpoint(dev):
    """a run started with ``fused_optimizer = False`` (transformers' Adafactor over self.parameters(), what
    the reference's checkpoints hold) must be resumable with the fused / hybrid optimizer: same parameter order, same
    per-parameter state layout.  One step each from the same loaded statistics gives the same weights."""
    import copy
    from transformers.optimization import Adafactor
    from diffassemble_amd.model.spatial_diffusion import GNN_Diffusion, ModelMeanType
    T, n = 100, 36
    torch.manual_seed(0)

    def build():
        m = GNN_Diffusion(steps=T, sampling="DDIM", rotation=True, model_mean_type=ModelMeanType.EPSILON,
                          visual_pretrained=False, backbone="resnet18equiv", freeze_backbone=False)
        dsd, esd = W.make_denoiser_state(T, 4, 4, seed=35), W.make_encoder_state(35)
        m.model.load_state_dict({**dsd, **{"visual_backbone." + k: v for k, v in esd.items()}}, strict=False)
        m = m.to(dev).train()
        m.model.precision = "fp32"
        return m

    rng = np.random.default_rng(9)
    x0 = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)).to(dev)
    noise = torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)).to(dev)
    t = torch.full((n,), 31, dtype=torch.int64, device=dev)
    patches = W.make_patches(n, 9).to(dev)
    ei, batch = W.dense_edge_index(n, True).to(dev), torch.zeros(n, dtype=torch.int64, device=dev)

    def step(m, opt):
        opt.zero_grad()
        m.p_losses(x0, t, noise=noise, loss_type="huber", cond=patches, edge_index=ei, batch=batch).backward()
        opt.step()

    ma = build()
    plain = Adafactor(ma.parameters())
    ma.model.train_engine(dev)                         # flat buffers bound before the first step (as configure_optimizers does)
    step(ma, plain)
    step(ma, plain)
    sd = copy.deepcopy(plain.state_dict())
    weights = {k: v.detach().clone() for k, v in ma.state_dict().items()}
    step(ma, plain)                                    # the run that kept going
    after_plain = {k: v.detach().clone() for k, v in ma.model.named_parameters()}

    mb = build()
    mb.load_state_dict(weights)
    opt = mb.configure_optimizers()
    assert type(opt).__name__ == "HybridAdafactor"
    opt.load_state_dict(sd)                            # plain layout: no "fused" / "rest" keys
    assert opt.fused.step_count == 2
    step(mb, opt)
    for k, v in mb.model.named_parameters():
        if v.requires_grad and after_plain[k].abs().max() > 0:
            assert rel(v, after_plain[k]) < 2e-5, k
