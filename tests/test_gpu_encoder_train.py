"""GPU parity tests of the piece encoder's TRAINING path (batch-statistics BatchNorm + backward):
the HIP primitives through the C ABI (da_enc_*, da_gemm_tn_f32) and the network walk (diffassemble_amd/encoder_train.py)
against torch autograd through the CPU oracle (oracle/encoder.py in training mode, itself pinned by the reference's own
train()-mode forward / backward, tests/golden/encoder_train_v1.npz).

Tolerances: single primitives 1e-5 (max-abs error / max-abs of the reference tensor) -- nothing discrete in them.  The
whole network: features 1e-5; gradients against the fp64 run of the oracle, digests to 2e-3 and full tensors to 2e-2,
because the backward through 20 ReLU layers is chaotic in fp32: a pre-activation within rounding of zero takes the
other side of the ReLU under a different summation order (the reference's own fp32 result differs from the fp64 oracle by
8e-4 and from the fp32 oracle by up to 8e-3 on single entries; tests/test_oracle.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import encoder as OE
from oracle import weights as W

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_train_v1.npz"))


def rel(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu-marked tests need a ROCm device"
    return torch.device("cuda:0")


def halo(x):
    """[B, C, H, W] -> zero-haloed NHWC [B, H+2, W+2, C]"""
    return F.pad(x.permute(0, 2, 3, 1), (0, 0, 1, 1, 1, 1)).contiguous()


def unhalo(x):
    return x[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)


def test_bn_forward_backward_primitives_vs_autograd(dev):
    from diffassemble_amd import _lib
    L = _lib.lib()
    torch.manual_seed(0)
    B, H, planes = 5, 8, 64
    y = torch.randn(B, planes, 4, H, H, dtype=torch.float64, requires_grad=True)
    gamma = (torch.rand(planes, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(planes, dtype=torch.float64).requires_grad_(True)
    res = torch.randn(B, planes, 4, H, H, dtype=torch.float64, requires_grad=True)
    z = F.relu(F.batch_norm(y, None, None, gamma, beta, True, 0.1, 1e-5) + res)
    dz = torch.randn_like(z)
    z.backward(dz)
    f = lambda t: halo(t.detach().float().reshape(B, planes * 4, H, H)).to(dev)  # noqa: E731
    Y, R, dZ = f(y), f(res), f(dz)
    Z, dY, dR = torch.zeros_like(Y), torch.zeros_like(Y), torch.zeros_like(Y)
    mean, var = torch.zeros(planes, device=dev), torch.zeros(planes, device=dev)
    dg, db = torch.zeros(planes, device=dev), torch.zeros(planes, device=dev)
    scratch = torch.empty(L.da_enc_train_scratch_bytes(B), dtype=torch.uint8, device=dev)
    st = _lib.stream_ptr(dev)
    g32, b32 = gamma.detach().float().to(dev), beta.detach().float().to(dev)
    _lib.check(L.da_enc_bn_stats(_lib.PREC_F32, B, H, planes * 4, _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(scratch), st))
    _lib.check(L.da_enc_bn_apply(_lib.PREC_F32, B, H, planes * 4, _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(var), _lib.ptr(g32), _lib.ptr(b32),
                                 _lib.ptr(R), 1, _lib.ptr(Z), st))
    _lib.check(L.da_enc_bn_backward(_lib.PREC_F32, B, H, planes * 4, _lib.ptr(dZ), _lib.ptr(Z), _lib.ptr(Y), _lib.ptr(mean), _lib.ptr(var),
                                    _lib.ptr(g32), 1, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dY), _lib.ptr(dR), _lib.ptr(scratch), st))
    yd = y.detach()
    assert rel(mean, yd.mean((0, 2, 3, 4))) < 1e-5 and rel(var, yd.var((0, 2, 3, 4), unbiased=False)) < 1e-5
    r5 = lambda t: unhalo(t).reshape(B, planes, 4, H, H)  # noqa: E731
    assert rel(r5(Z), z) < 1e-5
    assert rel(r5(dY), y.grad) < 1e-5 and rel(r5(dR), res.grad) < 1e-5
    assert rel(dg, gamma.grad) < 1e-5 and rel(db, beta.grad) < 1e-5
    for t in (Z, dY, dR):                                   # halos never written
        assert float(t[:, 0].abs().max()) == 0 and float(t[:, :, -1].abs().max()) == 0


@pytest.mark.parametrize("cin,cout,k,stride,H", [(32, 32, 3, 1, 8), (32, 64, 3, 2, 8), (32, 64, 1, 2, 8)])
def test_conv_dgrad_wgrad_primitives_vs_autograd(dev, cin, cout, k, stride, H):
    """One group convolution: forward (da_enc_conv on the packed bank), input gradient (the same kernel with flipped /
    transposed weights; zero-stuffed dY for stride 2) and parameter gradient (one TN GEMM per tap + the bank gather-sum)
    against torch autograd through the oracle's filter-bank + conv2d formulation."""
    from diffassemble_amd.encoder_train import EncoderTrainEngine as E
    from diffassemble_amd import _lib
    from diffassemble_amd.encoder import p4_filter_bank
    L = _lib.lib()
    torch.manual_seed(1)
    B, Ho = 3, H // stride
    w = (torch.randn(cout, cin, 4, k, k, dtype=torch.float64) * 0.1).requires_grad_(True)
    x = torch.randn(B, cin * 4, H, H, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x, OE.p4_filter_bank(w), None, stride=stride, padding=k // 2)
    dy = torch.randn_like(y)
    y.backward(dy)
    st = _lib.stream_ptr(dev)
    zero = torch.zeros(512, device=dev)
    bank = p4_filter_bank(w.detach().float().to(dev))
    X, dY = halo(x.detach().float()).to(dev), halo(dy.float()).to(dev)
    Y = torch.zeros(B, Ho + 2, Ho + 2, cout * 4, device=dev)
    _lib.check(L.da_enc_conv(_lib.PREC_F32, B, _lib.ptr(X), cin * 4, H, _lib.ptr(E._pack_fwd(bank)), _lib.ptr(zero), None, _lib.ptr(Y),
                             cout * 4, k, stride, 0, st))
    assert rel(unhalo(Y), y) < 1e-5
    if stride == 2:
        up = torch.zeros(B, H + 2, H + 2, cout * 4, device=dev)
        _lib.check(L.da_enc_upsample2(_lib.PREC_F32, B, Ho, cout * 4, _lib.ptr(dY), _lib.ptr(up), st))
        dY = up
    dX = torch.zeros_like(X)
    _lib.check(L.da_enc_conv(_lib.PREC_F32, B, _lib.ptr(dY), cout * 4, H, _lib.ptr(E._pack_dgrad(bank)), _lib.ptr(zero), None,
                             _lib.ptr(dX), cin * 4, k, 1, 0, st))
    assert rel(unhalo(dX), x.grad) < 1e-5
    # wgrad through the engine's helper (needs only these fields)
    eng = E.__new__(E)
    eng.lib, eng.device, eng._n, eng.precision = L, dev, B, "fp32"
    eng.gemm_scratch = torch.empty(16 << 20, device=dev)
    p = torch.nn.Parameter(w.detach().float().to(dev))
    eng.params = {"c.weight": p}
    src = E._pack_fwd(p4_filter_bank(torch.arange(p.numel(), device=dev).view(p.shape))).reshape(-1)
    eng._tables = {"c": torch.argsort(src, stable=True).to(torch.int32).view(p.numel(), 4).contiguous()}
    eng._wgrad("c", dY, X, cin * 4, cout * 4, k, H)
    assert rel(p.grad, w.grad) < 1e-5


@pytest.mark.parametrize("name,seed,n", [("tr_s0", 0, 4), ("tr_s1", 1, 6)])
def test_encoder_training_step_vs_oracle_and_reference(dev, name, seed, n):
    """Training-mode features, every parameter gradient and the running-statistics update of the whole P4 ResNet-18."""
    from diffassemble_amd.encoder_train import EncoderTrainEngine
    from diffassemble_amd.model.backbones.resnet_equivariant import ResNet18
    net = ResNet18(precision="fp32")
    net.load_state_dict(W.make_encoder_state(seed))
    net = net.to(dev).train()
    eng = EncoderTrainEngine(net, dev)
    x, G = W.make_patches(n, seed + 100), W.randn((n, 1088), seed + 200)
    feats = eng.forward(x.to(dev))
    assert rel(feats, GOLD[f"{name}/feats"]) < 1e-5
    eng.backward(G.to(dev))
    # oracle in fp64: the reference point for the gradients
    sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v))
          for k, v in W.make_encoder_state(seed).items()}
    OE.MEAN, OE.STD = OE.MEAN.double(), OE.STD.double()
    try:
        st = {}
        fo = OE.visual_features(sd, x.double(), stats=st)
        (fo * G.double()).sum().backward()
    finally:
        OE.MEAN, OE.STD = OE.MEAN.float(), OE.STD.float()
    assert rel(feats, fo) < 1e-5
    worst = 0.0
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        ref = sd[k].grad
        t = p.grad.double().cpu()
        assert abs(float(t.abs().sum()) - float(ref.abs().sum())) <= 2e-3 * float(ref.abs().sum()), k
        e = rel(t, ref)
        worst = max(worst, e)
        assert e < (1e-5 if k.startswith("linear") else 2e-2), (k, e)
        key = f"{name}/gstats/{k}"
        assert abs(float(t.abs().sum()) - GOLD[key][1]) <= 2e-3 * GOLD[key][1], k          # the reference's own backward
    sdn = net.state_dict()
    for key in [f for f in GOLD.files if f.startswith(f"{name}/running/")]:
        assert rel(sdn[key.split("/running/")[1]], GOLD[key]) < 1e-5, key
    assert int(sdn["bn1.num_batches_tracked"]) == 101
    # a second backward accumulates into .grad like autograd does
    g1 = net.conv1.weight.grad.clone()
    eng.forward(x.to(dev))
    eng.backward(G.to(dev))
    assert rel(net.conv1.weight.grad, 2 * g1) < 1e-4


@pytest.mark.parametrize("M,N,K", [(5000, 128, 128), (777, 544, 300), (2312, 128, 27), (64, 256, 512)])
def test_gemm_tn_bf16_vs_torch(dev, M, N, K):
    """C += A^T B with bf16 operands (the encoder's bf16 weight gradients): against the fp64 product of the SAME bf16
    values, so only the fp32 accumulation order differs (1e-5); ragged N / K, the row split and the += semantics."""
    from diffassemble_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(M + N)
    lda, ldb = (N + 7) // 8 * 8 + 8, (K + 7) // 8 * 8
    A = torch.randn(M, lda, generator=g, device=dev).to(torch.bfloat16)
    B = torch.randn(M, ldb, generator=g, device=dev).to(torch.bfloat16)
    C0 = torch.randn(N, K, generator=g, device=dev)
    C = C0.clone()
    scratch = torch.empty(16 << 20, device=dev)
    _lib.check(L.da_gemm_tn_bf16(M, N, K, _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C), K, _lib.ptr(scratch), _lib.stream_ptr(dev)))
    ref = C0.double() + A[:, :N].double().t() @ B[:, :K].double()
    assert rel(C, ref) < 1e-5


@pytest.mark.parametrize("seed,n", [(1, 6), (2, 48)])
def test_encoder_training_step_bf16(dev, seed, n):
    """The bf16 training mode (bf16 activations / activation gradients, bf16 matrix cores, fp32 BatchNorm arithmetic,
    master weights and parameter gradients) against the fp64 oracle: features to 5e-2 (the inference bound of 17 bf16-stored
    layers); every parameter gradient by direction (cosine) and size (norm ratio).  Entry-wise bounds mean little for a
    backward that is chaotic already in fp32 (module docstring): storing a pre-activation in bf16 moves it by 4e-3 of its
    size, so ~0.3 % of the ReLU decisions differ in every layer, each switching a whole gradient path: ~5 % of error per
    ReLU layer, adding in quadrature with depth (cosine 0.9998 at the linear heads, 0.97-0.99 in the last stage, ~0.95 at
    the stem).  With this test's loss (an independent random direction per piece) the per-piece gradients are incoherent,
    so the relative error does not shrink with the batch (6 and 48 pieces measure the same)."""
    from diffassemble_amd.encoder_train import EncoderTrainEngine
    from diffassemble_amd.model.backbones.resnet_equivariant import ResNet18
    net = ResNet18(precision="bf16")
    net.load_state_dict(W.make_encoder_state(seed))
    net = net.to(dev).train()
    eng = EncoderTrainEngine(net, dev, precision="bf16")
    x, G = W.make_patches(n, seed + 100), W.randn((n, 1088), seed + 200)
    feats = eng.forward(x.to(dev))
    eng.backward(G.to(dev))
    sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v))
          for k, v in W.make_encoder_state(seed).items()}
    OE.MEAN, OE.STD = OE.MEAN.double(), OE.STD.double()
    try:
        st = {}
        fo = OE.visual_features(sd, x.double(), stats=st)
        (fo * G.double()).sum().backward()
    finally:
        OE.MEAN, OE.STD = OE.MEAN.float(), OE.STD.float()
    assert feats.dtype == torch.float32 and rel(feats, fo) < 5e-2
    report = []
    for k, p in net.named_parameters():
        a, b = p.grad.double().cpu().flatten(), sd[k].grad.flatten()
        report.append((k, round(float((a @ b) / (a.norm() * b.norm() + 1e-30)), 4), round(float(a.norm() / b.norm()), 4)))
        assert p.grad.dtype == torch.float32
    print(report)
    heads = [r for r in report if r[0].startswith("linear")]
    last = [r for r in report if r[0].startswith("layer4")]
    assert min(r[1] for r in heads) > 0.999 and min(r[1] for r in last) > 0.96 and min(r[1] for r in report) > 0.9, report
    assert all(abs(r[2] - 1) < 0.15 for r in report), report
    assert rel(net.bn1.running_mean, st["bn1.running_mean"]) < 1e-2


@pytest.mark.parametrize("n", [1, 5])
def test_encoder_training_tiny_batches(dev, n):
    """One piece (BatchNorm statistics over a single piece's 4 x H x W values) and an odd batch: features and the last
    block's gradients against the fp64 oracle."""
    from diffassemble_amd.encoder_train import EncoderTrainEngine
    from diffassemble_amd.model.backbones.resnet_equivariant import ResNet18
    seed = 7
    net = ResNet18(precision="fp32")
    net.load_state_dict(W.make_encoder_state(seed))
    net = net.to(dev).train()
    eng = EncoderTrainEngine(net, dev)
    x, G = W.make_patches(n, seed + 100), W.randn((n, 1088), seed + 200)
    feats = eng.forward(x.to(dev))
    eng.backward(G.to(dev))
    sd = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v))
          for k, v in W.make_encoder_state(seed).items()}
    OE.MEAN, OE.STD = OE.MEAN.double(), OE.STD.double()
    try:
        fo = OE.visual_features(sd, x.double(), stats={})
        (fo * G.double()).sum().backward()
    finally:
        OE.MEAN, OE.STD = OE.MEAN.float(), OE.STD.float()
    assert rel(feats, fo) < 2e-5
    P = dict(net.named_parameters())
    for k in ("linear1.weight", "linear2.bias", "layer4.1.bn2.weight", "layer4.1.conv2.weight"):
        assert rel(P[k].grad, sd[k].grad) < (1e-5 if k.startswith("linear") else 2e-2), k
