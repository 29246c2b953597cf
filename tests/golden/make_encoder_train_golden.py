"""Generate tests/golden/encoder_train_v1.npz from the REFERENCE's own encoder code in TRAINING mode.

BUILD-CONTAINER ONLY (same imports and stubs as make_encoder_golden.py).  For each case the reference's
``Eff_GAT(model='resnet18equiv')`` is put in train() mode (BatchNorm3d on batch statistics, as inside ``training_step``,
spatial_diffusion.py:450), ``visual_features`` runs on seeded patches, the scalar  L = sum(feats * G)  (G seeded) is
back-propagated with torch autograd, and the fixture stores the reference's OUTPUTS:
  * ``feats`` [n, 1088];
  * per parameter of ``visual_backbone``: sum / abs-sum / square-sum of its gradient, plus the full gradient of a few
    small / structurally distinct ones (conv1.weight, bn1.*, layer2.0.shortcut.*, layer4.1.bn2.*, linear2.bias) and a
    slice of two big ones;
  * the BatchNorm running statistics after the step (momentum 0.1, unbiased variance) for three layers.
Weights, patches and G are regenerated from seeds by the tests (oracle/weights.py).

Run:  python tests/golden/make_encoder_train_golden.py
"""
import importlib
import os
import sys

sys.dont_write_bytecode = True
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ref_import import REF, install_stubs  # noqa: E402

from oracle import weights as W  # noqa: E402

torch.set_num_threads(8)
install_stubs()
sys.path.insert(0, REF)
eg = importlib.import_module("model.backbones.efficient_gat")

FULL = ("conv1.weight", "bn1.weight", "bn1.bias", "layer2.0.shortcut.0.weight", "layer2.0.shortcut.1.weight",
        "layer2.0.shortcut.1.bias", "layer4.1.bn2.weight", "layer4.1.bn2.bias", "linear2.bias")
SLICED = ("layer1.0.conv1.weight", "layer3.0.conv1.weight")
RUNNING = ("bn1", "layer2.0.shortcut.1", "layer4.1.bn2")


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).numpy()


OUT = {}
for name, seed, n in (("tr_s0", 0, 4), ("tr_s1", 1, 6)):
    net = eg.Eff_GAT(steps=10, input_channels=4, output_channels=4, model="resnet18equiv",
                     visual_pretrained=False, architecture="transformer")
    net.visual_backbone.load_state_dict(W.make_encoder_state(seed), strict=True)
    net.train()
    x = W.make_patches(n, seed + 100)
    G = W.randn((n, 1088), seed + 200)
    feats = net.visual_features(x)
    (feats * G).sum().backward()
    OUT[f"{name}/feats"] = feats.detach().numpy()
    for k, p in net.visual_backbone.named_parameters():
        OUT[f"{name}/gstats/{k}"] = stats(p.grad)
        if k in FULL:
            OUT[f"{name}/grad/{k}"] = p.grad.numpy()
        if k in SLICED:
            OUT[f"{name}/gslice/{k}"] = p.grad[:3, :5].numpy()
    sd = net.visual_backbone.state_dict()
    for k in RUNNING:
        OUT[f"{name}/running/{k}.running_mean"] = sd[k + ".running_mean"].numpy()
        OUT[f"{name}/running/{k}.running_var"] = sd[k + ".running_var"].numpy()
    print(name, feats.shape, float(feats.abs().mean()), len([1 for _ in net.visual_backbone.parameters()]))

path = os.path.join(os.path.dirname(__file__), "encoder_train_v1.npz")
np.savez_compressed(path, **OUT)
print("wrote", path, os.path.getsize(path))
