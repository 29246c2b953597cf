"""Generate tests/golden/encoder_v1.npz from the REFERENCE's own encoder code.

BUILD-CONTAINER ONLY: imports /root/reference/puzzle_diff/model/backbones/{efficient_gat,
resnet_equivariant}.py and the groupy package from where they lie (under the inert stubs of
ref_import.py), loads the seeded weights of oracle/weights.make_encoder_state into the reference's
``Eff_GAT(model='resnet18equiv').visual_backbone`` and stores the reference's OUTPUTS:
  * the filter-transformation index arrays of make_gconv_indices.py for k = 1, 3;
  * visual_features() of seeded patches (eval mode) + per-stage statistics and slices.
Weights and inputs are regenerated from seeds by the tests.

Run:  python tests/golden/make_encoder_golden.py
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
idx = importlib.import_module("model.backbones.groupy.gconv.make_gconv_indices")

OUT = {}
for k in (1, 3):
    OUT[f"inds/c4_z2_k{k}"] = idx.make_c4_z2_indices(k)
    OUT[f"inds/c4_p4_k{k}"] = idx.make_c4_p4_indices(k)


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).float().numpy()


for name, seed, n in (("enc_s0", 0, 3), ("enc_s1", 1, 5)):
    net = eg.Eff_GAT(steps=10, input_channels=4, output_channels=4, model="resnet18equiv",
                     visual_pretrained=False, architecture="transformer")
    sd = W.make_encoder_state(seed)
    missing, unexpected = net.visual_backbone.load_state_dict(sd, strict=True)
    net.eval()
    x = W.make_patches(n, seed + 100)
    stages = []
    hooks = [getattr(net.visual_backbone, f"layer{i}").register_forward_hook(lambda m, a, o: stages.append(o.detach()))
             for i in range(1, 5)]
    with torch.no_grad():
        feats = net.visual_features(x)
    for h in hooks:
        h.remove()
    OUT[f"{name}/feats"] = feats.numpy()
    for i, s in enumerate(stages):
        OUT[f"{name}/stage{i + 1}_shape"] = np.asarray(s.shape)
        OUT[f"{name}/stage{i + 1}_stats"] = stats(s)
        OUT[f"{name}/stage{i + 1}_slice"] = s[:, :4, :, :3, :5].numpy()
    print(name, feats.shape, float(feats.abs().mean()), [tuple(s.shape) for s in stages])

path = os.path.join(os.path.dirname(__file__), "encoder_v1.npz")
np.savez_compressed(path, **OUT)
print("wrote", path, os.path.getsize(path))
