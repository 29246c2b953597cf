"""The CPU oracle of the piece encoder (oracle/encoder.py) against the fixtures produced from the
reference's own ResNet18() / groupy code (tests/golden/make_encoder_golden.py).  CPU-only, fp32."""
import os

import numpy as np
import pytest
import torch

from oracle import encoder as E
from oracle import weights as W

RTOL = 1e-4
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "encoder_v1.npz"))


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def stats(t):
    t = t.double()
    return torch.stack([t.sum(), t.abs().sum(), (t * t).sum()]).float()


@pytest.mark.parametrize("k", [1, 3])
@pytest.mark.parametrize("stab", [1, 4])
def test_filter_bank_matches_reference_indices(k, stab):
    """The closed-form rotation / stabilizer shift of p4_filter_bank against the reference's index
    arrays (make_gconv_indices.py:15-40) applied the way trans_filter does (splitgconv2d.py:15-22)."""
    inds = GOLD[f"inds/c4_{'z2' if stab == 1 else 'p4'}_k{k}"].astype(np.int64)       # [4, S, k, k, 3]
    assert inds.shape == (4, stab, k, k, 3)
    w = torch.arange(5 * 3 * stab * k * k, dtype=torch.float32).reshape(5, 3, stab, k, k)
    flat = inds.reshape(-1, 3)
    ref = w[:, :, flat[:, 0], flat[:, 1], flat[:, 2]].reshape(5, 3, 4, stab, k, k).permute(0, 2, 1, 3, 4, 5)
    ref = ref.reshape(5 * 4, 3 * stab, k, k)
    assert torch.equal(E.p4_filter_bank(w), ref)


@pytest.mark.parametrize("name,seed,n", [("enc_s0", 0, 3), ("enc_s1", 1, 5)])
def test_visual_features_match_reference(name, seed, n):
    sd = W.make_encoder_state(seed)
    stages = []
    feats = E.visual_features(sd, W.make_patches(n, seed + 100), stages)
    assert rel_err(feats, GOLD[f"{name}/feats"]) < RTOL
    for i, s in enumerate(stages, start=1):
        assert list(s.shape) == list(GOLD[f"{name}/stage{i}_shape"])
        assert rel_err(stats(s), GOLD[f"{name}/stage{i}_stats"]) < RTOL, i
        assert rel_err(s[:, :4, :, :3, :5], GOLD[f"{name}/stage{i}_slice"]) < RTOL, i


def test_encoder_is_rotation_equivariant():
    """Domain property: rotating a piece by 90 degrees rotates every stage's feature maps and cyclically
    shifts the 4 stabilizer planes (what makes the encoder P4-equivariant)."""
    sd = W.make_encoder_state(2)
    x = (W.make_patches(2, 7) - E.MEAN) / E.STD
    a, b = [], []
    E.resnet18_p4(sd, x, a)
    E.resnet18_p4(sd, torch.rot90(x, 1, dims=(2, 3)), b)
    # stride-2 convs with padding 1 on even sizes sample off-centre, so exact equivariance holds for the
    # stride-1 stage only
    exp = torch.roll(torch.rot90(a[0], 1, dims=(3, 4)), shifts=1, dims=2)
    assert rel_err(b[0], exp) < 1e-5
