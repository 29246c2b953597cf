"""Oracle restatement of the 2D piece encoder (TEST INFRASTRUCTURE, oracle/__init__.py).

The reference's ``model='resnet18equiv'`` encoder: a P4 (90-degree rotation) group-equivariant
ResNet-18 over the 32x32 piece crops.  Pure torch fp32 on CPU.  ``sd`` has the key layout of the
reference's ``ResNet18().state_dict()`` (``Eff_GAT.visual_backbone.*`` in a checkpoint).

Followed, by file:line under /root/reference/puzzle_diff/model/backbones/:
  groupy/gconv/pytorch_gconv/splitgconv2d.py:15-22,70-92   trans_filter + conv2d on [B, C*S, H, W]
  groupy/gconv/make_gconv_indices.py:15-40                 which filter tap / stabilizer feeds which
  resnet_equivariant.py:14-38 (BasicBlock), :70-112 (ResNet), :113-114 (ResNet18 = [2,2,2,2])
  efficient_gat.py:149-189                                 normalise, encoder, cat(feats[2], feats[3])

The filter transformation is written in closed form instead of through the reference's index
arrays: output rotation r uses the filter rotated by r quarter turns, and -- for a P4 input -- its
stabilizer planes cyclically shifted by r:
    tw[o, r, i, s, u, v] = w[o, i, (s - r) mod 4, rot90^r(u, v)]
tests/test_oracle.py checks this against the reference's own index arrays (fixture).
"""
import torch
import torch.nn.functional as F

from .weights import encoder_conv_specs

BN_EPS = 1e-5          # nn.BatchNorm3d default (resnet_equivariant.py:22)


def p4_filter_bank(w):
    """w [O, I, S, k, k] (S = 1: Z2 input, S = 4: P4 input) -> conv2d weight [O*4, I*S, k, k]
    (splitgconv2d.py:15-22,71-75; output channel = o*4 + r, input channel = i*S + s)."""
    O, I, S, k, _ = w.shape
    banks = []
    for r in range(4):
        wr = torch.roll(w, shifts=r, dims=2) if S == 4 else w
        banks.append(torch.rot90(wr, r, dims=(3, 4)))
    tw = torch.stack(banks, dim=1)                       # [O, 4, I, S, k, k]
    return tw.reshape(O * 4, I * S, k, k).contiguous()


def gconv(sd, key, x, stride, padding):
    """SplitGConv2D.forward (splitgconv2d.py:70-92), bias=False everywhere in the ResNet."""
    B = x.shape[0]
    tw = p4_filter_bank(sd[key + ".weight"])
    y = F.conv2d(x.reshape(B, tw.shape[1], x.shape[-2], x.shape[-1]), tw, None, stride=stride, padding=padding)
    return y.reshape(B, tw.shape[0] // 4, 4, y.shape[-2], y.shape[-1])


def bn_eval(sd, key, x, stats=None):
    """nn.BatchNorm3d over [B, C, 4, H, W]: per plane C, shared by the 4 rotations.  Eval mode (running statistics);
    with ``stats`` (a dict) TRAINING mode: batch statistics over (B, 4, H, W), biased variance for the normalisation,
    and the running-statistics update torch applies (momentum 0.1, unbiased variance) recorded into ``stats``."""
    sh = (1, -1, 1, 1, 1)
    if stats is not None:
        mu, var = x.mean((0, 2, 3, 4)), x.var((0, 2, 3, 4), unbiased=False)
        n = x.numel() // x.shape[1]
        stats[key + ".running_mean"] = (0.9 * sd[key + ".running_mean"] + 0.1 * mu).detach()
        stats[key + ".running_var"] = (0.9 * sd[key + ".running_var"] + 0.1 * var * n / (n - 1)).detach()
        return (x - mu.view(sh)) * (torch.rsqrt(var + BN_EPS) * sd[key + ".weight"]).view(sh) + sd[key + ".bias"].view(sh)
    inv = torch.rsqrt(sd[key + ".running_var"] + BN_EPS) * sd[key + ".weight"]
    return (x - sd[key + ".running_mean"].view(sh)) * inv.view(sh) + sd[key + ".bias"].view(sh)


def basic_block(sd, p, x, stride, has_shortcut, stats=None):
    """resnet_equivariant.py:33-38."""
    out = F.relu(bn_eval(sd, p + "bn1", gconv(sd, p + "conv1", x, stride, 1), stats))
    out = bn_eval(sd, p + "bn2", gconv(sd, p + "conv2", out, 1, 1), stats)
    sc = bn_eval(sd, p + "shortcut.1", gconv(sd, p + "shortcut.0", x, stride, 0), stats) if has_shortcut else x
    return F.relu(out + sc)


def resnet18_p4(sd, x, collect=None, stats=None):
    """resnet_equivariant.py:93-112 -> [out1, out2, linear1(out3), linear2(out4)].  ``stats``: see bn_eval."""
    B = x.shape[0]
    out = F.relu(bn_eval(sd, "bn1", gconv(sd, "conv1", x, 1, 1), stats))
    outs = []
    for li in range(1, 5):
        for bi in range(2):
            p = f"layer{li}.{bi}."
            stride = 2 if (li > 1 and bi == 0) else 1
            out = basic_block(sd, p, out, stride, (p + "shortcut.0.weight") in sd, stats)
        outs.append(out)
        if collect is not None:
            collect.append(out)
    f3 = F.linear(outs[2].reshape(B, -1), sd["linear1.weight"], sd["linear1.bias"])
    f4 = F.linear(outs[3].reshape(B, -1), sd["linear2.weight"], sd["linear2.bias"])
    return [outs[0], outs[1], f3, f4]


MEAN = torch.tensor([0.4850, 0.4560, 0.4060])[None, :, None, None]     # efficient_gat.py:109-112
STD = torch.tensor([0.2290, 0.2240, 0.2250])[None, :, None, None]


def visual_features(sd, patch_rgb, collect=None, stats=None):
    """Eff_GAT.visual_features for model='resnet18equiv', all_equivariant=False
    (efficient_gat.py:149-189): [N, 3, 32, 32] -> patch_feats [N, 1088].  With ``stats`` (a dict) the BatchNorms run in
    TRAINING mode, as they do inside ``training_step`` (spatial_diffusion.py:450); torch autograd through this function
    is the gradient oracle of the encoder's backward."""
    feats = resnet18_p4(sd, (patch_rgb - MEAN) / STD, collect, stats)
    n = patch_rgb.shape[0]
    return torch.cat([feats[2].reshape(n, -1), feats[3].reshape(n, -1)], -1)


__all__ = ["p4_filter_bank", "gconv", "resnet18_p4", "visual_features", "encoder_conv_specs"]
