"""Python owner of the packed piece-encoder weights + workspace (C ABI: da_encoder_* in
include/diffassemble_hip.h).

Replaces, for ``model='resnet18equiv'``, ``Eff_GAT.visual_features``
(/root/reference/puzzle_diff/model/backbones/efficient_gat.py:149-189): the P4 group-equivariant
ResNet-18 of resnet_equivariant.py over the 32x32 piece crops, eval mode.  This file is host logic
that runs ONCE per checkpoint: it expands every group-convolution weight into its rotated filter
bank (what groupy's trans_filter recomputes on every forward, splitgconv2d.py:15-22,71-75), folds
the eval-mode BatchNorm into it and lays it out for the NHWC implicit-GEMM kernels.  All per-piece
arithmetic runs in libdiffassemble_hip.so; there is no fallback.
"""
import torch

from . import _lib

_PREC = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "bf16": _lib.PREC_BF16}
BN_EPS = 1e-5
PLANES = (32, 64, 64, 128)


def conv_keys():
    """(conv key, bn key) of the 19 P4ConvP4 layers in state-dict order: per BasicBlock conv1, conv2 and,
    for the first block of layers 2-4, shortcut (resnet_equivariant.py:20-31,77-81)."""
    keys = []
    for li in range(1, 5):
        for bi in range(2):
            p = f"layer{li}.{bi}."
            keys += [(p + "conv1", p + "bn1"), (p + "conv2", p + "bn2")]
            if li > 1 and bi == 0:
                keys.append((p + "shortcut.0", p + "shortcut.1"))
    return keys


def p4_filter_bank(w):
    """[O, I, S, k, k] -> conv2d weight [O*4, I*S, k, k]: output rotation r uses the filter turned by r
    quarter turns with, for a P4 input (S = 4), its stabilizer planes shifted cyclically by r
    (make_gconv_indices.py:15-40 in closed form; channel order o*4+r / i*S+s as splitgconv2d.py:72-83)."""
    O, I, S, k, _ = w.shape
    banks = [torch.rot90(torch.roll(w, shifts=r, dims=2) if S == 4 else w, r, dims=(3, 4)) for r in range(4)]
    return torch.stack(banks, dim=1).reshape(O * 4, I * S, k, k)


def _fold_bn(sd, bn, bank):
    """eval-mode BatchNorm3d over [B, C, 4, H, W] (one statistic per plane, shared by its 4 rotations)
    folded into the bank: returns (scaled bank, bias[C*4])."""
    inv = sd[bn + ".weight"].double() * torch.rsqrt(sd[bn + ".running_var"].double() + BN_EPS)
    bias = sd[bn + ".bias"].double() - sd[bn + ".running_mean"].double() * inv
    inv4, bias4 = inv.repeat_interleave(4), bias.repeat_interleave(4)
    return bank.double() * inv4.view(-1, 1, 1, 1), bias4


def _halo_linear(wt, C, H):
    """nn.Linear weight over the reference's NCHW flatten of [C, H, H] -> columns over the zero-haloed NHWC
    map [(H+2), (H+2), C]; halo columns are zero."""
    F = wt.shape[0]
    out = torch.zeros(F, H + 2, H + 2, C, dtype=wt.dtype, device=wt.device)
    out[:, 1:H + 1, 1:H + 1, :] = wt.reshape(F, C, H, H).permute(0, 2, 3, 1)
    return out.reshape(F, -1)


class EncoderEngine:
    """Packed P4 ResNet-18 encoder.  ``sd``: state dict with the reference's ``ResNet18()`` keys
    (``visual_backbone.*`` of an ``Eff_GAT`` checkpoint with the prefix stripped)."""

    FEATS = _lib.ENCODER_FEATS

    def __init__(self, sd, *, precision="bf16", device=None, chunk=None):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.DaError("EncoderEngine needs a ROCm device (no CPU path in diffassemble_amd)")
        self.lib = _lib.lib()
        self.precision, self.prec = precision, _PREC[precision]
        self.act_dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.chunk = chunk
        sd = {k: v.detach().to(self.device) for k, v in sd.items() if v.is_floating_point()}
        self._keep = []
        w = _lib.DaEncoderWeights()
        w.n_convs = _lib.ENCODER_CONVS

        def keep(t, dtype):
            t = t.to(dtype).contiguous()
            self._keep.append(t)
            return t.data_ptr()

        bank, bias = _fold_bn(sd, "bn1", p4_filter_bank(sd["conv1.weight"]))
        w.stem_w, w.stem_b = keep(bank.reshape(128, 27), torch.float32), keep(bias, torch.float32)
        for i, (ck, bk) in enumerate(conv_keys()):
            bank, bias = _fold_bn(sd, bk, p4_filter_bank(sd[ck + ".weight"]))
            # K ordered tap-major, channel-minor: [Cout, ky, kx, Cin] -- a tap is contiguous in NHWC
            w.conv_w[i] = keep(bank.permute(0, 2, 3, 1).reshape(bank.shape[0], -1), self.act_dtype)
            w.conv_b[i] = keep(bias, torch.float32)
        w.lin1_w = keep(_halo_linear(sd["linear1.weight"], 256, 8), self.act_dtype)
        w.lin1_b = keep(sd["linear1.bias"], torch.float32)
        w.lin2_w = keep(_halo_linear(sd["linear2.weight"], 512, 4), self.act_dtype)
        w.lin2_b = keep(sd["linear2.bias"], torch.float32)
        self.w = w
        self._ws = None
        self._ws_key = None

    def _chunk_for(self, n):
        if self.chunk:
            return int(self.chunk)
        return 1024 if n >= 1024 else max(8, (n + 7) // 8 * 8)

    def forward(self, patches, out=None):
        """patches [N, 3, 32, 32] fp32 in [0, 1] (device) -> patch_feats [N, 1088] in the act dtype."""
        if patches.device.type != "cuda":
            raise _lib.DaError("EncoderEngine.forward: patches must live on the ROCm device")
        assert patches.dim() == 4 and tuple(patches.shape[1:]) == (3, 32, 32), tuple(patches.shape)
        x = patches.detach().to(torch.float32).contiguous()
        n = x.shape[0]
        chunk = self._chunk_for(n)
        key = (n, chunk)
        fresh = self._ws_key != key
        if fresh:
            nbytes = self.lib.da_encoder_workspace_bytes(self.prec, n, chunk)
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        if out is None:
            out = torch.empty(n, self.FEATS, dtype=self.act_dtype, device=self.device)
        assert out.dtype == self.act_dtype and out.shape[0] == n and out.stride(1) == 1
        _lib.check(self.lib.da_encoder_forward(self.prec, self.w, n, _lib.ptr(x), _lib.ptr(out), out.stride(0),
                                               _lib.ptr(self._ws), self._ws.numel(), chunk, 1 if fresh else 0,
                                               _lib.stream_ptr(self.device)))
        return out
