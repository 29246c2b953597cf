"""Module surface of the reference's equivariant encoder
(/root/reference/puzzle_diff/model/backbones/resnet_equivariant.py): ``ResNet18()`` with the same
state-dict keys (``conv1.weight [32,3,1,3,3]``, ``layerL.B.conv{1,2}.weight [O,I,4,3,3]``,
``layerL.0.shortcut.{0,1}.*``, BatchNorm3d ``weight/bias/running_mean/running_var/num_batches_tracked``,
``linear1 [544,16384]``, ``linear2 [544,8192]``), so checkpoints load unchanged.

The modules only HOLD parameters.  The arithmetic of an eval-mode forward runs in libdiffassemble_hip.so through
``diffassemble_amd.encoder.EncoderEngine`` (BatchNorm folded, bf16 or fp32); in train() mode -- BatchNorm3d on batch
statistics, the reference's behaviour inside ``training_step`` -- forward AND backward run through
``diffassemble_amd.encoder_train.EncoderTrainEngine`` (fp32 HIP primitives), joined to torch autograd by
``_EncoderTrainFunction``: gradients land in ``param.grad``, running statistics are updated like torch does.  No torch
fallback in either mode.
"""
import math

import torch
import torch.nn as nn

from ...encoder import EncoderEngine


class _EncoderTrainFunction(torch.autograd.Function):
    """patch_feats = encoder(patches) in train() mode.  Parameter gradients are ADDED into ``param.grad`` by the engine;
    autograd only sees the ``anchor`` (one small parameter, zero gradient returned) so that the output joins the graph and
    a DistributedDataParallel wrapper's reducer gets the one hook it needs to close its iteration (see
    ``diffassemble_amd.train.DenoiserTrainFn``).  One forward must be followed by its backward before the next forward
    (the engine owns the activations)."""

    @staticmethod
    def forward(ctx, eng, patches, anchor):
        ctx.eng, ctx.anchor_shape = eng, anchor.shape
        return eng.forward(patches).clone()

    @staticmethod
    def backward(ctx, d_feats):
        ctx.eng.backward(d_feats)
        return None, None, torch.zeros(ctx.anchor_shape, dtype=torch.float32, device=d_feats.device)


class _GConv(nn.Module):
    """Parameter holder of groupy's SplitGConv2D (splitgconv2d.py:25-59): weight [O, I, S, k, k], no bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stab):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, stab, kernel_size, kernel_size))
        stdv = 1.0 / math.sqrt(in_channels * kernel_size * kernel_size)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)


def P4ConvZ2(i, o, k):
    return _GConv(i, o, k, 1)


def P4ConvP4(i, o, k):
    return _GConv(i, o, k, 4)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        self.conv1 = P4ConvP4(in_planes, planes, 3)
        self.bn1 = nn.BatchNorm3d(planes)
        self.conv2 = P4ConvP4(planes, planes, 3)
        self.bn2 = nn.BatchNorm3d(planes)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = nn.Sequential(P4ConvP4(in_planes, planes, 1), nn.BatchNorm3d(planes))


class ResNet(nn.Module):
    def __init__(self, num_blocks=(2, 2, 2, 2), precision="bf16"):
        super().__init__()
        assert tuple(num_blocks) == (2, 2, 2, 2), "only the ResNet18() the reference instantiates is built"
        self.precision = precision
        self.conv1 = P4ConvZ2(3, 32, 3)
        self.bn1 = nn.BatchNorm3d(32)
        cin = 32
        for li, (planes, stride) in enumerate(((32, 1), (64, 2), (64, 2), (128, 2)), start=1):
            blocks = []
            for s in (stride, 1):
                blocks.append(BasicBlock(cin, planes, s))
                cin = planes
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.linear1 = nn.Linear(64 * 4 * 8 * 8, 544)
        self.linear2 = nn.Linear(128 * 4 * 4 * 4, 544)
        self._engine = None
        self._engine_key = None
        self._train_engine = None
        # Set to True to run a FROZEN encoder on its RUNNING statistics while the LightningModule is in train mode (what
        # "frozen" usually means).  The reference's freeze_backbone=True still normalises with batch statistics
        # (efficient_gat.py:152-154 under model.train()), which is what happens here by default: train() mode = batch
        # statistics, with or without gradients.
        self.frozen_eval_stats = False
        # train() mode storage of activations / activation gradients: "fp32" (the reference's arithmetic, the parity mode)
        # or "bf16" (bf16 maps and matrix cores, fp32 BatchNorm arithmetic, master weights and parameter gradients: 2.8x
        # faster, half the activation memory; DIFFASSEMBLE_TRAIN_PRECISION sets the default)
        import os
        self.train_precision = os.environ.get("DIFFASSEMBLE_TRAIN_PRECISION", "fp32")

    def engine(self):
        """Packed weights, rebuilt when a parameter / buffer was replaced or modified in place."""
        sd = self.state_dict()
        key = (self.precision,) + tuple((t.data_ptr(), t._version) for t in sd.values())
        if self._engine is None or self._engine_key != key:
            dev = self.conv1.weight.device
            self._engine = EncoderEngine(sd, precision=self.precision, device=dev)
            self._engine_key = key
        return self._engine

    def patch_features(self, patch_rgb):
        """[N, 3, 32, 32] in [0, 1], NOT normalised (the kernel normalises) -> [N, 1088] =
        cat(linear1(out3), linear2(out4)), the two maps Eff_GAT.visual_features keeps."""
        if self.training and not self.frozen_eval_stats:
            te = self._train_engine
            if te is None or te.device != patch_rgb.device or te.precision != self.train_precision:
                from ...encoder_train import EncoderTrainEngine
                self._train_engine = EncoderTrainEngine(self, patch_rgb.device, precision=self.train_precision)
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                anchor = min((p for p in self.parameters() if p.requires_grad), key=lambda p: p.numel())
                return _EncoderTrainFunction.apply(self._train_engine, patch_rgb, anchor)
            return self._train_engine.forward(patch_rgb).clone()           # frozen (no_grad) encoder in train mode
        return self.engine().forward(patch_rgb)

    def forward(self, x):
        raise NotImplementedError(
            "ResNet.forward on pre-normalised input is not exposed: Eff_GAT.visual_features calls patch_features "
            "(normalisation is fused into the stem kernel)")


def ResNet18(precision="bf16"):
    return ResNet((2, 2, 2, 2), precision=precision)
