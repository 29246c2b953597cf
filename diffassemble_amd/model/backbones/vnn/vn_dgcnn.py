"""Module surface of the reference's 3D piece encoder (/root/reference/puzzle_diff/model/backbones/vnn/vn_dgcnn.py,
vn_layers.py): ``VN_DGCNN(feat_dim, inv)`` with the same state-dict keys (``convL.map_to_feat.weight``,
``convL.batchnorm.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``, ``convL.map_to_dir.weight``,
``VnInv.vn{1,2}.*``, ``VnInv.vn_lin.weight``, ``linear0.{weight,bias}``), so checkpoints load unchanged.

The modules only HOLD parameters; an eval-mode forward runs in libdiffassemble_hip.so through
``diffassemble_amd.pcd_encoder.PcdEncoderEngine`` (kNN, vector-neuron layers and pooling as HIP kernels; no torch
fallback).  Training-mode BatchNorm (batch statistics) and the backward through the encoder are not built."""
import torch.nn as nn

from ....pcd_encoder import PcdEncoderEngine


class VNBatchNorm(nn.Module):
    def __init__(self, num_features, dim):
        super().__init__()
        self.dim = dim
        self.bn = nn.BatchNorm1d(num_features) if dim in (3, 4) else nn.BatchNorm2d(num_features)


class VNLinearLeakyReLU(nn.Module):
    """Parameter holder of vn_layers.py:50-72."""

    def __init__(self, in_channels, out_channels, dim=5, share_nonlinearity=False, negative_slope=0.2):
        super().__init__()
        assert negative_slope == 0.2, "the kernels hard-wire the reference's slope 0.2"
        self.dim, self.negative_slope = dim, negative_slope
        self.map_to_feat = nn.Linear(in_channels, out_channels, bias=False)
        self.batchnorm = VNBatchNorm(out_channels, dim=dim)
        self.map_to_dir = nn.Linear(in_channels, 1 if share_nonlinearity else out_channels, bias=False)


class VNStdFeature(nn.Module):
    """Parameter holder of vn_layers.py:179-206: built by the reference, its result is discarded (vn_dgcnn.py:67-68)."""

    def __init__(self, in_channels, dim=4, normalize_frame=False):
        super().__init__()
        self.vn1 = VNLinearLeakyReLU(in_channels, in_channels // 2, dim=dim)
        self.vn2 = VNLinearLeakyReLU(in_channels // 2, in_channels // 4, dim=dim)
        self.vn_lin = nn.Linear(in_channels // 4, 2 if normalize_frame else 3, bias=False)


class VN_DGCNN(nn.Module):
    def __init__(self, feat_dim, inv=False):
        super().__init__()
        self.n_knn = 20
        self.inv = inv
        c = 64 // 3
        self.conv1 = VNLinearLeakyReLU(2, c)
        self.conv2 = VNLinearLeakyReLU(c, c)
        self.conv3 = VNLinearLeakyReLU(c * 2, c)
        self.conv4 = VNLinearLeakyReLU(c, c)
        self.conv5 = VNLinearLeakyReLU(c * 2, c)
        self.VnInv = VNStdFeature(2 * feat_dim, dim=3, normalize_frame=False)
        self.conv6 = VNLinearLeakyReLU(c * 3, feat_dim, dim=4, share_nonlinearity=True)
        self.linear0 = nn.Linear(3, 2 * feat_dim)
        self._engine, self._engine_key = None, None

    def engine(self):
        """Packed weights, rebuilt when a parameter / buffer was replaced or modified in place."""
        sd = self.state_dict()
        key = (self.inv,) + tuple((t.data_ptr(), t._version) for t in sd.values())
        if self._engine is None or self._engine_key != key:
            self._engine = PcdEncoderEngine(sd, inv=self.inv, device=self.conv1.map_to_feat.weight.device)
            self._engine_key = key
        return self._engine

    def forward(self, x):
        """vn_dgcnn.py:34-74.  x [P, N, 3] (N >= 20) -> [P, 6 feat_dim], or [P, 2 feat_dim] when ``inv``."""
        if self.training:
            raise NotImplementedError(
                "the HIP point-cloud encoder implements eval-mode BatchNorm only: call .eval() (sampling / validation) "
                "or pass precomputed pcd_feats when training")
        return self.engine().forward(x)
