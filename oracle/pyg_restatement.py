"""Restatement of the third-party arithmetic the reference's hot path calls.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at this boundary:
neither dependency is vendored under /root/reference nor pinned by it.

1. torch_geometric (``pyg``, unpinned: singularity/build/conda_env.yaml:12; with
   pytorch==1.12.1 that means PyG 2.1-2.3).  Call sites in the reference:
   puzzle_diff/model/backbones/Transformer_GNN.py:10-24,32,38 and
   puzzle_diff/model/backbones/exophormer_gnn.py:139-153,203,205.
   Restated from PyG's published ``TransformerConv`` (defaults concat=True, beta=False,
   dropout=0, edge_dim=None, bias=True, root_weight=True, aggr='add', flow
   source_to_target) and ``torch_geometric.utils.softmax``:
       alpha_e = <q_i, k_j> / sqrt(C)           (edge e = j -> i; edge_index[0]=j, [1]=i)
       alpha   = exp(alpha - segmax_i) / (segsum_i + 1e-16)
       out_i   = sum_e alpha_e * v_j ; concat heads ; + lin_skip(x_i)

2. pytorch3d.transforms (version unknown; not listed in conda_env.yaml).  Call sites:
   puzzle_diff/model/backbones/efficient_gat_3d.py:4,217 and
   puzzle_diff/model/spatial_diffusion_3d_test_double_diffusion.py:27,637-657.
   Restated from pytorch3d 0.7.x ``rotation_conversions.py`` (real-part-first
   quaternions, ``matrix_to_quaternion`` ending in ``standardize_quaternion``).
   Older releases did not standardise the sign; compare rotations modulo q == -q
   where that matters.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- PyG
def segment_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.softmax(src, index, num_nodes=N) for src [E, H]."""
    H = src.shape[1]
    idx = index[:, None].expand(-1, H)
    src_max = torch.full((num_nodes, H), float("-inf"), dtype=src.dtype)
    src_max = src_max.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros((num_nodes, H), dtype=src.dtype).index_add_(0, index, out) + 1e-16
    return out / out_sum.index_select(0, index)


def transformer_conv(x, edge_index, wq, bq, wk, bk, wv, bv, ws, bs, heads):
    """PyG TransformerConv.forward(x, edge_index, return_attention_weights=True).

    Weights are [out, in] like nn.Linear.  Returns (out [N, H*C], alpha [E, H])."""
    N = x.shape[0]
    HC = wq.shape[0]
    C = HC // heads
    q = F.linear(x, wq, bq).view(N, heads, C)
    k = F.linear(x, wk, bk).view(N, heads, C)
    v = F.linear(x, wv, bv).view(N, heads, C)
    src, dst = edge_index[0], edge_index[1]
    alpha = (q.index_select(0, dst) * k.index_select(0, src)).sum(-1) / math.sqrt(C)
    alpha = segment_softmax(alpha, dst, N)
    msg = v.index_select(0, src) * alpha[:, :, None]
    out = torch.zeros((N, heads, C), dtype=x.dtype).index_add_(0, dst, msg)
    out = out.reshape(N, HC) + F.linear(x, ws, bs)
    return out, alpha


class TransformerConv(nn.Module):
    """Module form with PyG's parameter names (lin_key/lin_query/lin_value/lin_skip),
    used (a) as the arithmetic stub when the reference is imported by
    tests/golden/make_golden.py and (b) to check state-dict key layout."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, beta=False,
                 dropout=0.0, edge_dim=None, bias=True, root_weight=True, **kwargs):
        super().__init__()
        assert concat and not beta and dropout == 0.0 and edge_dim is None and bias and root_weight
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin_key = nn.Linear(in_channels, heads * out_channels)
        self.lin_query = nn.Linear(in_channels, heads * out_channels)
        self.lin_value = nn.Linear(in_channels, heads * out_channels)
        self.lin_skip = nn.Linear(in_channels, heads * out_channels)

    def forward(self, x, edge_index, edge_attr=None, return_attention_weights=None):
        out, alpha = transformer_conv(
            x, edge_index,
            self.lin_query.weight, self.lin_query.bias,
            self.lin_key.weight, self.lin_key.bias,
            self.lin_value.weight, self.lin_value.bias,
            self.lin_skip.weight, self.lin_skip.bias, self.heads)
        if return_attention_weights:
            return out, (edge_index, alpha)
        return out


# ----------------------------------------------------------------------- pytorch3d
def _sqrt_positive_part(x):
    ret = torch.zeros_like(x)
    m = x > 0
    ret[m] = torch.sqrt(x[m])
    return ret


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(matrix):
    """pytorch3d.transforms.matrix_to_quaternion (0.7.x), real part first."""
    batch_dim = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(
        matrix.reshape(batch_dim + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([
        1.0 + m00 + m11 + m22,
        1.0 + m00 - m11 - m22,
        1.0 - m00 + m11 - m22,
        1.0 - m00 - m11 + m22], dim=-1))
    quat_by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1),
    ], dim=-2)
    flr = torch.tensor(0.1, dtype=q_abs.dtype)
    quat_candidates = quat_by_rijk / (2.0 * q_abs[..., None].max(flr))
    out = quat_candidates[F.one_hot(q_abs.argmax(dim=-1), num_classes=4) > 0.5, :].reshape(
        batch_dim + (4,))
    return standardize_quaternion(out)


def quaternion_to_matrix(quaternions):
    """pytorch3d.transforms.quaternion_to_matrix, real part first."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def quaternion_apply(quaternion, point):
    """pytorch3d.transforms.quaternion_apply: rotate points by unit quaternions (real part first), as the
    product q (0, p) q^-1 restricted to its vector part (restated from the published algorithm)."""
    w, u = quaternion[..., :1], quaternion[..., 1:]
    t = 2.0 * torch.cross(u, point, dim=-1)
    return point + w * t + torch.cross(u, t, dim=-1)


class _KNN:
    def __init__(self, dists, idx):
        self.dists, self.idx, self.knn = dists, idx, None


def knn_points(p1, p2, lengths1=None, lengths2=None, K=1, **kwargs):
    """pytorch3d.ops.knn_points for K = 1 on equal-length clouds: SQUARED euclidean distance to, and index of, the
    nearest point of p2 for every point of p1 (brute force)."""
    assert K == 1
    d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    dist, idx = d.min(-1)
    return _KNN(dist[..., None], idx[..., None])
