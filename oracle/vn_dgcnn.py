"""CPU restatement (TEST INFRASTRUCTURE, see oracle/__init__.py) of the reference's 3D piece encoder, eval mode:
``VN_DGCNN.forward`` (puzzle_diff/model/backbones/vnn/vn_dgcnn.py:34-74) with its helpers ``get_graph_feature`` (:84-111)
and ``knn`` (:114-120), over the vector-neuron layers of vnn/vn_layers.py (``VNLinearLeakyReLU`` :50-91, ``VNBatchNorm``
:133-154 with running statistics, ``mean_pool`` :175-176).  Pinned by tests/golden/golden_v3.npz ("pcd_enc/*", produced by
the reference's own module on the same seeded clouds and weights; tests/golden/make_golden_v3.py).

Arrays are numpy fp32.  A vector-neuron feature map is kept POINT-MAJOR here: [B, N, C, 3] (the reference holds
[B, C, 3, N]); only the layout differs, every arithmetic step follows the reference line by line."""
import numpy as np

EPS = np.float32(1e-6)       # vn_layers.py:11
BN_EPS = np.float32(1e-5)    # torch.nn.BatchNorm default
K_NN = 20                    # vn_dgcnn.py:9
SLOPE = np.float32(0.2)


def knn(x, k=K_NN):
    """vn_dgcnn.py:114-120.  x [B, N, F] -> idx [B, N, k]: the k largest of -|xi|^2 + 2 xi.xj - |xj|^2 (self included)."""
    inner = -2 * np.einsum("bif,bjf->bij", x, x, dtype=np.float32)
    xx = (x * x).sum(-1, dtype=np.float32)
    pd = -xx[:, :, None] - inner - xx[:, None, :]
    # topk returns the k largest sorted descending; argsort of the negated row (stable) matches it up to ties
    return np.argsort(-pd, axis=-1, kind="stable")[..., :k]


def graph_feature(x, k=K_NN, idx=None):
    """vn_dgcnn.py:84-111.  x [B, N, C, 3] -> [B, N, k, 2C, 3] = cat(x_j - x_i, x_i) over the k nearest j of i, the
    neighbours taken in the flattened C*3 feature space."""
    B, N, C, _ = x.shape
    if idx is None:
        idx = knn(x.reshape(B, N, C * 3), k)
    nb = np.take_along_axis(x[:, None, :, :, :], idx[:, :, :, None, None], axis=2)     # [B, N, k, C, 3]
    ctr = np.broadcast_to(x[:, :, None, :, :], nb.shape)
    return np.concatenate([nb - ctr, ctr], axis=3), idx


def vn_linear_leaky_relu(x, w_feat, w_dir, bn):
    """vn_layers.py:74-91 (eval).  x [..., Cin, 3]; w_feat [Cout, Cin]; w_dir [Cout | 1, Cin];
    bn = (weight, bias, running_mean, running_var), each [Cout]."""
    p = np.einsum("oc,...ck->...ok", w_feat, x).astype(np.float32)
    # VNBatchNorm (vn_layers.py:145-154): batch-norm the vector NORM, keep the direction
    norm = np.sqrt((p * p).sum(-1, dtype=np.float32)) + EPS
    g, b, mu, var = bn
    norm_bn = (norm - mu) / np.sqrt(var + BN_EPS) * g + b
    p = p / norm[..., None] * norm_bn[..., None]
    d = np.einsum("oc,...ck->...ok", w_dir, x).astype(np.float32)              # one shared direction when w_dir has 1 row
    dot = (p * d).sum(-1, keepdims=True, dtype=np.float32)
    mask = (dot >= 0).astype(np.float32)
    dsq = (d * d).sum(-1, keepdims=True, dtype=np.float32)
    return (SLOPE * p + (1 - SLOPE) * (mask * p + (1 - mask) * (p - (dot / (dsq + EPS)) * d))).astype(np.float32)


def _layer(sd, name):
    bn = tuple(np.asarray(sd[f"{name}.batchnorm.bn.{k}"], np.float32) for k in ("weight", "bias", "running_mean", "running_var"))
    return np.asarray(sd[f"{name}.map_to_feat.weight"], np.float32), np.asarray(sd[f"{name}.map_to_dir.weight"], np.float32), bn


def forward(sd, pts, inv=False, return_intermediates=False):
    """vn_dgcnn.py:34-74.  pts [B, N, 3] -> [B, 6 * feat_dim] (inv=False) or [B, 2 * feat_dim] (inv=True)."""
    sd = {k: np.asarray(v) for k, v in sd.items()}
    x = np.asarray(pts, np.float32)[:, :, None, :]                               # [B, N, 1, 3]
    f, idx1 = graph_feature(x)
    f = vn_linear_leaky_relu(f, *_layer(sd, "conv1"))
    f = vn_linear_leaky_relu(f, *_layer(sd, "conv2"))
    x1 = f.mean(2, dtype=np.float32)                                             # pool over the k neighbours
    f, idx2 = graph_feature(x1)
    f = vn_linear_leaky_relu(f, *_layer(sd, "conv3"))
    f = vn_linear_leaky_relu(f, *_layer(sd, "conv4"))
    x2 = f.mean(2, dtype=np.float32)
    f, idx3 = graph_feature(x2)
    f = vn_linear_leaky_relu(f, *_layer(sd, "conv5"))
    x3 = f.mean(2, dtype=np.float32)
    x123 = np.concatenate([x1, x2, x3], axis=2)                                  # [B, N, 63, 3]
    x = vn_linear_leaky_relu(x123, *_layer(sd, "conv6"))                         # [B, N, feat_dim, 3]
    xm = np.broadcast_to(x.mean(1, keepdims=True, dtype=np.float32), x.shape)
    x = np.concatenate([x, xm], axis=2).mean(1, dtype=np.float32)                # [B, 2 feat_dim, 3]
    if inv:
        # vn_dgcnn.py:68-69: linear0 on the pooled vectors, mean over channels (VnInv's result :67 is discarded)
        y = x @ np.asarray(sd["linear0.weight"], np.float32).T + np.asarray(sd["linear0.bias"], np.float32)
        out = y.mean(1, dtype=np.float32)
    else:
        out = x.reshape(x.shape[0], -1)
    if return_intermediates:
        return out, dict(idx1=idx1, idx2=idx2, idx3=idx3, x1=x1, x2=x2, x3=x3)
    return out


def chamfer_sq(a, b):
    """Two-sided nearest-neighbour squared distances (chamfer_distance.py:148-149 via knn_points K=1):
    a [P, N, 3], b [P, M, 3] -> (d_ab [P, N], d_ba [P, M])."""
    d = ((a[:, :, None, :] - b[:, None, :, :]) ** 2).sum(-1, dtype=np.float32)
    return d.min(2), d.min(1)
