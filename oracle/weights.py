"""Deterministic weights, graphs and synthetic inputs (TEST INFRASTRUCTURE, oracle/__init__.py).

Everything is drawn from numpy's PCG64 ``default_rng(seed)`` streams, which are stable
across machines and numpy versions, so fixtures only need to store seeds + outputs and
the GPU box regenerates bit-identical weights/inputs without /root/reference.
"""
import math

import numpy as np
import torch


def _uniform(rng, shape, bound):
    return torch.from_numpy(rng.uniform(-bound, bound, size=shape).astype(np.float32))


def _linear(rng, sd, name, fan_in, fan_out, gain=1.0):
    b = 1.0 / math.sqrt(fan_in)
    sd[name + ".weight"] = _uniform(rng, (fan_out, fan_in), b) * gain
    sd[name + ".bias"] = _uniform(rng, (fan_out,), b) * gain


def make_denoiser_state(steps, c_in, c_out=None, D=1152, hidden=128, variant="2d",
                        arch="transformer", virt_nodes=4, n_layers=4, heads=8, seed=0,
                        qk_gain=1.0):
    """State dict with the reference's ``Eff_GAT`` / ``Eff_GAT_3d`` key layout (live
    params only; the dead ``linear1/linear2`` and the encoders are not on the path).
    Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (torch's default scale); embeddings N(0,1).
    ``qk_gain`` multiplies the query/key projections to make the softmax peaky (the
    default init gives near-uniform attention, which hides indexing bugs)."""
    rng = np.random.default_rng(seed)
    sd = {}
    sd["time_emb.weight"] = torch.from_numpy(rng.standard_normal((steps, 32)).astype(np.float32))
    _linear(rng, sd, "pos_mlp.0", c_in, 16)
    _linear(rng, sd, "pos_mlp.2", 16, 32)
    _linear(rng, sd, "mlp.0", D, hidden)
    _linear(rng, sd, "mlp.2", hidden, D)
    dims = [D] + [32 * heads] * (n_layers - 1)
    outs = [32 * heads] * (n_layers - 1) + [heads * (D // heads)]
    for l in range(n_layers):
        p = f"gnn_backbone.module_list.{l}."
        _linear(rng, sd, p + "lin_key", dims[l], outs[l], qk_gain)
        _linear(rng, sd, p + "lin_query", dims[l], outs[l], qk_gain)
        _linear(rng, sd, p + "lin_value", dims[l], outs[l])
        _linear(rng, sd, p + "lin_skip", dims[l], outs[l])
    if arch == "exophormer" and virt_nodes > 0:
        sd["gnn_backbone.virt_node_embedding.weight"] = torch.from_numpy(
            rng.standard_normal((virt_nodes, D)).astype(np.float32))
    if variant == "2d":
        _linear(rng, sd, "final_mlp.0", D, 32)
        _linear(rng, sd, "final_mlp.2", 32, c_out)
    else:
        _linear(rng, sd, "mlp_t.0", D, 256)
        _linear(rng, sd, "mlp_t.2", 256, 3)
        _linear(rng, sd, "mlp_r.0", D, 256)
        _linear(rng, sd, "mlp_r.2", 256, 3)
    return sd


# ------------------------------------------------------------------------- graphs
def dense_edge_index(n, self_loops=True):
    """pyg.utils.dense_to_sparse(ones(n, n)) order (puzzle_dataset.py:609-614): row-major
    nonzero, edge_index[0] = row, [1] = col.  ``self_loops=False`` is the K_n the
    non-rotation dataset builds (puzzle_dataset.py:47-64 via generate_random_expander
    with degree >= n-1... `num_nodes <= 10` branch and the roll construction never emit
    i == j)."""
    r = torch.arange(n).repeat_interleave(n)
    c = torch.arange(n).repeat(n)
    if not self_loops:
        m = r != c
        r, c = r[m], c[m]
    return torch.stack([r, c])


def random_regular_edge_index(n, degree, rng):
    """generate_random_regular_graph, puzzle_dataset.py:115-152, restated: a random
    permutation rolled 1..degree//2 (+ a perfect matching when degree is odd),
    symmetrised.  Returns int64 [2, n*degree] with row 0 = senders, row 1 = receivers."""
    if (n * degree) % 2 != 0:
        raise TypeError("nodes * degree must be even")
    nodes = rng.permutation(np.arange(n))
    reps = degree // 2
    ns = np.hstack([np.roll(nodes, i + 1) for i in range(reps)]) if reps else np.zeros(0, np.int64)
    ei = np.vstack((np.tile(nodes, reps), ns))
    if degree % 2 == 1:
        ei = np.hstack((ei, np.vstack((nodes[: n // 2], nodes[n // 2:]))))
    s = np.concatenate([ei[0], ei[1]])
    r = np.concatenate([ei[1], ei[0]])
    return torch.from_numpy(np.stack([s, r]).astype(np.int64))


def collate(edge_indices, sizes):
    """PyG Batch collation of per-graph edge_index + the ``batch`` vector."""
    off, eis, batch = 0, [], []
    for g, (ei, n) in enumerate(zip(edge_indices, sizes)):
        eis.append(ei + off)
        batch.append(torch.full((n,), g, dtype=torch.long))
        off += n
    return torch.cat(eis, 1), torch.cat(batch)


# ------------------------------------------------------------------------- inputs
def make_inputs(n_nodes, c_in, feat_dim, seed=0, rot=None):
    """Synthetic puzzle: piece features ~ N(0,1) [N, feat_dim], x_T ~ N(0,1) [N, c_in]
    (no datasets; encoder bypassed)."""
    rng = np.random.default_rng(seed + 1_000_003)
    feats = torch.from_numpy(rng.standard_normal((n_nodes, feat_dim)).astype(np.float32))
    x = torch.from_numpy(rng.standard_normal((n_nodes, c_in)).astype(np.float32))
    return x, feats


def randn(shape, seed):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32))


# ------------------------------------------------------------------------- piece encoder
ENCODER_PLANES = (32, 64, 64, 128)     # resnet_equivariant.py:77-81 (in_planes 32, then 64, 64, 128)


def encoder_conv_specs():
    """(state-dict prefix, in planes, out planes, input stabilizer size, kernel, stride) of every group
    convolution of the reference's ``ResNet18()`` (resnet_equivariant.py:14-38,70-91,113-114), in
    forward order; each is followed by a BatchNorm3d under ``bn_key``."""
    specs = [dict(conv="conv1", bn="bn1", cin=3, cout=32, stab=1, k=3, stride=1)]
    cin = 32
    for li, planes in enumerate(ENCODER_PLANES, start=1):
        for bi in range(2):
            stride = (1 if li == 1 else 2) if bi == 0 else 1
            p = f"layer{li}.{bi}."
            specs.append(dict(conv=p + "conv1", bn=p + "bn1", cin=cin, cout=planes, stab=4, k=3, stride=stride))
            specs.append(dict(conv=p + "conv2", bn=p + "bn2", cin=planes, cout=planes, stab=4, k=3, stride=1))
            if stride != 1 or cin != planes:
                specs.append(dict(conv=p + "shortcut.0", bn=p + "shortcut.1", cin=cin, cout=planes, stab=4, k=1,
                                  stride=stride))
            cin = planes
    return specs


def make_encoder_state(seed=0):
    """State dict with the key layout of the reference's equivariant ``ResNet18()``
    (``Eff_GAT.visual_backbone`` when ``model='resnet18equiv'``).  Group-conv weights follow the
    reference's init scale U(+-1/sqrt(in*k*k)) (splitgconv2d.py:52-59); the BatchNorm affine and RUNNING
    statistics are randomised so that the eval-mode normalisation is pinned (the defaults 0 / 1 would
    make it an identity)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for s in encoder_conv_specs():
        bound = 1.0 / math.sqrt(s["cin"] * s["k"] * s["k"])
        sd[s["conv"] + ".weight"] = _uniform(rng, (s["cout"], s["cin"], s["stab"], s["k"], s["k"]), bound)
        c = s["cout"]
        sd[s["bn"] + ".weight"] = torch.from_numpy(rng.uniform(0.6, 1.4, size=c).astype(np.float32))
        sd[s["bn"] + ".bias"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
        sd[s["bn"] + ".running_mean"] = torch.from_numpy((0.1 * rng.standard_normal(c)).astype(np.float32))
        sd[s["bn"] + ".running_var"] = torch.from_numpy(rng.uniform(0.5, 1.5, size=c).astype(np.float32))
        sd[s["bn"] + ".num_batches_tracked"] = torch.tensor(100, dtype=torch.int64)
    _linear(rng, sd, "linear1", 64 * 4 * 8 * 8, 544)
    _linear(rng, sd, "linear2", 128 * 4 * 4 * 4, 544)
    return sd


def make_patches(n, seed=0):
    """[n, 3, 32, 32] fp32 in [0, 1] (puzzle_dataset.py:290-299 hands the model 32x32 RGB crops)."""
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.uniform(0.0, 1.0, size=(n, 3, 32, 32)).astype(np.float32))


def vn_dgcnn_layer_specs(feat_dim=128):
    """(name, in_channels, out_channels, shared_direction) of the VNLinearLeakyReLU layers of the reference's
    ``VN_DGCNN`` (vnn/vn_dgcnn.py:13-31), then the dead ``VnInv`` stack (vn_layers.py:193-206)."""
    c = 64 // 3
    live = [("conv1", 2, c, False), ("conv2", c, c, False), ("conv3", 2 * c, c, False), ("conv4", c, c, False),
            ("conv5", 2 * c, c, False), ("conv6", 3 * c, feat_dim, True)]
    dead = [("VnInv.vn1", 2 * feat_dim, feat_dim, False), ("VnInv.vn2", feat_dim, feat_dim // 2, False)]
    return live, dead


def make_vn_dgcnn_state(feat_dim=128, seed=0):
    """State dict with the key layout of the reference's ``VN_DGCNN(feat_dim)``.  Linear maps at torch's default scale
    U(+-1/sqrt(fan_in)); BatchNorm affine AND running statistics randomised (the running mean sits near the typical
    vector norm so that the normalised norms take both signs)."""
    rng = np.random.default_rng(seed)
    sd = {}
    live, dead = vn_dgcnn_layer_specs(feat_dim)
    for name, cin, cout, shared in live + dead:
        b = 1.0 / math.sqrt(cin)
        sd[f"{name}.map_to_feat.weight"] = _uniform(rng, (cout, cin), b)
        sd[f"{name}.map_to_dir.weight"] = _uniform(rng, (1 if shared else cout, cin), b)
        sd[f"{name}.batchnorm.bn.weight"] = torch.from_numpy(rng.uniform(0.6, 1.4, size=cout).astype(np.float32))
        sd[f"{name}.batchnorm.bn.bias"] = torch.from_numpy((0.5 + 0.2 * rng.standard_normal(cout)).astype(np.float32))
        sd[f"{name}.batchnorm.bn.running_mean"] = torch.from_numpy(rng.uniform(0.05, 0.4, size=cout).astype(np.float32))
        sd[f"{name}.batchnorm.bn.running_var"] = torch.from_numpy(rng.uniform(0.02, 0.2, size=cout).astype(np.float32))
        sd[f"{name}.batchnorm.bn.num_batches_tracked"] = torch.tensor(100, dtype=torch.int64)
    sd["VnInv.vn_lin.weight"] = _uniform(rng, (3, feat_dim // 2), 1.0 / math.sqrt(feat_dim // 2))
    _linear(rng, sd, "linear0", 3, 2 * feat_dim)
    return sd


def make_point_clouds(P, N, seed=0):
    """[P, N, 3] fp32: each fragment is an anisotropic blob with its own offset and scale, like the centred,
    unit-scaled fragments the reference's dataset hands over (1000 points each in the Breaking Bad configuration)."""
    rng = np.random.default_rng(seed)
    pts = rng.standard_normal((P, N, 3)) * rng.uniform(0.05, 0.4, size=(P, 1, 3))
    pts += 0.3 * rng.standard_normal((P, 1, 3))
    return torch.from_numpy(pts.astype(np.float32))
