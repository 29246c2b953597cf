"""Oracle restatement of the per-timestep denoiser (TEST INFRASTRUCTURE, oracle/__init__.py).

Pure torch fp32 on CPU.  ``sd`` is the denoiser's state dict with the reference's key
layout (``Eff_GAT.state_dict()`` keys, i.e. the checkpoint keys minus the leading
``model.``): ``time_emb.weight``, ``pos_mlp.{0,2}.*``, ``mlp.{0,2}.*``,
``gnn_backbone.module_list.{l}.lin_{key,query,value,skip}.*``,
``gnn_backbone.virt_node_embedding.weight``, ``final_mlp.{0,2}.*`` (2D) or
``mlp_t.{0,2}.*`` / ``mlp_r.{0,2}.*`` (3D).
"""
import torch
import torch.nn.functional as F

from . import so3
from .pyg_restatement import matrix_to_quaternion, transformer_conv

HEADS = 8


def _conv(sd, l, x, edge_index):
    p = f"gnn_backbone.module_list.{l}."
    return transformer_conv(
        x, edge_index,
        sd[p + "lin_query.weight"], sd[p + "lin_query.bias"],
        sd[p + "lin_key.weight"], sd[p + "lin_key.bias"],
        sd[p + "lin_value.weight"], sd[p + "lin_value.bias"],
        sd[p + "lin_skip.weight"], sd[p + "lin_skip.bias"], HEADS)


def n_layers_of(sd):
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("gnn_backbone.module_list."))


def transformer_gnn(sd, x, edge_index, collect=None):
    """puzzle_diff/model/backbones/Transformer_GNN.py:29-46 -- n_layers TransformerConv,
    erf-GELU after every layer but the last (:35); returns per-layer (edge_index, alpha)."""
    L = n_layers_of(sd)
    attentions = []
    for l in range(L):
        x, alpha = _conv(sd, l, x, edge_index)
        if collect is not None:
            collect.append(x)            # raw conv output (before the GELU)
        if l < L - 1:
            x = F.gelu(x)
        attentions.append((edge_index, alpha))
    return x, attentions


def exophormer_edges(edge_index, batch, virt_nodes):
    """Edge construction of puzzle_diff/model/backbones/exophormer_gnn.py:164-200,
    restated without the per-graph Python loop but with the SAME element-wise pairing
    quirk: ``src = cat[arange(N), virt_edges]`` and ``dst = cat[virt_edges, arange(N)]``
    are paired position by position after concatenation (so real node k is wired to
    ``virt_edges[k]`` only, virtual->virtual duplicates follow, and the last N entries
    are virtual->real).  ``batch`` there is the EXTENDED batch vector (:181-183): the
    V*G appended entries are ``arange(G).repeat(V)``, so graph i counts n_i + V nodes.
    Returns the extended edge_index [2, E + N + sum_i V*(n_i+V)]."""
    N = batch.numel()
    G = int(batch.max()) + 1
    V = virt_nodes
    counts = torch.bincount(batch, minlength=G) + V            # len(batch_ext[batch_ext == i])
    virt_edges = torch.cat([
        torch.arange(N + i * V, N + (i + 1) * V).repeat(int(counts[i])) for i in range(G)])
    ar = torch.arange(N)
    src = torch.cat([ar, virt_edges])
    dst = torch.cat([virt_edges, ar])
    return torch.hstack((edge_index, torch.stack((src, dst))))


def exophormer_gnn(sd, x, edge_index, batch, virt_nodes, collect=None):
    """exophormer_gnn.py:161-215 -- learned virtual-node rows appended per graph
    (:169-178; row order ``arange(V).repeat(G)``), extended edges (above), the same
    convs with NO activation in between (:202-203), virtual rows dropped (:209); only
    the last layer's attention is returned."""
    N = x.shape[0]
    if virt_nodes > 0:
        G = int(batch.max()) + 1
        emb = sd["gnn_backbone.virt_node_embedding.weight"]
        x = torch.cat((x, emb[torch.arange(virt_nodes).repeat(G)]))
        edge_index = exophormer_edges(edge_index, batch, virt_nodes)
    L = n_layers_of(sd)
    alpha = None
    for l in range(L):
        x, alpha = _conv(sd, l, x, edge_index)
        if collect is not None:
            collect.append(x)
    return x[:N], [(edge_index, alpha)]


def embed(sd, xy_pos, time, feats, leaky=False):
    """efficient_gat.py:131-135 (2D: mlp = Linear-GELU-Linear) /
    efficient_gat_3d.py:181-186,136-141 (3D: Linear-LeakyReLU(0.2)-Linear-LeakyReLU(0.2))."""
    time_feats = sd["time_emb.weight"][time]
    pos = F.linear(xy_pos, sd["pos_mlp.0.weight"], sd["pos_mlp.0.bias"])
    pos = F.linear(F.gelu(pos), sd["pos_mlp.2.weight"], sd["pos_mlp.2.bias"])
    comb = torch.cat([feats, pos, time_feats], -1)
    h = F.linear(comb, sd["mlp.0.weight"], sd["mlp.0.bias"])
    h = F.leaky_relu(h, 0.2) if leaky else F.gelu(h)
    out = F.linear(h, sd["mlp.2.weight"], sd["mlp.2.bias"])
    if leaky:
        out = F.leaky_relu(out, 0.2)
    return out


def _gnn(sd, combined, edge_index, batch, arch, virt_nodes, collect):
    if arch == "transformer":
        return transformer_gnn(sd, combined, edge_index, collect)
    if arch == "exophormer":
        return exophormer_gnn(sd, combined, edge_index, batch, virt_nodes, collect)
    raise ValueError(arch)


def eff_gat_forward_with_feats(sd, xy_pos, time, edge_index, patch_feats, batch,
                               arch="transformer", virt_nodes=4, collect=None):
    """Eff_GAT.forward_with_feats, efficient_gat.py:121-146.  Returns (out [N,c_out],
    attentions).  ``collect`` (a list) receives the intermediate activations."""
    combined = embed(sd, xy_pos, time, patch_feats, leaky=False)
    if collect is not None:
        collect.append(combined)
    feats, attentions = _gnn(sd, combined, edge_index, batch, arch, virt_nodes, collect)
    h = F.gelu(F.linear(feats + combined, sd["final_mlp.0.weight"], sd["final_mlp.0.bias"]))
    out = F.linear(h, sd["final_mlp.2.weight"], sd["final_mlp.2.bias"])
    return out, attentions


def eff_gat_3d_forward_with_feats(sd, xy_pos, time, edge_index, pcd_feats, batch,
                                  arch="transformer", virt_nodes=8, collect=None):
    """Eff_GAT_3d.forward_with_feats, efficient_gat_3d.py:173-220 (the default branch;
    ``use_vn_dgcnn_equiv_inv_mp`` is False in train_3d.py).  Output = hstack(quat wxyz
    [4], trans [3])."""
    combined = embed(sd, xy_pos, time, pcd_feats, leaky=True)
    if collect is not None:
        collect.append(combined)
    feats, attentions = _gnn(sd, combined, edge_index, batch, arch, virt_nodes, collect)
    z = feats + combined
    t_pred = F.linear(F.gelu(F.linear(z, sd["mlp_t.0.weight"], sd["mlp_t.0.bias"])),
                      sd["mlp_t.2.weight"], sd["mlp_t.2.bias"])
    r_pred = F.linear(F.gelu(F.linear(z, sd["mlp_r.0.weight"], sd["mlp_r.0.bias"])),
                      sd["mlp_r.2.weight"], sd["mlp_r.2.bias"])
    if collect is not None:
        collect.append(torch.cat([r_pred, t_pred], -1))
    q = matrix_to_quaternion(so3.skew_to_rmat(r_pred))
    q = F.normalize(q, p=2, dim=-1)
    return torch.hstack((q, t_pred)), attentions
