"""Counterpart of puzzle_diff/model/backbones/exophormer_gnn.py:132-215 (``Exophormer_GNN``;
the dead ``ExphormerFullLayer`` of :22-129 references an undefined class and is not kept)."""
import torch
from torch import nn

from ... import _lib
from ...graph_plan import build_plan
from .transformer_conv import TransformerConv


class Exophormer_GNN(nn.Module):
    arch = "exophormer"

    def __init__(self, input_size, hidden_dim, heads, output_size, n_layers=4, virt_nodes=4) -> None:
        super().__init__()
        self.module_list = nn.ModuleList(
            [TransformerConv(input_size, out_channels=hidden_dim // heads, heads=heads)]
            + [TransformerConv(hidden_dim, out_channels=hidden_dim // heads, heads=heads)
               for _ in range(n_layers - 2)]
            + [TransformerConv(hidden_dim, out_channels=output_size // heads, heads=heads, concat=True)])
        self.virt_nodes = virt_nodes
        if self.virt_nodes > 0:
            self.virt_node_embedding = nn.Embedding(virt_nodes, input_size)
        self.n_layers = n_layers

    @torch.no_grad()
    def forward(self, x, edge_index, move_to_cpu=False, batch=None, mean_value=False, *args, precision="fp32"):
        """Standalone use through the kernel-level ABI: virtual rows + the reference's extended
        edges (graph_plan.exophormer_edge_index), no activation between layers (:202-203)."""
        if mean_value:
            raise NotImplementedError("mean_value=True is never used by the reference drivers")
        n_real = x.shape[0]
        plan = build_plan(edge_index, batch, self.virt_nodes, detect_dense=False)
        if self.virt_nodes > 0:
            idx = torch.arange(self.virt_nodes, device=x.device).repeat(plan.n_graphs)
            x = torch.cat((x, self.virt_node_embedding.weight[idx].to(x.dtype)))
        atts = None
        for i in range(self.n_layers):
            last = i == self.n_layers - 1
            r = self.module_list[i](x, plan.edge_index, return_attention_weights=last, precision=precision,
                                    plan=plan, act=_lib.ACT_NONE)
            x, atts = r if last else (r, None)
        x = x[:n_real]
        attentions = [atts]
        if move_to_cpu:
            attentions = [(a[0].cpu().numpy(), a[1].cpu().numpy()) for a in attentions]
            x = x.cpu()
        return x, attentions
