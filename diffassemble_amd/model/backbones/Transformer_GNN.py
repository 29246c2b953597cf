"""Counterpart of puzzle_diff/model/backbones/Transformer_GNN.py:5-46."""
import torch
from torch import nn

from ... import _lib
from ...graph_plan import build_plan
from .transformer_conv import TransformerConv


class Transformer_GNN(nn.Module):
    arch = "transformer"
    virt_nodes = 0

    def __init__(self, input_size, hidden_dim, heads, output_size, n_layers=4) -> None:
        super().__init__()
        self.module_list = nn.ModuleList(
            [TransformerConv(input_size, out_channels=hidden_dim // heads, heads=heads)]
            + [TransformerConv(hidden_dim, out_channels=hidden_dim // heads, heads=heads)
               for _ in range(n_layers - 2)]
            + [TransformerConv(hidden_dim, heads=heads, concat=True, out_channels=output_size // heads)])
        self.n_layers = n_layers

    @torch.no_grad()
    def forward(self, x, edge_index, move_to_cpu=False, batch=None, *args, precision="fp32"):
        """Standalone use: layer by layer through the kernel-level ABI (GELU fused in the
        attention epilogue).  Inside ``Eff_GAT`` the whole stack runs in one
        ``da_denoiser_forward`` call instead."""
        if batch is None:
            batch = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
        plan = build_plan(edge_index, batch, 0, detect_dense=False)
        attentions = []
        for i in range(self.n_layers):
            act = _lib.ACT_GELU if i < self.n_layers - 1 else _lib.ACT_NONE
            x, atts = self.module_list[i](x, edge_index, return_attention_weights=True,
                                          precision=precision, plan=plan, act=act)
            attentions.append(atts)
        if move_to_cpu:
            attentions = [(a[0].cpu().numpy(), a[1].cpu().numpy()) for a in attentions]
            x = x.cpu()
        return x, attentions
