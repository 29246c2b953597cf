"""Parameter holder + standalone operator for one PyG ``TransformerConv`` as the reference
instantiates it (Transformer_GNN.py:10-24: concat=True, beta=False, root_weight=True,
bias=True, no edge features).  Parameter names match PyG's (``lin_key``, ``lin_query``,
``lin_value``, ``lin_skip``) so reference checkpoints load unchanged."""
import torch
import torch.nn as nn

from ... import _lib
from ... import engine as E
from ...graph_plan import build_plan


class TransformerConv(nn.Module):
    def __init__(self, in_channels, out_channels, heads=1, concat=True, **kwargs):
        super().__init__()
        if not concat or kwargs.get("beta") or kwargs.get("edge_dim") or kwargs.get("dropout"):
            raise NotImplementedError("only the configuration the reference uses is supported")
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin_key = nn.Linear(in_channels, heads * out_channels)
        self.lin_query = nn.Linear(in_channels, heads * out_channels)
        self.lin_value = nn.Linear(in_channels, heads * out_channels)
        self.lin_skip = nn.Linear(in_channels, heads * out_channels)

    def fused_weight(self):
        w = torch.cat([self.lin_query.weight, self.lin_key.weight, self.lin_value.weight, self.lin_skip.weight])
        b = torch.cat([self.lin_query.bias, self.lin_key.bias, self.lin_value.bias, self.lin_skip.bias])
        return w, b

    @torch.no_grad()
    def forward(self, x, edge_index, return_attention_weights=None, precision="fp32", plan=None,
                act=_lib.ACT_NONE):
        """Standalone layer through the kernel-level C ABI (da_linear + da_attn_csr); ``act`` is
        applied in the attention epilogue (the GELU of Transformer_GNN.py:35)."""
        if plan is None:
            batch = torch.zeros(x.shape[0], dtype=torch.long, device=x.device)
            plan = build_plan(edge_index, batch, 0, detect_dense=False)
        w, b = self.fused_weight()
        qkvs = E.linear(x, w, b, _lib.ACT_NONE, None, precision)
        res = E.attn_csr(plan, qkvs, self.heads, self.out_channels, None, act,
                         bool(return_attention_weights), precision)
        if return_attention_weights:
            return res[0].float(), (edge_index, res[1])
        return res.float()
