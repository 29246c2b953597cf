"""Counterpart of puzzle_diff/model/backbones/efficient_gat.py (``Eff_GAT``): same
constructor arguments, attribute and state-dict names; ``forward_with_feats`` runs as ONE
call into the HIP library (embedding -> mlp -> 4x graph attention -> residual -> head)."""
import torch
import torch.nn as nn
from torch import Tensor

from ._denoiser_base import DenoiserBase, default_precision
from .exophormer_gnn import Exophormer_GNN
from .Transformer_GNN import Transformer_GNN


class Eff_GAT(DenoiserBase):
    variant = "2d"

    def __init__(self, steps, input_channels=2, output_channels=2, n_layers=4, visual_pretrained=True,
                 freeze_backbone=False, model="efficientnet_b0", architecture="transformer", virt_nodes=4,
                 all_equivariant=False, return_attentions=True) -> None:
        super().__init__()
        # piece encoder (efficient_gat.py:37-42): NOT on the per-timestep path;
        # built only when timm is importable, otherwise callers must pass patch_feats.
        self.visual_backbone = None
        if model == "resnet18equiv":
            from .resnet_equivariant import ResNet18
            self.visual_backbone = ResNet18()          # P4-equivariant ResNet-18, HIP kernels (eval mode)
        else:
            try:
                import timm
                self.visual_backbone = timm.create_model(model, pretrained=visual_pretrained, features_only=True)
            except Exception:  # noqa: BLE001  (timm absent / no weights / no network)
                self.visual_backbone = None
        self.all_equivariant = all_equivariant
        self.model = model
        self.combined_features_dim = {"resnet18": 3136, "resnet50": 12352, "efficientnet_b0": 1088 + 32 + 32,
                                      "resnet18equiv": 1088 + 32 + 32}[model]
        self.input_channels, self.output_channels = input_channels, output_channels
        self.freeze_backbone = freeze_backbone
        self.return_attentions = return_attentions
        D = self.combined_features_dim
        if architecture == "transformer":
            self.gnn_backbone = Transformer_GNN(D, n_layers=n_layers, hidden_dim=32 * 8, heads=8, output_size=D)
        elif architecture == "exophormer":
            self.gnn_backbone = Exophormer_GNN(D, n_layers=n_layers, hidden_dim=32 * 8, heads=8, output_size=D,
                                               virt_nodes=virt_nodes)
        else:
            raise NotImplementedError(f"architecture={architecture!r}: the GCN ablation is out of scope")
        self.time_emb = nn.Embedding(steps, 32)
        self.pos_mlp = nn.Sequential(nn.Linear(input_channels, 16), nn.GELU(), nn.Linear(16, 32))
        self.final_mlp = nn.Sequential(nn.Linear(D, 32), nn.GELU(), nn.Linear(32, output_channels))
        self.mlp = nn.Sequential(nn.Linear(D, 128), nn.GELU(), nn.Linear(128, D))
        # dead parameters of the reference (efficient_gat.py:105-107), kept for checkpoint keys
        self.linear1 = nn.Linear(8192, 544)
        self.linear2 = nn.Linear(4096, 544)
        self.register_buffer("mean", torch.tensor([0.4850, 0.4560, 0.4060])[None, :, None, None])
        self.register_buffer("std", torch.tensor([0.2290, 0.2240, 0.2250])[None, :, None, None])

    def forward(self, xy_pos, time, patch_rgb, edge_index, batch):
        patch_feats = self.visual_features(patch_rgb)
        return self.forward_with_feats(xy_pos, time, patch_rgb, edge_index, patch_feats=patch_feats, batch=batch)

    def forward_with_feats(self, xy_pos: Tensor, time: Tensor, patch_rgb: Tensor, edge_index: Tensor,
                           patch_feats: Tensor, batch):
        """efficient_gat.py:121-146 -> (out [N, c_out] fp32, attentions)."""
        if self._wants_grad():
            return self._run_train(xy_pos, time, edge_index, patch_feats, batch)
        return self._run(xy_pos, time, edge_index, patch_feats, batch, self.return_attentions)

    def visual_features(self, patch_rgb):
        """efficient_gat.py:149-189: normalise, piece encoder, concat feature maps 2 and 3 -> [N, 1088].
        ``model='resnet18equiv'`` (the reference's in-tree P4-equivariant ResNet-18) runs in the HIP library: eval mode
        through da_encoder_forward (BatchNorm folded), train() mode -- batch statistics, trainable unless
        ``freeze_backbone`` -- through the da_enc_* training primitives with the backward joined to autograd
        (SURVEY 8f-2); the timm encoders are third-party and, when timm is importable, run as plain torch outside the
        accelerated path."""
        if self.visual_backbone is None:
            raise NotImplementedError(
                "no piece encoder available (timm / equivariant ResNet are outside the hot path): "
                "pass precomputed patch_feats [N, 1088]")
        if self.model == "resnet18equiv":
            self.visual_backbone.precision = getattr(self, "precision", None) or default_precision()
            if self.all_equivariant:
                # efficient_gat.py:156-158: patch_rgb [N, 4, 3, 32, 32] holds the four quarter-turn views of a piece;
                # the encoder runs on each and the outputs are averaged -- one batched call over 4 N crops here
                n = patch_rgb.shape[0]
                views = patch_rgb.transpose(0, 1).reshape(4 * n, *patch_rgb.shape[2:])
                return self.visual_backbone.patch_features(views).float().view(4, n, -1).mean(0)
            if self.freeze_backbone:                                   # efficient_gat.py:152-154
                with torch.no_grad():
                    feats = self.visual_backbone.patch_features(patch_rgb)
            else:
                feats = self.visual_backbone.patch_features(patch_rgb)    # normalise + encoder + cat, all in HIP
            return feats.float()
        patch_rgb = (patch_rgb - self.mean) / self.std
        if self.freeze_backbone:
            with torch.no_grad():
                feats = self.visual_backbone.forward(patch_rgb)
        else:
            feats = self.visual_backbone.forward(patch_rgb)
        n = patch_rgb.shape[0]
        return torch.cat([feats[2].reshape(n, -1), feats[3].reshape(n, -1)], -1)
