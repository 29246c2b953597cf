"""Counterpart of puzzle_diff/model/backbones/efficient_gat_3d.py (``Eff_GAT_3d``)."""
import torch
import torch.nn as nn
from torch import Tensor

from ._denoiser_base import DenoiserBase
from .exophormer_gnn import Exophormer_GNN
from .Transformer_GNN import Transformer_GNN

_FEAT_DIM = {"pointnet_inv": 1024, "pointnet": 128, "pointnet_plus": 256, "vn_dgcnn": 768,
             "vn_dgcnn_inv": 256, "vnn": 2104}


class Eff_GAT_3d(DenoiserBase):
    variant = "3d"

    def __init__(self, steps, input_channels=7, t_channels=3, r_channels=3, n_layers=4,
                 architecture="transformer", virt_nodes=8, backbone="pointnet", freeze_backbone=False,
                 use_vn_dgcnn_equiv_inv_mp=False, return_attentions=True) -> None:
        super().__init__()
        if use_vn_dgcnn_equiv_inv_mp:
            raise NotImplementedError("use_vn_dgcnn_equiv_inv_mp is False in train_3d.py; not on the path")
        if backbone not in _FEAT_DIM:
            raise Exception(f"Backbone not implemented {backbone}")
        self.use_vn_dgcnn_equiv_inv_mp = False
        # point-cloud encoders (efficient_gat_3d.py:73-97): once per sampling loop, not per step.  The
        # vector-neuron DGCNN the 3D configuration trains with (train_3d.py: backbone="vn_dgcnn") runs in the HIP library
        # (da_pcd_encoder_forward, eval mode); the PointNet variants are not built: pass pcd_feats.
        self.pcd_backbone = None
        if backbone in ("vn_dgcnn", "vn_dgcnn_inv"):
            from .vnn.vn_dgcnn import VN_DGCNN
            self.pcd_backbone = VN_DGCNN(feat_dim=128, inv=(backbone == "vn_dgcnn_inv"))
        feat_dim = _FEAT_DIM[backbone]
        self.combined_features_dim = feat_dim + 32 + 32
        self.gnn_feat_dim = self.combined_features_dim
        self.input_channels = input_channels
        self.freeze_backbone = freeze_backbone
        self.return_attentions = return_attentions
        D = self.gnn_feat_dim
        if D % 64 != 0:
            raise NotImplementedError(f"feature width {D} must be a multiple of 64 for 8 heads x C%8==0")
        if architecture == "transformer":
            self.gnn_backbone = Transformer_GNN(D, n_layers=n_layers, hidden_dim=32 * 8, heads=8, output_size=D)
        elif architecture == "exophormer":
            self.gnn_backbone = Exophormer_GNN(D, n_layers=n_layers, hidden_dim=32 * 8, heads=8, output_size=D,
                                               virt_nodes=virt_nodes)
        else:
            raise NotImplementedError(f"architecture={architecture!r}: the GCN ablation is out of scope")
        self.time_emb = nn.Embedding(steps, 32)
        self.pos_mlp = nn.Sequential(nn.Linear(input_channels, 16), nn.GELU(), nn.Linear(16, 32))
        self.mlp = nn.Sequential(nn.Linear(D, 256), nn.LeakyReLU(0.2), nn.Linear(256, D), nn.LeakyReLU(0.2))
        self.mlp_t = nn.Sequential(nn.Linear(D, 256), nn.GELU(), nn.Linear(256, t_channels))
        self.mlp_r = nn.Sequential(nn.Linear(D, 256), nn.GELU(), nn.Linear(256, r_channels))

    def forward(self, xy_pos, time, pcd, edge_index, batch):
        return self.forward_with_feats(xy_pos, time, edge_index, pcd_feats=self.pcd_features(pcd), batch=batch)

    def forward_with_feats(self, xy_pos: Tensor, time: Tensor, edge_index: Tensor, pcd_feats: Tensor, batch):
        """efficient_gat_3d.py:173-220 -> (hstack(unit quaternion wxyz, translation) [P, 7], attentions)."""
        return self._run(xy_pos, time, edge_index, pcd_feats, batch, self.return_attentions)

    def pcd_features(self, pcd):
        """efficient_gat_3d.py:230-236: pcd [P, N, 3] -> pcd_feats [P, feat_dim]."""
        if self.pcd_backbone is None:
            raise NotImplementedError(
                "only the VN-DGCNN point-cloud encoders are built (backbone='vn_dgcnn' / 'vn_dgcnn_inv'): pass pcd_feats")
        return self.pcd_backbone(pcd)
