"""Counterpart of puzzle_diff/model/spatial_diffusion_3d_test_double_diffusion.py (the SE(3)
``GNN_Diffusion`` that train_3d.py:19 imports): R^3 Gaussian diffusion on translations, SO(3)
diffusion on rotations (quaternion wxyz | translation per fragment).  On the hot path:
``forward_with_feats`` (:369-382), ``p_sample_ddim`` (:595-663) and ``p_sample_loop``
(:688-731) run in the HIP library; losses / metrics / mesh export (pytorch3d kNN, chamfer)
are training-side and out of scope."""
from functools import partial
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

from .. import _lib
from ..engine import Schedule
from .. import metrics3d
from ._lightning_compat import LightningModule, MeanMetric
from .backbones import Eff_GAT_3d
from .spatial_diffusion import (ModelMeanType, ModelScheduler, cosine_beta_schedule,  # noqa: F401
                                cosine_discrete_beta_schedule, extract, linear_beta_schedule)


def extract_rot(a, t, x_shape):
    b, *_ = t.shape
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def _metric_was_updated(m):
    """True when a MeanMetric has seen at least one update since its last reset -- for torchmetrics' class
    (``update_called`` / ``_update_count``) and for the stand-in of this package (``count``) alike."""
    for attr in ("update_called", "_update_called"):
        v = getattr(m, attr, None)
        if isinstance(v, bool):
            return v
    v = getattr(m, "_update_count", None)
    if v is not None:
        return int(v) > 0
    return float(getattr(m, "count", 0)) > 0


class GNN_Diffusion(LightningModule):
    def __init__(self, steps=600, inference_ratio=1, sampling="DDPM", learning_rate=1e-4,
                 save_and_sample_every=1000, classifier_free_prob=0, classifier_free_w=0, noise_weight=0.0,
                 model_mean_type: ModelMeanType = ModelMeanType.EPSILON, input_channels=7, output_channels=7,
                 scheduler: ModelScheduler = ModelScheduler.LINEAR, visual_pretrained: bool = True,
                 freeze_backbone: bool = True, n_layers: int = 4, loss_type="all", backbone="vnn",
                 max_epochs=200, use_vn_dgcnn_equiv_inv_mp: bool = False, max_num_part: int = 20,
                 use_6dof: bool = False, architecture="transformer", *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        if use_6dof:
            raise NotImplementedError("use_6dof is off in train_3d.py; not on the accelerated path")
        self.loss_type = loss_type
        self.visual_pretrained = visual_pretrained
        self.free_backbone = freeze_backbone
        self.model_mean_type = model_mean_type
        self.learning_rate = learning_rate
        self.save_and_sample_every = save_and_sample_every
        self.classifier_free_prob = classifier_free_prob
        self.classifier_free_w = classifier_free_w
        self.noise_weight = noise_weight
        self.backbone = backbone
        self.max_epochs = max_epochs
        self.use_vn_dgcnn_equiv_inv_mp = use_vn_dgcnn_equiv_inv_mp
        self.max_num_part = max_num_part
        self.use_6dof = use_6dof
        self.save_eval_images = False
        self.return_attentions = False
        self.use_hip_graph = True
        if sampling == "DDIM":          # the reference binds a sampler only for DDIM (:269-275)
            self.inference_ratio = inference_ratio
            self.p_sample = partial(self.p_sample, sampling_func=self.p_sample_ddim)
            self.eta = 0
        betas = {ModelScheduler.LINEAR: linear_beta_schedule, ModelScheduler.COSINE: cosine_beta_schedule,
                 ModelScheduler.COSINE_DISCRETE: cosine_discrete_beta_schedule}[scheduler](timesteps=steps)
        self.register_buffer("betas", betas)
        alphas = 1.0 - self.betas
        self.register_buffer("alphas", alphas)
        self.register_buffer("alphas_cumprod", torch.cumprod(self.alphas, axis=0))
        self.register_buffer("alphas_cumprod_prev", F.pad(self.alphas_cumprod[:-1], (1, 0), value=1.0))
        self.register_buffer("sqrt_recip_alphas", torch.sqrt(1.0 / self.alphas))
        self.register_buffer("identity", torch.eye(3))
        self.register_buffer("sqrt_alphas_cumprod", torch.sqrt(self.alphas_cumprod))
        self.register_buffer("sqrt_recip_alphas_cumprod",
                             torch.from_numpy(np.sqrt((1.0 / self.alphas_cumprod).numpy())))
        self.register_buffer("sqrt_recipm1_alphas_cumprod",
                             torch.from_numpy(np.sqrt((1.0 / self.alphas_cumprod - 1).numpy())))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - self.alphas_cumprod))
        self.register_buffer("posterior_variance",
                             self.betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod))
        self.steps = steps
        self.input_channels = input_channels
        self.architecture = architecture
        self.n_layers = n_layers
        self.init_backbone()
        self.save_hyperparameters()

    def init_backbone(self):
        self.model = Eff_GAT_3d(steps=self.steps, input_channels=self.input_channels, n_layers=self.n_layers,
                                architecture=self.architecture, backbone=self.backbone,
                                freeze_backbone=self.free_backbone,
                                use_vn_dgcnn_equiv_inv_mp=self.use_vn_dgcnn_equiv_inv_mp)

    def _mean_type(self):
        return _lib.MEAN_START_X if self.model_mean_type == ModelMeanType.START_X else _lib.MEAN_EPSILON

    def _schedule(self):
        key = (self.betas.data_ptr(), str(self.betas.device), self.steps)
        if getattr(self, "_sched_key", None) != key:
            self._sched = Schedule({k: getattr(self, k) for k in Schedule.KEYS}, self.betas.device)
            self._sched_key = key
        return self._sched

    def forward(self, xy_pos, time, patch_rgb, edge_index, batch) -> Any:
        return self.model(xy_pos, time, patch_rgb, edge_index, batch)

    def forward_with_feats(self, xy_pos: Tensor, time: Tensor, edge_index: Tensor, pcd_feats: Tensor, batch,
                           return_attentions=False) -> Any:
        """...double_diffusion.py:369-382 (always returns the pair, like the reference)."""
        self.model.return_attentions = bool(return_attentions)
        return self.model.forward_with_feats(xy_pos, time, edge_index, pcd_feats, batch)

    def pcd_features(self, pcd):
        return self.model.pcd_features(pcd)

    def q_sample_tr(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        return (extract(self.sqrt_alphas_cumprod, t) * x_start
                + extract(self.sqrt_one_minus_alphas_cumprod, t) * noise)

    @torch.no_grad()
    def p_sample_ddim(self, x, t, t_index, edge_index, pcd_feats, batch):
        """...double_diffusion.py:595-663."""
        model_output, attentions = self.forward_with_feats(x, t, edge_index, pcd_feats, batch,
                                                           return_attentions=self.return_attentions)
        prev = self.model.engine(x.device).ddim_step(self._schedule(), x, model_output, t, self.inference_ratio,
                                                     self._mean_type())
        return prev, attentions

    @torch.no_grad()
    def p_sample_loop(self, shape, cond, edge_index, batch, pcd_feats=None):
        """...double_diffusion.py:688-731: identity rotations + randn * noise_weight translations."""
        device = self.device
        b = shape[0]
        tr = torch.randn((b, 3), device=device) * self.noise_weight
        quat = torch.zeros((b, 4), device=device)
        quat[:, 0] = 1.0                              # matrix_to_quaternion(eye(3))
        img = torch.cat([quat, tr], dim=1)
        if pcd_feats is None:
            pcd_feats = self.pcd_features(cond)
        its = list(reversed(range(0, self.steps, self.inference_ratio)))
        if not self.return_attentions:
            eng = self.model.engine(device)
            plan = self.model._plan_for(eng, edge_index, batch)
            self.model._feat_key = None
            traj, _ = eng.sample_loop(plan, self._schedule(), img, pcd_feats, ratio=self.inference_ratio,
                                      mean_type=self._mean_type(), keep_trajectory=True,
                                      use_graph=self.use_hip_graph)
            self.model._release_dense_plan_key()          # do not pin this Batch's edge list until the next one is planned
            return list(traj.clone().unbind(0)), [None] * len(its)
        imgs, attentions = [], []
        for i in its:
            img, atts = self.p_sample(img, torch.full((b,), i, device=device, dtype=torch.long), i,
                                      edge_index=edge_index, pcd_feats=pcd_feats, batch=batch)
            attentions.append(atts)
            imgs.append(img)
        return imgs, attentions

    @torch.no_grad()
    def p_sample(self, x, t, t_index, edge_index, sampling_func, pcd_feats, batch):
        return sampling_func(x, t, t_index, edge_index, pcd_feats, batch)

    def p_losses(self, *args, **kwargs):
        raise NotImplementedError("3D training losses (pytorch3d kNN / chamfer) are out of scope: SURVEY.md 2 #2")

    # ------------------------------------------------------------------ Lightning hooks (inference callers)
    def initialize_torchmetrics(self, categories):
        """...double_diffusion.py:347-364: four MeanMetrics per category + their averages."""
        import torch.nn as nn
        metrics = {}
        for i in categories:
            for k in ("rmse_t", "rmse_r", "gd_r", "part_acc"):
                metrics[f"{k}_{i}"] = MeanMetric()
        self.metrics = nn.ModuleDict(metrics)
        self.avg_metrics = nn.ModuleDict({f"{k}_AVG": MeanMetric() for k in ("rmse_t", "rmse_r", "gd_r", "part_acc")})

    @torch.no_grad()
    def _eval_step(self, batch, batch_idx):
        """validation_step / test_step (:895-960, :1036-1080): one sampling loop for the whole Batch, then per object
        the four pose metrics against the ground-truth poses in ``batch.x`` (wandb / mesh dumps omitted).  Returns the
        final poses [P, 7]."""
        sampled_pos, _ = self.p_sample_loop(batch.x.shape, batch.pcds, batch.edge_index, batch=batch.batch,
                                            pcd_feats=getattr(batch, "pcd_feats", None))
        final_pos = sampled_pos[-1]
        G = int(batch.batch.max()) + 1
        for i in range(G):
            idx = torch.where(batch.batch == i)[0]
            gt_pos, pred_pos = batch.x[idx], final_pos[idx]
            pred_r, pred_t, gt_r, gt_t = pred_pos[:, :4], pred_pos[:, 4:7], gt_pos[:, :4], gt_pos[:, 4:]
            vals = {"rmse_t": metrics3d.trans_metrics(pred_t, gt_t),
                    "rmse_r": metrics3d.rot_metrics(pred_r, gt_r, "rmse"),
                    "gd_r": metrics3d.rot_metrics(pred_r, gt_r, "geodesic")}
            if getattr(batch, "pcds", None) is not None:
                vals["part_acc"] = metrics3d.calc_part_acc(batch.pcds[idx], pred_t, gt_t, pred_r, gt_r)
            if hasattr(self, "metrics"):
                cat = batch.category[i]
                for k, v in vals.items():
                    if f"{k}_{cat}" in self.metrics:
                        self.metrics[f"{k}_{cat}"].update(v)
        # (the step only UPDATES the per-category metrics; they are computed, logged and reset once per epoch in
        # validation_epoch_end -- computing them here would log a batch-weighted mean of running means and, under DDP,
        # trigger one metric sync per category per step)
        return final_pos

    def validation_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx)

    def test_step(self, batch, batch_idx):
        return self._eval_step(batch, batch_idx)

    @torch.no_grad()
    def prediction_step(self, batch, batch_idx):
        return self.p_sample_loop(batch.x.shape, batch.pcds, batch.edge_index, batch=batch.batch,
                                  pcd_feats=getattr(batch, "pcd_feats", None))

    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        return self.prediction_step(batch, batch_idx)

    def validation_epoch_end(self, outputs) -> None:
        """:1015-1031: every per-category metric feeds the matching *_AVG metric; the per-category metrics are reset
        (Lightning does that for logged Metric objects at epoch end)."""
        if not hasattr(self, "metrics"):
            return
        seen = {}
        for k in ("rmse_t", "rmse_r", "gd_r", "part_acc"):
            for name, m in self.metrics.items():
                if name.startswith(k + "_") and _metric_was_updated(m):        # categories without a sample stay out of the averages
                    seen[name] = m.compute()
                    self.avg_metrics[f"{k}_AVG"].update(float(seen[name]))
        self.log_dict({**seen, **{k: m.compute() for k, m in self.avg_metrics.items() if _metric_was_updated(m)}})
        for m in list(self.metrics.values()) + list(self.avg_metrics.values()):
            m.reset()

    def test_epoch_end(self, outputs) -> None:
        return self.validation_epoch_end(outputs)

    def configure_optimizers(self):
        from transformers.optimization import Adafactor
        return Adafactor(self.parameters())
