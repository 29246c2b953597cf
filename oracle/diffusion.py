"""Oracle restatement of the diffusion glue (TEST INFRASTRUCTURE, oracle/__init__.py).

2D: puzzle_diff/model/spatial_diffusion.py; 3D:
puzzle_diff/model/spatial_diffusion_3d_test_double_diffusion.py.  fp32 torch on CPU.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import denoiser, so3
from .pyg_restatement import matrix_to_quaternion, quaternion_to_matrix


# ------------------------------------------------------------------ schedules (a-1)
def linear_beta_schedule(timesteps):
    """spatial_diffusion.py:154-157."""
    return torch.linspace(0.0001, 0.02, timesteps)


def cosine_beta_schedule(timesteps, s=0.08):
    """spatial_diffusion.py:142-151."""
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.clip(betas, 0.0001, 0.9999)


def make_schedule(steps, scheduler="linear"):
    """The ten registered buffers of GNN_Diffusion.__init__, spatial_diffusion.py:282-321
    (same op order so the fp32 values are bit-identical)."""
    betas = {"linear": linear_beta_schedule, "cosine": cosine_beta_schedule}[scheduler](steps)
    alphas = 1.0 - betas
    alphas_cumprod = torch.cumprod(alphas, axis=0)
    alphas_cumprod_prev = F.pad(alphas_cumprod[:-1], (1, 0), value=1.0)
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": alphas_cumprod,
        "alphas_cumprod_prev": alphas_cumprod_prev,
        "sqrt_recip_alphas": torch.sqrt(1.0 / alphas),
        "sqrt_alphas_cumprod": torch.sqrt(alphas_cumprod),
        # the reference uses np.sqrt on these two (:300-305); numpy's and torch's fp32
        # sqrt differ by 1 ulp on a few entries, so follow it exactly
        "sqrt_recip_alphas_cumprod": torch.from_numpy(np.sqrt((1.0 / alphas_cumprod).numpy())),
        "sqrt_recipm1_alphas_cumprod": torch.from_numpy(np.sqrt((1.0 / alphas_cumprod - 1).numpy())),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - alphas_cumprod),
        "posterior_variance": betas * (1.0 - alphas_cumprod_prev) / (1.0 - alphas_cumprod),
    }


def extract(a, t):
    """spatial_diffusion.py:173-176: a.gather(-1, t)[:, None]."""
    return a.gather(-1, t)[:, None]


def q_sample(sch, x_start, t, noise):
    """spatial_diffusion.py:421-430."""
    return (extract(sch["sqrt_alphas_cumprod"], t) * x_start
            + extract(sch["sqrt_one_minus_alphas_cumprod"], t) * noise)


# ------------------------------------------------------------------ 2D sampling (a-9..a-11)
def _get_variance(sch, t, prev_t):
    """spatial_diffusion.py:528-546."""
    ap = extract(sch["alphas_cumprod"], t)
    ap_prev = extract(sch["alphas_cumprod"], prev_t) if bool((prev_t >= 0).all()) else ap * 0 + 1
    return ((1 - ap_prev) / (1 - ap)) * (1 - ap / ap_prev)


def _predict_eps_from_xstart(sch, x_t, t, x0):
    """spatial_diffusion.py:629-632."""
    return (extract(sch["sqrt_recip_alphas_cumprod"], t) * x_t - x0) / extract(
        sch["sqrt_recipm1_alphas_cumprod"], t)


def ddim_update(sch, x, t, model_output, inference_ratio, mean_type="START_X", eta=0.0, noise=None):
    """The algebra of p_sample_ddim after the model call, spatial_diffusion.py:555-566,
    603-627."""
    prev_t = t - inference_ratio
    ap = extract(sch["alphas_cumprod"], t)
    ap_prev = extract(sch["alphas_cumprod"], prev_t) if bool((prev_t >= 0).all()) else ap * 0 + 1
    beta = 1 - ap
    x0 = model_output if mean_type == "START_X" else (x - beta ** 0.5 * model_output) / ap ** 0.5
    eps = _predict_eps_from_xstart(sch, x, t, x0)
    std_eta = eta * _get_variance(sch, t, prev_t) ** 0.5
    prev = ap_prev ** 0.5 * x0 + (1 - ap_prev - std_eta ** 2) ** 0.5 * eps
    if eta > 0:
        prev = prev + std_eta * noise
    return prev


def ddpm_update(sch, x, t, t_index, model_output, noise=None):
    """p_sample_ddpm, spatial_diffusion.py:485-510."""
    mean = extract(sch["sqrt_recip_alphas"], t) * (
        x - extract(sch["betas"], t) * model_output / extract(sch["sqrt_one_minus_alphas_cumprod"], t))
    if t_index == 0:
        return mean
    return mean + torch.sqrt(extract(sch["posterior_variance"], t)) * noise


def p_sample_loop(sd, sch, x_init, edge_index, patch_feats, batch, steps, inference_ratio=1,
                  mean_type="START_X", arch="transformer", virt_nodes=4, sampling="DDIM",
                  noises=None, max_iters=None, classifier_free_w=0.0, classifier_free_prob=0.0):
    """p_sample_loop, spatial_diffusion.py:635-676 (x_init = randn*noise_weight is drawn by
    the caller so tests share it).  ``sampling='DDPM'`` runs the direct p_sample_ddpm
    algebra per step (the reference's loop itself raises for DDPM: p_sample_ddpm returns a
    bare tensor, :504-510, but :663 unpacks two values).  Returns (imgs, last attentions)."""
    x = x_init
    imgs, att = [], None
    b = x.shape[0]
    its = list(reversed(range(0, steps, inference_ratio)))
    if max_iters is not None:
        its = its[:max_iters]
    for k, i in enumerate(its):
        t = torch.full((b,), i, dtype=torch.long)
        out, att = denoiser.eff_gat_forward_with_feats(
            sd, x, t, edge_index, patch_feats, batch, arch, virt_nodes)
        if classifier_free_prob > 0.0:      # spatial_diffusion.py:568-589
            unc, _ = denoiser.eff_gat_forward_with_feats(
                sd, x, t, edge_index, torch.zeros_like(patch_feats), batch, arch, virt_nodes)
            out = (1 + classifier_free_w) * out - classifier_free_w * unc
        if sampling == "DDIM":
            x = ddim_update(sch, x, t, out, inference_ratio, mean_type)
        else:
            x = ddpm_update(sch, x, t, i, out, None if noises is None else noises[k])
        imgs.append(x)
    return imgs, att


def p_losses(sd, sch, x_start, t, noise, edge_index, patch_feats, batch, mean_type="EPSILON",
             arch="transformer", virt_nodes=4, loss_type="huber"):
    """p_losses, spatial_diffusion.py:432-483 with the encoder bypassed (patch_feats given)."""
    x_noisy = q_sample(sch, x_start, t, noise)
    pred, _ = denoiser.eff_gat_forward_with_feats(
        sd, x_noisy, t, edge_index, patch_feats, batch, arch, virt_nodes)
    target = x_start if mean_type == "START_X" else noise
    fn = {"l1": F.l1_loss, "l2": F.mse_loss, "huber": F.smooth_l1_loss}[loss_type]
    return fn(target, pred)


# ------------------------------------------------------------------ 3D sampling (a-14)
def ddim_update_3d(sch, x, t, model_output, inference_ratio, mean_type="START_X"):
    """spatial_diffusion_3d_test_double_diffusion.py:595-685: translation as in 2D,
    rotation through so3_scale / log_rmat.  x, model_output = [P, 7] (quat wxyz, trans)."""
    prev_t = t - inference_ratio
    ap = extract(sch["alphas_cumprod"], t)
    ap_prev = extract(sch["alphas_cumprod"], prev_t) if bool((prev_t >= 0).all()) else ap * 0 + 1
    beta = 1 - ap
    x0 = model_output if mean_type == "START_X" else (x - beta ** 0.5 * model_output) / ap ** 0.5
    x0_tr, x0_r, x_tr, x_q = x0[:, 4:], x0[:, :4], x[:, 4:], x[:, :4]
    eps_tr = _predict_eps_from_xstart(sch, x_tr, t, x0_tr)
    sr = sch["sqrt_recip_alphas_cumprod"].gather(-1, t)
    srm1 = sch["sqrt_recipm1_alphas_cumprod"].gather(-1, t)
    x_t_term = so3.so3_scale(quaternion_to_matrix(x_q), sr / srm1)          # :670-677
    x0_term = so3.so3_scale(quaternion_to_matrix(x0_r), 1 / srm1)           # :679-682
    eps_rot = matrix_to_quaternion(x_t_term @ x0_term.transpose(-1, -2))    # :637-639,685
    dir_tr = (1 - ap_prev) ** 0.5 * eps_tr
    dir_rot = so3.so3_scale(quaternion_to_matrix(eps_rot), ((1 - ap_prev) ** 0.5).view(-1))
    prev_tr = ap_prev ** 0.5 * x0_tr + dir_tr
    prev_r = matrix_to_quaternion(
        so3.so3_scale(quaternion_to_matrix(x0_r), (ap_prev ** 0.5).view(-1)) @ dir_rot)
    return torch.cat([prev_r, prev_tr], 1)


def p_sample_loop_3d(sd, sch, x_init, edge_index, pcd_feats, batch, steps, inference_ratio=1,
                     mean_type="START_X", arch="transformer", virt_nodes=8, max_iters=None):
    """p_sample_loop, …double_diffusion.py:688-731.  x_init = [identity quat | randn*nw]."""
    x = x_init
    imgs, att = [], None
    b = x.shape[0]
    its = list(reversed(range(0, steps, inference_ratio)))
    if max_iters is not None:
        its = its[:max_iters]
    for i in its:
        t = torch.full((b,), i, dtype=torch.long)
        out, att = denoiser.eff_gat_3d_forward_with_feats(
            sd, x, t, edge_index, pcd_feats, batch, arch, virt_nodes)
        x = ddim_update_3d(sch, x, t, out, inference_ratio, mean_type)
        imgs.append(x)
    return imgs, att
