"""Evaluation metrics of the 3D ``validation_step`` / ``test_step`` (host glue: torch ops on whatever device the
sampled poses live on; runs once per Batch AFTER the sampling loop, not on the per-timestep path).

Counterparts of puzzle_diff/model/utils_3d.py ``trans_metrics`` (:362-383), ``rot_metrics`` (:415-450, 'rmse' and
'geodesic'), ``geodesic_distance`` (:916-945) and ``calc_part_acc`` (:1089-1129).  The reference routes the last one
through pytorch3d's CUDA ``knn_points`` (model/chamfer_distance.py:148-149), which does not exist on ROCm: on the device
the K = 1 search both ways is ``da_nearest_sq`` (diffassemble_amd/csrc/da_pcd_encoder.hip; b tiles staged in LDS, the
[P, N, N] distance tensor is never formed).  Host tensors take the cdist route below.  Poses are (unit quaternion wxyz |
translation) rows, [P, 7]; fragments [P, N, 3]."""
import math

import torch


def _rotate(q, v):
    """points v [P, N, 3] by unit quaternions q [P, 4] (real part first)."""
    w, u = q[:, None, :1], q[:, None, 1:].expand(-1, v.shape[1], -1)
    t = 2.0 * torch.cross(u, v, dim=-1)
    return v + w * t + torch.cross(u, t, dim=-1)


def _rmat(q):
    r, i, j, k = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    return torch.stack((1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r),
                        s * (i * j + k * r), 1 - s * (i * i + k * k), s * (j * k - i * r),
                        s * (i * k - j * r), s * (j * k + i * r), 1 - s * (i * i + j * j)), -1).reshape(q.shape[:-1] + (3, 3))


def trans_metrics(t1, t2):
    """RMSE over xyz per part, mean over parts (utils_3d.py:362-383, metric='rmse')."""
    return ((t1 - t2).pow(2).mean(-1) ** 0.5).mean()


def _euler_zyx_deg(q):
    q0, q1, q2, q3 = q.unbind(-1)
    x = torch.atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2))
    y = torch.asin(torch.clamp(2 * (q0 * q2 - q1 * q3), -1, 1))
    z = torch.atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))
    return torch.stack((x, y, z), -1) * (180.0 / math.pi)


def rot_metrics(q1, q2, metric="rmse"):
    """utils_3d.py:415-450: 'rmse' of the zyx Euler angles in degrees (differences wrap at 360) or 'geodesic'."""
    if metric == "geodesic":
        tr = torch.einsum("bij,bij->b", _rmat(q1), _rmat(q2))
        return torch.acos(torch.clamp(0.5 * (tr - 1), -1 + 1e-6, 1 - 1e-6)).mean()
    d = (_euler_zyx_deg(q1) - _euler_zyx_deg(q2)).abs()
    d = torch.minimum(d, 360.0 - d)
    return (d.pow(2).mean(-1) ** 0.5).mean()


def calc_part_acc(pts, t1, t2, q1, q2, thr=0.01):
    """utils_3d.py:1089-1129: parts whose two-sided mean squared Chamfer distance between the two posed copies of the
    fragment is below ``thr``, as a fraction of the parts."""
    a = _rotate(q1, pts) + t1[:, None, :]
    b = _rotate(q2, pts) + t2[:, None, :]
    if a.device.type == "cuda":
        from .pcd_encoder import nearest_sq
        d_ab, d_ba = nearest_sq(a, b)
        loss = d_ab.mean(1) + d_ba.mean(1)
    else:                                       # poses already moved to the host by the caller
        d = torch.cdist(a, b, compute_mode="donot_use_mm_for_euclid_dist").pow(2)   # exact differences: the threshold sits near 0
        loss = d.min(2)[0].mean(1) + d.min(1)[0].mean(1)
    return (loss < thr).sum() / loss.numel()
