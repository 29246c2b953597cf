"""SO(3) helpers the 3D path calls (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restated from puzzle_diff/model/backbones/efficient_gat_3d.py:30-45 (vec2skew,
skew_to_rmat) and puzzle_diff/model/utils_3d.py:1018-1071 (log_rmat, so3_scale).
"""
import torch


def vec2skew(vec):
    """efficient_gat_3d.py:30-35 / utils_3d.py: [.., 3] -> [.., 3, 3] with
    S[2,1]=v0, S[2,0]=-v1, S[1,0]=v2, antisymmetrised."""
    z = torch.zeros_like(vec[..., 0])
    v0, v1, v2 = vec[..., 0], vec[..., 1], vec[..., 2]
    return torch.stack([
        torch.stack([z, -v2, v1], -1),
        torch.stack([v2, z, -v0], -1),
        torch.stack([-v1, v0, z], -1)], -2)


def skew2vec(skew):
    """utils_3d.py skew2vec: inverse of vec2skew."""
    return torch.stack([skew[..., 2, 1], -skew[..., 2, 0], skew[..., 1, 0]], -1)


def skew_to_rmat(v):
    """efficient_gat_3d.py:38-45 (check=False): matrix_exp(vec2skew(v))."""
    return torch.matrix_exp(vec2skew(v))


def log_rmat(r_mat):
    """utils_3d.py:1018-1046.  atan2 form; rotations by exactly 0 get a zero log; NaNs
    (rotation by pi) fall back to the eigh axis."""
    skew_mat = r_mat - r_mat.transpose(-1, -2)
    sk_vec = skew2vec(skew_mat)
    s_angle = sk_vec.norm(p=2, dim=-1) / 2
    c_angle = (torch.einsum("...ii", r_mat) - 1) / 2
    angle = torch.atan2(s_angle, c_angle)
    scale = angle / (2 * s_angle)
    scale = torch.where(angle == 0.0, torch.zeros_like(scale), scale)
    log_r = scale[..., None, None] * skew_mat
    nanlocs = log_r[..., 0, 0].isnan()
    if nanlocs.any():
        _, eigvec = torch.linalg.eigh(r_mat[nanlocs])
        nan_axes = eigvec[..., -1, :]
        log_r = log_r.clone()
        log_r[nanlocs] = vec2skew(angle[nanlocs][..., None] * nan_axes)
    return log_r


def so3_scale(rmat, scalars):
    """utils_3d.py:1049-1061: matrix_exp(scalars * log_rmat(rmat))."""
    return torch.matrix_exp(log_rmat(rmat) * scalars[..., None, None])
