"""SE(3) exponential map with the semantics of the reference's util/pytorch3d.py:107-191 (itself taken from
pytorch3d): squared-norm clamp at eps=1e-4, Rodrigues rotation, the "V" matrix for the translation, row-vector
convention of the returned 4x4 (``transform[:, 3, :3] = T``).  Written in closed form (no hat-matrix bmm)."""
import torch


def _skew_terms(r: torch.Tensor):
    x, y, z = r.unbind(-1)
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=-1).reshape(-1, 3, 3)
    # K @ K = r r^T - |r|^2 I
    K2 = r[:, :, None] * r[:, None, :] - (r * r).sum(-1)[:, None, None] * torch.eye(3, dtype=r.dtype, device=r.device)
    return K, K2


def se3_exp_map(log_transform: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    if log_transform.ndim != 2 or log_transform.shape[1] != 6:
        raise ValueError("Expected input to be of shape (N, 6).")
    v, r = log_transform[:, :3], log_transform[:, 3:]
    theta = torch.clamp((r * r).sum(1), eps).sqrt()
    K, K2 = _skew_terms(r)
    eye = torch.eye(3, dtype=r.dtype, device=r.device)[None]
    fac1 = (1.0 / theta) * theta.sin()
    fac2 = (1.0 / theta) * (1.0 / theta) * (1.0 - theta.cos())
    R = fac1[:, None, None] * K + fac2[:, None, None] * K2 + eye
    V = eye + K * ((1 - torch.cos(theta)) / (theta ** 2))[:, None, None] \
        + K2 * ((theta - torch.sin(theta)) / (theta ** 3))[:, None, None]
    T = torch.bmm(V, v[:, :, None])[:, :, 0]
    out = torch.zeros(log_transform.shape[0], 4, 4, dtype=log_transform.dtype, device=log_transform.device)
    out[:, :3, :3] = R.transpose(1, 2)      # row-vector convention (the reference returns transform.permute(0,2,1))
    out[:, 3, :3] = T
    out[:, 3, 3] = 1.0
    return out
