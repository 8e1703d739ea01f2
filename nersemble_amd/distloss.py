"""``flatten_eff_distloss`` with torch_efficient_distloss's signature (reference call: models/base.py:245-247),
backed by one HIP segmented-scan kernel (forward value and analytic dL/dw in the same pass)."""
import torch

from ._lib import check, lib, ptr, stream
from .nerfacc import pack_info


class _FlattenEffDistLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, m, interval, packed):
        wf = w.detach().to(torch.float32).contiguous()
        mf = m.detach().to(torch.float32).contiguous()
        iv = interval.detach().to(torch.float32).contiguous()
        R = packed.shape[0]
        ray_loss = torch.empty((R,), dtype=torch.float32, device=wf.device)
        grad_w = torch.empty_like(wf)
        check(lib().nsx_distloss(ptr(wf), ptr(mf), ptr(iv), ptr(packed), R, R, 1, 1.0, ptr(ray_loss), ptr(grad_w),
                                 stream()), "nsx_distloss")
        ctx.save_for_backward(grad_w)
        return ray_loss.sum()

    @staticmethod
    def backward(ctx, g):
        (grad_w,) = ctx.saved_tensors
        return grad_w * g, None, None, None


def flatten_eff_distloss(w: torch.Tensor, m: torch.Tensor, interval: torch.Tensor, ray_id: torch.Tensor,
                         packed_info: torch.Tensor = None) -> torch.Tensor:
    """w, m, interval: [N]; ray_id: sorted long [N].  loss = (sum 1/3 interval w^2 + 2 w (m Wpre - WMpre)) / n_rays
    with n_rays = ray_id.max()+1 (torch_efficient_distloss semantics)."""
    if w.numel() == 0:
        return w.sum()
    n_rays_t = ray_id.max() + 1                                   # stays on device (no sync)
    if packed_info is None:
        packed_info = pack_info(ray_id, int(n_rays_t.item()))
    # a caller-provided packed_info may cover more (empty) rays than ray_id.max()+1: they contribute nothing
    return _FlattenEffDistLoss.apply(w, m, interval, packed_info.contiguous()) / n_rays_t


class _SampleLosses(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, t0, t1, packed, depth_targets, eps, max_ray, n_rays):
        wf = w.detach().to(torch.float32).contiguous()
        a = t0.detach().to(torch.float32).contiguous()
        b = t1.detach().to(torch.float32).contiguous()
        dt = depth_targets.detach().to(torch.float32).contiguous() if depth_targets is not None else None
        R = packed.shape[0]
        per_ray = torch.zeros((R, 5), dtype=torch.float32, device=wf.device)
        if wf.numel() > 0:
            check(lib().nsx_sample_losses_fwd(ptr(wf), ptr(a), ptr(b), ptr(packed), R, ptr(dt), float(eps), int(max_ray),
                                              ptr(per_ray), stream()), "nsx_sample_losses_fwd")
        sums = per_ray.sum(dim=0)
        ctx.save_for_backward(wf, a, b, packed, dt, sums)
        ctx.cfg = (float(eps), int(max_ray), int(n_rays))
        one = torch.ones((), dtype=torch.float32, device=wf.device)
        out = torch.stack([sums[0] / n_rays, sums[1] / torch.maximum(sums[2], one), sums[3] / torch.maximum(sums[4], one)])
        return out

    @staticmethod
    def backward(ctx, g):
        wf, a, b, packed, dt, sums = ctx.saved_tensors
        eps, max_ray, n_rays = ctx.cfg
        gw = torch.zeros_like(wf)
        if wf.numel() > 0:
            check(lib().nsx_sample_losses_bwd(ptr(wf), ptr(a), ptr(b), ptr(packed), packed.shape[0], ptr(dt), eps, max_ray,
                                              n_rays, ptr(sums.contiguous()), ptr(g.to(torch.float32).contiguous()),
                                              ptr(gw), stream()), "nsx_sample_losses_bwd")
        return gw, None, None, None, None, None, None, None


def fused_sample_losses(weights: torch.Tensor, t_starts: torch.Tensor, t_ends: torch.Tensor, packed_info: torch.Tensor,
                        depth_targets, eps: float, max_ray: int, n_rays: int) -> torch.Tensor:
    """Returns a [3] tensor (dist, empty, near) -- un-weighted (the caller multiplies by the lambdas)."""
    return _SampleLosses.apply(weights, t_starts, t_ends, packed_info.to(torch.int64).contiguous(), depth_targets, eps,
                               max_ray, n_rays)
