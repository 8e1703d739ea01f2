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
