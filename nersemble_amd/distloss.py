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


# ---- all loss terms + metrics of a training step in two launches (csrc/losses.hip) --------------------------------
LOSS_OUT = 24
(LOSS_RGB, LOSS_ALPHA, LOSS_DEPTH, LOSS_DIST, LOSS_EMPTY, LOSS_NEAR, LOSS_TOTAL, LOSS_PSNR, LOSS_PSNR_MASKED,
 LOSS_NUM_SAMPLES, LOSS_SAMPLE_SUMS) = range(11)


class _StepLosses(torch.autograd.Function):
    """out[LOSS_OUT] = every loss term of models/base.py + nersemble_instant_ngp.py:366-407, their sum and the
    metrics of :409-422.  cfg = (use_masked_rgb, alpha_thr, l_alpha, l_depth, l_dist, l_empty, l_near, eps, max_ray)."""

    @staticmethod
    def forward(ctx, rgb, accumulation, depth, weights, t0, t1, packed, image, alpha_map, depth_targets, cfg):
        f32 = torch.float32
        rgbf = rgb.detach().to(f32).contiguous()
        accf = accumulation.detach().to(f32).reshape(-1).contiguous()
        depf = depth.detach().to(f32).reshape(-1).contiguous()
        img = image.detach().to(f32).contiguous()
        am = alpha_map.reshape(-1).contiguous() if alpha_map is not None else None
        dt = depth_targets.detach().to(f32).reshape(-1).contiguous() if depth_targets is not None else None
        wf = weights.detach().to(f32).reshape(-1).contiguous()
        a = t0.detach().to(f32).reshape(-1).contiguous()
        b = t1.detach().to(f32).reshape(-1).contiguous()
        R = rgbf.shape[0]
        use_masked, thr, l_alpha, l_depth, l_dist, l_empty, l_near, eps, max_ray = cfg
        per_ray = torch.empty((R, 5), dtype=f32, device=rgbf.device)
        check(lib().nsx_sample_losses_fwd(ptr(wf), ptr(a), ptr(b), ptr(packed), R, ptr(dt), float(eps), int(max_ray),
                                          ptr(per_ray), stream()), "nsx_sample_losses_fwd")
        out = torch.empty((LOSS_OUT,), dtype=f32, device=rgbf.device)
        check(lib().nsx_ray_losses_fwd(ptr(rgbf), ptr(accf), ptr(depf), ptr(img), ptr(am), ptr(dt), ptr(per_ray),
                                       ptr(packed), R, int(use_masked), float(thr), float(l_alpha), float(l_depth),
                                       float(l_dist), float(l_empty), float(l_near), ptr(out), stream()),
              "nsx_ray_losses_fwd")
        ctx.save_for_backward(rgbf, accf, depf, img, am, dt, wf, a, b, packed, out)
        ctx.cfg = cfg
        ctx.shapes = (rgb.shape, accumulation.shape, depth.shape, weights.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        rgbf, accf, depf, img, am, dt, wf, a, b, packed, out = ctx.saved_tensors
        use_masked, thr, l_alpha, l_depth, l_dist, l_empty, l_near, eps, max_ray = ctx.cfg
        R = rgbf.shape[0]
        g = g.to(torch.float32).contiguous()
        g_rgb, g_acc, g_dep = torch.empty_like(rgbf), torch.empty_like(accf), torch.empty_like(depf)
        g3 = torch.empty((3,), dtype=torch.float32, device=rgbf.device)
        check(lib().nsx_ray_losses_bwd(ptr(rgbf), ptr(accf), ptr(depf), ptr(img), ptr(am), ptr(dt), R, int(use_masked),
                                       float(thr), float(l_alpha), float(l_depth), float(l_dist), float(l_empty),
                                       float(l_near), R, ptr(out), ptr(g), ptr(g_rgb), ptr(g_acc), ptr(g_dep), ptr(g3),
                                       stream()), "nsx_ray_losses_bwd")
        gw = torch.empty_like(wf)
        if wf.numel() > 0:
            sums = out[LOSS_SAMPLE_SUMS:LOSS_SAMPLE_SUMS + 5]
            check(lib().nsx_sample_losses_bwd(ptr(wf), ptr(a), ptr(b), ptr(packed), R, ptr(dt), float(eps), int(max_ray),
                                              R, ptr(sums), ptr(g3), ptr(gw), stream()), "nsx_sample_losses_bwd")
        s_rgb, s_acc, s_dep, s_w = ctx.shapes
        return (g_rgb.reshape(s_rgb), g_acc.reshape(s_acc), g_dep.reshape(s_dep), gw.reshape(s_w),
                None, None, None, None, None, None, None)


def fused_step_losses(rgb, accumulation, depth, weights, t_starts, t_ends, packed_info, image, alpha_map, depth_targets,
                      *, use_masked_rgb: bool, alpha_mask_threshold: float, lambda_alpha: float, lambda_depth: float,
                      lambda_dist: float, lambda_empty: float, lambda_near: float, eps: float, max_ray: int
                      ) -> torch.Tensor:
    """Returns the [LOSS_OUT] vector of ``nsx_ray_losses_fwd`` (index constants LOSS_* above); differentiable w.r.t.
    rgb, accumulation, depth and weights."""
    cfg = (bool(use_masked_rgb), float(alpha_mask_threshold), float(lambda_alpha or 0.0), float(lambda_depth or 0.0),
           float(lambda_dist), float(lambda_empty), float(lambda_near), float(eps), int(max_ray))
    return _StepLosses.apply(rgb, accumulation, depth, weights, t_starts, t_ends,
                             packed_info.to(torch.int64).contiguous(), image, alpha_map, depth_targets, cfg)
