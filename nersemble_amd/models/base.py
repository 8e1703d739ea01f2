"""BaseModel losses -- mirror of the reference's nerfstudio/models/base.py:15-249 (BaseModelConfig fields, masked
RGB MSE, alpha L1, empty / near / depth losses, efficient distortion loss, eps-depth scheduler).  The distortion
loss runs on the native segmented-scan kernel (``nersemble_amd.distloss.flatten_eff_distloss``)."""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import nn
from torch.distributions import Normal

from ..distloss import flatten_eff_distloss
from ..engine.generic_scheduler import GenericScheduler
from ..rays import RaySamples


@dataclass
class BaseModelConfig:
    use_masked_rgb_loss: bool = False
    alpha_mask_threshold: float = 0.5
    lambda_alpha_loss: float = 0
    lambda_empty_loss: float = 0
    lambda_near_loss: float = 0
    lambda_depth_loss: float = 0
    eps_depth_initial: float = 0.9
    eps_depth_final: float = 0.01
    eps_depth_begin_step: int = 0
    eps_depth_end_step: int = 10000
    lambda_dist_loss: float = 0
    dist_loss_max_rays: int = 5000


def _masked_mean(values: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """mean of values[mask] (0 when the mask is empty) without materialising the selection."""
    m = mask.to(values.dtype)
    return (values * m).sum() / m.sum().clamp(min=1.0)


def select_dist_loss_samples(ray_indices, weights, starts, ends, max_rays: int):
    """models/base.py:233-243: keep samples of rays with index < max_rays; midpoints and intervals."""
    w = weights.squeeze(1)
    keep = ray_indices < max_rays
    ray_id = ray_indices[keep]
    w_small = w[keep]
    e = ends[keep].squeeze(-1)
    s = starts[keep].squeeze(-1)
    return w_small, (e + s) * 0.5, e - s, ray_id


class BaseModel(nn.Module):
    config: BaseModelConfig

    def populate_modules(self):
        if self.config.lambda_empty_loss > 0 or self.config.lambda_near_loss > 0:
            self.sched_eps_depth = GenericScheduler(init_value=self.config.eps_depth_initial,
                                                    final_value=self.config.eps_depth_final,
                                                    begin_step=self.config.eps_depth_begin_step,
                                                    end_step=self.config.eps_depth_end_step)
        else:
            self.sched_eps_depth = None

    @property
    def device(self):
        return next(self.parameters()).device

    @staticmethod
    def get_alpha_per_ray(batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        assert "alpha_map" in batch
        return batch["alpha_map"].squeeze(1) / 255.

    def get_masked_rgb_loss(self, batch, rgb_pred: torch.Tensor) -> torch.Tensor:
        image = batch["image"].to(rgb_pred.device)
        if self.config.use_masked_rgb_loss and "alpha_map" in batch:
            mask = self.get_alpha_per_ray(batch) > self.config.alpha_mask_threshold
            # == MSELoss()(image[mask], rgb_pred[mask]) without the boolean-index copy and its host sync
            return _masked_mean(((image - rgb_pred) ** 2).mean(dim=-1), mask)
        return torch.nn.functional.mse_loss(image, rgb_pred)

    def get_alpha_loss(self, batch, accumulation: torch.Tensor) -> Optional[torch.Tensor]:
        if self.config.lambda_alpha_loss is None or self.config.lambda_alpha_loss <= 0:
            return None
        acc = accumulation.squeeze(1)
        alpha = self.get_alpha_per_ray(batch)
        bg = alpha < 1
        # the reference returns None when no background ray is present (:129); the masked mean is 0 then, which
        # leaves the summed loss unchanged and needs no host synchronisation
        return _masked_mean((acc - alpha).abs(), bg) * self.config.lambda_alpha_loss

    def get_near_and_empty_loss(self, batch, ray_samples: RaySamples, ray_indices, weights, accumulation
                                ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        near_loss = empty_loss = None
        if (self.config.lambda_empty_loss > 0 or self.config.lambda_near_loss > 0) and self.training:
            eps = self.sched_eps_depth.value
            depth_targets = batch["depth_maps"]
            starts = ray_samples.frustums.starts.squeeze(1)
            ends = ray_samples.frustums.ends.squeeze(1)
            midpoints = (starts + ends) * 0.5
            target = depth_targets[ray_indices]
            w = weights.squeeze(1)
            if self.config.lambda_empty_loss > 0:
                very_near = (target > 0) & (midpoints < target - eps)
                empty_loss = self.config.lambda_empty_loss * _masked_mean(w ** 2, very_near)
            if self.config.lambda_near_loss > 0:
                near = (target > 0) & (target - eps <= midpoints) & (midpoints <= target + eps)
                # accumulated weight up to and including each sample, per ray (base.py:176-190 does this with a
                # global cumsum minus the value at each ray's head); heads found with a running maximum, no sync
                csum = w.cumsum(dim=0)
                is_head = torch.ones_like(ray_indices, dtype=torch.bool)
                is_head[1:] = ray_indices[1:] != ray_indices[:-1]
                pos = torch.arange(w.shape[0], device=w.device)
                head_idx = torch.cummax(torch.where(is_head, pos, torch.zeros_like(pos)), dim=0).values
                accumulated = csum - csum[head_idx] + w[head_idx]
                expected = Normal(0, (eps / 3) ** 2).cdf(midpoints - target)     # sigma = (eps/3)^2 as in :162
                near_loss = self.config.lambda_near_loss * _masked_mean((accumulated - expected) ** 2, near)
        return near_loss, empty_loss

    def get_depth_loss(self, batch, depths: torch.Tensor) -> Optional[torch.Tensor]:
        if not (self.config.lambda_depth_loss > 0 and self.training):
            return None
        target = batch["depth_maps"]
        pred = depths.squeeze()
        mask = target > 0
        return _masked_mean((target - pred) ** 2, mask) * self.config.lambda_depth_loss

    def get_dist_loss(self, ray_samples: RaySamples, ray_indices, weights, num_rays: Optional[int] = None,
                      packed_info: Optional[torch.Tensor] = None):
        if not self.config.lambda_dist_loss > 0:
            return None
        max_rays = self.config.dist_loss_max_rays
        if num_rays is not None and num_rays <= max_rays:
            # every ray qualifies (R = 4096 < 5000): no boolean-mask copy, reuse the step's packed_info
            w = weights.squeeze(1)
            s = ray_samples.frustums.starts.squeeze(-1)
            e = ray_samples.frustums.ends.squeeze(-1)
            return self.config.lambda_dist_loss * flatten_eff_distloss(w, (e + s) * 0.5, e - s, ray_indices,
                                                                       packed_info=packed_info)
        w, m, iv, rid = select_dist_loss_samples(ray_indices, weights, ray_samples.frustums.starts,
                                                 ray_samples.frustums.ends, max_rays)
        return self.config.lambda_dist_loss * flatten_eff_distloss(w, m, iv, rid)
