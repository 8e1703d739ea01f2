"""Packed-sample renderers: nerfstudio 0.3.1's RGBRenderer / DepthRenderer('expected') / AccumulationRenderer
restated for the packed (ray_indices, num_rays) path the reference uses (nersemble_instant_ngp.py:149-152,
:334-343), plus the reference's own DeformationRenderer
(model_components/nersemble_deformation_renderer.py:8-29).  All reductions are native segmented sums
(``accumulate_along_rays``), not index_add."""
from typing import Optional

import torch
from torch import nn, Tensor

from .. import nerfacc
from ..rays import RaySamples


class AccumulationRenderer(nn.Module):
    def forward(self, weights: Tensor, ray_indices: Tensor, num_rays: int, packed_info: Optional[Tensor] = None):
        return nerfacc.accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)


class RGBRenderer(nn.Module):
    def __init__(self, background_color="white"):
        super().__init__()
        if background_color not in ("white", "black"):
            raise NotImplementedError(f"background_color={background_color!r}: NeRSemble trains with a constant background "
                                      "(scripts/train/train_nersemble.py:193 sets 'white'); nerfstudio's 'random' / "
                                      "'last_sample' modes are not part of the path")
        self.background_color = background_color

    def forward(self, rgb: Tensor, weights: Tensor, ray_indices: Tensor, num_rays: int,
                packed_info: Optional[Tensor] = None) -> Tensor:
        comp = nerfacc.accumulate_along_rays(weights[..., 0], rgb, ray_indices, num_rays, packed_info)
        acc = nerfacc.accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)
        bg = 1.0 if self.background_color == "white" else 0.0
        comp = comp + bg * (1.0 - acc)
        if not self.training:
            comp = torch.clamp(comp, min=0.0, max=1.0)
        return comp


class DepthRenderer(nn.Module):
    def __init__(self, method: str = "expected"):
        super().__init__()
        assert method == "expected"
        self.method = method

    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices: Tensor, num_rays: int,
                packed_info: Optional[Tensor] = None) -> Tensor:
        eps = 1e-10
        steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
        depth = nerfacc.accumulate_along_rays(weights[..., 0], steps, ray_indices, num_rays, packed_info)
        acc = nerfacc.accumulate_along_rays(weights[..., 0], None, ray_indices, num_rays, packed_info)
        depth = depth / (acc + eps)
        return torch.clip(depth, steps.min(), steps.max())


class DeformationRenderer(nn.Module):
    def forward(self, weights: Tensor, ray_samples: RaySamples, ray_indices: Optional[Tensor] = None,
                num_rays: Optional[int] = None, packed_info: Optional[Tensor] = None) -> Tensor:
        offsets = ray_samples.frustums.offsets
        if ray_indices is not None and num_rays is not None:
            return nerfacc.accumulate_along_rays(weights.squeeze(1), offsets, ray_indices, num_rays, packed_info)
        eps = 1e-10
        return torch.sum(weights * offsets, dim=-2) / (torch.sum(weights, -2) + eps)
