"""NeRSembleVolumetricSampler -- mirror of the reference's model_components/nersemble_volumetric_sampler.py:13-135
(on nerfstudio's VolumetricSampler): optional view-frustum culling AND-ed into ``binaries[0]``, then
``OccGridEstimator.sampling`` (native traversal + sigma_fn visibility pruning), packed ``RaySamples`` +
``ray_indices`` with the 1-fake-sample fallback."""
from typing import Callable, List, Optional, Tuple

import torch
from torch import nn, Tensor

from .. import functional as F
from ..nerfacc import OccGridEstimator
from ..rays import Frustums, RayBundle, RaySamples
from .frustum import TorchFrustum, visibility_grid


class NeRSembleVolumetricSampler(nn.Module):

    def __init__(self, occupancy_grid: OccGridEstimator, density_fn: Optional[Callable] = None,
                 scene_aabb: Optional[Tensor] = None, camera_frustums: Optional[List[TorchFrustum]] = None,
                 view_frustum_culling: Optional[int] = None):
        super().__init__()
        assert occupancy_grid is not None
        self.density_fn = density_fn
        self.occupancy_grid = occupancy_grid
        self.camera_frustums = camera_frustums
        self.view_frustum_culling = view_frustum_culling
        self.camera_frustum_grid = None
        if camera_frustums is not None and view_frustum_culling is not None:
            self.camera_frustum_grid = visibility_grid(camera_frustums, scene_aabb, self.occupancy_grid.resolution,
                                                       view_frustum_culling, device="cpu")

    def get_sigma_fn(self, origins: Tensor, directions: Tensor, times: Optional[Tensor] = None) -> Optional[Callable]:
        """nerfstudio VolumetricSampler.get_sigma_fn: None in eval mode / without a density_fn."""
        if self.density_fn is None or not self.training:
            return None
        density_fn = self.density_fn

        def sigma_fn(t_starts, t_ends, ray_indices):
            if origins.is_cuda:
                positions = F.sample_positions(origins, directions, t_starts, t_ends, ray_indices)
            else:
                positions = origins[ray_indices] + directions[ray_indices] * (t_starts + t_ends)[:, None] / 2.0
            if times is None:
                return density_fn(positions).squeeze(-1)
            return density_fn(positions, times[ray_indices]).squeeze(-1)

        return sigma_fn

    def forward(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0,
                far_plane: Optional[float] = None, alpha_thre: float = 0.01, cone_angle: float = 0.0,
                early_stop_eps: float = 1e-4) -> Tuple[RaySamples, Tensor]:
        rays_o = ray_bundle.origins.contiguous()
        rays_d = ray_bundle.directions.contiguous()
        times = ray_bundle.times
        t_min = t_max = None
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            t_min = ray_bundle.nears.contiguous().reshape(-1)
            t_max = ray_bundle.fars.contiguous().reshape(-1)
        if far_plane is None:
            far_plane = 1e10
        camera_indices = ray_bundle.camera_indices.contiguous() if ray_bundle.camera_indices is not None else None

        if self.camera_frustum_grid is not None:
            if self.camera_frustum_grid.device != self.occupancy_grid.binaries.device:
                self.camera_frustum_grid = self.camera_frustum_grid.to(self.occupancy_grid.binaries.device)
            # view-frustum culling only on the coarsest level (:90-93)
            self.occupancy_grid.binaries[0] = self.occupancy_grid.binaries[0] & self.camera_frustum_grid

        ray_indices, starts, ends = self.occupancy_grid.sampling(
            rays_o=rays_o, rays_d=rays_d, t_min=t_min, t_max=t_max,
            sigma_fn=self.get_sigma_fn(rays_o, rays_d, times), render_step_size=render_step_size,
            near_plane=near_plane, far_plane=far_plane, stratified=self.training, cone_angle=cone_angle,
            alpha_thre=alpha_thre, early_stop_eps=early_stop_eps)
        if starts.shape[0] == 0:
            # a single fake sample so downstream shapes stay valid (:109-115)
            ray_indices = torch.zeros((1,), dtype=torch.long, device=rays_o.device)
            starts = torch.ones((1,), dtype=starts.dtype, device=rays_o.device)
            ends = torch.ones((1,), dtype=ends.dtype, device=rays_o.device)

        origins = rays_o[ray_indices]
        dirs = rays_d[ray_indices]
        if camera_indices is not None:
            camera_indices = camera_indices[ray_indices]
        ray_samples = RaySamples(
            frustums=Frustums(origins=origins, directions=dirs, starts=starts[..., None], ends=ends[..., None],
                              pixel_area=torch.zeros_like(origins[:, :1])),
            camera_indices=camera_indices)
        if ray_bundle.times is not None:
            ray_samples.times = ray_bundle.times[ray_indices]
        return ray_samples, ray_indices
