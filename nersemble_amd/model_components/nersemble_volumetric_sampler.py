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

    # set by the model: times [R, 1] -> int32 timesteps [R] (``round(times * (T - 1))``, nersemble_instant_ngp.py:249)
    timestep_fn: Optional[Callable] = None

    def get_sigma_fn(self, origins: Tensor, directions: Tensor, times: Optional[Tensor] = None,
                     ray_timesteps: Optional[Tensor] = None) -> Optional[Callable]:
        """nerfstudio VolumetricSampler.get_sigma_fn: None in eval mode / without a density_fn.

        ``ray_timesteps`` (native extension): the rays' integer timesteps if the bundle carries them
        (``metadata["timesteps"]``, what ``_add_metadata_to_ray_bundle`` attaches, nersemble_datamanager.py:68-74).  The
        reference gathers ``times[ray_indices]`` per SAMPLE and the model rounds them to timesteps per sample
        (nersemble_instant_ngp.py:249) -- four sample-sized launches; rounding commutes with the gather, so here the
        per-ray timesteps (from the metadata, else rounded once per ray by ``timestep_fn``) are gathered in one launch and
        handed to the density function as ``timesteps=``."""
        if self.density_fn is None or not self.training:
            return None
        density_fn = self.density_fn
        ray_ts = None
        if times is not None and origins.is_cuda and getattr(density_fn, "accepts_timesteps", False):
            # the reference's priority: the rays' ``times`` decide (``round(times * (T - 1))``, nersemble_instant_ngp.py:249;
            # the main pass does the same, :300-318).  Timesteps that travel as metadata are trusted only after they have
            # been SEEN to agree with the rounded times (first call, then every 256th: one small comparison with a host
            # read) -- a dataparser with start_timestep / skip_timesteps carries original frame ids there, not indices
            have_md = ray_timesteps is not None and ray_timesteps.numel() == origins.shape[0]
            n = self._md_timestep_checks = getattr(self, "_md_timestep_checks", -1) + 1
            if have_md and self.timestep_fn is not None and n % 256 == 0:
                rounded = self.timestep_fn(times).reshape(-1).to(torch.int32)
                self._md_timesteps_agree = bool(torch.equal(rounded, ray_timesteps.reshape(-1).to(torch.int32)))
            if have_md and (self.timestep_fn is None or getattr(self, "_md_timesteps_agree", False)):
                ray_ts = ray_timesteps.reshape(-1).to(torch.int32).contiguous()
            elif self.timestep_fn is not None:
                ray_ts = self.timestep_fn(times).reshape(-1).to(torch.int32).contiguous()

        def sigma_fn(t_starts, t_ends, ray_indices):
            if origins.is_cuda:
                positions = F.sample_positions(origins, directions, t_starts, t_ends, ray_indices)
            else:
                positions = origins[ray_indices] + directions[ray_indices] * (t_starts + t_ends)[:, None] / 2.0
            if times is None:
                return density_fn(positions).squeeze(-1)
            if ray_ts is not None:
                return density_fn(positions, None, timesteps=F.gather_rows(ray_indices, ray_ts)[0]).squeeze(-1)
            return density_fn(positions, times[ray_indices]).squeeze(-1)

        return sigma_fn

    def _cull_to_camera_frusta(self) -> None:
        """Cells seen by fewer than ``view_frustum_culling`` training cameras never hold samples (reference :90-93)."""
        grid = self.camera_frustum_grid
        if grid is None:
            return
        binaries = self.occupancy_grid.binaries
        if grid.device != binaries.device:
            grid = self.camera_frustum_grid = grid.to(binaries.device)
        stamp = (binaries.data_ptr(), binaries._version)
        if getattr(self, "_culled_stamp", None) == stamp:
            return                                   # nobody wrote the grid since it was culled: the AND is idempotent
        binaries[0] = binaries[0] & grid
        self._culled_stamp = (binaries.data_ptr(), binaries._version)

    def _zero_column(self, like: Tensor) -> Tensor:
        """A [1, 1] zero of ``like``'s dtype / device (the packed samples' unused ``pixel_area`` is a broadcast view of it:
        the reference writes zeros there, :125; one fill launch per step otherwise)."""
        z = getattr(self, "_zero11", None)
        if z is None or z.device != like.device or z.dtype != like.dtype:
            z = self._zero11 = torch.zeros((1, 1), dtype=like.dtype, device=like.device)
        return z

    # per-ray entries of ``ray_bundle.metadata`` that the model wants per sample (gathered in the same launch)
    sample_metadata_keys = ("image_index",)

    def _packed_samples(self, bundle: RayBundle, o: Tensor, d: Tensor, ray_indices: Tensor, t0: Tensor, t1: Tensor,
                        n_dev: Optional[Tensor] = None) -> RaySamples:
        """One RaySamples row per marched interval, rays gathered through ``ray_indices`` (flattened / packed layout).
        All per-ray arrays (origins, directions, camera indices, times, requested metadata) go through ONE gather
        launch on the device (the reference issues one advanced-indexing op per field, :117-134)."""
        named = [("origins", o), ("directions", d)]
        if bundle.camera_indices is not None:
            named.append(("camera_indices", bundle.camera_indices.contiguous()))
        if bundle.times is not None:
            named.append(("times", bundle.times))
        for key in self.sample_metadata_keys:
            value = bundle.metadata.get(key) if bundle.metadata else None
            if isinstance(value, Tensor) and value.shape[0] == o.shape[0]:
                named.append(("metadata:" + key, value))
        if o.is_cuda:
            from .._lib import device_count
            with device_count(n_dev, ray_indices.shape[0]):          # (n_dev None: every row)
                rows = F.gather_rows(ray_indices, *[t for _, t in named], zero_fill=n_dev is not None)
        else:
            rows = [t[ray_indices] for _, t in named]
        got = dict(zip([n for n, _ in named], rows))
        samples = RaySamples(
            frustums=Frustums(origins=got["origins"], directions=got["directions"], starts=t0[..., None],
                              ends=t1[..., None],
                              pixel_area=self._zero_column(o).expand(ray_indices.shape[0], 1)),
            camera_indices=got.get("camera_indices"))
        if "times" in got:
            samples.times = got["times"]
        extra = {n[len("metadata:"):]: v for n, v in got.items() if n.startswith("metadata:")}
        if extra:
            samples.metadata = extra
        return samples

    @staticmethod
    def _march_inputs(ray_bundle: RayBundle):
        o, d = ray_bundle.origins.contiguous(), ray_bundle.directions.contiguous()
        per_ray_near = per_ray_far = None
        if ray_bundle.nears is not None and ray_bundle.fars is not None:
            per_ray_near = ray_bundle.nears.contiguous().reshape(-1)
            per_ray_far = ray_bundle.fars.contiguous().reshape(-1)
        return o, d, per_ray_near, per_ray_far

    def prefetch(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0,
                 far_plane: Optional[float] = None) -> bool:
        """Start the traversal's counting pass for a later ``forward`` on this very bundle with these arguments
        (``OccGridEstimator.prefetch_march``).  False when there is nothing to gain (CPU, per-ray far planes)."""
        o, d, per_ray_near, per_ray_far = self._march_inputs(ray_bundle)
        if per_ray_far is not None or not o.is_cuda:
            return False
        self._cull_to_camera_frusta()
        return self.occupancy_grid.prefetch_march(o, d, near_plane=near_plane,
                                                  far_plane=1e10 if far_plane is None else far_plane, t_min=per_ray_near,
                                                  render_step_size=render_step_size, stratified=self.training)

    def forward(self, ray_bundle: RayBundle, render_step_size: float, near_plane: float = 0.0,
                far_plane: Optional[float] = None, alpha_thre: float = 0.01, cone_angle: float = 0.0,
                early_stop_eps: float = 1e-4, device_counts: bool = False) -> Tuple[RaySamples, Tensor]:
        """``device_counts`` (native extension, training fast path): see ``OccGridEstimator.sampling`` -- the returned
        arrays keep the marched capacity, ``self.occupancy_grid.last_n_kept`` holds the number of valid rows."""
        o, d, per_ray_near, per_ray_far = self._march_inputs(ray_bundle)
        self._cull_to_camera_frusta()
        ray_indices, t0, t1 = self.occupancy_grid.sampling(
            rays_o=o, rays_d=d, t_min=per_ray_near, t_max=per_ray_far,
            sigma_fn=self.get_sigma_fn(o, d, ray_bundle.times, (ray_bundle.metadata or {}).get("timesteps")),
            render_step_size=render_step_size,
            near_plane=near_plane, far_plane=1e10 if far_plane is None else far_plane, stratified=self.training,
            cone_angle=cone_angle, alpha_thre=alpha_thre, early_stop_eps=early_stop_eps, device_counts=device_counts)
        n_dev = self.occupancy_grid.last_n_kept if device_counts else None
        if n_dev is not None:
            return self._packed_samples(ray_bundle, o, d, ray_indices, t0, t1, n_dev), ray_indices
        if t0.shape[0] == 0:
            # nothing survived: one dummy interval on ray 0 keeps every downstream shape valid (reference :109-115)
            ray_indices = torch.zeros((1,), dtype=torch.long, device=o.device)
            t0 = torch.ones((1,), dtype=t0.dtype, device=o.device)
            t1 = torch.ones((1,), dtype=t1.dtype, device=o.device)
        return self._packed_samples(ray_bundle, o, d, ray_indices, t0, t1), ray_indices
