"""Minimal perspective ``Cameras`` container -- the slice of nerfstudio 0.3.1's ``nerfstudio.cameras.cameras.Cameras``
that the evaluation / trajectory-rendering callers of the path use (``util/render.py:29-36`` in the reference:
``rescale_output_resolution``, ``to``, ``size``, ``generate_rays(camera_indices=i)``; ``evaluate_nersemble.py:95-98``:
``.times``).  nerfstudio is a third-party dependency that is not installed here: restated from its documented
behaviour (OpenGL poses, pixel centres at +0.5, unit directions, ``pixel_area`` from the neighbouring-pixel direction
differences) -- PARITY UNPINNED.  Only full-image ray generation for pinhole cameras without distortion is provided,
which is what the reference configures (``nersemble_dataparser.py:237-244``: all distortion parameters zero)."""
from typing import Optional

import torch
from torch import Tensor

from .rays import RayBundle


def _column(v, n: int, dtype) -> Tensor:
    t = torch.as_tensor(v, dtype=dtype).reshape(-1, 1)
    return t.expand(n, 1).clone() if t.shape[0] == 1 else t


class Cameras:
    def __init__(self, camera_to_worlds: Tensor, fx, fy, cx, cy, width, height, times: Optional[Tensor] = None):
        self.camera_to_worlds = torch.as_tensor(camera_to_worlds, dtype=torch.float32)[..., :3, :4].reshape(-1, 3, 4)
        n = self.camera_to_worlds.shape[0]
        self.fx, self.fy = _column(fx, n, torch.float32), _column(fy, n, torch.float32)
        self.cx, self.cy = _column(cx, n, torch.float32), _column(cy, n, torch.float32)
        self.width, self.height = _column(width, n, torch.int64), _column(height, n, torch.int64)
        self.times = None if times is None else torch.as_tensor(times, dtype=torch.float32).reshape(n, 1)

    @property
    def size(self) -> int:
        return self.camera_to_worlds.shape[0]

    def __len__(self) -> int:
        return self.size

    @property
    def device(self):
        return self.camera_to_worlds.device

    def to(self, device) -> "Cameras":
        out = object.__new__(Cameras)
        for k, v in vars(self).items():
            setattr(out, k, v.to(device) if isinstance(v, Tensor) else v)
        return out

    def rescale_output_resolution(self, scaling_factor: float) -> None:
        """In place, like nerfstudio: intrinsics scale with the factor, image sizes are floored."""
        self.fx, self.fy = self.fx * scaling_factor, self.fy * scaling_factor
        self.cx, self.cy = self.cx * scaling_factor, self.cy * scaling_factor
        self.width = (self.width * scaling_factor).to(torch.int64)
        self.height = (self.height * scaling_factor).to(torch.int64)

    def generate_rays(self, camera_indices, coords: Optional[Tensor] = None) -> RayBundle:
        """``generate_rays(i)``: all rays of camera ``i`` as an image-shaped bundle ``[H, W, ...]`` (row-major).
        ``generate_rays(camera_indices [N, 1], coords [N, 2])``: one ray per row, ``coords`` = (y, x) in pixels with
        the pixel centre already added (what nerfstudio's ``RayGenerator`` passes)."""
        if coords is not None:
            cams = torch.as_tensor(camera_indices, device=self.device).reshape(-1).long()
            return self._rays_at(cams, coords[:, 0].to(self.device), coords[:, 1].to(self.device))
        i = int(camera_indices)
        h, w = int(self.height[i, 0]), int(self.width[i, 0])
        dev = self.device
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32) + 0.5,
                                torch.arange(w, device=dev, dtype=torch.float32) + 0.5, indexing="ij")
        flat = self._rays_at(torch.full((h * w,), i, device=dev, dtype=torch.long), ys.reshape(-1), xs.reshape(-1))
        return RayBundle(origins=flat.origins.view(h, w, 3), directions=flat.directions.view(h, w, 3),
                         pixel_area=flat.pixel_area.view(h, w, 1), camera_indices=flat.camera_indices.view(h, w, 1),
                         times=None if flat.times is None else flat.times.view(h, w, 1))

    def _rays_at(self, cams: Tensor, ys: Tensor, xs: Tensor) -> RayBundle:
        if self.camera_to_worlds.is_cuda:
            return self._rays_at_native(cams, ys, xs)
        # host mirror of csrc/raygen.hip (camera containers that live on the CPU: tests, the evaluation tools' set-up)
        rot = self.camera_to_worlds[cams, :, :3]                       # [N, 3, 3]
        fx, fy, cx, cy = (t[cams, 0] for t in (self.fx, self.fy, self.cx, self.cy))

        def unit_dirs(y, x):
            d = torch.stack([(x - cx) / fx, -(y - cy) / fy, -torch.ones_like(x)], dim=-1)
            d = torch.einsum("nij,nj->ni", rot, d)
            return d / d.norm(dim=-1, keepdim=True)

        directions = unit_dirs(ys, xs)
        dx = (directions - unit_dirs(ys, xs + 1)).norm(dim=-1, keepdim=True)
        dy = (directions - unit_dirs(ys + 1, xs)).norm(dim=-1, keepdim=True)
        times = None if self.times is None else self.times[cams]
        return RayBundle(origins=self.camera_to_worlds[cams, :, 3].contiguous(), directions=directions.contiguous(),
                         pixel_area=dx * dy, camera_indices=cams[:, None], times=times)

    def _rays_at_native(self, cams: Tensor, ys: Tensor, xs: Tensor) -> RayBundle:
        """One launch of ``nsx_generate_rays`` (csrc/raygen.hip) instead of ~35 torch kernels."""
        from ._lib import check, lib, ptr, stream
        dev = self.device
        cams = cams.to(device=dev, dtype=torch.int64).contiguous()
        ys = ys.to(device=dev, dtype=torch.float32).contiguous()
        xs = xs.to(device=dev, dtype=torch.float32).contiguous()
        n = cams.shape[0]
        origins = torch.empty((n, 3), dtype=torch.float32, device=dev)
        directions = torch.empty((n, 3), dtype=torch.float32, device=dev)
        area = torch.empty((n, 1), dtype=torch.float32, device=dev)
        c2w = self.camera_to_worlds.contiguous()
        fx, fy, cx, cy = (t.reshape(-1).contiguous() for t in (self.fx, self.fy, self.cx, self.cy))
        check(lib().nsx_generate_rays(ptr(c2w, torch.float32), ptr(fx, torch.float32), ptr(fy, torch.float32),
                                      ptr(cx, torch.float32), ptr(cy, torch.float32), self.size, ptr(cams), ptr(ys), ptr(xs),
                                      n, ptr(origins), ptr(directions), ptr(area), stream()), "nsx_generate_rays")
        times = None if self.times is None else self.times[cams]
        return RayBundle(origins=origins, directions=directions, pixel_area=area, camera_indices=cams[:, None], times=times)


class RayGenerator:
    """nerfstudio's ``RayGenerator``: (camera, y, x) pixel indices -> ``RayBundle`` (pixel centres at +0.5); the
    module the datamanager calls on the pixel sampler's ``indices`` (``VanillaDataManager.next_train``, UPSTREAM)."""

    def __init__(self, cameras: Cameras):
        self.cameras = cameras

    def __call__(self, ray_indices: Tensor) -> RayBundle:
        c, y, x = ray_indices[:, 0], ray_indices[:, 1], ray_indices[:, 2]
        coords = torch.stack([y, x], dim=-1).to(torch.float32) + 0.5
        return self.cameras.generate_rays(camera_indices=c.unsqueeze(-1), coords=coords)
