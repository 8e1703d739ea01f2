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

    def _directions(self, i: int, ys: Tensor, xs: Tensor) -> Tensor:
        d = torch.stack([(xs - self.cx[i, 0]) / self.fx[i, 0], -(ys - self.cy[i, 0]) / self.fy[i, 0],
                         -torch.ones_like(xs)], dim=-1)
        return d @ self.camera_to_worlds[i, :, :3].T

    def generate_rays(self, camera_indices: int) -> RayBundle:
        """All rays of camera ``camera_indices`` as an image-shaped bundle ``[H, W, ...]`` (row-major)."""
        i = int(camera_indices)
        h, w = int(self.height[i, 0]), int(self.width[i, 0])
        dev = self.device
        ys, xs = torch.meshgrid(torch.arange(h, device=dev, dtype=torch.float32) + 0.5,
                                torch.arange(w, device=dev, dtype=torch.float32) + 0.5, indexing="ij")
        raw = self._directions(i, ys, xs)
        directions = raw / raw.norm(dim=-1, keepdim=True)
        right = self._directions(i, ys, xs + 1)
        down = self._directions(i, ys + 1, xs)
        dx = (directions - right / right.norm(dim=-1, keepdim=True)).norm(dim=-1, keepdim=True)
        dy = (directions - down / down.norm(dim=-1, keepdim=True)).norm(dim=-1, keepdim=True)
        origins = self.camera_to_worlds[i, :, 3].expand(h, w, 3).contiguous()
        times = None if self.times is None else self.times[i].expand(h, w, 1).contiguous()
        return RayBundle(origins=origins, directions=directions.contiguous(), pixel_area=dx * dy,
                         camera_indices=torch.full((h, w, 1), i, device=dev, dtype=torch.long), times=times)
