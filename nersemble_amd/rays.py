"""Minimal stand-ins for the nerfstudio 0.3.1 ray containers the path touches (nerfstudio is a third-party
dependency of the reference and is not installed here): ``Frustums``, ``RaySamples``, ``RayBundle``, ``SceneBox``
(nerfstudio.cameras.rays / nerfstudio.data.scene_box).  Only the fields and methods the hot
path uses are provided; names and semantics follow nerfstudio so the field/sampler/model mirrors read like the
reference."""
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
from torch import Tensor


@dataclass
class Frustums:
    origins: Tensor            # [..., 3]
    directions: Tensor         # [..., 3]
    starts: Tensor             # [..., 1]
    ends: Tensor               # [..., 1]
    pixel_area: Tensor         # [..., 1]
    offsets: Optional[Tensor] = None
    base_positions: Optional[Tensor] = None     # cache: origins + directions * (starts + ends) / 2, without offsets

    def get_positions(self) -> Tensor:
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos

    def set_offsets(self, offsets: Tensor) -> None:
        self.offsets = offsets


@dataclass
class RaySamples:
    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None

    def __len__(self) -> int:
        return self.frustums.origins.shape[0]


@dataclass
class RayBundle:
    origins: Tensor
    directions: Tensor
    pixel_area: Optional[Tensor] = None
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = field(default_factory=dict)
    times: Optional[Tensor] = None

    def __len__(self) -> int:
        return self.origins.numel() // self.origins.shape[-1]

    @property
    def shape(self):
        return tuple(self.origins.shape[:-1])

    def _map(self, fn) -> "RayBundle":
        def ap(t):
            return fn(t) if isinstance(t, Tensor) else t
        # metadata keys starting with "_" are per-batch (not per-ray) entries and are passed through untouched
        md = {k: (v if k.startswith("_") else ap(v)) for k, v in self.metadata.items()}
        return RayBundle(ap(self.origins), ap(self.directions), ap(self.pixel_area), ap(self.camera_indices),
                         ap(self.nears), ap(self.fars), md, ap(self.times))

    def flatten(self) -> "RayBundle":
        """Image-shaped bundle [H, W, ...] -> [H*W, ...] (row-major), nerfstudio's ``RayBundle.flatten``."""
        lead = len(self.shape)
        return self._map(lambda t: t.reshape(-1, *t.shape[lead:]))

    def get_row_major_sliced_ray_bundle(self, start_idx: int, end_idx: int) -> "RayBundle":
        """Rays ``start_idx:end_idx`` of the flattened bundle (nerfstudio API used by
        ``Model.get_outputs_for_camera_ray_bundle``)."""
        return self.flatten()[start_idx:end_idx]

    def to(self, device) -> "RayBundle":
        def mv(t):
            return t.to(device) if isinstance(t, Tensor) else t
        return RayBundle(mv(self.origins), mv(self.directions), mv(self.pixel_area), mv(self.camera_indices),
                         mv(self.nears), mv(self.fars), {k: mv(v) for k, v in self.metadata.items()}, mv(self.times))

    def __getitem__(self, idx) -> "RayBundle":
        def sl(t):
            return t[idx] if isinstance(t, Tensor) else t
        # metadata keys starting with "_" are per-batch (not per-ray) entries and are passed through unsliced
        md = {k: (v if k.startswith("_") else sl(v)) for k, v in self.metadata.items()}
        return RayBundle(sl(self.origins), sl(self.directions), sl(self.pixel_area), sl(self.camera_indices),
                         sl(self.nears), sl(self.fars), md, sl(self.times))


@dataclass
class SceneBox:
    aabb: Tensor   # [2, 3]

    @staticmethod
    def get_normalized_positions(positions: Tensor, aabb: Tensor) -> Tensor:
        aabb_lengths = aabb[1] - aabb[0]
        return (positions - aabb[0]) / aabb_lengths
