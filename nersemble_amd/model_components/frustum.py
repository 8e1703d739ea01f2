"""View-frustum half-space test used for the 128^3 culling grid (reference: model_components/frustum.py and
nersemble_volumetric_sampler.py:28-42).  Device-agnostic restatement: a frustum is the intersection of the
half-spaces {x : n_i . x + d_i >= 0}; ``contains_points`` tests all planes at once."""
from typing import Sequence

import torch


class TorchFrustum:
    def __init__(self, normals: torch.Tensor, offsets: torch.Tensor):
        self.normals = normals.float()        # [P, 3] inward normals
        self.offsets = offsets.float()        # [P]

    @staticmethod
    def from_camera(cam_to_world: torch.Tensor, fx: float, fy: float, cx: float, cy: float, width: int, height: int,
                    near: float, far: float) -> "TorchFrustum":
        """Pinhole frustum in the OpenGL convention (camera looks along -z, +y up)."""
        R, t = cam_to_world[:3, :3].float(), cam_to_world[:3, 3].float()
        # corner ray directions in camera space
        xs = torch.tensor([(0 - cx) / fx, (width - cx) / fx])
        ys = torch.tensor([-(0 - cy) / fy, -(height - cy) / fy])
        tl = torch.tensor([xs[0], ys[0], -1.0]); tr = torch.tensor([xs[1], ys[0], -1.0])
        bl = torch.tensor([xs[0], ys[1], -1.0]); br = torch.tensor([xs[1], ys[1], -1.0])
        # inward normals of the four side planes (through the camera centre), then near / far
        normals_c = torch.stack([torch.linalg.cross(bl, tl), torch.linalg.cross(tr, br),
                                 torch.linalg.cross(tl, tr), torch.linalg.cross(br, bl),
                                 torch.tensor([0.0, 0.0, -1.0]), torch.tensor([0.0, 0.0, 1.0])])
        normals_c = normals_c / normals_c.norm(dim=1, keepdim=True)
        offs_c = torch.tensor([0.0, 0.0, 0.0, 0.0, -near, far])
        normals_w = normals_c @ R.T
        offsets_w = offs_c - (normals_w * t[None]).sum(1)
        return TorchFrustum(normals_w, offsets_w)

    def to(self, device) -> "TorchFrustum":
        return TorchFrustum(self.normals.to(device), self.offsets.to(device))

    def contains_points(self, points: torch.Tensor) -> torch.Tensor:
        d = points @ self.normals.to(points.device).T + self.offsets.to(points.device)[None]
        return (d >= 0).all(dim=1)


def visibility_grid(frustums: Sequence[TorchFrustum], scene_aabb: torch.Tensor, resolution, min_views: int,
                    device) -> torch.Tensor:
    """bool [rx,ry,rz]: lattice points seen by >= min_views frustums (nersemble_volumetric_sampler.py:28-41)."""
    rx, ry, rz = [int(r) for r in resolution]
    gx, gy, gz = torch.meshgrid(torch.linspace(float(scene_aabb[0][0]), float(scene_aabb[1][0]), steps=rx),
                                torch.linspace(float(scene_aabb[0][1]), float(scene_aabb[1][1]), steps=ry),
                                torch.linspace(float(scene_aabb[0][2]), float(scene_aabb[1][2]), steps=rz),
                                indexing="ij")
    pts = torch.stack([gx, gy, gz], dim=-1).view(-1, 3).to(device)
    count = torch.zeros(pts.shape[0], dtype=torch.int32, device=device)
    for f in frustums:
        count += f.contains_points(pts).int()
    return (count >= min_views).view(rx, ry, rz)
