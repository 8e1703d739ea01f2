"""Camera view frusta for the 128^3 culling grid of the volumetric sampler.

Mirror of the reference's ``model_components/frustum.py`` (``TorchFrustum(cam_to_world, intrinsics, image_dimensions)``
with ``contains`` / ``contains_points``, :104-145) as the dataparser builds it (``nersemble_dataparser.py:249-256``: one
frustum per camera, pose in the OpenCV convention -- x right, y down, z forward) and as the sampler consumes it
(``nersemble_volumetric_sampler.py:28-41``).

A frustum here is the intersection of the FOUR side half-spaces through the camera centre (no near / far plane, like
the reference).  Construction: the image corners (0,0), (W,0), (0,H), (W,H) at depth 1 are un-projected with K^-1 and
moved to world space; the edge vectors from the camera centre to neighbouring corners, crossed in the cyclic order
top -> right -> bottom -> left, give inward normals (right-handed frame); a point is inside when
``n_i . (x - c) >= 0`` for all four unit normals.  Everything is evaluated in the dtype of the pose (the dataparser hands
over float64), with points promoted the way torch promotes ``float32 - float64``.

Pinned by ``tests/test_glue_cpu.py::test_frustum_matches_reference`` against the reference's numpy ``Frustum``
(fixtures in ``tests/golden/frustum.npz``).  Device-agnostic: the reference moves its planes to ``.cuda()`` at
construction; here they follow the points.
"""
from typing import Sequence, Tuple

import torch


class TorchFrustum:
    def __init__(self, cam_to_world: torch.Tensor, intrinsics: torch.Tensor, image_dimensions: Tuple[int, int]):
        cam_to_world = torch.as_tensor(cam_to_world)
        dt = cam_to_world.dtype if cam_to_world.dtype.is_floating_point else torch.float64
        pose = cam_to_world.to(dt)
        k_inv = torch.linalg.inv(torch.as_tensor(intrinsics).to(dt))
        img_w, img_h = image_dimensions
        centre = pose[:3, 3]
        # pixel corners (homogeneous, depth 1): top-left, top-right, bottom-right, bottom-left (cyclic)
        px = torch.tensor([[0, 0, 1], [img_w, 0, 1], [img_w, img_h, 1], [0, img_h, 1]], dtype=dt)
        corners_cam = px @ k_inv.T                                     # [4,3]
        corners_world = corners_cam @ pose[:3, :3].T + centre[None]
        edges = corners_world - centre[None]                           # tl, tr, br, bl
        # top = tl x tr, right = tr x br, bottom = br x bl, left = bl x tl
        normals = torch.linalg.cross(edges, edges.roll(-1, dims=0))
        self.normals = normals / normals.norm(dim=1, keepdim=True)     # [4,3] inward, unit
        self.centre = centre.clone()                                   # every plane passes through the camera centre

    @staticmethod
    def from_camera(cam_to_world_gl: torch.Tensor, fx: float, fy: float, cx: float, cy: float, width: int,
                    height: int) -> "TorchFrustum":
        """Convenience for rigs that keep OpenGL poses (camera looks along -z, +y up), e.g. ``data/synthetic.py``:
        flips the y / z camera axes to the OpenCV convention the reference's dataparser converts to (:253)."""
        pose = torch.as_tensor(cam_to_world_gl).double().clone()
        pose[:3, 1:3] = -pose[:3, 1:3]
        k = torch.tensor([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]], dtype=torch.float64)
        return TorchFrustum(pose, k, (width, height))

    def to(self, device) -> "TorchFrustum":
        out = object.__new__(TorchFrustum)
        out.normals, out.centre = self.normals.to(device), self.centre.to(device)
        return out

    def signed_distances(self, points: torch.Tensor) -> torch.Tensor:
        """[B,4] distances to the four side planes (>= 0 inside)."""
        n, c = self.normals.to(points.device), self.centre.to(points.device)
        diff = points[:, None, :] - c[None, None, :]                   # promotes to the plane dtype
        return (diff * n[None]).sum(-1)

    def contains_points(self, points: torch.Tensor) -> torch.Tensor:
        return (self.signed_distances(points) >= 0).all(dim=1)

    def contains(self, point: torch.Tensor) -> bool:
        return bool(self.contains_points(torch.as_tensor(point).reshape(1, 3)).item())


def visibility_grid(frustums: Sequence[TorchFrustum], scene_aabb: torch.Tensor, resolution, min_views: int,
                    device) -> torch.Tensor:
    """bool [rx,ry,rz]: lattice points (``linspace`` over the box, end points included) seen by >= ``min_views``
    frusta (nersemble_volumetric_sampler.py:28-41)."""
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
