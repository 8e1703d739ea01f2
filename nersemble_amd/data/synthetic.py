"""Synthetic stand-in for the (gated, absent) NeRSemble dataset, shaped like the reference's datamanager
contract (nersemble_datamanager.py:68-81, nersemble_pixel_sampler.py:51-64): ``next_train(step)`` returns a
``RayBundle`` (origins, directions, camera_indices, times, metadata timesteps/cam_ids) and a batch dict with
``image [R,3]``, ``alpha_map [R,1] uint8``, ``depth_maps [R]``.

Rig: 16 pinhole cameras on a ~100 degree frontal arc at radius 9 scene units looking at the box
centre, 1100x1604 images, 12 train / 4 eval; per batch 24 (camera, timestep) images are drawn and 4096 pixels
sampled uniformly.  Ground truth is analytic: an ellipsoid "head" (semi-axes 0.9, 1.1, 1.2) with a smooth
time-varying rotation and procedural albedo on a white background, so PSNR is meaningful without data.
Everything is generated on the device with a seeded torch.Generator (seed 19980801, train_nersemble.py:116).
"""
import math
from typing import Dict, Tuple

import torch

from ..engine.parallel import shard_seed
from ..model_components.frustum import TorchFrustum
from ..rays import RayBundle


def _look_at(eye: torch.Tensor, target: torch.Tensor, up=(0.0, 0.0, 1.0)) -> torch.Tensor:
    f = target - eye
    f = f / f.norm()
    upv = torch.tensor(up)
    s = torch.linalg.cross(f, upv)
    s = s / s.norm()
    u = torch.linalg.cross(s, f)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = s, u, -f, eye        # OpenGL: camera looks along -z
    return c2w


class SyntheticNeRSembleData:
    def __init__(self, scene_box: torch.Tensor, n_timesteps: int, n_rays: int = 4096, n_images_per_batch: int = 24,
                 device="cuda", seed: int = 19980801, rank: int = 0, width: int = 1100, height: int = 1604,
                 n_cameras: int = 16, radius: float = 9.0, focal: float = 2200.0):
        self.device = torch.device(device)
        self.scene_box = scene_box.float()
        self.n_timesteps, self.n_rays, self.n_images = n_timesteps, n_rays, n_images_per_batch
        self.width, self.height, self.focal = width, height, focal
        self.gen = torch.Generator(device=self.device).manual_seed(shard_seed(seed, rank))
        center = (self.scene_box[0] + self.scene_box[1]) / 2
        angles = torch.linspace(-50.0, 50.0, n_cameras) * math.pi / 180.0
        elev = torch.tensor([0.15, -0.15]).repeat(n_cameras // 2 + 1)[:n_cameras]
        # the head faces -y: cameras on an arc in front of it
        eyes = torch.stack([center[0] + radius * torch.sin(angles) * torch.cos(elev),
                            center[1] - radius * torch.cos(angles) * torch.cos(elev),
                            center[2] + radius * torch.sin(elev)], dim=1)
        self.c2w = torch.stack([_look_at(e, center) for e in eyes]).to(self.device)      # [16,4,4]
        self.train_cams = torch.tensor([c for c in range(n_cameras) if c % 4 != 3], device=self.device)  # 12 train
        self.eval_cams = torch.tensor([c for c in range(n_cameras) if c % 4 == 3], device=self.device)   # 4 eval
        self.center = center.to(self.device)
        self.semi_axes = torch.tensor([0.9, 1.1, 1.2], device=self.device)
        self.camera_frustums = [TorchFrustum.from_camera(self.c2w[c].cpu(), focal, focal, width / 2, height / 2, width,
                                                         height) for c in range(n_cameras)]

    # ---- analytic scene ---------------------------------------------------------------------------
    def _rotation(self, times: torch.Tensor) -> torch.Tensor:
        """Smooth time-varying rotation about z (head shake) + small nod about x. times [N] in [0,1]."""
        a = 0.35 * torch.sin(2 * math.pi * times)
        b = 0.15 * torch.sin(4 * math.pi * times + 0.5)
        ca, sa, cb, sb = torch.cos(a), torch.sin(a), torch.cos(b), torch.sin(b)
        zero, one = torch.zeros_like(a), torch.ones_like(a)
        Rz = torch.stack([ca, -sa, zero, sa, ca, zero, zero, zero, one], -1).reshape(-1, 3, 3)
        Rx = torch.stack([one, zero, zero, zero, cb, -sb, zero, sb, cb], -1).reshape(-1, 3, 3)
        return Rz @ Rx

    def render_ground_truth(self, origins, directions, times) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Closed-form ray / rotating-ellipsoid intersection -> (rgb [N,3], alpha uint8 [N,1], depth [N])."""
        R = self._rotation(times.reshape(-1))
        o = torch.einsum("nij,nj->ni", R.transpose(1, 2), origins - self.center) / self.semi_axes
        d = torch.einsum("nij,nj->ni", R.transpose(1, 2), directions) / self.semi_axes
        A = (d * d).sum(-1)
        Bq = 2 * (o * d).sum(-1)
        Cq = (o * o).sum(-1) - 1.0
        disc = Bq * Bq - 4 * A * Cq
        hit = disc > 0
        t = (-Bq - torch.sqrt(disc.clamp(min=0))) / (2 * A)
        hit = hit & (t > 0)
        p = o + d * t[:, None]                                    # unit-sphere surface point (canonical)
        albedo = 0.5 + 0.5 * torch.stack([torch.sin(5.0 * p[:, 0] + 1.0), torch.sin(7.0 * p[:, 1] + 2.0),
                                          torch.sin(3.0 * p[:, 2] * p[:, 0] + 0.3)], -1)
        shade = (0.6 + 0.4 * p[:, 2].clamp(-1, 1))[:, None]
        rgb = torch.where(hit[:, None], (albedo * shade).clamp(0, 1), torch.ones_like(albedo))
        alpha = (hit.to(torch.uint8) * 255)[:, None]
        depth = torch.where(hit, t, torch.zeros_like(t))
        return rgb, alpha, depth

    # ---- rays -------------------------------------------------------------------------------------
    def _rays_for(self, cam_ids, ys, xs):
        c2w = self.c2w[cam_ids]
        dirs_c = torch.stack([(xs + 0.5 - self.width / 2) / self.focal, -(ys + 0.5 - self.height / 2) / self.focal,
                              -torch.ones_like(xs)], -1)
        dirs_w = torch.einsum("nij,nj->ni", c2w[:, :3, :3], dirs_c)
        dirs_w = dirs_w / dirs_w.norm(dim=-1, keepdim=True)
        return c2w[:, :3, 3].contiguous(), dirs_w.contiguous()

    def eval_cameras(self, timesteps, downscale: int = 16):
        """``Cameras`` of the 4 held-out views x the given timesteps (what nerfstudio's eval dataloader iterates)."""
        from ..cameras import Cameras
        cams = self.eval_cams.repeat(len(timesteps))
        times = torch.tensor([t / max(self.n_timesteps - 1, 1) for t in timesteps for _ in self.eval_cams])
        cameras = Cameras(self.c2w[cams].cpu(), self.focal, self.focal, self.width / 2, self.height / 2, self.width,
                          self.height, times=times)
        cameras.rescale_output_resolution(1.0 / downscale)
        return cameras

    def image_dataset(self, images, downscale: int = 16, max_cached_items: int = -1,
                      use_cache_compression: bool = False):
        """``InMemoryInputDataset`` over the given (camera, timestep) images of the rig, rendered analytically on first
        use: the stand-in for the reference's ``NeRSembleInputDataset`` (whose loader decodes the gated dataset's files).
        Item keys as there: ``image``, ``alpha_map`` (uint8), ``depth_maps`` (per pixel), ``timesteps`` / ``cam_ids``
        (per image), ``image_idx``; one camera per image."""
        from ..cameras import Cameras
        from .dataset import InMemoryInputDataset
        images = [(int(c), int(t)) for c, t in images]
        cams = torch.tensor([c for c, _ in images], device=self.c2w.device)
        times = torch.tensor([t / max(self.n_timesteps - 1, 1) for _, t in images])
        cameras = Cameras(self.c2w[cams].cpu(), self.focal, self.focal, self.width / 2, self.height / 2, self.width,
                          self.height, times=times)
        cameras.rescale_output_resolution(1.0 / downscale)
        on_device = cameras.to(self.device)

        def load(image_idx: int):
            bundle = on_device.generate_rays(image_idx)
            h, w = bundle.shape
            flat = bundle.flatten()
            rgb, alpha, depth = self.render_ground_truth(flat.origins, flat.directions, flat.times.reshape(-1))
            cam, timestep = images[image_idx]
            return {"image_idx": image_idx, "image": rgb.view(h, w, 3), "alpha_map": alpha.view(h, w, 1),
                    "depth_maps": depth.view(h, w), "timesteps": timestep, "cam_ids": cam}

        return InMemoryInputDataset(load, cameras, max_cached_items=max_cached_items,
                                    use_cache_compression=use_cache_compression,
                                    metadata={"camera_frustums": self.camera_frustums})

    def eval_views(self, timesteps, downscale: int = 16):
        """(camera_ray_bundle [H, W], batch) per held-out view, like ``fixed_indices_eval_dataloader``: ``batch`` has
        ``image [H,W,3]``, ``alpha_map [H,W,1] uint8``, ``depth_maps [H,W]`` and ``cam_ids`` = index of the eval cam."""
        cameras = self.eval_cameras(timesteps, downscale).to(self.device)
        for i in range(cameras.size):
            bundle = cameras.generate_rays(i)
            h, w = bundle.shape
            flat = bundle.flatten()
            rgb, alpha, depth = self.render_ground_truth(flat.origins, flat.directions, flat.times.reshape(-1))
            yield bundle, {"image": rgb.view(h, w, 3), "alpha_map": alpha.view(h, w, 1), "depth_maps": depth.view(h, w),
                           "cam_ids": torch.tensor(i % len(self.eval_cams))}

    def next_train(self, step: int) -> Tuple[RayBundle, Dict[str, torch.Tensor]]:
        g, dev = self.gen, self.device
        img_cams = self.train_cams[torch.randint(0, len(self.train_cams), (self.n_images,), device=dev, generator=g)]
        img_ts = torch.randint(0, self.n_timesteps, (self.n_images,), device=dev, generator=g)
        which = torch.randint(0, self.n_images, (self.n_rays,), device=dev, generator=g)
        which, _ = torch.sort(which)
        ys = torch.randint(0, self.height, (self.n_rays,), device=dev, generator=g).float()
        xs = torch.randint(0, self.width, (self.n_rays,), device=dev, generator=g).float()
        cam_ids, timesteps = img_cams[which], img_ts[which]
        times = timesteps.float() / max(self.n_timesteps - 1, 1)
        origins, directions = self._rays_for(cam_ids, ys, xs)
        rgb, alpha, depth = self.render_ground_truth(origins, directions, times)
        bundle = RayBundle(origins=origins, directions=directions, pixel_area=torch.ones_like(origins[:, :1]),
                           camera_indices=cam_ids[:, None], times=times[:, None],
                           metadata={"timesteps": timesteps[:, None].int(), "cam_ids": cam_ids[:, None],
                                     # which of the batch's images each ray comes from + the images' timesteps: the
                                     # pixel sampler already has both (nersemble_pixel_sampler.py:51-64 gathers
                                     # per-image attributes by the image index c); the model uses them as code slots
                                     "image_index": which[:, None].int(), "_image_timesteps": img_ts.int()})
        batch = {"image": rgb, "alpha_map": alpha, "depth_maps": depth,
                 "indices": torch.stack([which, ys.long(), xs.long()], -1)}
        return bundle, batch

    def eval_image_rays(self, cam: int, timestep: int, downscale: int = 16):
        h, w = self.height // downscale, self.width // downscale
        ys, xs = torch.meshgrid(torch.arange(h, device=self.device).float() * downscale,
                                torch.arange(w, device=self.device).float() * downscale, indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        cam_ids = torch.full_like(ys, cam, dtype=torch.long)
        times = torch.full_like(ys, timestep / max(self.n_timesteps - 1, 1))
        origins, directions = self._rays_for(cam_ids, ys, xs)
        rgb, alpha, depth = self.render_ground_truth(origins, directions, times)
        bundle = RayBundle(origins=origins, directions=directions, pixel_area=torch.ones_like(origins[:, :1]),
                           camera_indices=cam_ids[:, None], times=times[:, None])
        return bundle, {"image": rgb, "alpha_map": alpha, "depth_maps": depth}, (h, w)
