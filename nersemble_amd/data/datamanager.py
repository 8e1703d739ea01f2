"""``NeRSembleVanillaDataManager`` -- the component in front of the hot path.

Mirror of the reference's ``datamanager/nersemble_datamanager.py:14-118`` on the slice of nerfstudio 0.3.1's
``VanillaDataManager`` it relies on (UPSTREAM, restated: ``CacheDataloader`` = draw ``num_images_to_sample_from`` images,
reuse them ``num_times_to_repeat_images`` times; ``next_train`` = pixel sampler -> ``RayGenerator``;
``FixedIndicesEvalDataloader`` = every evaluation image as a full camera ray bundle): same method names, the same
``(RayBundle, batch)`` results with the per-image attributes ``depth_map`` / ``timesteps`` / ``cam_ids`` copied per ray
into ``ray_bundle.metadata`` (``[R, 1]``), configuration values of ``train_nersemble.py:172-179``.

Device-first differences: the image batch, the sampled indices and every gather stay on the device the dataset's
images live on (the reference samples on the host, ``c.cpu()``, and moves 4096 rays per step); the ray bundle also
carries ``image_index`` (ray -> image of the cached batch) and ``_image_timesteps`` so that the model indexes its
<= 24 time codes per step instead of gathering ``[S, H]`` rows (models/nersemble_instant_ngp.py).
"""
from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Tuple

import torch
from torch import Tensor

from ..cameras import RayGenerator
from ..rays import RayBundle
from .dataset import InMemoryInputDataset
from .pixel_sampler import ADDITIONAL_METADATA, NeRSemblePixelSampler, add_metadata_to_ray_bundle


def _collate(items: List[Dict], device) -> Dict[str, Tensor]:
    batch = {}
    for key in items[0]:
        values = [torch.as_tensor(item[key]) for item in items]
        batch[key] = torch.stack(values).to(device)
    return batch


class CacheDataloader:
    """Endless iterator over image batches: ``num_images_to_sample_from`` random images of the dataset (all of them if
    -1 or not fewer than the dataset), collated to ``[N, H, W, ...]`` tensors on ``device`` and handed out
    ``num_times_to_repeat_images`` times before the next draw."""

    def __init__(self, dataset: InMemoryInputDataset, num_images_to_sample_from: int = -1,
                 num_times_to_repeat_images: int = -1, device="cpu", generator: Optional[torch.Generator] = None):
        self.dataset, self.device, self.generator = dataset, device, generator
        n = len(dataset)
        self.cache_all_images = num_images_to_sample_from == -1 or num_images_to_sample_from >= n
        self.num_images_to_sample_from = n if self.cache_all_images else num_images_to_sample_from
        self.num_times_to_repeat_images = num_times_to_repeat_images
        self.num_repeated = num_times_to_repeat_images          # forces a draw on the first request
        self.cached_collated_batch: Optional[Dict[str, Tensor]] = None

    def _draw(self) -> Dict[str, Tensor]:
        n = len(self.dataset)
        if self.cache_all_images:
            chosen = list(range(n))
        else:
            chosen = torch.randperm(n, generator=self.generator)[:self.num_images_to_sample_from].tolist()
        return _collate([self.dataset[i] for i in chosen], self.device)

    def __iter__(self) -> Iterator[Dict[str, Tensor]]:
        while True:
            if self.cache_all_images and self.cached_collated_batch is not None:
                pass
            elif self.cached_collated_batch is None or (self.num_times_to_repeat_images != -1
                                                        and self.num_repeated >= self.num_times_to_repeat_images):
                self.num_repeated = 0
                self.cached_collated_batch = self._draw()
            self.num_repeated += 1
            yield self.cached_collated_batch


class FixedIndicesEvalDataloader:
    """Every image of the dataset (or ``image_indices``) as ``(camera_ray_bundle [H, W], batch)``."""

    def __init__(self, dataset: InMemoryInputDataset, image_indices: Optional[List[int]] = None, device="cpu"):
        self.dataset, self.device = dataset, device
        self.cameras = dataset.cameras.to(device)
        self.image_indices = list(range(len(dataset))) if image_indices is None else list(image_indices)

    def __len__(self) -> int:
        return len(self.image_indices)

    def get_data_from_image_idx(self, image_idx: int) -> Tuple[RayBundle, Dict]:
        bundle = self.cameras.generate_rays(camera_indices=image_idx)
        # tensors move to the device; per-image scalars (timesteps, cam_ids, image_idx) stay Python numbers so that
        # ``_add_metadata_to_ray_bundle`` broadcasts them over the image (nersemble_datamanager.py:71-73)
        batch = {k: (v.to(self.device) if isinstance(v, Tensor) else v) for k, v in self.dataset[image_idx].items()}
        return bundle, batch

    def __iter__(self):
        for i in self.image_indices:
            yield self.get_data_from_image_idx(i)


@dataclass
class NeRSembleVanillaDataManagerConfig:
    train_num_rays_per_batch: int = 4096                 # train_nersemble.py:101,172
    eval_num_rays_per_batch: int = 1024                  # :173
    train_num_images_to_sample_from: int = 24            # :174
    train_num_times_to_repeat_images: int = 20           # :175
    eval_num_images_to_sample_from: int = 36             # :176
    eval_num_times_to_repeat_images: int = -1
    patch_size: int = 1
    max_cached_items: int = -1                           # nersemble_datamanager.py:22-23 (held by the datasets)
    use_cache_compression: bool = False


class NeRSembleVanillaDataManager:
    config: NeRSembleVanillaDataManagerConfig

    def __init__(self, config: NeRSembleVanillaDataManagerConfig, train_dataset: InMemoryInputDataset,
                 eval_dataset: Optional[InMemoryInputDataset] = None, device="cpu",
                 generator: Optional[torch.Generator] = None):
        self.config, self.device = config, torch.device(device)
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.train_count = self.eval_count = 0
        self.setup_train(generator)
        if eval_dataset is not None:
            self.setup_eval(generator)

    # ---- set-up (VanillaDataManager.setup_train / setup_eval) ------------------------------------------------------
    def _get_pixel_sampler(self, dataset: InMemoryInputDataset, *args, **kwargs) -> NeRSemblePixelSampler:
        """Per-image metadata (timestep, camera id, depth map) needs the NeRSemble sampler (:59-66)."""
        if self.config.patch_size > 1:
            raise NotImplementedError()
        return NeRSemblePixelSampler(*args, additional_metadata=ADDITIONAL_METADATA, **kwargs)

    def setup_train(self, generator=None) -> None:
        cfg = self.config
        self.train_image_dataloader = CacheDataloader(self.train_dataset, cfg.train_num_images_to_sample_from,
                                                      cfg.train_num_times_to_repeat_images, self.device, generator)
        self.iter_train_image_dataloader = iter(self.train_image_dataloader)
        self.train_pixel_sampler = self._get_pixel_sampler(self.train_dataset, cfg.train_num_rays_per_batch)
        self.train_ray_generator = RayGenerator(self.train_dataset.cameras.to(self.device))
        # for logging train images during evaluation as well (:31-39)
        self.train_dataloader = FixedIndicesEvalDataloader(self.train_dataset, device=self.device)

    def setup_eval(self, generator=None) -> None:
        cfg = self.config
        self.eval_image_dataloader = CacheDataloader(self.eval_dataset, cfg.eval_num_images_to_sample_from,
                                                     cfg.eval_num_times_to_repeat_images, self.device, generator)
        self.iter_eval_image_dataloader = iter(self.eval_image_dataloader)
        self.eval_pixel_sampler = self._get_pixel_sampler(self.eval_dataset, cfg.eval_num_rays_per_batch)
        self.eval_ray_generator = RayGenerator(self.eval_dataset.cameras.to(self.device))
        self.fixed_indices_eval_dataloader = FixedIndicesEvalDataloader(self.eval_dataset, device=self.device)
        self.eval_dataloader = self.fixed_indices_eval_dataloader
        self._eval_image_cursor = 0

    # ---- batches ---------------------------------------------------------------------------------------------------
    def _add_metadata_to_ray_bundle(self, ray_bundle: RayBundle, batch: Dict) -> None:
        add_metadata_to_ray_bundle(ray_bundle, batch)

    def _rays_from(self, image_batch: Dict, sampler: NeRSemblePixelSampler, generator: RayGenerator):
        batch = sampler.sample(image_batch)
        ray_bundle = generator(batch["indices"])
        self._add_metadata_to_ray_bundle(ray_bundle, batch)
        # ray -> position of its image in the cached batch, and the batch's per-image timesteps (code slots)
        local = getattr(sampler, "last_local_image_index", None)
        if local is not None and local.shape[0] != batch["indices"].shape[0]:
            local = None
        if "timesteps" in image_batch:
            if local is not None:
                ray_bundle.metadata["image_index"] = local.to(torch.int32)[:, None]
            else:
                image_idx = image_batch["image_idx"].reshape(-1)
                lookup = torch.full((int(image_idx.max()) + 1,), -1, dtype=torch.int32, device=image_idx.device)
                lookup[image_idx] = torch.arange(image_idx.numel(), dtype=torch.int32, device=image_idx.device)
                ray_bundle.metadata["image_index"] = lookup[batch["indices"][:, 0]][:, None]
            cached = getattr(self, "_timesteps_i32", None)     # (one conversion per image batch: 20 steps share it)
            if cached is None or cached[0] is not image_batch["timesteps"]:
                cached = self._timesteps_i32 = (image_batch["timesteps"], image_batch["timesteps"].reshape(-1).int())
            ray_bundle.metadata["_image_timesteps"] = cached[1]
        return ray_bundle, batch

    def next_train(self, step: int) -> Tuple[RayBundle, Dict]:
        self.train_count += 1
        return self._rays_from(next(self.iter_train_image_dataloader), self.train_pixel_sampler,
                               self.train_ray_generator)

    def next_eval(self, step: int) -> Tuple[RayBundle, Dict]:
        self.eval_count += 1
        return self._rays_from(next(self.iter_eval_image_dataloader), self.eval_pixel_sampler,
                               self.eval_ray_generator)

    def next_eval_image(self, step: int) -> Tuple[int, RayBundle, Dict]:
        loader = self.fixed_indices_eval_dataloader
        image_idx = loader.image_indices[self._eval_image_cursor % len(loader)]
        self._eval_image_cursor += 1
        camera_ray_bundle, batch = loader.get_data_from_image_idx(image_idx)
        self._add_metadata_to_ray_bundle(camera_ray_bundle, batch)
        return image_idx, camera_ray_bundle, batch

    def next_train_image(self, step: int) -> Tuple[int, RayBundle, Dict]:
        for camera_ray_bundle, batch in self.train_dataloader:
            image_idx = int(camera_ray_bundle.camera_indices[0, 0, 0])
            self._add_metadata_to_ray_bundle(camera_ray_bundle, batch)
            return image_idx, camera_ray_bundle, batch
        raise ValueError("No more train images")

    def get_train_rays_per_batch(self) -> int:
        return self.config.train_num_rays_per_batch

    def get_eval_rays_per_batch(self) -> int:
        return self.config.eval_num_rays_per_batch
