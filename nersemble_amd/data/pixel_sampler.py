"""Pixel sampling + per-image metadata on the device -- SURVEY.md 8(f) rank 2 (the step in front of the hot path).

Mirrors ``nersemble/nerfstudio/data/nersemble_pixel_sampler.py:8-69`` (on nerfstudio's ``PixelSampler``) and
``NeRSembleVanillaDataManager._add_metadata_to_ray_bundle`` (``datamanager/nersemble_datamanager.py:14,68-74``):
from a batch of images pick ``num_rays_per_batch`` random (image, y, x) triples, gather every per-pixel entry at
them, treat the per-image entries (``image_idx`` and ``depth_map`` / ``timesteps`` / ``cam_ids``) per image, and attach
the latter to the ray bundle's metadata.  The reference moves the sampled indices to the host before indexing
(``c.cpu()`` at :48) -- a device->host sync and a host-indexed gather per step; here everything stays where the image
batch lives.  Uniform sampling follows nerfstudio 0.3.1's ``PixelSampler.sample_method``
(``floor(rand(n, 3) * [num_images, H, W])``; restated, nerfstudio is not installed).
"""
from typing import Dict, List, Optional

import torch
from torch import Tensor

ADDITIONAL_METADATA = ["depth_map", "timesteps", "cam_ids"]      # datamanager/nersemble_datamanager.py:14
_SCALE_CACHE = {}


class NeRSemblePixelSampler:
    def __init__(self, num_rays_per_batch: int, keep_full_image: bool = False,
                 additional_metadata: Optional[List[str]] = None) -> None:
        self.num_rays_per_batch = num_rays_per_batch
        self.keep_full_image = keep_full_image
        self._additional_metadata = list(additional_metadata) if additional_metadata is not None else []
        self._per_image_attributes = ["image_idx"] + self._additional_metadata

    def set_num_rays_per_batch(self, num_rays_per_batch: int) -> None:
        self.num_rays_per_batch = num_rays_per_batch

    @staticmethod
    def sample_method(batch_size: int, num_images: int, image_height: int, image_width: int,
                      mask: Optional[Tensor] = None, device="cpu") -> Tensor:
        """[batch_size, 3] long (image, y, x): uniform over the batch, or uniform over the non-zero mask pixels."""
        if isinstance(mask, Tensor):
            nonzero = torch.nonzero(mask[..., 0], as_tuple=False)
            chosen = torch.randint(0, nonzero.shape[0], (batch_size,), device=nonzero.device)
            return nonzero[chosen]
        # (the three extents as a device tensor, made once per image-batch shape: `torch.tensor([...], device=...)` is a
        # pageable host-to-device copy, ordered behind everything queued on the stream -- it blocked the host for the whole
        # backlog of the previous training step, 5.6 ms per `next_train` in the timed loop)
        key = (int(num_images), int(image_height), int(image_width), str(device))
        scale = _SCALE_CACHE.get(key)
        if scale is None:
            if len(_SCALE_CACHE) > 16:
                _SCALE_CACHE.clear()
            scale = _SCALE_CACHE[key] = torch.tensor([num_images, image_height, image_width], device=device)
        return torch.floor(torch.rand((batch_size, 3), device=device) * scale).long()

    def collate_image_dataset_batch(self, batch: Dict, num_rays_per_batch: int, keep_full_image: bool = False) -> Dict:
        device = batch["image"].device
        num_images, image_height, image_width, _ = batch["image"].shape
        indices = self.sample_method(num_rays_per_batch, num_images, image_height, image_width,
                                     mask=batch.get("mask"), device=device)
        return self.collate_at(batch, indices, keep_full_image)

    def collate_at(self, batch: Dict, indices: Tensor, keep_full_image: bool = False) -> Dict:
        """The gather half of ``collate_image_dataset_batch`` for given batch-local (image, y, x) triples
        (nersemble_pixel_sampler.py:44-66)."""
        num_rays_per_batch = indices.shape[0]
        c, y, x = indices[:, 0], indices[:, 1], indices[:, 2]            # stay on the device (reference: .cpu())
        collated = {key: value[c, y, x] for key, value in batch.items()
                    if key not in self._per_image_attributes and value is not None}
        assert collated["image"].shape[0] == num_rays_per_batch
        for key in self._additional_metadata:
            if key in batch:
                collated[key] = batch[key][c]
        absolute = indices.clone()
        absolute[:, 0] = batch["image_idx"][c]          # batch-local image number -> dataset image index
        collated["indices"] = absolute
        # (native extension, outside the returned dict -- its keys are the reference's) the batch-local image number
        # itself: the model's code slot of the ray; the datamanager puts it into the ray bundle's metadata instead of
        # inverting ``image_idx`` through a lookup table every step
        self.last_local_image_index = c
        if keep_full_image:
            collated["full_image"] = batch["image"]
        return collated

    def sample(self, image_batch: Dict) -> Dict:
        return self.collate_image_dataset_batch(image_batch, self.num_rays_per_batch, self.keep_full_image)


def add_metadata_to_ray_bundle(ray_bundle, batch: Dict) -> None:
    """``_add_metadata_to_ray_bundle`` (datamanager/nersemble_datamanager.py:68-74): per-ray copies of the per-image
    attributes, shape [..., 1], in ``ray_bundle.metadata``."""
    if ray_bundle.metadata is None:
        ray_bundle.metadata = {}
    for key in ADDITIONAL_METADATA:
        if key in batch:
            value = batch[key]
            if not isinstance(value, Tensor):
                value = torch.tensor(value).repeat(*ray_bundle.origins.shape[:-1]).to(ray_bundle.origins.device)
            ray_bundle.metadata[key] = value.unsqueeze(-1)
