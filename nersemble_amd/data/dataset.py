"""Image dataset with an in-memory cache -- mirror of the reference's ``InMemoryInputDataset`` / ``NeRSembleInputDataset``
(``dataset/nersemble_dataset.py:13-58,113-128``) on nerfstudio's ``InputDataset``: ``dataset[i]`` is a dict with
``image_idx``, ``image [H, W, 3]`` float in [0, 1] and the per-image extras the model trains with (``alpha_map
[H, W, 1] uint8``, ``depth_map [H, W]``, ``timesteps``, ``cam_ids``); ``dataset.cameras`` holds one camera per image,
``dataset.metadata["camera_frustums"]`` the training frusta (``nersemble_datamanager.py:50-51``).

Items are produced by a ``loader(image_idx) -> dict`` (the reference decodes PNGs and depth / alpha files there: dataset
I/O is out of scope, SURVEY.md 2 -- ``data/synthetic.py`` provides an analytic loader) and cached on first use, up to
``max_cached_items`` (-1 = no limit), optionally with the image held as uint8 (``use_cache_compression``: lossy, exactly
the reference's ``(image * 255).round()`` / ``/ 255``).
"""
from typing import Callable, Dict, Optional

import torch

from ..cameras import Cameras


class InMemoryInputDataset:
    def __init__(self, loader: Callable[[int], Dict], cameras: Cameras, max_cached_items: int = -1,
                 use_cache_compression: bool = False, metadata: Optional[Dict] = None):
        self._loader = loader
        self.cameras = cameras
        self.metadata = dict(metadata or {})
        self._cached_items: Dict[int, Dict] = {}
        self._max_cached_items = max_cached_items
        self._use_cache_compression = use_cache_compression

    def __len__(self) -> int:
        return self.cameras.size

    def _compress(self, item: Dict) -> Dict:
        if not self._use_cache_compression:
            return item
        item = dict(item)
        item["image"] = (item["image"] * 255).round().to(torch.uint8)
        return item

    def _uncompress(self, item: Dict) -> Dict:
        if not self._use_cache_compression:
            return item
        item = dict(item)
        item["image"] = item["image"].float() / 255.0
        return item

    def __getitem__(self, image_idx: int) -> Dict:
        image_idx = int(image_idx)
        if image_idx in self._cached_items:
            return self._uncompress(self._cached_items[image_idx])
        item = self._loader(image_idx)
        item.setdefault("image_idx", image_idx)
        if self._max_cached_items == -1 or len(self._cached_items) < self._max_cached_items:
            self._cached_items[image_idx] = self._compress(item)
        return item


NeRSembleInputDataset = InMemoryInputDataset
