"""Scalar-to-colour maps for the logged / rendered evaluation images (``get_image_metrics_and_images``,
``nersemble_instant_ngp.py:430-440,469``; ``util/render.py:47-54``).  Restatement of the slice of nerfstudio 0.3.1's
``nerfstudio.utils.colormaps`` those call sites use (``ColormapOptions``, ``apply_colormap``, ``apply_depth_colormap``)
plus a stand-in for dreifus' ``apply_scene_flow_colormap``: both are third-party packages that are not installed ->
PARITY UNPINNED; visualisation only, no metric depends on them.  Colour tables come from matplotlib (as in nerfstudio)."""
from dataclasses import dataclass
from functools import lru_cache
from typing import Optional

import torch
from torch import Tensor


@dataclass(frozen=True)
class ColormapOptions:
    colormap: str = "default"
    normalize: bool = False
    colormap_min: float = 0
    colormap_max: float = 1
    invert: bool = False


@lru_cache(maxsize=None)
def _table(name: str) -> Tensor:
    import matplotlib
    return torch.tensor(matplotlib.colormaps[name].colors, dtype=torch.float32)      # [256, 3]


def apply_float_colormap(image: Tensor, colormap: str = "viridis") -> Tensor:
    """[..., 1] in [0, 1] -> [..., 3] through a 256-entry table ("default" = turbo, "gray" = replicate)."""
    colormap = "turbo" if colormap == "default" else colormap
    image = torch.nan_to_num(image, 0)
    if colormap == "gray":
        return image.repeat_interleave(3, dim=-1)
    index = (image * 255).long()
    if int(index.min()) < 0 or int(index.max()) > 255:
        raise ValueError("colormap input outside [0, 1]")
    return _table(colormap).to(image.device)[index[..., 0]]


def apply_colormap(image: Tensor, colormap_options: ColormapOptions = ColormapOptions(), eps: float = 1e-9) -> Tensor:
    if image.shape[-1] == 3:
        return image
    if image.dtype == torch.bool:
        out = torch.ones(image.shape[:-1] + (3,), device=image.device)
        out[~image[..., 0]] = 0.0
        return out
    if image.shape[-1] != 1 or not torch.is_floating_point(image):
        raise NotImplementedError("only 1-channel float, boolean and RGB images are colour-mapped on this path")
    o = colormap_options
    x = image
    if o.normalize:
        x = x - x.min()
        x = x / (x.max() + eps)
    x = (x * (o.colormap_max - o.colormap_min) + o.colormap_min).clip(0, 1)
    if o.invert:
        x = 1 - x
    return apply_float_colormap(x, colormap=o.colormap)


def apply_depth_colormap(depth: Tensor, accumulation: Optional[Tensor] = None, near_plane: Optional[float] = None,
                         far_plane: Optional[float] = None,
                         colormap_options: ColormapOptions = ColormapOptions()) -> Tensor:
    """Depth scaled to [near, far] (default: its own range), colour-mapped, faded to white by 1 - accumulation."""
    near = near_plane or float(depth.min())
    far = far_plane or float(depth.max())
    colored = apply_colormap(((depth - near) / (far - near + 1e-10)).clip(0, 1), colormap_options=colormap_options)
    if accumulation is not None:
        colored = colored * accumulation + (1 - accumulation)
    return colored


def apply_scene_flow_colormap(flow: Tensor) -> Tensor:
    """[..., 3] offsets -> [..., 3] colours: direction as hue-like RGB around mid-grey, magnitude as saturation
    (normalised by the image's largest offset).  Stand-in for ``dreifus.util.colormap.apply_scene_flow_colormap``."""
    scale = flow.norm(dim=-1, keepdim=True).max().clamp(min=1e-12)
    return (0.5 + 0.5 * flow / scale).clip(0, 1)
