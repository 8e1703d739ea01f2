"""Fixed-point codecs of the dataset's depth / normal maps -- the data format on the input side of the path
(``batch["depth_maps"]`` reaches ``get_depth_loss`` / ``get_near_and_empty_loss`` decoded by these).

API mirror of the reference's ``util/quantization.py`` (``Quantizer :33-72``, ``DepthQuantizer :75-87``,
``NormalsQuantizer :90-120``, ``to_spherical / to_cartesian :6-30``): same class names, constructor arguments and
``encode`` / ``decode`` results on numpy arrays, bit for bit (``tests/test_glue_cpu.py::test_quantizers_match_reference``
against ``tests/golden/dataformat.npz``, produced by the reference's own classes).

The code: B bits give 2^B levels; when ``separate_mask`` is set level 0 means "no measurement" and the value range
[min, max] is spread over the remaining 2^B - 2 steps:

    level(v) = round(max(0, v - min) * gain) + first        gain = (2^B - 1 - first) / (max - min), first = 0 or 1
    value(q) = (float32(q) - first) / gain + min

Pixels equal to ``mask_value`` (all channels, for multi-channel maps) are stored as level 0 / restored as
``mask_value``.  Host-side numpy like the reference: it runs once per image in the data loader, not per sample.
"""
from dataclasses import dataclass
from typing import Union

import numpy as np

ArrayOrFloat = Union[np.ndarray, float]


# ---- spherical coordinates of unit normals ---------------------------------------------------------------------------
def to_spherical(cartesian_points: np.ndarray) -> np.ndarray:
    """[..., 3] xyz -> [..., 3] (radius, polar angle measured from +z, azimuth in the xy plane)."""
    px, py, pz = np.moveaxis(cartesian_points, -1, 0)
    planar = np.sqrt(px * px + py * py)
    return np.stack((np.linalg.norm(cartesian_points, axis=-1, ord=2), np.arctan2(planar, pz), np.arctan2(py, px)), -1)


def to_cartesian(spherical_coordinates: np.ndarray) -> np.ndarray:
    rad, polar, azimuth = np.moveaxis(spherical_coordinates, -1, 0)
    ring = np.sin(polar)
    return np.stack((rad * np.cos(azimuth) * ring, rad * np.sin(azimuth) * ring, rad * np.cos(polar)), -1)


def _per_pixel(flags: np.ndarray, combine) -> np.ndarray:
    """Masks of [H, W, C] maps are per pixel (over the channels); [H, W] maps are masked element-wise."""
    return combine(flags, axis=-1) if flags.ndim > 2 else flags


@dataclass
class _LinearCode:
    """The affine value <-> level map shared by all codecs."""
    lowest: ArrayOrFloat
    highest: ArrayOrFloat
    bits: int
    first_level: int                       # 1 when level 0 is reserved for "no measurement"

    @property
    def levels(self) -> int:
        return 1 << self.bits

    @property
    def gain(self) -> ArrayOrFloat:
        return (self.levels - 1 - self.first_level) / (self.highest - self.lowest)

    def to_levels(self, values: np.ndarray, measured: np.ndarray) -> np.ndarray:
        real = np.maximum(0, values - self.lowest) * self.gain + self.first_level
        if real.min() < self.first_level or real.max() >= self.levels:
            raise AssertionError("value outside the quantiser's range")
        real[~measured] = 0
        return real.round().astype(np.uint8 if self.bits == 8 else np.uint16)

    def to_values(self, levels: np.ndarray) -> np.ndarray:
        return (levels.astype(np.float32) - self.first_level) / self.gain + self.lowest


class Quantizer:
    def __init__(self, min_values: ArrayOrFloat, max_values: ArrayOrFloat, bits: int, mask_value: ArrayOrFloat = 0,
                 separate_mask: bool = True):
        self._code = _LinearCode(min_values, max_values, bits, 1 if separate_mask else 0)
        self._mask_value = mask_value
        # the reference's attribute names, for code that reads them
        self._min_values, self._max_values, self._bits, self._separate_mask = min_values, max_values, bits, separate_mask
        self._mask_offset, self._n_buckets, self._scale_factor = self._code.first_level, self._code.levels, self._code.gain

    def encode(self, values: np.ndarray) -> np.ndarray:
        return self._code.to_levels(values, _per_pixel(values != self._mask_value, np.any))

    def decode(self, quantized_values: np.ndarray) -> np.ndarray:
        restored = self._code.to_values(quantized_values)
        restored[_per_pixel(quantized_values == self._mask_value, np.all)] = self._mask_value
        return restored


class DepthQuantizer(Quantizer):
    """16-bit depth in metres over [0, 2]; anything farther is an outlier and is stored as "no measurement"."""

    def __init__(self, min_values: float = 0, max_values: float = 2, bits: int = 16, separate_mask: bool = True):
        super().__init__(min_values, max_values, bits, separate_mask=separate_mask)

    def encode(self, values: np.ndarray) -> np.ndarray:
        np.putmask(values, values > self._max_values, self._mask_value)      # in place, as the reference does
        return super().encode(values)


class NormalsQuantizer(Quantizer):
    """Unit normals as 8-bit spherical coordinates (radius, polar angle in [pi/3, pi], azimuth in [-pi, pi])."""

    def __init__(self):
        super().__init__(np.array([0, 1 / 3 * np.pi, -np.pi]), np.array([1, np.pi, np.pi]), 8)

    def encode(self, values: np.ndarray) -> np.ndarray:
        return super().encode(to_spherical(values))

    def decode(self, quantized_values: np.ndarray) -> np.ndarray:
        angles = super().decode(quantized_values)
        present = _per_pixel(quantized_values != 0, np.any)
        normals = np.zeros_like(angles)
        normals[present] = to_cartesian(angles[present])
        return normals
