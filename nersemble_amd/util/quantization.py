"""Fixed-point codecs of the dataset's depth / normal maps -- the data format on the input side of the path
(``batch["depth_maps"]`` reaches ``get_depth_loss`` / ``get_near_and_empty_loss`` decoded by these).

Mirror of the reference's ``util/quantization.py`` (``Quantizer :33-72``, ``DepthQuantizer :75-87``,
``NormalsQuantizer :90-120``, ``to_spherical / to_cartesian :6-30``): same class names, constructor arguments and
``encode`` / ``decode`` results on numpy arrays.  Code 0 is reserved for "no measurement" when ``separate_mask`` is
set, so B bits hold 2^B - 2 value steps between ``min_values`` and ``max_values``:

    encode(v) = round(max(0, v - min) * s) + off      (0 where v is the mask value)       s = (2^B - 1 - off)/(max - min)
    decode(q) = (float32(q) - off) / s + min          (mask value where q is the mask value)

Host-side numpy, like the reference -- it runs once per image in the data loader, not per sample.
Pinned by ``tests/test_glue_cpu.py::test_quantizers_match_reference`` (``tests/golden/dataformat.npz``).
"""
from typing import Union

import numpy as np

ArrayOrFloat = Union[np.ndarray, float]


def to_spherical(cartesian_points: np.ndarray) -> np.ndarray:
    """[...,3] xyz -> [...,3] (radius, polar angle from +z, azimuth in the xy plane)."""
    x, y, z = (cartesian_points[..., i] for i in range(3))
    return np.stack([np.linalg.norm(cartesian_points, axis=-1, ord=2), np.arctan2(np.sqrt(x * x + y * y), z),
                     np.arctan2(y, x)], axis=-1)


def to_cartesian(spherical_coordinates: np.ndarray) -> np.ndarray:
    radius, theta, phi = (spherical_coordinates[..., i] for i in range(3))
    ring = np.sin(theta)
    return np.stack([radius * np.cos(phi) * ring, radius * np.sin(phi) * ring, radius * np.cos(theta)], axis=-1)


def _pixel_mask(flags: np.ndarray, reduce) -> np.ndarray:
    """Multi-channel maps ([H,W,C]) are masked per pixel, single-channel ones per element."""
    return reduce(flags, axis=-1) if flags.ndim > 2 else flags


class Quantizer:
    def __init__(self, min_values: ArrayOrFloat, max_values: ArrayOrFloat, bits: int, mask_value: ArrayOrFloat = 0,
                 separate_mask: bool = True):
        self._min_values, self._max_values = min_values, max_values
        self._bits, self._mask_value, self._separate_mask = bits, mask_value, separate_mask
        self._mask_offset = int(bool(separate_mask))
        self._n_buckets = 1 << bits
        self._scale_factor = (self._n_buckets - 1 - self._mask_offset) / (max_values - min_values)

    def encode(self, values: np.ndarray) -> np.ndarray:
        valid = _pixel_mask(values != self._mask_value, np.any)
        codes = np.maximum(0, values - self._min_values) * self._scale_factor + self._mask_offset
        if codes.min() < self._mask_offset or codes.max() >= self._n_buckets:
            raise AssertionError("value outside the quantiser's range")
        codes[~valid] = 0
        return codes.round().astype(np.uint8 if self._bits == 8 else np.uint16)

    def decode(self, quantized_values: np.ndarray) -> np.ndarray:
        empty = _pixel_mask(quantized_values == self._mask_value, np.all)
        values = (quantized_values.astype(np.float32) - self._mask_offset) / self._scale_factor + self._min_values
        values[empty] = self._mask_value
        return values


class DepthQuantizer(Quantizer):
    """16-bit depth in metres over [0, 2]; anything farther is an outlier and is stored as "no measurement"."""

    def __init__(self, min_values: float = 0, max_values: float = 2, bits: int = 16, separate_mask: bool = True):
        super().__init__(min_values=min_values, max_values=max_values, bits=bits, separate_mask=separate_mask)

    def encode(self, values: np.ndarray) -> np.ndarray:
        values[values > self._max_values] = self._mask_value        # in place, as the reference does
        return super().encode(values)


class NormalsQuantizer(Quantizer):
    """Unit normals as 8-bit spherical coordinates (radius, theta in [pi/3, pi], phi in [-pi, pi])."""

    def __init__(self):
        super().__init__(min_values=np.array([0, 1 / 3 * np.pi, -np.pi]), max_values=np.array([1, np.pi, np.pi]),
                         bits=8)

    def encode(self, values: np.ndarray) -> np.ndarray:
        return super().encode(to_spherical(values))

    def decode(self, quantized_values: np.ndarray) -> np.ndarray:
        present = _pixel_mask(quantized_values != 0, np.any)
        spherical = super().decode(quantized_values)
        normals = np.zeros_like(spherical)
        normals[present] = to_cartesian(spherical[present])
        return normals
