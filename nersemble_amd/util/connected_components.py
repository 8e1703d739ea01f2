"""Occupancy-grid floater filter -- mirror of the reference's ``nersemble/util/connected_components.py:28-139``
(``extract_top_k_connected_component`` / ``filter_occupancy_grid``; called before evaluation / rendering,
scripts/evaluate/evaluate_nersemble.py:68-73): sigmoid -> 8-bit quantisation -> Gaussian blur -> threshold -> largest
6-connected component -> Gaussian dilation -> AND into ``occupancy_grid.binaries``.

The reference does this on the host with numpy / scipy.ndimage / cc3d.  Here the 128^3 grid stays on the device:

  * the 8-bit quantisation is done with numpy exactly as the reference writes it (2 M values, bit-identical);
  * ``scipy.ndimage.gaussian_filter`` on an INTEGER array is restated bit-exactly: float64 separable passes in scipy's
    summation order (centre tap, then symmetric pairs from the outermost inwards), 'reflect' boundary
    (d c b a | a b c d | d c b a), radius int(4 sigma + 0.5), and -- because scipy writes every axis pass back into the
    integer output array -- truncation after EVERY pass;
  * the largest 6-connected component (cc3d.largest_k(k=1, connectivity=6)) by label propagation.

One-off evaluation-time transform: plain torch ops, no kernel of its own.
"""
from typing import List

import numpy as np
import torch
from torch import Tensor


def _gaussian_kernel1d(sigma: float, radius: int) -> np.ndarray:
    """scipy.ndimage._filters._gaussian_kernel1d (order 0), float64."""
    x = np.arange(-radius, radius + 1)
    phi_x = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi_x / phi_x.sum()


def _reflect_index(n: int, r: int, device) -> Tensor:
    idx = torch.arange(-r, n + r, device=device)
    # scipy 'reflect' with as many folds as needed (radius may exceed the axis length)
    period = 2 * n
    idx = torch.remainder(idx, period)
    return torch.where(idx >= n, period - 1 - idx, idx)


def gaussian_filter_integer(x: Tensor, sigma: float, truncate: float = 4.0) -> Tensor:
    """``scipy.ndimage.gaussian_filter(x, sigma)`` for an integer-typed ndarray x (any integer dtype), returned as
    float64 holding the integer values."""
    radius = int(truncate * float(sigma) + 0.5)
    w = _gaussian_kernel1d(float(sigma), radius)
    out = x.to(torch.float64)
    for axis in range(out.dim()):
        v = out.movedim(axis, -1)
        n = v.shape[-1]
        p = v.index_select(-1, _reflect_index(n, radius, v.device))
        tmp = p[..., radius:radius + n] * float(w[radius])
        for jj in range(-radius, 0):
            tmp = tmp + (p[..., radius + jj:radius + jj + n] + p[..., radius - jj:radius - jj + n]) * float(w[jj + radius])
        out = torch.trunc(tmp).movedim(-1, axis).contiguous()
    return out


def largest_connected_component(mask: Tensor) -> Tensor:
    """Largest 6-connected component of a boolean 3-D volume (empty mask -> empty result).  Ties: the component
    containing the smallest linear index."""
    assert mask.dim() == 3 and mask.dtype == torch.bool
    n = mask.numel()
    if not bool(mask.any()):
        return torch.zeros_like(mask)
    labels = (torch.arange(1, n + 1, device=mask.device, dtype=torch.int32).reshape(mask.shape)) * mask
    check_every = 8
    it = 0
    while True:
        new = labels
        for dim in range(3):
            z = torch.zeros_like(labels.narrow(dim, 0, 1))
            lo = torch.cat([labels.narrow(dim, 1, labels.shape[dim] - 1), z], dim=dim)      # neighbour at +1
            hi = torch.cat([z, labels.narrow(dim, 0, labels.shape[dim] - 1)], dim=dim)      # neighbour at -1
            new = torch.maximum(new, torch.maximum(lo, hi))
        new = new * mask
        it += 1
        if it % check_every == 0 and bool(torch.equal(new, labels)):
            break
        labels = new
    counts = torch.bincount(labels[mask].to(torch.int64), minlength=n + 1)
    best = counts.max()
    # every component is labelled by its LARGEST linear index; ties -> the component with the smallest minimum index
    cands = (counts == best).nonzero()[:, 0]
    if cands.numel() > 1:
        lin = torch.arange(1, n + 1, device=mask.device, dtype=torch.int64).reshape(mask.shape)
        firsts = torch.stack([lin[labels == c].min() for c in cands.tolist()])
        winner = cands[firsts.argmin()]
    else:
        winner = cands[0]
    return labels == winner.to(labels.dtype)


def extract_top_k_connected_component(density_grid: Tensor, threshold: float = 0.6, sigma_thinning: float = 1,
                                      sigma_erosion: float = 2, K: int = 1) -> List[Tensor]:
    """connected_components.py:28-99 for K = 1 (the only value the reference uses): list with one {0,1} uint8 volume."""
    if K != 1:
        raise NotImplementedError("the reference only ever extracts the largest component (connected_components.py:129)")
    dev = density_grid.device
    # :52-55 -- sigmoid and 8-bit quantisation exactly as written there (numpy, the array's own float dtype)
    d = density_grid.detach().cpu().numpy()
    d = 1 / (1 + np.exp(-d))
    q = ((d - 0.5) * 2 * (255)).astype(np.uint8)
    g = gaussian_filter_integer(torch.from_numpy(q).to(dev), sigma_thinning)             # :58
    binary = g >= 255 * threshold                                                         # :62-63
    cc = largest_connected_component(binary)                                              # :76-80, connectivity 6
    grown = gaussian_filter_integer(cc.to(torch.int64) * 100, sigma_erosion)              # :88
    return [(grown > 0).to(torch.uint8)]                                                  # :90-92


@torch.no_grad()
def filter_occupancy_grid(occupancy_grid, threshold: float = 0.6, sigma_thinning: float = 1,
                          sigma_erosion: float = 5) -> None:
    """connected_components.py:102-139: keep only the largest connected blob of the occupancy grid (removes floaters)."""
    resolution = occupancy_grid.resolution
    try:
        resolution = [int(r) for r in resolution]
    except TypeError:
        resolution = [int(resolution)] * 3
    dens = occupancy_grid.occs.reshape(*resolution)
    largest = extract_top_k_connected_component(dens, threshold=threshold, sigma_thinning=sigma_thinning,
                                                sigma_erosion=sigma_erosion)[0]
    filtered = (largest > 0).to(device=occupancy_grid.binaries.device, dtype=occupancy_grid.binaries.dtype)
    occupancy_grid.binaries[0] = occupancy_grid.binaries[0] & filtered
