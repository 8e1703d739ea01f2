"""``chunked`` -- mirror of the reference's util/chunker.py:7-28 (same contract: aligned slices along dim 0,
None / 0-d tensors pass through, a single list yields bare items)."""
from math import ceil

import torch


def _passthrough(v) -> bool:
    return v is None or (isinstance(v, torch.Tensor) and v.dim() == 0)


def chunked(max_chunk_size: int, *lists):
    size = len(lists[0])
    for v in lists:
        assert _passthrough(v) or len(v) == size, "all chunked lists need the same length"
    if size <= max_chunk_size:
        # one chunk: hand the tensors through (a full-range slice is the same data, minus one dispatch each)
        parts = list(lists)
        yield parts[0] if len(parts) == 1 else tuple(parts)
        return
    for k in range(ceil(size / max_chunk_size)):
        lo, hi = k * max_chunk_size, (k + 1) * max_chunk_size
        parts = [v if _passthrough(v) else v[lo:hi] for v in lists]
        yield parts[0] if len(parts) == 1 else tuple(parts)
