"""Data-parallel gradient averaging: one process per GPU, RCCL all-reduce over xGMI (backend "nccl" on ROCm).

The reference is single-GPU (scripts/train/train_nersemble.py:272-274 hard-codes world_size = 1); rays are
independent until the loss mean, so each rank draws its own rays and holds a full replica; the only exchange is
the sum of gradients.  Large buffers (the 1.6 GB hash-table gradient) go out as their own
collective; small tensors are flattened into one bucket so a step issues O(1) collectives.
"""
from typing import Iterable, Optional, Sequence

import torch
import torch.distributed as dist

SMALL_BUCKET_ELEMS = 1 << 22
_NATIVE_PIECES = 16          # NSX_MAX_BUCKET_PIECES (include/nsx.h)


def all_reduce_gradients(params: Iterable[torch.nn.Parameter], world_size: int, group=None,
                         takes_part: Optional[Sequence[bool]] = None, force: bool = False, arena=None,
                         extra_flags: Optional[torch.Tensor] = None, native_comm=None):
    """In-place average of ``p.grad`` over all ranks.  A parameter without a gradient on this rank contributes zeros
    (every rank must join every collective).  What it is left with afterwards follows from whether the parameter took
    part in the step on ANY rank:

    * no rank had a gradient: it stays WITHOUT one -- an optimizer must not count a step for it (``torch.optim.Adam`` keeps
      ``step`` per parameter and creates it with the first real gradient: ``time_embedding.weight`` starts when the
      coarse-to-fine window opens, train_nersemble.py:77-78; its first update then takes the bias corrections of step 1 on
      one GPU and must do so on eight);
    * some rank had one (this rank's rays produced no samples, say): it keeps the AVERAGED gradient and every rank steps
      identically.  (Round 4 dropped the gradient here: the rank skipped an update the others applied -- the replicas
      diverged and the step after raised.)

    The global count is on the device when the decision is made.  ``takes_part`` -- the previous step's global counts > 0, per
    parameter, identical on every rank (the trainer reads the counts one step late, off the critical path) -- stands in for
    it: membership changes only where the schedule says so, on all ranks at once.  ``None`` (the first step; callers without
    a history) reads this step's counts on the host -- one blocking read.  The remaining hole -- a parameter that starts to
    take part in exactly the step in which this rank has no gradient at all -- is DETECTED: the number of ranks that dropped a
    gradient travels in the same bucket, and ``check_gradient_presence`` raises on every rank when a parameter was stepped
    by some ranks and dropped by others.

    ``arena`` (round 6): the persistent buffer in which the native training step deposits its parameter gradients as views
    (``engine.native_step._GradBuffers``): ``flat`` (1-D fp32 device tensor), ``slack`` (elements guaranteed behind ``fixed``), ``fixed`` (its leading elements that hold the
    leaf parameters' gradients -- the two fused MLPs and the deformation tensors; a constant of the model) and
    ``slot_of(p)`` (the view a parameter's gradient lives in, or None).  The wire format of the bucket is then
    ``flat[:fixed] | gradients without a slot (the embeddings') | presence counts | extra_flags`` -- the same length on every
    rank whatever path its step took -- and the slotted gradients are reduced WHERE THEY ARE: one collective and ~10 small
    launches per step instead of a ``cat`` of ~40 gradients and as many copies back (~80 launches on a
    path the host paces).  A rank whose step did not run the native drivers (nothing marched: the per-kernel path) copies its
    gradients into their slots first.  Every rank must pass an arena of the same model, or none (the trainer decides from
    the configuration): without one the small gradients are flattened into one bucket as before.

    ``native_comm``: an ``nsx_comm`` over the ranks of ``group`` (the level-parallel exchange's, csrc/comm.hip): the arena's
    bucket is then summed by the library on the current stream (``nsx_comm_all_reduce_sum``) -- no work object, no side
    stream to wait for.

    ``extra_flags``: a small fp32 device tensor summed over the ranks in the same bucket (the level-parallel optimizer's
    non-finite flag: a step is skipped on every rank or on none) -- returned as the third value.

    Returns (``[2, len(params)]`` device counts: ranks that held a gradient, ranks that dropped theirs; the summed
    ``extra_flags`` or None).  (None, None) for one process."""
    if world_size <= 1 and not force:             # (force: a one-rank group still issues its collectives -- emulated ranks)
        return None, None
    params = [p for p in params if p.requires_grad]
    if not params:
        return None, None
    absent = [p.grad is None for p in params]
    if takes_part is not None and len(takes_part) != len(params):
        takes_part = None
    for p, a in zip(params, absent):
        if a:
            p.grad = torch.zeros_like(p)
    big = [p for p in params if p.grad.numel() >= SMALL_BUCKET_ELEMS]
    small = [p for p in params if p.grad.numel() < SMALL_BUCKET_ELEMS]
    handles = [dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group, async_op=True) for p in big]
    dev = params[0].grad.device
    n = len(params)
    inv = 1.0 / world_size
    # what this rank will do with an absent gradient, decided BEFORE the exchange when a history exists
    drops = [a and takes_part is not None and not takes_part[i] for i, a in enumerate(absent)]
    tail = _presence_tail(tuple(absent), tuple(drops), dev)
    n_flags = int(extra_flags.numel()) if extra_flags is not None else 0
    flags = None
    inside, outside = [], small
    if arena is not None:
        if arena.flat.device != dev:
            raise RuntimeError("all_reduce_gradients: the gradient arena lives on another device than the gradients")
        slots = [arena.slot_of(p) for p in small]
        inside = [p for p, sl in zip(small, slots) if sl is not None]
        outside = [p for p, sl in zip(small, slots) if sl is None]
        need = sum(p.grad.numel() for p in outside) + 2 * n + n_flags
        if need > int(arena.slack):
            # (sizes of the model, the same on every rank: e.g. a DENSE table gradient among the small tensors of a test
            # model) -- the flattened bucket for everybody
            arena, inside, outside, slots = None, [], small, [None] * len(small)
        for p, sl in zip(small, slots):
            if sl is not None and p.grad.data_ptr() != sl.data_ptr():
                sl.copy_(p.grad.reshape(sl.shape))                 # (a step that did not deposit its gradients)
                p.grad = sl.view(p.shape)
    if arena is not None:
        flat = arena.flat
        fixed = int(arena.fixed)
        # on the device the bucket's tail is written, and read back, by ONE launch each (nsx_bucket_pack / _unpack) instead of a
        # copy per piece: ~7 launches fewer on a step the host paces (the CPU route below is what the gloo tests hold it to)
        native = (flat.is_cuda and len(outside) + 2 <= _NATIVE_PIECES and all(p.grad.is_contiguous() and
                  p.grad.dtype == torch.float32 for p in outside)
                  and (extra_flags is None or (extra_flags.is_contiguous() and extra_flags.dtype == torch.float32)))
        off = fixed
        if native:
            import ctypes as C
            from .._lib import check, lib, stream
            pieces = [p.grad for p in outside] + [tail] + ([extra_flags] if n_flags else [])
            ptrs = (C.c_void_p * len(pieces))(*[t.data_ptr() for t in pieces])
            sizes = (C.c_int64 * len(pieces))(*[t.numel() for t in pieces])
            check(lib().nsx_bucket_pack(flat.data_ptr() + 4 * fixed, ptrs, sizes, len(pieces), stream()), "nsx_bucket_pack")
            n_grad = fixed + sum(p.grad.numel() for p in outside)
            off = n_grad + 2 * n + n_flags
        else:
            for p in outside:                                        # (two embeddings' gradients, typically)
                k = p.grad.numel()
                flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
            n_grad = off
            flat[off:off + 2 * n].copy_(tail)
            off += 2 * n
            if n_flags:
                flat[off:off + n_flags].copy_(extra_flags.reshape(-1).to(torch.float32))
                off += n_flags
        span = flat[:off]
        if native_comm is not None:
            from .._lib import check, lib, stream
            check(lib().nsx_comm_all_reduce_sum(native_comm, span.data_ptr(), off, stream()), "nsx_comm_all_reduce_sum")
        else:
            handles.append(dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for h in handles:
            h.wait()
        aux = flat[n_grad:off]                                   # counts (+ flags): not averaged; views of the arena -- their
        #                                                          readers (the inf check, the copy to the host) run this step
        counts = aux[:2 * n].view(2, n)
        if n_flags:
            flags = aux[2 * n:]
        if native:
            m = len(outside)
            check(lib().nsx_bucket_unpack(flat.data_ptr(), fixed, float(inv), flat.data_ptr() + 4 * fixed, ptrs, sizes, m,
                                          stream()), "nsx_bucket_unpack")
        else:
            if inv != 1.0:
                flat[:n_grad].mul_(inv)
            off = fixed
            for p in outside:
                k = p.grad.numel()
                p.grad.copy_(flat[off:off + k].view_as(p.grad))
                off += k
    else:
        pieces = [p.grad.reshape(-1).float() for p in small] + [tail]
        if n_flags:
            pieces.append(extra_flags.reshape(-1).to(torch.float32))
        flat = torch.cat(pieces)
        handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
        for h in handles:
            h.wait()
        aux = flat[flat.numel() - 2 * n - n_flags:].clone()
        counts = aux[:2 * n].view(2, n)
        if n_flags:
            flags = aux[2 * n:]
        flat.mul_(inv)
        off = 0
        for p in small:
            k = p.grad.numel()
            p.grad.copy_(flat[off:off + k].view_as(p.grad))
            off += k
    for p in big:
        p.grad.mul_(inv)
    if takes_part is None and any(absent):
        present_any = (counts[0] > 0).tolist()               # one blocking read (first step / no history)
        drops = [a and not present_any[i] for i, a in enumerate(absent)]
    for p, d in zip(params, drops):
        if d:
            p.grad = None                   # (zeros went into the sum; no rank stepped the parameter)
    return counts, flags


_TAILS = {}


def _presence_tail(absent, drops, dev) -> torch.Tensor:
    """[held ... | dropped ...] of this rank as a device tensor; the pattern is the same step after step (cached)."""
    key = (absent, drops, str(dev))
    t = _TAILS.get(key)
    if t is None:
        if len(_TAILS) > 64:
            _TAILS.clear()
        t = _TAILS[key] = torch.tensor([0.0 if a else 1.0 for a in absent] + [1.0 if d else 0.0 for d in drops],
                                       dtype=torch.float32).to(dev)
    return t


def check_gradient_presence(counts, world_size: int) -> None:
    """``counts``: what ``all_reduce_gradients`` returned, on the host (``[2][n]``: ranks with a gradient, ranks that
    dropped theirs).  A parameter some rank had a gradient for must not have been dropped by another: the first would have
    stepped it, the second not."""
    held, dropped = counts[0], counts[1]
    bad = [i for i, (h, d) in enumerate(zip(held, dropped)) if h > 0 and d > 0]
    if bad:
        raise RuntimeError(f"data-parallel step: parameters {bad} (positions in the all-reduced list) received a gradient on "
                           f"{[held[i] for i in bad]} of {world_size} ranks while {[dropped[i] for i in bad]} ranks left them "
                           f"without one -- the replicas have diverged (per-parameter Adam step counts differ)")


def global_normaliser_scales(local_counts: torch.Tensor, n_eff_local: torch.Tensor, n_rays_local: int,
                             world_size: int, rank: int, group=None) -> torch.Tensor:
    """Strong scaling of ONE ray batch sliced over the ranks: the reference's loss terms are masked means
    whose denominators count rays / samples of the WHOLE batch (models/base.py:110-113 masked MSE, :120-134 alpha,
    :158-202 empty / near, :206-222 depth) and the distortion loss divides by ``ray_id.max() + 1`` of the whole batch
    (:235-247).  A rank only sees its slice, so every term ``sum_local / count_local`` is re-weighted by

        scale = world * count_local / count_global          (count clamped to >= 1 like the reference's empty-mask case)

    after which the usual gradient AVERAGE over the ranks (all-reduce sum / world) yields exactly the gradient of the
    single-process loss on the union batch, and the mean of the ranks' losses is that loss.

    local_counts [K] fp32: the raw local denominators of the K masked means; n_eff_local: this rank's raw
    ``ray_id.max() + 1`` (0 when the slice has no samples); the slices are consecutive blocks of ``n_rays_local`` rays in
    rank order.  Returns [K + 1] scales (the last one for the distortion term).  One small all-gather."""
    k = local_counts.numel()
    mine = torch.cat([local_counts.reshape(-1).to(torch.float32), n_eff_local.reshape(1).to(torch.float32)])
    if world_size > 1:
        everyone = torch.empty((world_size * (k + 1),), dtype=torch.float32, device=mine.device)
        dist.all_gather_into_tensor(everyone, mine.contiguous(), group=group)
        everyone = everyone.view(world_size, k + 1)
    else:
        everyone = mine.view(1, k + 1)
    one = torch.ones((), dtype=torch.float32, device=mine.device)
    count_global = torch.maximum(everyone[:, :k].sum(dim=0), one)
    count_local = torch.maximum(mine[:k], one)
    # index + 1 of the last ray of the WHOLE batch that carries samples (slices without samples do not take part)
    offsets = torch.arange(world_size, device=mine.device, dtype=torch.float32) * float(n_rays_local)
    last = torch.where(everyone[:, k] > 0, offsets + everyone[:, k], torch.zeros_like(offsets))
    n_eff_global = torch.maximum(last.max(), one)
    return torch.cat([world_size * count_local / count_global,
                      (world_size * torch.maximum(mine[k], one) / n_eff_global).reshape(1)])


def shard_seed(base_seed: int, rank: int) -> int:
    """Rank-specific RNG stream for ray sampling (identical model init comes from the un-sharded base seed)."""
    return base_seed + 7919 * rank
