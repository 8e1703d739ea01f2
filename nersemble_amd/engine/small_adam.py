"""torch.optim.Adam for the SMALL parameter groups (mlp_base / mlp_head, the time embeddings, the deformation field:
~0.27 M parameters in ~25 tensors) as two native launches per step for ALL groups together.

The reference steps three ``torch.optim.Adam`` objects through ``GradScaler`` (nersemble_trainer.py:185-203,
hyper-parameters train_nersemble.py:243-256).  On the device that is negligible work, but on the host it is three
``_amp_foreach_non_finite_check_and_unscale_`` calls plus three fused-Adam steps with their optimizer hooks, step-tensor
bookkeeping and profiler ranges: ~0.5 ms per step, which is what bounds a step once the occupancy grid has pruned the
scene (the step is host-bound there).  ``SmallGroupAdam`` keeps torch's optimizer interface (``param_groups`` for the
LR schedulers, ``state`` / ``state_dict`` with ``exp_avg`` / ``exp_avg_sq`` / ``step``), and ``step_groups`` runs

    nsx_multi_unscale_check   GradScaler.unscale_: found_inf per group, gradients * inv_scale in place
    nsx_multi_adam            Adam per group; a group whose found_inf is set is skipped, as GradScaler.step does

Semantics are torch.optim.Adam's (no amsgrad, no weight decay); parameters without a gradient are left alone.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from .._lib import AdamGroup, NSX_MAX_GROUPS, NSX_MAX_TENSORS, TensorRef, check, lib, ptr, stream


class SmallGroupAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps))
        self.step_count = 0                   # host-side (the kernels take bias corrections by value)
        for p in self._params():
            if p.dtype != torch.float32:
                raise TypeError("SmallGroupAdam: fp32 parameters only")

    def _params(self) -> List[torch.nn.Parameter]:
        return [p for g in self.param_groups for p in g["params"]]

    def _moments(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st["exp_avg"], st["exp_avg_sq"]

    def rollback_step(self) -> None:
        """The last step was skipped on the device (inf / NaN gradient): it does not count."""
        self.step_count = max(0, self.step_count - 1)

    def step(self, closure=None):             # pragma: no cover - the trainer steps all groups together
        raise RuntimeError("SmallGroupAdam steps through step_groups([...]) (all groups in one launch)")

    # torch's Adam state-dict shape: per-parameter {"step", "exp_avg", "exp_avg_sq"}
    def state_dict(self):
        for p in self._params():
            self._moments(p)
            self.state[p]["step"] = torch.tensor(float(self.step_count))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [int(st["step"]) for st in self.state.values() if "step" in st]
        self.step_count = max(steps) if steps else 0


def _table(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int]):
    refs = (TensorRef * NSX_MAX_TENSORS)()
    n = 0
    for opt, grp in zip(optimizers, group_of):
        for p in opt._params():
            if n >= NSX_MAX_TENSORS:
                raise RuntimeError(f"more than {NSX_MAX_TENSORS} small parameter tensors")
            g = p.grad
            if g is not None and not g.is_contiguous():
                p.grad = g = g.contiguous()
            m, v = opt._moments(p)
            if not (p.is_cuda and p.is_contiguous()):
                raise RuntimeError("SmallGroupAdam: contiguous device parameters only")
            r = refs[n]
            # raw pointers without a dispatch per tensor (Tensor.data / detach() are torch ops)
            r.param, r.exp_avg, r.exp_avg_sq = p.data_ptr(), m.data_ptr(), v.data_ptr()
            r.grad = g.data_ptr() if g is not None else None
            r.n, r.group = p.numel(), grp
            n += 1
    return refs, n


def unscale_and_check_groups(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int], n_groups: int,
                             found_inf: torch.Tensor, inv_scale: Optional[torch.Tensor]):
    """found_inf [n_groups] (device, fp32): set to 1 for every group that holds a non-finite gradient; gradients are
    unscaled in place.  Returns the tensor table for ``adam_groups`` (the pointers stay valid for this step)."""
    assert n_groups <= NSX_MAX_GROUPS and found_inf.numel() >= n_groups
    refs, n = _table(optimizers, group_of)
    if n:
        check(lib().nsx_multi_unscale_check(refs, n, n_groups, ptr(inv_scale), ptr(found_inf), stream()),
              "nsx_multi_unscale_check")
    return refs, n


def adam_groups(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int], n_groups: int, table,
                found_inf: Optional[torch.Tensor]) -> None:
    refs, n = table
    if not n:
        return
    groups = (AdamGroup * NSX_MAX_GROUPS)()
    seen = set()
    for opt, grp in zip(optimizers, group_of):
        if grp in seen:
            raise RuntimeError("one SmallGroupAdam per group")
        seen.add(grp)
        if not any(p.grad is not None for p in opt._params()):
            opt_step = max(opt.step_count, 1)                     # nothing to do for this group: the kernel skips it
        else:
            opt.step_count += 1
            opt_step = opt.step_count
        pg = opt.param_groups[0]
        groups[grp].lr, (groups[grp].beta1, groups[grp].beta2), groups[grp].eps = pg["lr"], pg["betas"], pg["eps"]
        groups[grp].step = opt_step
    for k in range(n_groups):
        if k not in seen:
            groups[k].step = 1
    check(lib().nsx_multi_adam(refs, n, groups, n_groups, ptr(found_inf), stream()), "nsx_multi_adam")
    # what torch's optimizer-step hooks would have announced: cached parameter packs (deformation MFMA fragments, the
    # evaluation pre-blend) are stale now
    from ..field_components.deformation_field import _OPTIMIZER_STEPS
    _OPTIMIZER_STEPS[0] += 1
