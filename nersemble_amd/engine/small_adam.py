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
from typing import Dict, List, Optional, Sequence

import ctypes as C

import torch

from .._lib import AdamGroup, NSX_MAX_GROUPS, NSX_MAX_TENSORS, TensorRef, check, lib, ptr, stream


class SmallGroupAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps))
        # torch.optim.Adam keeps ``step`` PER PARAMETER and creates it when the parameter first has a gradient: a tensor
        # that starts late (time_embedding.weight gets its first gradient when the coarse-to-fine window opens at step
        # 40 000, train_nersemble.py:77-78) takes its first update with the bias corrections of step 1, not of step
        # 40 001.  Host-side counts (the kernel takes them by value in nsx_tensor_ref.step).
        self.steps: Dict[torch.nn.Parameter, int] = {}
        self._last_stepped: List[torch.nn.Parameter] = []
        self._layout_version = 0           # bumped when parameters / moment tensors are replaced (the cached tensor table)
        for p in self._params():
            if p.dtype != torch.float32:
                raise TypeError("SmallGroupAdam: fp32 parameters only")

    @property
    def step_count(self) -> int:
        """Steps taken by the group = the largest per-parameter count."""
        return max(self.steps.values(), default=0)

    def _params(self) -> List[torch.nn.Parameter]:
        return [p for g in self.param_groups for p in g["params"]]

    def clear_grads(self) -> None:
        """``zero_grad(set_to_none=True)`` without torch.optim's profiler scope (see HashTableAdam.clear_grads)."""
        for g in self.param_groups:
            for p in g["params"]:
                p.grad = None

    def _moments(self, p):
        st = self.state[p]
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st["exp_avg"], st["exp_avg_sq"]

    def advance(self, p) -> int:
        """Count one step of ``p`` (called for the tensors that have a gradient); returns its 1-based step."""
        n = self.steps.get(p, 0) + 1
        self.steps[p] = n
        self._last_stepped.append(p)
        return n

    def rollback_step(self) -> None:
        """The last step was skipped on the device (inf / NaN gradient): it does not count, for any of its tensors."""
        for p in self._last_stepped:
            self.steps[p] = max(0, self.steps.get(p, 0) - 1)
        self._last_stepped = []

    def rollback_params(self, params) -> None:
        """The last step left these tensors alone on the device (data-parallel run: no rank had a gradient for them --
        ``nsx_multi_adam_present``): their step does not count.  They leave ``_last_stepped``, so a ``rollback_step`` of the
        same step does not take a second count from them."""
        gone = {id(p) for p in params}
        keep = []
        for p in self._last_stepped:
            if id(p) in gone:
                self.steps[p] = max(0, self.steps.get(p, 0) - 1)
            else:
                keep.append(p)
        self._last_stepped = keep

    def step(self, closure=None):             # pragma: no cover - the trainer steps all groups together
        raise RuntimeError("SmallGroupAdam steps through step_groups([...]) (all groups in one launch)")

    # torch's Adam state-dict shape: per-parameter {"step", "exp_avg", "exp_avg_sq"}; a parameter that never had a
    # gradient has no state (as in torch)
    def state_dict(self):
        for p in self._params():
            n = self.steps.get(p, 0)
            if n > 0:
                self._moments(p)
                self.state[p]["step"] = torch.tensor(float(n))
            elif p in self.state and "step" not in self.state[p]:
                del self.state[p]              # moments allocated for the tensor table, never used
                self._layout_version += 1      # (... which must not keep their addresses)
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self.steps = {p: int(st["step"]) for p, st in self.state.items() if "step" in st}
        self._last_stepped = []
        self._layout_version += 1

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._layout_version = getattr(self, "_layout_version", 0) + 1


_TABLE_CACHE = {}


def _table(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int]):
    """The by-value tensor table of the two native launches.  Parameters, moments, sizes and groups do not change from step
    to step: that part is built once per set of optimizers (and again after ``load_state_dict``, which replaces the moment
    tensors); per step only the gradient pointers are read -- ~25 tensors x (attribute + data_ptr) instead of rebuilding
    every reference (0.07 ms of a step that the host paces)."""
    key = tuple((id(o), o._layout_version, g) for o, g in zip(optimizers, group_of))
    hit = _TABLE_CACHE.get("t")
    if hit is None or hit[0] != key:
        refs = (TensorRef * NSX_MAX_TENSORS)()
        owners = []
        n = 0
        for opt, grp in zip(optimizers, group_of):
            for p in opt._params():
                if n >= NSX_MAX_TENSORS:
                    raise RuntimeError(f"more than {NSX_MAX_TENSORS} small parameter tensors")
                m, v = opt._moments(p)
                if not (p.is_cuda and p.is_contiguous()):
                    raise RuntimeError("SmallGroupAdam: contiguous device parameters only")
                r = refs[n]
                # raw pointers without a dispatch per tensor (Tensor.data / detach() are torch ops)
                r.param, r.exp_avg, r.exp_avg_sq = p.data_ptr(), m.data_ptr(), v.data_ptr()
                r.n, r.group, r.step = p.numel(), grp, 0
                owners.append((opt, p))
                n += 1
        hit = (key, refs, n, owners)
        _TABLE_CACHE["t"] = hit
    _, refs, n, owners = hit
    for i, (opt, p) in enumerate(owners):
        if refs[i].param != p.data_ptr():
            # the parameter's storage was replaced (``model.to()``, a master-weight swap): the cached raw pointers are stale
            # -- rebuild the table (advisor, round 4: the kernels would otherwise read and write freed memory)
            _TABLE_CACHE.pop("t", None)
            return _table(optimizers, group_of)
        g = p.grad
        if g is None:
            refs[i].grad = None
        else:
            if not g.is_contiguous():
                p.grad = g = g.contiguous()
            refs[i].grad = g.data_ptr()
        refs[i].step = 0
    return refs, n, owners


def unscale_and_check_groups(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int], n_groups: int,
                             found_inf: torch.Tensor, inv_scale: Optional[torch.Tensor]):
    """found_inf [n_groups] (device, fp32): set to 1 for every group that holds a non-finite gradient; gradients are
    unscaled in place.  Returns the tensor table for ``adam_groups`` (the pointers stay valid for this step)."""
    assert n_groups <= NSX_MAX_GROUPS and found_inf.numel() >= n_groups
    refs, n, owners = _table(optimizers, group_of)
    if n:
        check(lib().nsx_multi_unscale_check(refs, n, n_groups, ptr(inv_scale), ptr(found_inf), stream()),
              "nsx_multi_unscale_check")
    return refs, n, owners


def adam_groups(optimizers: Sequence[SmallGroupAdam], group_of: Sequence[int], n_groups: int, table,
                found_inf: Optional[torch.Tensor], present: Optional[torch.Tensor] = None, present_index=None) -> None:
    """``present`` / ``present_index`` (data-parallel runs): the device vector of how many ranks held a gradient per parameter
    (``all_reduce_gradients``' counts[0]) and ``{id(parameter): its element}`` -- a parameter nobody had a gradient for is
    left alone by the kernel although every rank carries the zeros it joined the all-reduce with; the trainer takes the
    host-side count back when the counts arrive (``SmallGroupAdam.rollback_params``)."""
    refs, n, owners = table
    if not n:
        return
    groups = (AdamGroup * NSX_MAX_GROUPS)()
    seen = set()
    for opt, grp in zip(optimizers, group_of):
        if grp in seen:
            raise RuntimeError("one SmallGroupAdam per group")
        seen.add(grp)
        opt._last_stepped = []
        pg = opt.param_groups[0]
        groups[grp].lr, (groups[grp].beta1, groups[grp].beta2), groups[grp].eps = pg["lr"], pg["betas"], pg["eps"]
        groups[grp].step = 1                                      # (every tensor carries its own count)
    for i, (opt, p) in enumerate(owners):
        if refs[i].grad:
            refs[i].step = opt.advance(p)                         # counted on the host before the device decides to skip
    for k in range(n_groups):
        if k not in seen:
            groups[k].step = 1
    if present is not None and present_index:
        at = (C.c_int32 * n)(*[present_index.get(id(p), -1) for _, p in owners])
        check(lib().nsx_multi_adam_present(refs, n, groups, n_groups, ptr(found_inf), ptr(present), at, int(present.numel()),
                                           stream()), "nsx_multi_adam_present")
    else:
        check(lib().nsx_multi_adam(refs, n, groups, n_groups, ptr(found_inf), stream()), "nsx_multi_adam")
    # what torch's optimizer-step hooks would have announced: cached parameter packs (deformation MFMA fragments, the
    # evaluation pre-blend) are stale now
    from ..field_components.deformation_field import _OPTIMIZER_STEPS
    _OPTIMIZER_STEPS[0] += 1
