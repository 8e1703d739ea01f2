"""Native optimizer step for the hash tables + a device-side GradScaler.

``HashTableAdam`` is torch.optim.Adam (no amsgrad / weight decay; reference hyper-parameters
scripts/train/train_nersemble.py:243-246) for ``HashEnsemble.tables`` as ONE libnsx kernel that forms the table
gradient on the fly from the factored gradient (``functional.FactoredGradSink``), unscales it, updates the moments
and the fp32 master and writes the fp16 working copy.  ``NativeGradScaler`` keeps torch.amp.GradScaler's algorithm
(scale 65536, x2 every 2000 clean steps, x0.5 on inf/NaN; per-optimizer skip; nersemble_trainer.py:185-203) but holds
scale / found_inf on the device so the table gradient never needs its own unscale pass.
"""
import ctypes as C
import os
from typing import List, Optional

import torch

from .. import _lib
from .. import functional as F
from .._lib import check, lib, ptr, stream
from ..field_components.hash_ensemble import HashEnsemble


class HashTableAdam(torch.optim.Optimizer):
    writes_half_tables = True      # HashEnsemble keeps its fp16 working copy; this optimizer refreshes it itself

    def __init__(self, hash_ensemble: HashEnsemble, lr: float = 5e-3, betas=(0.9, 0.999), eps: float = 1e-15,
                 factored: bool = True):
        self.he = hash_ensemble
        super().__init__([hash_ensemble.tables], dict(lr=lr, betas=betas, eps=eps))
        self.factored = factored
        if factored:
            hash_ensemble.grad_sink = F.FactoredGradSink()
        # early step (armed by the trainer for one backward): see arm_early_step
        self._early = None
        self.stepped_early = False
        # compact first-grid phase of the HashEnsemble (field_components/hash_ensemble.py): the moments of grid 0 as
        # contiguous [entry][f] arrays while it lasts
        self._compact_state = None
        hash_ensemble._compact_listeners = [self._on_first_grid_phase]      # (the optimizer that owns the tables now)

    @torch.no_grad()
    def _on_first_grid_phase(self, what: str) -> None:
        """enter: take column 0 of the moments; sync / leave: write it back (the HashEnsemble has ordered the current
        stream after the last optimizer pass before it calls)."""
        st = self._state()
        if what == "enter":
            w = self.he._compact["width"]            # grids of the compact copy (1: first-grid phase; 2 ... 16: window ramp)
            self._compact_state = {"width": w, "exp_avg": st["exp_avg"][:, :, 0:w].contiguous(),
                                   "exp_avg_sq": st["exp_avg_sq"][:, :, 0:w].contiguous()}
        elif self._compact_state is not None:
            w = self._compact_state["width"]
            st["exp_avg"][:, :, 0:w].copy_(self._compact_state["exp_avg"])
            st["exp_avg_sq"][:, :, 0:w].copy_(self._compact_state["exp_avg_sq"])
            if what == "leave":
                self._compact_state = None

    # ---- the step started from inside the backward -----------------------------------------------------------
    def arm_early_step(self, found_inf: torch.Tensor, inv_scale: torch.Tensor, side_stream) -> None:
        """For the coming backward: as soon as the HashEnsemble's backward has completed G -- and handed over the
        gradients of the rest of this optimizer's group (``FactoredGradSink.arrived(group_grads=...)``) -- check them for
        inf / NaN into ``found_inf`` and launch the table step on ``side_stream``, i.e. BESIDE the deformation field's
        backward and the small groups' optimizer work instead of after them.  GradScaler semantics are kept: the group is
        skipped as a whole iff any of its gradients is non-finite (the trainer merges ``found_inf`` into the flags the
        small-group kernels and the scale update read)."""
        sink = self.he.grad_sink
        if sink is None:
            return
        self._early = (found_inf, inv_scale, side_stream)
        self.stepped_early = False
        sink.on_complete = self._early_step

    # the optimizer pass clears the pieces of G it finds non-zero while it reads them (nsx_adam_hash_factored_consume):
    # no 1.6 GB fill in front of the next backward's scatter.  Off (A/B measurements): the backward clears G itself.
    consume_gradient = os.environ.get("NSX_ADAM_CONSUMES_G", "1") == "1"
    consume_always_below_bytes = 1 << 29     # (compact first-grid phase: a G of a few planes, HashEnsemble.first_grid_planes)
    consume_density_limit = 0.5      # ... while the scatter is expected to touch less than this share of G's sectors

    def disarm_early_step(self) -> None:
        self._early = None
        if self.he.grad_sink is not None:
            self.he.grad_sink.on_complete = None

    @torch.no_grad()
    def _early_step(self) -> None:
        sink = self.he.grad_sink
        if self._early is None or sink.group_grads is None or self.he.tables.grad is not None or len(sink.entries) != 1:
            return                                           # not everything is known here: the trainer steps later
        found_inf, inv_scale, side_stream = self._early
        for g in sink.group_grads:
            check(lib().nsx_check_finite(ptr(g), g.numel(), ptr(found_inf), stream()), "nsx_check_finite")
        self.check_finite(found_inf)
        self.step(found_inf=found_inf, inv_scale=inv_scale, side_stream=side_stream)
        self.stepped_early = True

    def _state(self):
        p = self.he.tables
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def check_finite(self, found_inf: torch.Tensor) -> None:
        """found_inf[0] = 1 if the pending table gradient holds an inf/NaN."""
        sink = self.he.grad_sink
        if sink is not None and sink.entries:
            sink.wait_scatter()            # G and the flag are complete once the scatter stream has caught up
            # the backward kernel flagged every non-finite value it added to G: no pass over the 1.2 GB buffer
            torch.maximum(found_inf, sink.nonfinite.to(found_inf.dtype), out=found_inf)
        g = self.he.tables.grad
        if g is not None:
            check(lib().nsx_check_finite(ptr(g.contiguous()), g.numel(), ptr(found_inf), stream()), "nsx_check_finite")

    def step(self, found_inf: Optional[torch.Tensor] = None, inv_scale: Optional[torch.Tensor] = None,
             side_stream: Optional["torch.cuda.Stream"] = None):
        """``side_stream``: run the 12 GB pass there instead of on the current stream.  Nothing after it in the step
        depends on the tables (the other groups' optimizers, the scaler update, the next step's ray marching), so the
        current stream carries on; ``HashEnsemble.wait_tables`` orders the next reader of the tables after it."""
        return self.step_unhooked(found_inf, inv_scale, side_stream)

    @torch.no_grad()
    def step_unhooked(self, found_inf: Optional[torch.Tensor] = None, inv_scale: Optional[torch.Tensor] = None,
                      side_stream: Optional["torch.cuda.Stream"] = None):
        """``step`` without torch.optim's wrapper around it (profiler scope, pre / post hook lists: ~30 us of host time
        per call).  The trainer calls this one; its small-parameter optimizer announces the step to the caches that
        listen for optimizer steps (engine/small_adam.py::adam_groups), and this optimizer writes the fp16 working tables
        itself (``writes_half_tables``), which is all HashEnsemble's post-step hook wants to know."""
        he, p = self.he, self.he.tables
        sink = he.grad_sink
        entries = sink.entries if sink is not None else []
        if not entries and p.grad is None:
            return
        if sink is not None and p.is_cuda:
            sink.wait_scatter()
        if side_stream is None or not p.is_cuda:
            return self._step_now(found_inf, inv_scale)
        he.wait_tables()
        side_stream.wait_stream(torch.cuda.current_stream(p.device))
        # (every G as well: a second G of a step comes from the caching allocator on the main stream and is dropped by
        # sink.clear() right after the launch -- without the record the allocator could hand it out while it is read)
        keep = [t for e in entries for t in (e["G"], e["code"], e["window"]) if t is not None]
        keep += [t for t in (found_inf, inv_scale, p.grad) if t is not None]
        with torch.cuda.stream(side_stream):
            self._step_now(found_inf, inv_scale)
            done = torch.cuda.Event()
            done.record(side_stream)
        for t in keep:                        # allocated on the main stream, still being read on the side stream
            t.record_stream(side_stream)
        he._tables_ready = done
        if sink is not None:
            sink.consumed = done                  # G has been read once this event has passed (FactoredGradSink.clear_ahead)

    def _step_now(self, found_inf, inv_scale):
        he, p = self.he, self.he.tables
        group = self.param_groups[0]
        sink = he.grad_sink
        entries = sink.entries if sink is not None else []
        st = self._state()
        st["step"] += 1
        b1, b2 = group["betas"]
        comp = he._compact
        if comp is not None and comp["width"] >= 2 and len(entries) == 1 and p.grad is None:
            # compact window-ramp layout: the first `width` grids alone (HashEnsemble.compact_width); the gradient's codes
            # and window are the full layout's, the kernel reads their first `width` entries
            e, cs, w = entries[0], self._compact_state, comp["width"]
            sparse = 0 < sink.samples_scattered * 80 < self.consume_density_limit * (e["G"].numel() // 8)
            consume = self.consume_gradient and sparse and sink.is_persistent(e["G"])
            fn = lib().nsx_adam_hash_factored_consume if consume else lib().nsx_adam_hash_factored
            check(fn(ptr(e["G"]), e["n_rows"], ptr(e["code"]), e["code"].stride(0), ptr(e["window"]), w, C.byref(he.geom),
                     ptr(comp["master"]), ptr(cs["exp_avg"]), ptr(cs["exp_avg_sq"]), ptr(comp["f16"]), group["lr"], b1, b2,
                     group["eps"], st["step"], ptr(inv_scale), ptr(found_inf), stream()), "nsx_adam_hash_factored")
            if consume:
                sink.mark_cleared(e["G"])
            sink.clear()
            return
        if comp is not None and len(entries) == 1 and p.grad is None and he.is_first_grid_code(entries[0]["code"]):
            # compact first-grid phase: the step of grid 0 alone, on its contiguous copy (H = 1 pass, 0.7 GB instead of
            # 11.7 GB at the reference geometry); the other grids have zero gradient and zero moments -- Adam leaves them
            # where they are
            e, cs = entries[0], self._compact_state
            # (a G of a few planes -- HashEnsemble.first_grid_planes -- is cleared in passing whatever its density: the pass
            # is a fraction of a millisecond either way, a separate fill would sit in front of the next scatter)
            sparse = (0 < sink.samples_scattered * 80 < self.consume_density_limit * (e["G"].numel() // 8)
                      or e["G"].numel() * 4 <= self.consume_always_below_bytes)
            consume = self.consume_gradient and sparse and sink.is_persistent(e["G"])
            fn = lib().nsx_adam_hash_factored_consume if consume else lib().nsx_adam_hash_factored
            check(fn(ptr(e["G"]), e["n_rows"], ptr(e["code"]), e["code"].stride(0), None, 1, C.byref(he.geom),
                     ptr(comp["master"]),
                     ptr(cs["exp_avg"]), ptr(cs["exp_avg_sq"]), ptr(comp["f16"]), group["lr"], b1, b2, group["eps"],
                     st["step"], ptr(inv_scale), ptr(found_inf), stream()), "nsx_adam_hash_factored")
            if consume:
                sink.mark_cleared(e["G"])
            sink.clear()
            return
        if comp is not None:
            raise RuntimeError("HashTableAdam: the HashEnsemble is in its compact first-grid phase but the pending gradient "
                               "is not that phase's (call hash_ensemble.leave_first_grid_phase() before mixing paths)")
        f16 = he.half_tables()            # make sure the working copy exists on the right device
        if len(entries) == 1 and p.grad is None:
            e = entries[0]
            # (worth it while the scatter touches a minority of G's 32-byte sectors -- about 80 per sample; on a densely
            # written G the clearing stores cost what the fill they replace costs, and the plain pass is the one whose
            # bytes bench.py prices)
            sparse = 0 < sink.samples_scattered * 80 < self.consume_density_limit * (e["G"].numel() // 8)
            consume = self.consume_gradient and sparse and sink.is_persistent(e["G"])
            fn = lib().nsx_adam_hash_factored_consume if consume else lib().nsx_adam_hash_factored
            check(fn(ptr(e["G"]), e["n_rows"], ptr(e["code"]), e["code"].stride(0), ptr(e["window"]), he.n_hash_encodings,
                     C.byref(he.geom), ptr(p.data), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(f16), group["lr"], b1,
                     b2, group["eps"], st["step"], ptr(inv_scale), ptr(found_inf), stream()), "nsx_adam_hash_factored")
            if consume:
                sink.mark_cleared(e["G"])              # (event on the stream the optimizer runs on)
        else:
            grad = p.grad.contiguous() if p.grad is not None else torch.zeros_like(p)
            for e in entries:      # several code tables in one step: expand each into the dense gradient
                check(lib().nsx_hash_grad_expand(ptr(e["G"]), e["n_rows"], ptr(e["code"]), e["code"].stride(0),
                                                 ptr(e["window"]), he.n_hash_encodings, C.byref(he.geom), ptr(grad), 1,
                                                 stream()), "nsx_hash_grad_expand")
            check(lib().nsx_adam_dense(ptr(grad), grad.numel(), ptr(p.data), ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]),
                                       ptr(f16), group["lr"], b1, b2, group["eps"], st["step"], ptr(inv_scale),
                                       ptr(found_inf), stream()), "nsx_adam_dense")
        if sink is not None:
            sink.clear()
        p._version_bump = None
        he.mark_half_synced()

    def rollback_step(self) -> None:
        """The last ``step()`` was skipped on the device (inf/NaN): it must not count (torch.optim.Adam semantics)."""
        st = self.state.get(self.he.tables)
        if st:
            st["step"] = max(0, st["step"] - 1)

    # ---- checkpointing: moments in the reference's parameter layout (one flat tensor per tcnn encoding) -------------
    def table_state(self) -> dict:
        self.he.sync_first_grid()
        self.he.wait_tables()
        st = self._state()
        return {"step": int(st["step"]), "lr": float(self.param_groups[0]["lr"]),
                "exp_avg": self.he.to_tcnn_layout(st["exp_avg"]), "exp_avg_sq": self.he.to_tcnn_layout(st["exp_avg_sq"])}

    def load_table_state(self, state: dict) -> None:
        self.he.leave_first_grid_phase()
        self.he.wait_tables()
        st = self._state()
        st["step"] = int(state["step"])
        self.param_groups[0]["lr"] = float(state.get("lr", self.param_groups[0]["lr"]))
        if state.get("exp_avg") is None:                 # a checkpoint written before the first optimizer step
            st["exp_avg"].zero_()
            st["exp_avg_sq"].zero_()
        else:
            st["exp_avg"].copy_(self.he.from_tcnn_layout(state["exp_avg"]))
            st["exp_avg_sq"].copy_(self.he.from_tcnn_layout(state["exp_avg_sq"]))

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        if self.he.grad_sink is not None:
            self.he.grad_sink.clear()

    def clear_grads(self) -> None:
        """``zero_grad(set_to_none=True)`` without torch.optim's profiler scope and per-group bookkeeping (the trainer calls
        it every step; four optimizers' ``zero_grad`` cost the host 55 us of a 1.6 ms step)."""
        self.he.tables.grad = None
        if self.he.grad_sink is not None:
            self.he.grad_sink.clear()


class NativeGradScaler:
    """torch.amp.GradScaler's algorithm with device-resident state and explicit found_inf plumbing."""

    def __init__(self, device, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5,
                 growth_interval: int = 2000, enabled: bool = True):
        self.enabled = enabled
        self.growth_factor, self.backoff_factor, self.growth_interval = growth_factor, backoff_factor, growth_interval
        self._scale = torch.full((), init_scale if enabled else 1.0, dtype=torch.float32, device=device)
        self._growth_tracker = torch.full((), 0, dtype=torch.int32, device=device)

    def scale(self, loss: torch.Tensor) -> torch.Tensor:
        return loss * self._scale if self.enabled else loss

    def _cached(self, key, make):
        """Device tensors derived from the scale, rebuilt only when ``_scale`` was written (its version counter: the
        update kernel, ``load_state_dict`` and callers that poke the tensor all bump it)."""
        cache = self.__dict__.setdefault("_derived", {})
        stamp = (self._scale._version, self._scale.data_ptr())
        hit = cache.get(key)
        if hit is None or hit[0] != stamp:
            hit = cache[key] = (stamp, make())
        return hit[1]

    def inv_scale(self) -> torch.Tensor:
        """1 / scale as a device tensor of shape [1] (three small launches, taken off the step's critical path: the
        trainer asks for it right after the scale update, beside the table optimizer's pass)."""
        return self._cached("inv", lambda: self._scale.double().reciprocal().float().reshape(1))

    def loss_grad_vector(self, n: int, index: int) -> torch.Tensor:
        """The gradient that ``scale(v[index]).backward()`` hands to the producer of the vector ``v`` [n]: zeros with the
        scale at ``index``.  Starting the backward from it (``torch.autograd.backward([v], [g])``) skips autograd's
        select / mul / ones nodes -- five tiny launches in front of every backward."""
        def make():
            g = torch.zeros((n,), dtype=torch.float32, device=self._scale.device)
            g[index] = self._scale if self.enabled else 1.0
            return g
        return self._cached(("grad", n, index), make)

    def unscale_and_check(self, grads: List[torch.Tensor], found_inf: torch.Tensor, inv_scale: torch.Tensor) -> None:
        if grads:
            torch._amp_foreach_non_finite_check_and_unscale_(grads, found_inf, inv_scale)

    def update(self, found_infs: List[torch.Tensor]) -> torch.Tensor:
        """One scale update from the found_inf flags of all groups; returns their sum (device, shape [1])."""
        total = found_infs[0].clone()
        for f in found_infs[1:]:
            total += f
        if self.enabled:
            torch._amp_update_scale_(self._scale, self._growth_tracker, total.reshape(()), self.growth_factor,
                                     self.backoff_factor, self.growth_interval)
            if self._scale.is_cuda:
                self.inv_scale()              # the next step's reciprocal now (queued behind the update, off its path)
        return total

    def update_native(self, found_all: torch.Tensor, found_copy: Optional[torch.Tensor], next_flags: torch.Tensor,
                      next_inv: torch.Tensor) -> None:
        """``update`` as ONE launch (``nsx_grad_scaler_update``): sum of the flags, scale / growth-tracker update, the
        flags copied to ``found_copy``, the NEXT step's flag buffer cleared and its 1 / scale written, and the one-hot loss
        gradients (``loss_grad_vector``) rewritten IN PLACE.  The torch route (a clone, two adds, ``_amp_update_scale_``,
        three ops for the reciprocal, a fill for the next step's flags) costs ~0.09 ms of host time per step, which is
        what paces a step once the occupancy grid has pruned the scene.  ``found_all`` / this step's 1 / scale are left
        alone: the table optimizer reads them on its own stream while this kernel runs (the caller alternates two slots).
        The kernel does not bump ``_scale``'s version counter, so the cached loss gradients stay the current ones; a
        torch-side write to ``_scale`` (``load_state_dict``) bumps it and they are rebuilt."""
        cache = self.__dict__.setdefault("_derived", {})
        cache.pop("inv", None)                               # (the cached reciprocal is not kept current on this route)
        stamp = (self._scale._version, self._scale.data_ptr())
        ptrs = tuple(t.data_ptr() + 4 * key[2] for key, (st, t) in cache.items()
                     if isinstance(key, tuple) and key[0] == "grad" and st == stamp)
        if len(ptrs) > _lib.NSX_MAX_SCALE_MIRRORS:
            for key in [k for k in cache if isinstance(k, tuple) and k[0] == "grad"]:
                del cache[key]                               # (rebuilt on their next use)
            ptrs = ()
        arr = self.__dict__.get("_mirror_arr")
        if arr is None or arr[0] != ptrs:
            arr = self._mirror_arr = (ptrs, (C.c_void_p * max(1, len(ptrs)))(*ptrs))
        check(lib().nsx_grad_scaler_update(ptr(found_all), found_all.numel(), ptr(self._scale), ptr(self._growth_tracker),
                                           ptr(next_inv), ptr(found_copy), ptr(next_flags), arr[1], len(ptrs),
                                           float(self.growth_factor), float(self.backoff_factor),
                                           int(self.growth_interval), int(self.enabled), stream()),
              "nsx_grad_scaler_update")

    def get_scale(self) -> float:
        return float(self._scale.item())

    def state_dict(self) -> dict:
        """torch.amp.GradScaler.state_dict()'s keys."""
        return {"scale": self.get_scale(), "growth_factor": self.growth_factor, "backoff_factor": self.backoff_factor,
                "growth_interval": self.growth_interval, "_growth_tracker": int(self._growth_tracker.item())}

    def load_state_dict(self, state: dict) -> None:
        self._scale.fill_(float(state["scale"]))
        self._growth_tracker.fill_(int(state.get("_growth_tracker", 0)))
        self.growth_factor = state.get("growth_factor", self.growth_factor)
        self.backoff_factor = state.get("backoff_factor", self.backoff_factor)
        self.growth_interval = state.get("growth_interval", self.growth_interval)
