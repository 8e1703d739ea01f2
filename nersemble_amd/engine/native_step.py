"""A training step's sampler and main pass enqueued by the native step drivers (csrc/step.hip, include/nsx.h
"training-step drivers") instead of one ctypes call per kernel.

What runs on the device is what ``NeRSembleNGPModel.fused_train_forward`` runs -- ``NeRSembleVolumetricSampler.forward``
(nersemble_volumetric_sampler.py:95-134) with its sigma_fn density pass (nersemble_instant_ngp.py:235-266), then
``get_outputs`` + ``get_loss_dict`` + ``get_metrics_dict`` (:280-422) on the kept samples and their backward -- the same
kernels with the same arguments in the same order (tests/test_native_step_gpu.py holds the two paths together bit for bit
in the forward).  What changes is the host side: ~45 marshalled native calls, ~60 ``torch.empty`` and the RaySamples /
Frustums objects of a step become 6 native calls (plan, sampler, forward, three backward stages) on 5 workspaces whose
sub-buffers the C side carves.  Python keeps what is policy, not launching: the traversal's counting pass (possibly issued
a step ahead), the code-row lookups that autograd differentiates, the factored-gradient sink, streams and events.

Covers the configuration the reference trains (occupancy grid on, deformation field on, time codes per image, fp16
mixed precision, the sigma pass's forward values reused); anything else -> ``None`` and the caller takes the per-kernel path.
"""
import ctypes as C
import os
from typing import Dict, Optional

import torch

from .. import _lib
from .. import distloss as dl
from .. import functional as F
from .._lib import check, lib, stream


def _view(buf: torch.Tensor, off: int, shape, dtype) -> torch.Tensor:
    n = 1
    for d in shape:
        n *= int(d)
    nbytes = n * F._ITEMSIZE[dtype]
    return buf[off:off + nbytes].view(dtype).view(tuple(shape))


class _LazyView:
    """A tensor view of a workspace region, made when somebody needs the tensor (``OccGridEstimator.last_n_kept``)."""
    __slots__ = ("buf", "off", "shape", "dtype", "_t")

    def __init__(self, buf, off, shape, dtype):
        self.buf, self.off, self.shape, self.dtype, self._t = buf, off, shape, dtype, None

    def tensor(self) -> torch.Tensor:
        if self._t is None:
            self._t = _view(self.buf, self.off, self.shape, self.dtype)
        return self._t

    def __getattr__(self, name):                    # (behaves like the tensor for the occasional reader)
        return getattr(self.tensor(), name)

    def __getitem__(self, i):
        return self.tensor()[i]


class _StepState:
    """Everything one step's drivers share: the plan, the argument structs, the workspaces (kept alive until the backward
    has been enqueued), the tensors the factored-gradient sink and the optimizer look at afterwards."""
    __slots__ = ("plan", "sample", "main", "ws_sample", "ws_fwd", "out", "S", "R", "n_rows", "H", "he", "first_grid", "hash_planes",
                 "main_code", "main_window", "keep", "grads", "code_width", "lp", "lp_ex")


class _GradBuffers:
    """The parameter gradients a step's backward produces (mlp_head, mlp_base, the 16 deformation tensors, the batch's
    deformation-code rows, its hash-code rows) in ONE persistent fp32 buffer with ready-made views -- laid out by the plan
    (``g_*`` offsets; they depend on the number of code rows and grids only).  Stage 0 of the backward clears it, the
    views of the LEAF parameters (the two fused MLPs, the 16 deformation tensors) become their ``.grad`` directly
    (``_deposit``; what DDP does with ``gradient_as_bucket_view``), the optimizers have read them (same stream) before the
    next backward writes again.  Handing them to autograd instead costs a clone each: the views are referenced from here,
    so ``AccumulateGrad`` copies instead of stealing -- 18 device copies of ~5 us in a row at the end of every backward,
    with their allocations and launches on the host (``profiles/r04_timeline_steady_state_compact.txt``)."""

    SLACK = 1 << 18

    def __init__(self, plan, n_rows: int, H: int, code_deform_shape, deform_shapes, head_hidden: int, base_hidden: int,
                 device):
        f32 = torch.float32
        # (+ slack behind the plan's bytes: a data-parallel step all-reduces the gradients HERE and lets the few gradients
        # that live elsewhere, the presence counts and the flags ride behind them -- engine/parallel.py)
        self.used = plan.grad_bytes // 4
        self.flat = torch.empty((self.used + self.SLACK,), dtype=f32, device=device)
        # [0, fixed): the leaf parameters' gradients (mlp_head, mlp_base, the deformation tensors) -- offsets that depend on
        # the model only, not on the batch; behind it the two code-row gradients autograd consumes inside the backward
        self.fixed = plan.g_code_deform // 4
        self.slack = self.SLACK        # elements behind ``fixed`` every rank's buffer has, whatever its batch
        self.slots = {}                # id(parameter) -> its view (filled by NativeStep: it knows the leaves)

        def cut(off, n):
            return self.flat[off // 4:off // 4 + n]

        self.d_head = cut(plan.g_head, F.mlp_param_count(head_hidden))
        self.d_base = cut(plan.g_base, F.mlp_param_count(base_hidden))
        gparams = cut(plan.g_deform, F.deform_param_count())
        sizes = []
        for shp in deform_shapes:
            n = 1
            for d in shp:
                n *= d
            sizes.append(n)
        self.deform = [gp if len(shp) == 1 else gp.view(shp) for gp, shp in zip(torch.split(gparams, sizes), deform_shapes)]
        self.gtable = cut(plan.g_code_deform, code_deform_shape[0] * code_deform_shape[1]).view(code_deform_shape)
        self.g_code_hash = cut(plan.g_code_hash, n_rows * H).view(n_rows, H)

    def slot_of(self, p):
        return self.slots.get(id(p))


_DEPOSIT = os.environ.get("NSX_GRAD_DEPOSIT", "1") != "0"


def _aliases(t: Optional[torch.Tensor], flat: torch.Tensor) -> bool:
    if t is None or t.device != flat.device:
        return False
    lo = flat.data_ptr()
    return lo <= t.data_ptr() < lo + flat.numel() * 4


def _has_grad_hooks(p: torch.Tensor) -> bool:
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


def _deposit(p: torch.Tensor, view: torch.Tensor) -> None:
    """``AccumulateGrad`` for a leaf whose gradient already sits in its final place.

    Contract (what DDP's ``gradient_as_bucket_view`` also asks of its users): ``p.grad`` is a VIEW of the step's persistent
    gradient buffer -- stage 0 of the next backward clears it, so a reference kept across steps (gradient logging) must be
    cloned by its holder; tensor hooks and post-accumulate hooks do not see a deposited gradient, therefore a leaf that has
    any takes the autograd route (``_NativeMain.backward``), as does every leaf when ``NSX_GRAD_DEPOSIT=0``."""
    if p.grad is None:
        p.grad = view
    else:
        p.grad.add_(view)


class _NativeMain(torch.autograd.Function):
    """The kept samples' main pass (forward: nsx_step_main_fwd, backward: nsx_step_main_bwd stages 0-2) as one autograd
    node with the inputs of ``engine.fused_pass._MainPass``: hash tables, the two fused MLPs, the batch's conditioned
    hash-code rows, its deformation-code rows, the 16 deformation tensors."""

    @staticmethod
    def forward(ctx, st: _StepState, tables_master, base_params, head_params, code_hash, code_deform, *deform_params):
        dev = code_deform.device
        st.ws_fwd = torch.empty((st.plan.fwd_bytes,), dtype=torch.uint8, device=dev)
        st.out = torch.empty((dl.LOSS_OUT,), dtype=torch.float32, device=dev)
        m = st.main
        m.ws_fwd, m.out = st.ws_fwd.data_ptr(), st.out.data_ptr()
        check(lib().nsx_step_main_fwd(C.byref(m), stream()), "nsx_step_main_fwd")
        ctx.st = st
        ctx.leaves = (base_params, head_params) + tuple(deform_params)
        ctx.sink = st.he.grad_sink
        ctx.announced = ctx.needs_input_grad[1] and ctx.sink is not None
        if ctx.announced:
            ctx.sink.expect()
        return st.out

    @staticmethod
    def backward(ctx, g_out):
        st: _StepState = ctx.st
        plan, m, sink = st.plan, st.main, ctx.sink
        dev = g_out.device
        g = g_out.to(torch.float32).contiguous()
        gb: _GradBuffers = st.grads
        leaves = ctx.leaves
        for p in leaves:                 # a gradient accumulated by an earlier backward lives in the buffer stage 0 clears
            if _aliases(p.grad, gb.flat):
                p.grad = p.grad.clone()
        ws_bwd = torch.empty((plan.bwd_bytes,), dtype=torch.uint8, device=dev)
        need_tab = ctx.needs_input_grad[1]
        need_code = bool(ctx.needs_input_grad[4]) and not st.first_grid    # (the code is the constant one in that phase)
        m.grad_out, m.ws_bwd, m.grads = g.data_ptr(), ws_bwd.data_ptr(), gb.flat.data_ptr()
        m.need_code_grad = 1 if need_code else 0
        L, s = lib(), stream()
        check(L.nsx_step_main_bwd(C.byref(m), 0, s), "nsx_step_main_bwd stage 0")
        if st.lp is not None:
            # level-parallel exchange instead of stage 1: dL/dfeatures travels to the levels' owners, whose gradient planes
            # take the table gradient; dL/dx and the code-row gradient come back summed (engine/level_parallel.py)
            S = st.S
            st.lp.backward(_view(st.ws_fwd, plan.f_pn, (S, 3), torch.float32), _view(st.ws_sample, plan.k_slot, (S,), torch.int32),
                           _view(ws_bwd, plan.b_dout, (S, 32), torch.float32),
                           n_dev=_view(st.ws_sample, plan.n_kept, (1,), torch.int64), need_table=bool(need_tab), ex=st.lp_ex,
                           dx_out=_view(ws_bwd, plan.b_dx, (S, 3), torch.float32), dcode_out=gb.g_code_hash)
            check(L.nsx_step_main_bwd(C.byref(m), 2, s), "nsx_step_main_bwd stage 2")
            return _NativeMain._hand_over(ctx, st, gb, leaves, need_code, dev)
        # the factored table gradient of the step (cleared ahead / left clean by the optimizer / cleared here)
        G = None
        if need_tab:
            if st.hash_planes:
                # (compact first-grid phase: every row's code is one -- the planes only spread the atomics)
                G = sink.buffer_for(st.he.first_grid_code(st.hash_planes), None, st.hash_planes, st.he.geom.total_entries,
                                    n_samples=st.S)
            else:
                G = sink.buffer_for(st.main_code, st.main_window, st.n_rows, st.he.geom.total_entries, n_samples=st.S)
        m.G = G.data_ptr() if G is not None else None
        m.nonfinite = sink.nonfinite.data_ptr() if G is not None else None
        m.scatter_separately = 1 if (F.scatter_alone(st.H) and G is not None) else 0
        check(L.nsx_step_main_bwd(C.byref(m), 1, s), "nsx_step_main_bwd stage 1")
        if need_tab and ctx.announced:
            # G is complete, and so are the gradients of the two fused MLPs (the rest of the tables' optimizer group)
            sink.arrived(group_grads=[gb.d_base, gb.d_head] if sink.on_complete is not None else None)
        check(L.nsx_step_main_bwd(C.byref(m), 2, s), "nsx_step_main_bwd stage 2")
        return _NativeMain._hand_over(ctx, st, gb, leaves, need_code, dev)

    @staticmethod
    def _hand_over(ctx, st, gb, leaves, need_code, dev):
        ctx.st = None                    # the workspaces go back to the allocator with this node
        g_code = gb.g_code_hash if need_code else None
        if g_code is not None and st.H != st.code_width:
            full = torch.zeros((st.n_rows, st.code_width), dtype=torch.float32, device=dev)   # (compact window-ramp layout)
            full[:, :st.H] = g_code
            g_code = full
        if not _DEPOSIT or any(_has_grad_hooks(p) for p in leaves):
            # autograd's own accumulation (it clones the referenced views): asked for (A/B knob), or a leaf carries hooks
            # that a deposited gradient would bypass
            ctx.leaves = None
            return (None, None, gb.d_base, gb.d_head, g_code, gb.gtable, *gb.deform)
        # leaf parameters: the views ARE the gradients (see _GradBuffers); the two code lookups are differentiated by autograd
        for k, (p, view) in enumerate(zip(leaves, (gb.d_base, gb.d_head, *gb.deform))):
            if ctx.needs_input_grad[2 + k if k < 2 else 4 + k]:
                _deposit(p, view.view(p.shape) if view.shape != p.shape else view)
        ctx.leaves = None
        return (None, None, None, None, g_code, gb.gtable) + (None,) * len(gb.deform)


class LazyVectorDict(dict):
    """name -> element of one device vector, selected on first access (a ``select`` dispatch per entry otherwise, every
    step, for entries nobody may read)."""

    def __init__(self, vector, terms):
        super().__init__()
        self._vector, self._index = vector, dict(terms)
        for name, _ in terms:
            super().__setitem__(name, None)

    def __getitem__(self, k):
        v = super().__getitem__(k)
        if v is None and k in self._index:
            v = self._vector[self._index[k]]
            super().__setitem__(k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def values(self):
        return [self[k] for k in self.keys()]

    def items(self):
        return [(k, self[k]) for k in self.keys()]


class LazyLossDict(LazyVectorDict):
    """``loss_dict`` of a natively driven step: the terms of ``get_loss_dict`` (nersemble_instant_ngp.py:366-407) as
    elements of the fused loss vector; ``total`` = their sum as the kernel formed it (``reduce(add, values)`` order),
    ``fused`` / ``total_index`` for the trainer's direct backward start."""

    def __init__(self, fused, terms, total_index):
        super().__init__(fused, terms)
        self.fused, self.total_index = fused, total_index
        self._total = None

    @property
    def total(self):
        if self._total is None:
            self._total = self.fused[self.total_index]
        return self._total


class LazyOutputs(dict):
    """``get_outputs``' dict for a natively driven step: the per-ray / per-sample tensors are views of the step's
    workspaces, made when somebody asks (the trainer does not)."""

    def __init__(self, make):
        super().__init__()
        self._make = make

    def _fill(self):
        if self._make is not None:
            make, self._make = self._make, None
            super().update(make())

    def __getitem__(self, k):
        self._fill()
        return super().__getitem__(k)

    def __contains__(self, k):
        self._fill()
        return super().__contains__(k)

    def get(self, k, default=None):
        self._fill()
        return super().get(k, default)

    def keys(self):
        self._fill()
        return super().keys()

    def items(self):
        self._fill()
        return super().items()

    def values(self):
        self._fill()
        return super().values()

    def __iter__(self):
        self._fill()
        return super().__iter__()

    def __len__(self):
        self._fill()
        return super().__len__()


class NativeStep:
    def __init__(self, model):
        self.model = model
        self._plan_cls = _lib.step_struct("nsx_step_plan")
        self._sample_cls = _lib.step_struct("nsx_step_sample")
        self._main_cls = _lib.step_struct("nsx_step_main")
        self._ones_codes = {}
        self._prof_state = (0, -1)
        self._grad_buffers = {}
        self.last_grads = None             # the step's parameter-gradient buffer (the data-parallel all-reduce works in it)

    def forward(self, ray_bundle, batch: Dict[str, torch.Tensor]):
        """(loss_dict, metrics_dict, outputs) of ``fused_train_forward`` -- or None when this step is outside what the
        drivers cover (the caller then takes the per-kernel path).  The caller has established that the bundle's metadata
        code rows stand for its times (``NeRSembleNGPModel._metadata_rows_trusted``): the sigma_fn pass and the main pass
        both index the batch's compacted code tables with the per-ray slot, and the sampler driver re-checks every ray
        against ``times`` on the device (``nsx_check_code_rows``, a sticky flag read at the model's periodic check)."""
        model = self.model
        cfg = model.config
        md = ray_bundle.metadata or {}
        he = model.field.hash_ensemble
        if not (model.reuse_sigma_pass and model.device_sample_counts and "image_index" in md and "_image_timesteps" in md
                and (cfg.alpha_thre > 0 or cfg.early_stop_eps > 0) and not cfg.disable_occupancy_grid
                and (he.grad_sink is not None or he.level_parallel is not None)
                and ray_bundle.nears is None and ray_bundle.fars is None
                and (ray_bundle.times is not None or "timesteps" in md)):
            return None
        uniq = md["_image_timesteps"].reshape(-1).int()
        n_rows = int(uniq.shape[0])
        if n_rows > _lib.NSX_MAX_SLOTS or he.geom.n_levels != 16 or model.field.mlp_base.n_output_dims != 16:
            return None
        alpha_map = batch.get("alpha_map")
        R = len(ray_bundle)
        if alpha_map is not None and not (alpha_map.dtype == torch.uint8 and alpha_map.numel() == R):
            return None
        from ..models.nersemble_instant_ngp import LossDict
        dev = ray_bundle.origins.device
        window_hash = model.sched_window_hash_encodings.value if model.sched_window_hash_encodings is not None else None
        window_deform = model.sched_window_deform.value if model.sched_window_deform is not None else None
        df = model.deformation_field
        emb_d = model.time_embedding_deformation if model.time_embedding_deformation is not None else model.time_embedding
        # the batch's code rows (two embedding lookups + the window conditioning; autograd differentiates them): queued
        # before the sampler so that they run beside the table optimizer instead of behind it
        if window_hash is not None and window_hash == 1 and he.disable_initial_hash_ensemble:
            # hash_ensemble.py:121-123: the code is replaced by ones while the window is 1 -- the lookup's result would not
            # be read, and no gradient reaches the time codes
            key = ("rows", n_rows, he.n_hash_encodings, str(dev))
            ones = self._ones_codes.get(key)
            if ones is None:
                ones = self._ones_codes[key] = torch.ones((n_rows, he.n_hash_encodings), dtype=torch.float32, device=dev)
            code_hash, window = ones, he.window_tensor(window_hash, dev)
        else:
            code_hash, window = he._conditioned(model.time_embedding(uniq), window_hash, dev)
        code_deform = emb_d(uniq)
        # -- traversal, pass 1 (possibly prefetched)
        o = ray_bundle.origins.to(torch.float32).contiguous()
        d = ray_bundle.directions.to(torch.float32).contiguous()
        model.sampler._cull_to_camera_frusta()
        grid = model.occupancy_grid
        far = 1e10 if cfg.far_plane is None else float(cfg.far_plane)
        near_planes, packed_march, S = grid.counted_march(o, d, cfg.near_plane, far, cfg.render_step_size,
                                                          stratified=model.sampler.training)
        if S <= 0:
            return None                     # (nothing marched: the per-kernel path owns the one-fake-sample fallback)
        # level-parallel run: the host-side size exchange of the step starts NOW (gloo, a background thread) and is collected
        # after the sampler's front has been enqueued
        lp_pending = he.level_parallel.exchange_sizes_begin(S, n_rows) if he.level_parallel is not None else None
        grid.last_keep_index, grid.last_n_marched, grid.last_n_kept = None, S, None
        ray_slots = md["image_index"].reshape(-1).to(torch.int32).contiguous()
        ray_times = None
        if ray_bundle.times is not None and ray_bundle.times.numel() == R:
            ray_times = ray_bundle.times.reshape(-1).to(torch.float32).contiguous()
        rows_flag = model._rows_flag_tensor(dev)
        alpha_thre_dev = grid._alpha_threshold(float(cfg.alpha_thre))
        # -- what the HashEnsemble kernels see: the H grids with conditioned codes and the window -- or, in the compact
        # first-grid phase (HashEnsemble.first_grid_phase), the contiguous copy of grid 0 with a constant code of one
        lp = he.level_parallel
        width = he.compact_width(window_hash) if lp is None else 0
        first = width == 1
        # (the previous step's table optimizer may still be running on its stream: only the HashEnsemble kernel waits for it --
        # the traversal and the deformation field of this step run beside it, as on the per-kernel path)
        if first:
            comp = he.enter_first_grid_phase()
            tables, Hk = comp["f16"], 1
            main_code, main_window = he.first_grid_code(n_rows), None
        else:
            if width >= 2:
                # window ramp: the first `width` grids as a contiguous copy, codes and window as in the full layout
                tables, Hk = he.enter_compact(width)["f16"], width
            else:
                he.leave_first_grid_phase()
                tables, Hk = he.half_tables(wait=False), he.n_hash_encodings
            main_code, main_window = code_hash.detach().contiguous(), window
            if lp is not None and main_window is None:
                raise RuntimeError("level-parallel HashEnsemble needs the window schedule (window_hash_encodings)")
        # (the sigma_fn pass reads the SAME rows as the main pass: the batch's conditioned code rows through the per-ray slot;
        # rounds 3-4 conditioned the dataset's whole [T, H] table for it every step)
        code_d = code_deform.detach().contiguous()
        mb, mh = model.field.mlp_base, model.field.mlp_head
        packed_w = df.packed_params()
        w7 = F.deform_window7(window_deform)
        L = lib()
        prof = _lib.profiler
        want_prof = (1, prof.tag if prof.tag is not None else -1) if prof.enabled else (0, -1)
        if want_prof != self._prof_state:                # HIP events around the drivers' kernel calls (bench.py)
            L.nsx_step_profile(*want_prof)
            self._prof_state = want_prof
        plan = self._plan_cls()
        check(L.nsx_step_plan_make(S, R, n_rows, Hk, mb.n_hidden_mats, mh.n_hidden_mats, C.byref(plan)), "nsx_step_plan_make")
        ws_sample = torch.empty((plan.sample_bytes,), dtype=torch.uint8, device=dev)
        binary = grid.binaries[0].contiguous().view(torch.uint8)
        base_w16, head_w16 = mb.half_weights(), mh.half_weights()
        field_aabb, deform_aabb, occ_aabb = model.field._aabb6(), df._aabb6(), grid._aabb6()
        a = self._sample_cls()
        a.origins, a.directions, a.near_planes = o.data_ptr(), d.data_ptr(), near_planes.data_ptr()
        a.packed_march, a.binaries = packed_march.data_ptr(), binary.data_ptr()
        kept = grid.last_march_stash               # the counting pass ran a step ahead and kept the samples' starts
        a.march_stash, a.march_stash_cap = (kept[0].data_ptr(), kept[1]) if kept is not None else (None, 0)
        a.ray_slots = ray_slots.data_ptr()
        if ray_times is not None:
            a.ray_times, a.row_timesteps, a.rows_flag = ray_times.data_ptr(), uniq.data_ptr(), rows_flag.data_ptr()
        a.deform_packed, a.deform_codes = packed_w.data_ptr(), code_d.data_ptr()
        a.tables, a.geom = tables.data_ptr(), C.addressof(he.geom)
        a.hash_codes = main_code.data_ptr()
        a.hash_window = main_window.data_ptr() if main_window is not None else None
        a.base_w16, a.alpha_thre_dev = base_w16.data_ptr(), alpha_thre_dev.data_ptr()
        a.window7_host = C.addressof(w7) if w7 is not None else None
        a.ws, a.plan = ws_sample.data_ptr(), C.addressof(plan)
        tables_event = he.take_tables_event()
        a.tables_ready_event = tables_event.cuda_event if tables_event is not None else None
        a.R, a.S = R, S
        a.deform_code_stride, a.hash_code_stride = code_d.stride(0), main_code.stride(0)
        a.grid_res, a.H = grid._res, Hk
        a.n_code_rows, a.n_timesteps = n_rows, int(cfg.n_timesteps)
        a.base_hidden, a.base_out_dim, a.base_act = mb.n_hidden_mats, mb.n_output_dims, mb.out_act
        a.far_plane, a.step, a.early_stop_eps = far, float(cfg.render_step_size), float(cfg.early_stop_eps)
        for i in range(6):
            a.occ_aabb[i], a.deform_aabb[i], a.field_aabb[i] = occ_aabb[i], deform_aabb[i], field_aabb[i]
        lp_ex = None
        if lp is None:
            check(L.nsx_step_sample_run(C.byref(a), stream()), "nsx_step_sample_run")
        else:
            # level-parallel: the sampler up to the normalised positions, the sample exchange (this rank's levels for every
            # rank's samples; the features of ALL levels come back into the workspace), the rest of the sampler
            a.phase = 1
            check(L.nsx_step_sample_run(C.byref(a), stream()), "nsx_step_sample_run (front)")
            lp_ex = lp.exchange_sizes_end(lp_pending)            # (started as soon as S was known: see above)
            if tables_event is not None:
                torch.cuda.current_stream(dev).wait_event(tables_event)
            lp.features(_view(ws_sample, plan.m_pn, (S, 3), torch.float32), main_code,
                        _view(ws_sample, plan.m_slot, (S,), torch.int32), main_window, ex=lp_ex,
                        out=_view(ws_sample, plan.m_feat, (S, 32), torch.float16))
            a.phase = 2
            check(L.nsx_step_sample_run(C.byref(a), stream()), "nsx_step_sample_run (back)")
        if not first and he.grad_sink is not None:
            # (the sampler's sigma_fn pass is queued: from here to the HashEnsemble's backward only small kernels run)
            he.grad_sink.clear_ahead(n_rows, he.geom.total_entries, dev)
        # -- the kept samples' main pass: one autograd node
        m = self._main_cls()
        m.ws_sample = ws_sample.data_ptr()
        image_t = batch["image"].to(torch.float32).contiguous()
        m.image = image_t.data_ptr()
        amap = alpha_map.reshape(-1).contiguous() if alpha_map is not None else None
        m.alpha_map = amap.data_ptr() if amap is not None else None
        depth_t = batch["depth_maps"].to(torch.float32).reshape(-1).contiguous()
        m.depth_targets = depth_t.data_ptr()
        m.tables, m.geom = tables.data_ptr(), C.addressof(he.geom)
        m.code_hash = main_code.data_ptr()
        m.hash_window = main_window.data_ptr() if main_window is not None else None
        m.deform_packed, m.code_deform = packed_w.data_ptr(), code_d.data_ptr()
        m.base_w16, m.head_w16 = base_w16.data_ptr(), head_w16.data_ptr()
        m.window7_host = C.addressof(w7) if w7 is not None else None
        m.plan = C.addressof(plan)
        m.R, m.S = R, S
        m.code_hash_stride, m.code_deform_stride = main_code.stride(0), code_d.stride(0)
        m.max_ray = int(cfg.dist_loss_max_rays)
        m.H, m.n_code_rows = Hk, n_rows
        hash_planes = he.first_grid_planes(n_rows, S) if first else 0
        m.hash_planes = hash_planes
        m.base_hidden, m.base_out_dim, m.base_act = mb.n_hidden_mats, mb.n_output_dims, mb.out_act
        m.head_hidden, m.head_act, m.geo_dim = mh.n_hidden_mats, mh.out_act, model.field.geo_feat_dim
        m.use_masked = 1 if cfg.use_masked_rgb_loss else 0
        m.background = 1.0 if cfg.background_color == "white" else 0.0
        m.thr, m.l_alpha = float(cfg.alpha_mask_threshold), float(cfg.lambda_alpha_loss or 0.0)
        m.l_depth, m.l_dist = float(cfg.lambda_depth_loss or 0.0), float(cfg.lambda_dist_loss)
        m.l_empty, m.l_near = float(cfg.lambda_empty_loss), float(cfg.lambda_near_loss)
        m.eps = float(model.sched_eps_depth.value)
        for i in range(6):
            m.field_aabb[i], m.deform_aabb[i] = field_aabb[i], deform_aabb[i]
        st = _StepState()
        st.plan, st.sample, st.main, st.ws_sample = plan, a, m, ws_sample
        st.S, st.R, st.n_rows, st.H, st.he, st.first_grid = S, R, n_rows, Hk, he, first
        st.main_code, st.main_window, st.hash_planes = main_code, main_window, hash_planes
        st.code_width = int(code_hash.shape[1])
        st.lp, st.lp_ex = lp, lp_ex
        deform_params = df.ordered_params()
        gkey = (n_rows, Hk, str(dev), plan.grad_bytes)
        st.grads = self._grad_buffers.get(gkey)
        if st.grads is None:
            self._grad_buffers = {gkey: _GradBuffers(plan, n_rows, Hk, tuple(code_d.shape),
                                                     [tuple(p.shape) for p in deform_params], mh.n_hidden_mats,
                                                     mb.n_hidden_mats, dev)}
            st.grads = self._grad_buffers[gkey]
            st.grads.slots = self._slots(st.grads, mb, mh, deform_params)
        self.last_grads = st.grads
        # every tensor a raw pointer above borrows lives at least as long as the step's state
        st.keep = (o, d, near_planes, packed_march, binary, ray_slots, ray_times, uniq, rows_flag, packed_w, tables,
                   main_code, main_window, base_w16, head_w16, alpha_thre_dev, w7, image_t, amap, depth_t, code_d, he.geom,
                   tables_event)
        fused = _NativeMain.apply(st, he.tables, mb.params, mh.params, code_hash, code_deform, *deform_params)
        grid.last_n_kept = _LazyView(ws_sample, plan.n_kept, (1,), torch.int64)
        terms = [("rgb_loss", dl.LOSS_RGB)]
        if alpha_map is not None and cfg.lambda_alpha_loss is not None and cfg.lambda_alpha_loss > 0:
            terms.append(("alpha_loss", dl.LOSS_ALPHA))
        if cfg.lambda_dist_loss > 0:
            terms.append(("dist_loss", dl.LOSS_DIST))
        terms += [("empty_loss", dl.LOSS_EMPTY), ("near_loss", dl.LOSS_NEAR)]
        if cfg.lambda_depth_loss > 0:
            terms.append(("depth_loss", dl.LOSS_DEPTH))
        if model.global_loss_normalisers is not None:
            loss_dict = LossDict()
            for name, idx in terms:
                loss_dict[name] = fused[idx]
            loss_dict.total = fused[dl.LOSS_TOTAL]
            model._apply_global_normalisers(loss_dict, fused, R)
        else:
            # the entries are elements of ONE vector: selected when somebody reads them (the trainer starts the backward at
            # the vector itself, NativeGradScaler.loss_grad_vector, and hands the dict on)
            loss_dict = LazyLossDict(fused, terms, dl.LOSS_TOTAL)
        mterms = [("psnr", dl.LOSS_PSNR), ("num_samples_per_batch", dl.LOSS_NUM_SAMPLES)]
        if alpha_map is not None:
            mterms.append(("psnr_masked", dl.LOSS_PSNR_MASKED))
        metrics = LazyVectorDict(fused.detach(), mterms)
        return loss_dict, metrics, LazyOutputs(lambda: self._outputs(st))

    @staticmethod
    def _slots(gb, mb, mh, deform_params) -> dict:
        return {id(mb.params): gb.d_base, id(mh.params): gb.d_head, **{id(p): v for p, v in zip(deform_params, gb.deform)}}

    def grad_arena(self):
        """The persistent parameter-gradient buffer (``_GradBuffers``) for the data-parallel all-reduce -- the last step's, or
        (a rank that has not run the drivers yet) one laid out for a single code row: the leaf gradients' offsets depend on
        the model only."""
        if self.last_grads is not None:
            return self.last_grads
        model = self.model
        he, df = model.field.hash_ensemble, model.deformation_field
        mb, mh = model.field.mlp_base, model.field.mlp_head
        plan = self._plan_cls()
        check(lib().nsx_step_plan_make(1, 1, 1, he.n_hash_encodings, mb.n_hidden_mats, mh.n_hidden_mats, C.byref(plan)),
              "nsx_step_plan_make")
        deform_params = df.ordered_params()
        gb = _GradBuffers(plan, 1, he.n_hash_encodings, (1, 128), [tuple(p.shape) for p in deform_params], mh.n_hidden_mats,
                          mb.n_hidden_mats, mb.params.device)
        gb.slots = self._slots(gb, mb, mh, deform_params)
        self.last_grads = gb
        return gb

    @staticmethod
    def _outputs(st: _StepState) -> dict:
        """The ``get_outputs`` dict (nersemble_instant_ngp.py:345-362) as views of the step's workspaces; per-sample
        arrays keep the marched capacity, their first ``n_kept`` rows are valid."""
        from ..rays import Frustums, RaySamples
        p, ws, wf, S, R = st.plan, st.ws_sample, st.ws_fwd, st.S, st.R
        f32 = torch.float32
        offsets = _view(ws, p.k_off, (S, 3), f32)
        fr = Frustums(origins=_view(ws, p.k_org, (S, 3), f32), directions=_view(ws, p.k_dir, (S, 3), f32),
                      starts=_view(ws, p.k_t0, (S, 1), f32), ends=_view(ws, p.k_t1, (S, 1), f32),
                      pixel_area=torch.zeros((1, 1), device=ws.device).expand(S, 1), offsets=offsets)
        samples = RaySamples(frustums=fr, metadata={"image_index": _view(ws, p.k_slot, (S, 1), torch.int32)})
        packed = _view(ws, p.k_packed, (R, 2), torch.int64)
        return {"rgb": _view(wf, p.f_rgb, (R, 3), f32), "accumulation": _view(wf, p.f_acc, (R, 1), f32),
                "depth": _view(wf, p.f_depth, (R, 1), f32), "num_samples_per_ray": packed[:, 1],
                "ray_samples": (samples,), "ray_indices": (_view(ws, p.k_ri, (S,), torch.int64),),
                "weights": (_view(wf, p.f_w, (S, 1), f32),), "packed_info": (packed,),
                "deformation": _view(wf, p.f_aux, (R, 3), f32), "n_kept": _view(ws, p.n_kept, (1,), torch.int64)}
