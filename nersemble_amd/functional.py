"""torch.autograd glue over the C ABI of libnsx.so (include/nsx.h).

Every function here hands raw device pointers to the native library; there is no eager/CPU
fallback.  Gradients between kernels travel in fp32 (they are a negligible fraction of the gather
traffic), activations in fp16 like the reference's tcnn path.
"""
import ctypes as C
import os
from typing import Optional

import torch

from . import _lib
from ._lib import GridGeom, check, lib, ndev, ptr, stream


def scatter_alone(H: int) -> bool:
    """Padded grid counts whose fused backward has ONE lane per sample (table rows of <= 16 bytes: H <= 4): its 16
    (corner, feature) gradient items per sample and level leave as 16 instructions of 64 unrelated sectors each, while
    ``nsx_hash_ensemble_bwd_scatter`` keeps the 8-lanes-per-sample mapping whose (feature, x) neighbours share a sector.
    Measured at 0.95 M samples: H = 1 5.5 -> 1.8 ms, H = 4 (window ramp, compact width 4) 7.9 ms fused."""
    return H <= 4


# ------------------------------------------------------------------------------------------------
# table layout conversion (checkpoint compatibility with the reference's tcnn state dict)
# ------------------------------------------------------------------------------------------------
def tcnn_param_shape(H: int, geom: GridGeom):
    total = 2 * H
    f_enc = 8 if total >= 8 else total
    c = (total + 7) // 8
    return c, geom.total_entries, f_enc


def tables_from_tcnn(tcnn_params: torch.Tensor, H: int, geom: GridGeom, want_master: bool = True):
    """tcnn layout fp32 ``[C, total, F_enc]`` -> native ``[total, 2, Hp]`` (fp16 working copy, fp32 master)."""
    c, total, f_enc = tcnn_param_shape(H, geom)
    assert tcnn_params.numel() == c * total * f_enc, "tcnn params have the wrong size"
    src = tcnn_params.detach().to(torch.float32).contiguous()
    Hp = _lib.padded_grids(H)
    f16 = torch.empty((total, 2, Hp), dtype=torch.float16, device=src.device)
    master = torch.empty((total, 2, Hp), dtype=torch.float32, device=src.device) if want_master else None
    check(lib().nsx_tables_from_tcnn(ptr(src), H, C.byref(geom), ptr(f16), ptr(master), stream()),
          "nsx_tables_from_tcnn")
    return f16, master


def tables_to_tcnn(native_f32: torch.Tensor, H: int, geom: GridGeom) -> torch.Tensor:
    c, total, f_enc = tcnn_param_shape(H, geom)
    out = torch.empty((c, total, f_enc), dtype=torch.float32, device=native_f32.device)
    check(lib().nsx_tables_to_tcnn(ptr(native_f32.detach().contiguous(), torch.float32), H, C.byref(geom), ptr(out),
                                   stream()), "nsx_tables_to_tcnn")
    return out


def hash_indices(x: torch.Tensor, geom: GridGeom) -> torch.Tensor:
    """uint32 level-local entry indices ``[B, L, 8]`` (returned as int64 for convenience)."""
    x = x.detach().to(torch.float32).contiguous()
    out = torch.empty((x.shape[0], geom.n_levels, 8), dtype=torch.int32, device=x.device)
    check(lib().nsx_hash_indices(ptr(x), x.shape[0], C.byref(geom), ptr(out), stream()), "nsx_hash_indices")
    return out.to(torch.int64) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------
# HashEnsemble
# ------------------------------------------------------------------------------------------------
def _hash_ensemble_fwd_raw(x, tables_f16, H, geom, code, code_index, window):
    B = x.shape[0]
    out = torch.empty((B, 2 * geom.n_levels), dtype=torch.float16, device=x.device)
    check(lib().nsx_hash_ensemble_fwd(ptr(x, torch.float32), B, ptr(tables_f16, torch.float16), H, C.byref(geom),
                                      ptr(code, torch.float32), code.stride(0), ptr(code_index, torch.int32),
                                      ptr(window, torch.float32), ptr(out), ndev(B), stream()), "nsx_hash_ensemble_fwd")
    return out


_CODESUM_SCRATCH = {}


def codesum_scratch(n_rows: int, H: int, device) -> torch.Tensor:
    """Block partials of the in-kernel code-gradient sums (nsx_hash_ensemble_bwd_codesum); one cached buffer per
    device, grown on demand (stream-ordered reuse: every user launches on torch's current stream)."""
    n = int(lib().nsx_hash_codesum_scratch_floats(int(n_rows), int(H)))
    key = str(device)
    buf = _CODESUM_SCRATCH.get(key)
    if buf is None or buf.numel() < n:
        buf = _CODESUM_SCRATCH[key] = torch.empty((n,), dtype=torch.float32, device=device)
    return buf


class FactoredGradSink:
    """Collects the factored table gradient G[e][slot][f] (+ the code rows it factors through) instead of a dense
    1.6 GB table gradient; consumed by ``engine.hash_adam.HashTableAdam`` which forms the gradient on the fly.
    Backward calls of one step that share the code table (the chunks of a step) accumulate into one G."""

    def __init__(self):
        self.entries = []            # dicts: G, code, window, n_rows, key
        self._cache = {}
        self.nonfinite = None        # device float: set by the backward kernel when it adds an inf/NaN to a G
        self.pending = 0             # forwards recorded for autograd whose backward has not run yet
        self.pre_cleared = None      # (G, event): the cached buffer was cleared ahead of time on another stream
        self.consumed = None         # event after the LAST reader of G on another stream (the table optimizer's pass):
        #                              whoever rewrites the cached buffer off the main stream orders itself behind it
        self.samples_scattered = 0   # samples whose gradient the step's backward calls add to G (0: unknown)
        self.clear_ahead_enabled = os.environ.get("NSX_CLEAR_G_AHEAD", "1") == "1"
        self._fill_stream = None
        self.on_complete = None      # called inside the backward once the LAST pending one has added its share to G
        # Optional (NSX_SPLIT_SCATTER=1): the scatter into G as its own kernel on its own stream beside the gather half of
        # the HashEnsemble backward and the deformation field's backward (which only needs the gather's dL/dx).
        # MEASURED SLOWER than the fused kernel (9.3-9.5 vs 8.7 ms per step): the memory-side atomics
        # that bound the scatter also slow every bandwidth-bound kernel that runs beside them, and the fused kernel
        # already hides its table reads under them.  Kept for the stand-alone timings of the two halves (bench.py
        # kernels_alone) and as a tested entry point.  Consumers of G / ``nonfinite`` call ``wait_scatter()``.
        self.split_scatter = os.environ.get("NSX_SPLIT_SCATTER", "0") == "1"
        self.scatter_blocks_per_cu = int(os.environ.get("NSX_SCATTER_BLOCKS", "8"))
        self.scatter_stream = None
        self.scatter_done = None
        self.group_grads = None

    def wait_scatter(self, stream=None) -> None:
        """Order ``stream`` (default: the current one) after the scatter kernels launched so far."""
        ev = self.scatter_done
        if ev is not None:
            (stream if stream is not None else torch.cuda.current_stream()).wait_event(ev)

    def expect(self) -> None:
        self.pending += 1

    def arrived(self, group_grads=None) -> None:
        """One recorded forward has scattered its gradient; the table gradient of the step is complete when none is
        left (the data-parallel optimizer starts its reduce-scatter from here, the single-GPU one its whole step --
        beside the rest of the backward).  ``group_grads``: the (scaled) gradients of the OTHER parameters of the
        tables' optimizer group if the caller has them already (GradScaler skips a group as a whole); without them the
        completion hook must not step on its own."""
        if self.pending > 0:
            self.pending -= 1
        self.group_grads = group_grads if self.pending == 0 else None
        if self.pending == 0 and self.on_complete is not None and self.entries:
            self.on_complete()

    def buffer_for(self, code: torch.Tensor, window: Optional[torch.Tensor], n_rows: int, total_entries: int,
                   zero: bool = True, n_samples: int = 0):
        """The G this backward adds to.  ``zero=False``: a NEW buffer is returned un-cleared with ``fresh`` set in its
        entry -- the caller clears it right in front of its scatter, on the scatter's stream (freshly written zero lines
        are what the atomics then hit in the Infinity Cache)."""
        key = (code.data_ptr(), n_rows, None if window is None else window.data_ptr())
        if not self.entries:
            self.samples_scattered = 0
        self.samples_scattered += int(n_samples)        # (an upper bound under device-side counts: the capacity)
        for e in self.entries:
            if e["key"] == key:
                return e["G"]
        if self.nonfinite is None or self.nonfinite.device != code.device:
            self.nonfinite = torch.zeros((1,), dtype=torch.float32, device=code.device)
        elif not self.entries:
            self.nonfinite.zero_()             # first backward of a step
        G = self._cache.get(n_rows)
        if G is None or G.device != code.device:
            G = torch.empty((n_rows, total_entries, 2), dtype=torch.float32, device=code.device)
            self._cache = {n_rows: G}          # keep at most one persistent buffer
        if any(e["G"] is G for e in self.entries):
            G = torch.empty_like(G)
        pre = self.pre_cleared
        cleared = pre is not None and pre[0] is G
        if cleared:
            # left all zeros by the optimizer pass that consumed it (mark_cleared)
            self.pre_cleared = None
            torch.cuda.current_stream(G.device).wait_event(pre[1])
        elif zero:
            G.zero_()
        if pre is not None and not cleared and not self.entries:
            self.pre_cleared = None            # a buffer cleared ahead for a step that then asked for another one
        self.entries.append({"G": G, "code": code, "window": window, "n_rows": n_rows, "key": key,
                             "fresh": not zero and not cleared})
        return G

    def clear_ahead(self, n_rows: int, total_entries: int, device) -> None:
        """Clear the step's gradient buffer on a side stream, starting NOW (everything queued on the current stream so far
        is waited for): called once the sampler's sigma_fn pass is queued, the 1.6 GB fill then runs beside the small
        kernels of the main pass's forward and the head of its backward instead of in front of the scatter.  Nothing
        to do when the optimizer left the buffer clean (mark_cleared) or a backward of this step already holds it."""
        if not self.clear_ahead_enabled or self.entries or self.pre_cleared is not None:
            return
        G = self._cache.get(n_rows)
        if G is None or G.device != torch.device(device) or G.shape[1] != total_entries:
            G = torch.empty((n_rows, total_entries, 2), dtype=torch.float32, device=device)
            self._cache = {n_rows: G}
        main = torch.cuda.current_stream(G.device)
        side = self._fill_stream
        if side is None or side.device != G.device:
            side = self._fill_stream = torch.cuda.Stream(G.device)
        side.wait_stream(main)
        if self.consumed is not None:
            # the previous step's table optimizer reads this very buffer on ITS stream; the main stream is ordered behind
            # it only through HashEnsemble.wait_tables() in the sigma_fn forward -- do not rely on that call having run
            side.wait_event(self.consumed)
        with torch.cuda.stream(side):
            G.zero_()
            ev = torch.cuda.Event()
            ev.record(side)
        G.record_stream(side)
        self.pre_cleared = (G, ev)

    def is_persistent(self, G: torch.Tensor) -> bool:
        """Is ``G`` the buffer the next step's ``buffer_for`` will hand out again?"""
        return any(G is c for c in self._cache.values())

    def mark_cleared(self, G: torch.Tensor) -> None:
        """The table optimizer consumed ``G`` and left it all zeros (nsx_adam_hash_factored_consume), on the CURRENT
        stream: the next ``buffer_for`` waits for that point instead of clearing the buffer."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(G.device))
        self.pre_cleared = (G, ev)

    def clear(self):
        self.entries = []
        self.pending = 0
        self.scatter_done = None


class _HashEnsembleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tables_master, tables_f16, code, code_index, window, H, geom, sink=None, precomputed=None):
        ctx.sink = sink
        x = x.detach().to(torch.float32).contiguous()
        code_c = code.detach().to(torch.float32).contiguous()
        if precomputed is not None:
            # forward value already produced by the no-grad sigma_fn pass of the same step on the same inputs
            out = precomputed.detach().clone() if precomputed.requires_grad else precomputed.detach()
        else:
            out = _hash_ensemble_fwd_raw(x, tables_f16, H, geom, code_c, code_index, window)
        ctx.save_for_backward(x, tables_f16, code_c, code_index, window)
        ctx.H, ctx.geom = H, geom
        ctx.master_shape = tables_master.shape
        ctx.code_rows = code.shape[0]
        # a forward whose backward will add to the sink's G (the sink counts them to know when G is complete)
        ctx.announced = (sink is not None and ctx.needs_input_grad[1] and code_index is not None
                         and code.shape[0] <= _lib.NSX_MAX_SLOTS)
        if ctx.announced:
            sink.expect()
        return out

    @staticmethod
    def backward(ctx, dout):
        x, tables_f16, code, code_index, window = ctx.saved_tensors
        H, geom = ctx.H, ctx.geom
        B = x.shape[0]
        need_x, need_tab, _, need_code = ctx.needs_input_grad[0], ctx.needs_input_grad[1], None, ctx.needs_input_grad[3]
        dout = dout.to(torch.float32).contiguous()
        n_rows = code.shape[0]
        factored = code_index is not None and n_rows <= _lib.NSX_MAX_SLOTS
        # code rows shared by many samples: their gradient is summed per row inside the kernel (no [B, H] tensor)
        dcode_rows = torch.empty((n_rows, H), dtype=torch.float32, device=x.device) if (need_code and factored) else None
        dcode_s = torch.empty((B, H), dtype=torch.float32, device=x.device) if (need_code and not factored) else None
        dx = torch.empty((B, 3), dtype=torch.float32, device=x.device) if need_x else None
        if factored:
            # factored table gradient: scatter 2 scalars per corner into G[e][slot][f], then expand with the codes
            dtab = None
            G = None
            use_sink = ctx.sink is not None and need_tab
            split = use_sink and ctx.sink.split_scatter and x.is_cuda
            if use_sink:
                G = ctx.sink.buffer_for(code, window, n_rows, geom.total_entries, zero=not split, n_samples=x.shape[0])
            elif need_tab:
                G = torch.zeros((n_rows, geom.total_entries, 2), dtype=torch.float32, device=x.device)
            if split:
                sink = ctx.sink
                if sink.scatter_stream is None or sink.scatter_stream.device != x.device:
                    sink.scatter_stream = torch.cuda.Stream(x.device)
                side, cur = sink.scatter_stream, torch.cuda.current_stream(x.device)
                side.wait_stream(cur)                                   # dout / x / nonfinite are ready on `cur`
                entry = next(e for e in sink.entries if e["G"] is G)
                with torch.cuda.stream(side):
                    if entry["fresh"]:
                        G.zero_()
                        entry["fresh"] = False
                    check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), B, C.byref(geom), n_rows, ptr(code_index), ptr(dout),
                                                              ptr(G), ptr(sink.nonfinite), sink.scatter_blocks_per_cu,
                                                              ndev(B),
                                                              stream()), "nsx_hash_ensemble_bwd_scatter")
                    sink.scatter_done = torch.cuda.Event()
                    sink.scatter_done.record(side)
                for t in (x, dout, code_index, G):                      # allocated on `cur`, read on `side`
                    t.record_stream(side)
                G = None                                                # the gather half below adds nothing to G
            elif G is not None and scatter_alone(H) and x.is_cuda and use_sink:
                # <= 4 grids: the fused kernel's instance has one lane per sample and issues the 16 (corner, feature)
                # items of a sample and level as 16 instructions of unrelated sectors; the stand-alone scatter keeps the
                # 8-lanes-per-sample mapping whose neighbouring items share a sector (3x fewer sector atomics)
                check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), B, C.byref(geom), n_rows, ptr(code_index), ptr(dout),
                                                          ptr(G), ptr(ctx.sink.nonfinite), 8, ndev(B), stream()),
                      "nsx_hash_ensemble_bwd_scatter")
                G = None
            nonfinite = ptr(ctx.sink.nonfinite) if (use_sink and G is not None) else None
            if dcode_rows is not None:
                check(lib().nsx_hash_ensemble_bwd_codesum(ptr(x), B, ptr(tables_f16), H, C.byref(geom), ptr(code),
                                                          code.stride(0), n_rows, ptr(code_index), ptr(window),
                                                          ptr(dout), ptr(G), ptr(dcode_rows),
                                                          ptr(codesum_scratch(n_rows, H, x.device)), ptr(dx), nonfinite,
                                                          ndev(B),
                                                          stream()), "nsx_hash_ensemble_bwd_codesum")
            else:
                check(lib().nsx_hash_ensemble_bwd_factored(ptr(x), B, ptr(tables_f16), H, C.byref(geom), ptr(code),
                                                           code.stride(0), n_rows, ptr(code_index), ptr(window),
                                                           ptr(dout), ptr(G), None, ptr(dx), nonfinite, ndev(B), stream()),
                      "nsx_hash_ensemble_bwd_factored")
            if use_sink and ctx.announced:
                ctx.sink.arrived()
            if need_tab and not use_sink:
                dtab = torch.empty(ctx.master_shape, dtype=torch.float32, device=x.device)
                check(lib().nsx_hash_grad_expand(ptr(G), n_rows, ptr(code), code.stride(0), ptr(window), H,
                                                 C.byref(geom), ptr(dtab), 0, stream()), "nsx_hash_grad_expand")
                del G
        else:
            dtab = torch.zeros(ctx.master_shape, dtype=torch.float32, device=x.device) if need_tab else None
            check(lib().nsx_hash_ensemble_bwd(ptr(x), B, ptr(tables_f16), H, C.byref(geom), ptr(code),
                                              code.stride(0), ptr(code_index), ptr(window), ptr(dout), ptr(dtab),
                                              ptr(dcode_s), ptr(dx), ndev(B), stream()), "nsx_hash_ensemble_bwd")
        dcode = None
        if need_code:
            if dcode_rows is not None:
                dcode = dcode_rows                       # window chain rule and the per-row sums done by the kernels
                if dcode.shape[1] != code.shape[1]:      # compact window-ramp layout: the kernel ran with H < code width
                    full = torch.zeros((n_rows, code.shape[1]), dtype=torch.float32, device=x.device)
                    full[:, :H] = dcode
                    dcode = full
            else:
                if window is not None:
                    dcode_s = dcode_s * window[None, :H]
                if dcode_s.shape[1] != code.shape[1]:
                    dcode_s = torch.nn.functional.pad(dcode_s, (0, code.shape[1] - dcode_s.shape[1]))
                if code_index is not None:               # more rows than NSX_MAX_SLOTS: per-sample gradient + index_add_
                    dcode = torch.zeros((ctx.code_rows, H), dtype=torch.float32, device=x.device)
                    dcode.index_add_(0, code_index.to(torch.int64), dcode_s)
                else:
                    dcode = dcode_s
        return dx, dtab, None, dcode, None, None, None, None, None, None


def hash_ensemble(x: torch.Tensor, tables_master: torch.Tensor, tables_f16: torch.Tensor, code: torch.Tensor,
                  H: int, geom: GridGeom, code_index: Optional[torch.Tensor] = None,
                  window: Optional[torch.Tensor] = None, sink: Optional[FactoredGradSink] = None,
                  precomputed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused HashEnsemble forward (hash_ensemble.py:93-158), differentiable w.r.t. x, tables_master, code.

    x [B,3] fp32 in [0,1); code fp32 rows of H values (row b, or row code_index[b]); window [H] fp32 or None.
    Returns [B, 2*n_levels] fp16.
    """
    if code_index is not None:
        code_index = code_index.to(torch.int32).contiguous()
    if window is not None:
        window = window.to(device=x.device, dtype=torch.float32).contiguous()
    return _HashEnsembleFn.apply(x, tables_master, tables_f16, code, code_index, window, H, geom, sink, precomputed)


# ------------------------------------------------------------------------------------------------
# fully fused MLPs (tcnn FullyFusedMLP equivalents)
# ------------------------------------------------------------------------------------------------
def mlp_param_count(n_hidden_mats: int) -> int:
    return int(lib().nsx_mlp_param_count(n_hidden_mats))


def _seg(t, width_dtype):
    return t


def f32_to_f16(params: torch.Tensor) -> torch.Tensor:
    """Flat fp32 parameters -> fp16 (round to nearest even), one launch."""
    src = params.detach().contiguous()
    w16 = torch.empty(src.numel(), dtype=torch.float16, device=src.device)
    check(lib().nsx_f32_to_f16(ptr(src, torch.float32), ptr(w16), src.numel(), stream()), "nsx_f32_to_f16")
    return w16


class _FusedMLPFn(torch.autograd.Function):
    """out = MLP([a * a_mul + a_add (fp32 segment), b[:, b_off:b_off+b_dim] (fp16 segment)]); see include/nsx.h."""

    @staticmethod
    def forward(ctx, params, a, b, n_hidden_mats, a_mul, a_add, b_off, b_dim, n_out, out_act, precomputed=None,
                w16=None):
        dev = params.device
        if w16 is None:
            w16 = f32_to_f16(params)
        a_c = a.detach().to(torch.float32).contiguous() if a is not None else None
        b_c = b.detach().to(torch.float16).contiguous() if b is not None else None
        B = a_c.shape[0] if a_c is not None else b_c.shape[0]
        a_dim = a_c.shape[1] if a_c is not None else 0
        if precomputed is not None:
            out = precomputed.detach()
        else:
            out = torch.empty((B, n_out), dtype=torch.float16, device=dev)
            check(lib().nsx_mlp_fwd(ptr(w16), n_hidden_mats, B,
                                    ptr(a_c), a_c.stride(0) if a_c is not None else 0, a_dim, a_mul, a_add,
                                    ptr(b_c), b_c.stride(0) if b_c is not None else 0, b_off,
                                    b_dim if b_c is not None else 0, n_out, out_act, ptr(out), out.stride(0), ndev(B), stream()),
                  "nsx_mlp_fwd")
        ctx.save_for_backward(w16, a_c, b_c)
        ctx.cfg = (n_hidden_mats, a_mul, a_add, b_off, b_dim, n_out, out_act, a_dim)
        ctx.b_shape = b.shape if b is not None else None
        return out

    @staticmethod
    def backward(ctx, dout):
        w16, a_c, b_c = ctx.saved_tensors
        n_hidden_mats, a_mul, a_add, b_off, b_dim, n_out, out_act, a_dim = ctx.cfg
        dev = w16.device
        dout = dout.to(torch.float16).contiguous()
        B = dout.shape[0]
        dW = torch.zeros(w16.numel(), dtype=torch.float32, device=dev)
        need_a = a_c is not None and ctx.needs_input_grad[1]
        need_b = b_c is not None and ctx.needs_input_grad[2]
        da = torch.empty((B, a_dim), dtype=torch.float32, device=dev) if need_a else None
        db = torch.zeros(ctx.b_shape, dtype=torch.float16, device=dev) if need_b else None
        check(lib().nsx_mlp_bwd(ptr(w16), n_hidden_mats, B,
                                ptr(a_c), a_c.stride(0) if a_c is not None else 0, a_dim, a_mul, a_add,
                                ptr(b_c), b_c.stride(0) if b_c is not None else 0, b_off, b_dim if b_c is not None else 0,
                                n_out, out_act, ptr(dout), dout.stride(0), ptr(dW), ptr(da), ptr(db), None, ndev(B), stream()),
              "nsx_mlp_bwd")
        return dW, da, db, None, None, None, None, None, None, None, None, None


def fused_mlp(params: torch.Tensor, n_hidden_mats: int, n_out: int, out_act: int = 0,
              a: Optional[torch.Tensor] = None, a_mul: float = 1.0, a_add: float = 0.0,
              b: Optional[torch.Tensor] = None, b_off: int = 0, b_dim: Optional[int] = None,
              precomputed: Optional[torch.Tensor] = None, w16: Optional[torch.Tensor] = None) -> torch.Tensor:
    """tcnn FullyFusedMLP equivalent (width 64, 1 + n_hidden_mats hidden layers, no biases).
    params: flat fp32 [W0 | Wh | Wo]; ``w16``: their fp16 copy if the caller already has it; returns [B, n_out] fp16."""
    if b is not None and b_dim is None:
        b_dim = b.shape[1] - b_off
    return _FusedMLPFn.apply(params, a, b, n_hidden_mats, float(a_mul), float(a_add), int(b_off),
                             int(b_dim or 0), int(n_out), int(out_act), precomputed, w16)


# ------------------------------------------------------------------------------------------------
# fused SE(3) deformation field
# ------------------------------------------------------------------------------------------------
def deform_param_count() -> int:
    return int(lib().nsx_deform_param_count())


_WINDOW7_CACHE = {}


def deform_window7(windows_param, n_freq: int = 7):
    """Per-frequency cosine window (windowed_nerf_encoding.py:76-92) as a host float array; None -> no window."""
    if windows_param is None:
        return None
    key = (float(windows_param), n_freq)
    hit = _WINDOW7_CACHE.get(key)
    if hit is not None:
        return hit
    import numpy as np
    bands = np.linspace(0.0, n_freq - 1, n_freq, dtype=np.float32)
    x = np.clip(np.float32(windows_param) - bands, 0, 1)
    w = (0.5 * (1 - np.cos(np.float32(np.pi) * x))).astype(np.float32)
    if len(_WINDOW7_CACHE) > 64:
        _WINDOW7_CACHE.clear()
    _WINDOW7_CACHE[key] = (C.c_float * 7)(*[float(v) for v in w])
    return _WINDOW7_CACHE[key]


class _DeformFn(torch.autograd.Function):
    """packed: the fp16 MFMA fragments of the 16 nn.Linear tensors (deform_pack); params: those tensors themselves, in
    include/nsx.h order -- inputs only so that autograd routes their gradients (views of one flat buffer) back."""

    @staticmethod
    def forward(ctx, packed, code, positions, code_slot, aabb6, window7, precomputed, *params):
        dev = positions.device
        pos = positions.detach().to(torch.float32).contiguous()
        code_c = code.detach().to(torch.float32).contiguous()
        S = pos.shape[0]
        if code_slot is None and code_c.shape[0] == 1 and S != 1 and any(ctx.needs_input_grad):
            code_slot = torch.zeros((S,), dtype=torch.int32, device=dev)     # (the backward indexes the table per sample)
        if precomputed is not None:
            off = precomputed.detach()
        else:
            off = torch.empty((S, 3), dtype=torch.float32, device=dev)
            if code_slot is not None or (code_c.shape[0] == 1 and S != 1):
                # codes are rows of a table (always, in the model): the code columns of the two input layers are factored
                # through the row (nsx_deform_fwd_rows, the default since round 5); a one-row table without slots is ONE code
                # for every sample (an evaluation image's timestep)
                off = deform_fwd_rows(packed, pos, aabb6, code_c, code_slot, window7, out=off)
            else:
                check(lib().nsx_deform_fwd(ptr(packed), ptr(pos), S, aabb6, ptr(code_c), code_c.stride(0), ptr(code_slot),
                                           window7, ptr(off), ndev(S), stream()), "nsx_deform_fwd")
        ctx.save_for_backward(packed, pos, code_c, code_slot)
        ctx.aabb6, ctx.window7 = aabb6, window7
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return off

    @staticmethod
    def backward(ctx, goff):
        packed, pos, code_c, code_slot = ctx.saved_tensors
        dev = pos.device
        S = pos.shape[0]
        goff = goff.to(torch.float32).contiguous()
        n_params = int(lib().nsx_deform_param_count())
        gparams = torch.zeros(n_params, dtype=torch.float32, device=dev)
        need_code = ctx.needs_input_grad[1]
        gtable = gsamples = None
        if need_code:
            if code_slot is not None and code_c.shape[0] <= 128:
                gtable = torch.zeros_like(code_c)
            else:
                gsamples = torch.empty((S, 128), dtype=torch.float32, device=dev)
        scratch = torch.empty(int(lib().nsx_deform_scratch_bytes(S)), dtype=torch.uint8, device=dev)
        check(lib().nsx_deform_bwd(ptr(packed), ptr(pos), S, ctx.aabb6, ptr(code_c), code_c.stride(0), ptr(code_slot),
                                   code_c.shape[0] if gtable is not None else 0, ctx.window7, ptr(goff), ptr(scratch),
                                   ptr(gparams), ptr(gtable), ptr(gsamples), ndev(S), stream()), "nsx_deform_bwd")
        gcode = None
        if need_code:
            if gtable is not None:
                gcode = gtable
            elif code_slot is not None:
                gcode = torch.zeros_like(code_c).index_add_(0, code_slot.long(), gsamples)
            else:
                gcode = gsamples
        sizes = []
        for shp in ctx.param_shapes:
            n = 1
            for d in shp:
                n *= d
            sizes.append(n)
        assert sum(sizes) == n_params
        # views of the flat gradient, nsx.h order (one split dispatch; biases are already 1-D)
        grads = [g if len(shp) == 1 else g.view(shp) for g, shp in zip(torch.split(gparams, sizes), ctx.param_shapes)]
        return (None, gcode, None, None, None, None, None, *grads)


@torch.no_grad()
def deform_fwd_rows(packed: torch.Tensor, positions: torch.Tensor, aabb6, code_table: torch.Tensor,
                    code_slot: Optional[torch.Tensor], window7, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``nsx_deform_fwd_rows`` (values only): offsets [S,3] of ``positions`` whose warp code is row ``code_slot[s]`` of
    ``code_table`` (``code_slot`` None: a one-row table, every sample takes it)."""
    S, n_rows = int(positions.shape[0]), int(code_table.shape[0])
    off = out if out is not None else torch.empty((S, 3), dtype=torch.float32, device=positions.device)
    terms = torch.empty((int(lib().nsx_deform_terms_floats(n_rows)),), dtype=torch.float32, device=positions.device)
    check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(positions, torch.float32), S, aabb6, ptr(code_table, torch.float32),
                                    code_table.stride(0), ptr(code_slot), n_rows, window7, ptr(off), ptr(terms), ndev(S),
                                    stream()), "nsx_deform_fwd_rows")
    return off


def deform_pack(flat_params: torch.Tensor) -> torch.Tensor:
    """fp32 flat parameters (include/nsx.h order) -> packed fp16 MFMA fragments + fp16-rounded biases (device buffer)."""
    params = flat_params.detach().to(torch.float32).contiguous()
    packed = torch.empty(_deform_pack_bytes(), dtype=torch.uint8, device=params.device)
    check(lib().nsx_deform_pack(ptr(params), ptr(packed), stream()), "nsx_deform_pack")
    return packed


def deform_pack_tensors(params16) -> torch.Tensor:
    """As ``deform_pack`` but straight from the 16 parameter tensors (include/nsx.h order), without concatenating."""
    assert len(params16) == 16
    for t in params16:
        if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
            return deform_pack(torch.cat([p.detach().float().reshape(-1) for p in params16]))
    packed = torch.empty(_deform_pack_bytes(), dtype=torch.uint8, device=params16[0].device)
    arr = (C.c_void_p * 16)(*[t.data_ptr() for t in params16])
    check(lib().nsx_deform_pack_tensors(arr, ptr(packed), stream()), "nsx_deform_pack_tensors")
    return packed


_PACK_BYTES = None


def _deform_pack_bytes() -> int:
    global _PACK_BYTES
    if _PACK_BYTES is None:
        _PACK_BYTES = int(lib().nsx_deform_pack_bytes())
    return _PACK_BYTES


def deform_offsets(params, packed: torch.Tensor, positions: torch.Tensor, code: torch.Tensor, aabb6,
                   windows_param=None, code_slot: Optional[torch.Tensor] = None,
                   precomputed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Fused SE(3) deformation (deformation_field.py:148-166): offsets [S,3] fp32 in normalised space.
    params: the 16 nn.Linear tensors in include/nsx.h order (gradient routing); packed: ``deform_pack`` of their
    current values; code: [S,128] per-sample codes, or a code table with per-sample row indices ``code_slot``;
    aabb6: ctypes float[6] (host)."""
    if code_slot is not None:
        code_slot = code_slot.to(torch.int32).contiguous()
    return _DeformFn.apply(packed, code, positions, code_slot, aabb6, deform_window7(windows_param), precomputed,
                           *params)


# ------------------------------------------------------------------------------------------------
# field glue: sample positions, scene-box normalisation + selector, trunc_exp density epilogue
# ------------------------------------------------------------------------------------------------
def sample_positions(origins: torch.Tensor, directions: Optional[torch.Tensor], t_starts: Optional[torch.Tensor],
                     t_ends: Optional[torch.Tensor], ray_indices: Optional[torch.Tensor] = None) -> torch.Tensor:
    """World-space sample midpoints origins + directions * (t_starts + t_ends) / 2 (no gradient: rays are data)."""
    S = t_starts.shape[0] if t_starts is not None else origins.shape[0]
    o = origins.detach().to(torch.float32).contiguous()
    d = directions.detach().to(torch.float32).contiguous() if directions is not None else None
    t0 = t_starts.detach().reshape(-1).to(torch.float32).contiguous() if t_starts is not None else None
    t1 = t_ends.detach().reshape(-1).to(torch.float32).contiguous() if t_ends is not None else None
    ri = ray_indices.to(torch.int64).contiguous() if ray_indices is not None else None
    out = torch.empty((S, 3), dtype=torch.float32, device=o.device)
    check(lib().nsx_sample_positions(ptr(o), ptr(d), ptr(ri), ptr(t0), ptr(t1), None, S, None, ptr(out), None, None,
                                     ndev(S),
                                     stream()), "nsx_sample_positions")
    return out


class _NormalisedPositionsFn(torch.autograd.Function):
    """(positions [+ offsets]) -> scene-box-normalised positions masked by the in-box selector, and the selector."""

    @staticmethod
    def forward(ctx, positions, offsets, aabb6):
        p = positions.detach().to(torch.float32).contiguous()
        off = offsets.detach().to(torch.float32).contiguous() if offsets is not None else None
        S = p.shape[0]
        pn = torch.empty((S, 3), dtype=torch.float32, device=p.device)
        sel = torch.empty((S,), dtype=torch.uint8, device=p.device)
        check(lib().nsx_sample_positions(ptr(p), None, None, None, None, ptr(off), S, aabb6, None, ptr(pn), ptr(sel),
                                         ndev(S),
                                         stream()), "nsx_sample_positions")
        ctx.save_for_backward(sel)
        ctx.aabb6 = aabb6
        ctx.has_off = offsets is not None
        ctx.mark_non_differentiable(sel)
        return pn, sel

    @staticmethod
    def backward(ctx, g, _gsel):
        (sel,) = ctx.saved_tensors
        need_p, need_o = ctx.needs_input_grad[0], ctx.has_off and ctx.needs_input_grad[1]
        if not (need_p or need_o):
            return None, None, None
        g = g.to(torch.float32).contiguous()
        dpos = torch.empty_like(g)
        check(lib().nsx_normalise_bwd(ptr(g), ptr(sel), g.shape[0], ctx.aabb6, ptr(dpos), ndev(g.shape[0]), stream()),
              "nsx_normalise_bwd")
        return (dpos if need_p else None), (dpos if need_o else None), None


def normalised_positions(positions: torch.Tensor, offsets: Optional[torch.Tensor], aabb6):
    return _NormalisedPositionsFn.apply(positions, offsets, aabb6)


class _DensityFn(torch.autograd.Function):
    """density = trunc_exp(base_out[:, 0].float()) * selector."""

    @staticmethod
    def forward(ctx, base_out, sel):
        b = base_out.detach()
        assert b.dtype == torch.float16 and b.stride(1) == 1
        S = b.shape[0]
        dens = torch.empty((S, 1), dtype=torch.float32, device=b.device)
        check(lib().nsx_density_fwd(ptr(b) if b.is_contiguous() else C.c_void_p(b.data_ptr()), b.stride(0), ptr(sel), S,
                                    ptr(dens), ndev(S), stream()), "nsx_density_fwd")
        ctx.save_for_backward(b, sel)
        return dens

    @staticmethod
    def backward(ctx, g):
        b, sel = ctx.saved_tensors
        g = g.reshape(-1).to(torch.float32).contiguous()
        db = torch.zeros(b.shape, dtype=torch.float16, device=b.device)
        check(lib().nsx_density_bwd(C.c_void_p(b.data_ptr()), b.stride(0), ptr(sel), ptr(g), b.shape[0], ptr(db),
                                    ndev(b.shape[0]),
                                    stream()), "nsx_density_bwd")
        return db, None


def density_from_base(base_out: torch.Tensor, selector: torch.Tensor) -> torch.Tensor:
    return _DensityFn.apply(base_out, selector)


# ------------------------------------------------------------------------------------------------
# plain tcnn-shaped HashGrid encoding (compatibility path for the reference's own HashEnsemble module)
# ------------------------------------------------------------------------------------------------
class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, F_enc, geom):
        xx = x.detach().to(torch.float32).contiguous()
        t16 = params.detach().to(torch.float16).contiguous()
        B = xx.shape[0]
        out = torch.empty((B, geom.n_levels * F_enc), dtype=torch.float16, device=xx.device)
        check(lib().nsx_hashgrid_fwd(ptr(xx), B, ptr(t16), F_enc, C.byref(geom), ptr(out), stream()), "nsx_hashgrid_fwd")
        ctx.save_for_backward(xx, t16)
        ctx.F_enc, ctx.geom = F_enc, geom
        return out

    @staticmethod
    def backward(ctx, dout):
        xx, t16 = ctx.saved_tensors
        d = dout.to(torch.float16).contiguous()
        dtab = torch.zeros(t16.shape, dtype=torch.float32, device=xx.device) if ctx.needs_input_grad[1] else None
        dx = torch.zeros_like(xx) if ctx.needs_input_grad[0] else None
        check(lib().nsx_hashgrid_bwd(ptr(xx), xx.shape[0], ptr(t16), ctx.F_enc, C.byref(ctx.geom), ptr(d), ptr(dtab),
                                     ptr(dx), stream()), "nsx_hashgrid_bwd")
        return dx, dtab, None, None


_ITEMSIZE = {torch.float32: 4, torch.float16: 2, torch.bfloat16: 2, torch.float64: 8, torch.int64: 8, torch.int32: 4,
             torch.int16: 2, torch.uint8: 1, torch.int8: 1, torch.bool: 1}


def zeros_many(specs, device):
    """Zero tensors of the given (shape, dtype) specs carved out of ONE buffer: one fill launch instead of len(specs)
    (the steady-state step is a chain of small dependent kernels; every launch on it costs ~5 us of device time).
    Each tensor starts on a 256-byte boundary."""
    sizes, total = [], 0
    for shape, dtype in specs:
        n = 1
        for d in shape:
            n *= int(d)
        nbytes = n * _ITEMSIZE[dtype]
        sizes.append((total, nbytes))
        total += (nbytes + 255) // 256 * 256
    buf = torch.zeros((max(total, 1),), dtype=torch.uint8, device=device)
    return [buf[off:off + nb].view(dtype).view(tuple(shape)) for (off, nb), (shape, dtype) in zip(sizes, specs)]


@torch.no_grad()
def gather_rows(index: torch.Tensor, *tensors: torch.Tensor, zero_fill: bool = False):
    """[t[index] for t in tensors] (rows along dim 0) in one native launch; no autograd (values only).
    Under ``_lib.device_count`` the kernel writes the first ``n_dev`` rows and zeros the rest itself (``zero_fill`` is
    kept for callers' readability: no separate fill is launched)."""
    idx = index.to(torch.int64).contiguous()
    n = idx.shape[0]
    srcs = [(t.detach() if t.requires_grad else t).contiguous() for t in tensors]
    outs = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in srcs]
    k = len(srcs)
    if n > 0 and k > 0:
        rb = []
        for t in srcs:                                   # bytes per row (no tensor op: shapes only)
            row = t.element_size()
            for d in t.shape[1:]:
                row *= int(d)
            rb.append(row)
        if any(b % 4 for b in rb) or k > _lib.NSX_MAX_GATHER:
            if _lib.profiler.counted_capacity is not None or zero_fill:
                raise RuntimeError("gather_rows: rows that are not multiples of 4 bytes (or more than NSX_MAX_GATHER arrays) "
                                   "cannot be gathered under a device-side count")
            return tuple(t.index_select(0, idx) for t in srcs)
        src_arr = (C.c_void_p * k)(*[t.data_ptr() for t in srcs])
        dst_arr = (C.c_void_p * k)(*[t.data_ptr() for t in outs])
        rb_arr = (C.c_int64 * k)(*rb)
        check(lib().nsx_gather_rows(k, src_arr, rb_arr, dst_arr, ptr(idx), n, ndev(n), stream()), "nsx_gather_rows")
    return tuple(outs)


@torch.no_grad()
def tables_preblend(tables_f16: torch.Tensor, H: int, geom: GridGeom, code_row: torch.Tensor,
                    window: Optional[torch.Tensor]) -> torch.Tensor:
    """[total,2,Hp] fp16 tables x one [H] code row (x window) -> [total,2] fp16 blended grid (eval fast path)."""
    out = torch.empty((geom.total_entries, 2), dtype=torch.float16, device=tables_f16.device)
    code = code_row.detach().to(torch.float32).reshape(-1).contiguous()
    check(lib().nsx_tables_preblend(ptr(tables_f16), H, C.byref(geom), ptr(code), ptr(window), ptr(out), stream()),
          "nsx_tables_preblend")
    return out


@torch.no_grad()
def hashgrid_fwd_f16(x: torch.Tensor, table_f16: torch.Tensor, F_enc: int, geom: GridGeom) -> torch.Tensor:
    """Forward only, on an fp16 table in tcnn layout [total][F_enc] (no conversion, no autograd)."""
    xx = x.detach().to(torch.float32).contiguous()
    B = xx.shape[0]
    out = torch.empty((B, geom.n_levels * F_enc), dtype=torch.float16, device=xx.device)
    if B > 0:
        check(lib().nsx_hashgrid_fwd(ptr(xx), B, ptr(table_f16), F_enc, C.byref(geom), ptr(out), stream()),
              "nsx_hashgrid_fwd")
    return out


@torch.no_grad()
def density_fused(positions_world: torch.Tensor, offsets: Optional[torch.Tensor], aabb6, table_f16: torch.Tensor,
                  geom: GridGeom, base_w16: torch.Tensor, base_hidden_mats: int, want_base_out: bool = True):
    """``nsx_density_fused_fwd``: (world positions + offsets) -> scene-box normalisation + selector -> lookup in the
    pre-blended 2-feature grid -> mlp_base -> (density [S,1] fp32, base_out [S,16] fp16 or None) in one launch; values only."""
    pos = positions_world.detach().to(torch.float32).contiguous()
    off = offsets.detach().to(torch.float32).contiguous() if offsets is not None else None
    S = pos.shape[0]
    density = torch.empty((S, 1), dtype=torch.float32, device=pos.device)
    base_out = torch.empty((S, 16), dtype=torch.float16, device=pos.device) if want_base_out else None
    if S > 0:
        check(lib().nsx_density_fused_fwd(ptr(pos), ptr(off), S, aabb6, ptr(table_f16, torch.float16), C.byref(geom),
                                          ptr(base_w16, torch.float16), int(base_hidden_mats), ptr(base_out), 16, ptr(density),
                                          ndev(S), stream()), "nsx_density_fused_fwd")
    return density, base_out


def hashgrid_encoding(x: torch.Tensor, params: torch.Tensor, F_enc: int, geom: GridGeom) -> torch.Tensor:
    """x [B,3] in [0,1), params flat fp32 (tcnn layout [total][F_enc]) -> [B, n_levels*F_enc] fp16."""
    return _HashGridFn.apply(x, params, F_enc, geom)
