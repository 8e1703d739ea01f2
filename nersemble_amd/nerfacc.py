"""nerfacc-shaped operators backed by libnsx.so (include/nsx.h).

Mirrors the part of nerfacc 0.5.2 the reference imports: ``OccGridEstimator`` (``sampling``,
``update_every_n_steps``, attributes ``binaries`` / ``occs`` / ``aabbs`` / ``resolution``), ``pack_info``,
``render_weight_from_density``, ``render_visibility_from_density``, ``accumulate_along_rays``.
Call sites in the reference: nersemble_volumetric_sampler.py:95-108, nersemble_instant_ngp.py:133-137,
:185-196, :325-331, nersemble_deformation_renderer.py:22-25.  The traversal, the per-ray scans and the grid
update (cell selection, jitter, EMA-max, threshold) are HIP kernels.
"""
import ctypes as C
import math
import os
from typing import Callable, Optional, Tuple

import torch
from torch import nn, Tensor

from ._lib import check, lib, ndev, ptr, stream


# ------------------------------------------------------------------------------------------------
# packed-ray helpers
# ------------------------------------------------------------------------------------------------
def pack_info(ray_indices: Tensor, n_rays: Optional[int] = None) -> Tensor:
    """[n_rays, 2] int64 (start, count) from sorted ray indices (nerfacc.pack_info)."""
    assert ray_indices.dim() == 1
    ray_indices = ray_indices.to(torch.int64).contiguous()
    if n_rays is None:
        n_rays = int(ray_indices.max().item()) + 1 if ray_indices.numel() else 0
    dev = ray_indices.device
    counts = torch.zeros((n_rays,), dtype=torch.int64, device=dev)
    check(lib().nsx_ray_histogram(ptr(ray_indices), ray_indices.numel(), n_rays, ptr(counts), ndev(ray_indices.numel()),
                                  stream()),
          "nsx_ray_histogram")
    packed = torch.empty((n_rays, 2), dtype=torch.int64, device=dev)
    total = torch.empty((1,), dtype=torch.int64, device=dev)
    check(lib().nsx_pack_info(ptr(counts), n_rays, ptr(packed), ptr(total), stream()), "nsx_pack_info")
    return packed


def _packed(packed_info, ray_indices, n_rays):
    if packed_info is not None:
        return packed_info.to(torch.int64).contiguous()
    assert ray_indices is not None and n_rays is not None, "need packed_info or (ray_indices, n_rays)"
    return pack_info(ray_indices, n_rays)


class _RenderWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, packed):
        t0 = t_starts.detach().to(torch.float32).contiguous()
        t1 = t_ends.detach().to(torch.float32).contiguous()
        sg = sigmas.detach().to(torch.float32).contiguous()
        w, T, a = torch.empty_like(sg), torch.empty_like(sg), torch.empty_like(sg)
        if sg.numel() > 0:
            check(lib().nsx_render_weights_fwd(ptr(t0), ptr(t1), ptr(sg), ptr(packed), packed.shape[0], ptr(w), ptr(T),
                                                   ptr(a), None, 0.0, 0.0, None, stream()), "nsx_render_weights_fwd")
        ctx.save_for_backward(t0, t1, sg, packed)
        ctx.mark_non_differentiable(T, a)
        return w, T, a

    @staticmethod
    def backward(ctx, gw, gT, ga):
        t0, t1, sg, packed = ctx.saved_tensors
        gw = gw.to(torch.float32).contiguous()
        ds = torch.zeros_like(sg)
        check(lib().nsx_render_weights_bwd(ptr(t0), ptr(t1), ptr(sg), ptr(packed), packed.shape[0], ptr(gw), ptr(ds),
                                           stream()), "nsx_render_weights_bwd")
        return None, None, ds, None


def render_weight_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, packed_info: Optional[Tensor] = None,
                               ray_indices: Optional[Tensor] = None, n_rays: Optional[int] = None
                               ) -> Tuple[Tensor, Tensor, Tensor]:
    """(weights, transmittance, alphas) for packed samples; differentiable w.r.t. sigmas (through weights)."""
    return _RenderWeights.apply(t_starts, t_ends, sigmas, _packed(packed_info, ray_indices, n_rays))


@torch.no_grad()
def render_visibility_from_density(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor,
                                   packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                                   n_rays: Optional[int] = None, early_stop_eps: float = 1e-4,
                                   alpha_thre=0.0) -> Tensor:
    """``alpha_thre`` may be a float or a 1-element device tensor."""
    packed = _packed(packed_info, ray_indices, n_rays)
    t0, t1 = t_starts.to(torch.float32).contiguous(), t_ends.to(torch.float32).contiguous()
    sg = sigmas.to(torch.float32).contiguous()
    vis = torch.empty(sg.shape, dtype=torch.uint8, device=sg.device)
    if sg.numel() == 0:
        return vis.bool()
    thre_dev = alpha_thre if isinstance(alpha_thre, Tensor) else None
    check(lib().nsx_render_weights_fwd(ptr(t0), ptr(t1), ptr(sg), ptr(packed), packed.shape[0], None, None, None,
                                       ptr(vis), float(early_stop_eps), 0.0 if thre_dev is not None else float(alpha_thre),
                                       ptr(thre_dev), stream()),
          "nsx_render_weights_fwd")
    return vis.bool()


class _Accumulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, values, packed, ray_indices):
        w = weights.detach().to(torch.float32).contiguous()
        v = values.detach().to(torch.float32).contiguous() if values is not None else None
        Cc = v.shape[1] if v is not None else 1
        out = torch.empty((packed.shape[0], Cc), dtype=torch.float32, device=w.device)
        check(lib().nsx_accumulate_fwd(ptr(w), ptr(v), Cc, ptr(packed), packed.shape[0], ptr(out), stream()),
              "nsx_accumulate_fwd")
        ctx.save_for_backward(w, v, ray_indices)
        ctx.Cc = Cc
        return out

    @staticmethod
    def backward(ctx, g):
        w, v, ray_indices = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        dw = torch.empty_like(w) if ctx.needs_input_grad[0] else None
        dv = torch.empty_like(v) if (v is not None and ctx.needs_input_grad[1]) else None
        check(lib().nsx_accumulate_bwd(ptr(w), ptr(v), ctx.Cc, ptr(ray_indices), w.numel(), ptr(g), ptr(dw), ptr(dv),
                                       stream()), "nsx_accumulate_bwd")
        return dw, dv, None, None


def accumulate_along_rays(weights: Tensor, values: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                          n_rays: Optional[int] = None, packed_info: Optional[Tensor] = None) -> Tensor:
    """out[r] = sum_{i in ray r} weights[i] * values[i] (values None -> sum of weights); weights [S], values [S, C]."""
    assert weights.dim() == 1, "packed weights must be [S]"
    assert ray_indices is not None, "packed mode needs ray_indices"
    ray_indices = ray_indices.to(torch.int64).contiguous()
    packed = _packed(packed_info, ray_indices, n_rays)
    if values is not None and values.shape[1] not in (1, 3):
        # generic channel count: split into supported widths
        outs = [accumulate_along_rays(weights, values[:, c:c + 1].contiguous(), ray_indices, n_rays, packed)
                for c in range(values.shape[1])]
        return torch.cat(outs, dim=1)
    return _Accumulate.apply(weights, values, packed, ray_indices)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t_starts, t_ends, sigmas, rgb, aux, packed, background):
        t0 = t_starts.detach().to(torch.float32).contiguous()
        t1 = t_ends.detach().to(torch.float32).contiguous()
        sg = sigmas.detach().to(torch.float32).contiguous()
        c = rgb.detach().to(torch.float32).contiguous()
        ax = aux.detach().to(torch.float32).contiguous() if aux is not None else None
        R, dev = packed.shape[0], sg.device
        w = torch.empty_like(sg)
        rgb_ray = torch.empty((R, 3), dtype=torch.float32, device=dev)
        acc = torch.empty((R, 1), dtype=torch.float32, device=dev)
        depth = torch.empty((R, 1), dtype=torch.float32, device=dev)
        aux_ray = torch.empty((R, 3), dtype=torch.float32, device=dev) if ax is not None else None
        clip = torch.empty((2,), dtype=torch.float32, device=dev)
        if sg.numel() > 0:
            check(lib().nsx_composite_fwd(ptr(t0), ptr(t1), ptr(sg), ptr(c), ptr(ax), ptr(packed), R, float(background),
                                          ptr(clip), ptr(w), ptr(rgb_ray), ptr(acc), ptr(depth), ptr(aux_ray), stream()),
                  "nsx_composite_fwd")
        ctx.save_for_backward(t0, t1, sg, c, packed, clip, acc, depth)
        ctx.background = float(background)
        if aux_ray is not None:
            ctx.mark_non_differentiable(aux_ray)
        return w, rgb_ray, acc, depth, aux_ray

    @staticmethod
    def backward(ctx, g_w, g_rgb, g_acc, g_depth, g_aux):
        t0, t1, sg, c, packed, clip, acc, depth = ctx.saved_tensors
        def prep(g):
            return g.to(torch.float32).contiguous() if g is not None else None
        g_w, g_rgb, g_acc, g_depth = prep(g_w), prep(g_rgb), prep(g_acc), prep(g_depth)
        ds = torch.zeros_like(sg)
        dc = torch.zeros_like(c) if ctx.needs_input_grad[3] else None
        if sg.numel() > 0:
            check(lib().nsx_composite_bwd(ptr(t0), ptr(t1), ptr(sg), ptr(c), ptr(packed), packed.shape[0], ctx.background,
                                          ptr(clip), ptr(acc), ptr(depth), ptr(g_w), ptr(g_rgb), ptr(g_acc), ptr(g_depth),
                                          ptr(ds), ptr(dc), stream()), "nsx_composite_bwd")
        return None, None, ds, dc, None, None, None


def composite(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, rgb: Tensor, packed_info: Tensor,
              background: float = 1.0, aux: Optional[Tensor] = None):
    """Fused render_weight_from_density + RGB / accumulation / expected-depth renderers (+ aux accumulation):
    returns (weights [S], rgb [R,3], accumulation [R,1], depth [R,1], aux [R,3] | None)."""
    return _Composite.apply(t_starts, t_ends, sigmas, rgb, aux, packed_info.to(torch.int64).contiguous(), background)


# ------------------------------------------------------------------------------------------------
# occupancy grid
# ------------------------------------------------------------------------------------------------
class OccGridEstimator(nn.Module):
    """Occupancy grid (single level) with nerfacc 0.5.2's interface; traversal runs in libnsx."""

    DIM: int = 3

    def __init__(self, roi_aabb, resolution=128, levels: int = 1):
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        roi_aabb = torch.as_tensor(roi_aabb, dtype=torch.float32).flatten()
        assert resolution.shape[0] == self.DIM and roi_aabb.shape[0] == 2 * self.DIM
        if levels != 1 or len(set(resolution.tolist())) != 1:
            raise NotImplementedError("native traversal supports grid_levels=1 and cubic resolution "
                                      "(the configuration NeRSemble trains with, train_nersemble.py:100)")
        center, half = (roi_aabb[:3] + roi_aabb[3:]) / 2, (roi_aabb[3:] - roi_aabb[:3]) / 2
        aabbs = torch.stack([torch.cat([center - half * 2 ** i, center + half * 2 ** i]) for i in range(levels)])
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", aabbs)
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        self._res = int(resolution[0])               # host copy (the buffer lives on the device)
        self._aabb_host_key = None

    @property
    def device(self) -> torch.device:
        return self.occs.device

    def _aabb6(self):
        """Host copy of ``aabbs[0]`` for the kernels' by-value box argument, rebuilt whenever the buffer was replaced or
        written (``load_state_dict``, ``.to()``): the kernels always march the box the state dict reports."""
        key = (self.aabbs.data_ptr(), self.aabbs._version)
        if self._aabb_host_key != key:
            self._aabb_host = (C.c_float * 6)(*[float(v) for v in self.aabbs[0].tolist()])
            self._aabb_host_key = key
        return self._aabb_host

    def _occs_mean(self) -> Tensor:
        """``occs.mean()`` as a device scalar, recomputed only when ``occs`` was written (every 16 steps, or by a
        caller that edits the grid): a 2 M-element reduction otherwise launched in front of every sampling call."""
        key = (self.occs._version, self.occs.data_ptr())
        if getattr(self, "_occs_mean_key", None) != key:
            self._occs_mean_value = self.occs.mean()
            self._occs_mean_key = key
        return self._occs_mean_value

    def _alpha_threshold(self, alpha_thre: float) -> Tensor:
        """``min(alpha_thre, occs.mean())`` (nerfacc's sampling) as a device tensor of shape [1], cached like the mean: it
        changes when the grid is written, every 16 steps, not with every sampling call."""
        mean = self._occs_mean()
        key = (self._occs_mean_key, alpha_thre)
        if getattr(self, "_alpha_thre_key", None) != key:
            self._alpha_thre_value = torch.clamp(mean, max=alpha_thre).reshape(1).float()
            self._alpha_thre_key = key
        return self._alpha_thre_value

    # ---- traversal -------------------------------------------------------------------------------
    @staticmethod
    def _near_planes(rays_o: Tensor, near_plane: float, t_min: Optional[Tensor], render_step_size: float,
                     stratified: bool) -> Tensor:
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if stratified:
            near_planes = near_planes + torch.rand_like(near_planes) * render_step_size
        return near_planes

    def _march_key(self, rays_o, rays_d, near_plane, far_plane, render_step_size, stratified, t_min):
        return (rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], float(near_plane), float(far_plane),
                float(render_step_size), bool(stratified), None if t_min is None else t_min.data_ptr(),
                self.binaries.data_ptr(), self.binaries._version)

    @torch.no_grad()
    def prefetch_march(self, rays_o: Tensor, rays_d: Tensor, near_plane: float = 0.0, far_plane: float = 1e10,
                       t_min: Optional[Tensor] = None, render_step_size: float = 1e-3, stratified: bool = False) -> bool:
        """Pass 1 of the two-pass traversal for a LATER ``sampling()`` call on the same rays: the (jittered) near planes,
        the per-ray sample counts, their prefix sums, and the copy of the total into pinned host memory -- on a side
        stream, beside whatever the device is doing.  Marching depends on the rays and on the occupancy grid only, not
        on the model's weights, so a training loop that knows its next ray batch (a loader always does) issues this one
        step ahead; the later ``sampling()`` finds the count on the host and never waits for the device.  The result is
        used only if rays, arguments and the grid (same storage, same version) are exactly those of the later call."""
        if not rays_o.is_cuda:
            return False
        rays_o = rays_o.to(torch.float32).contiguous()
        rays_d = rays_d.to(torch.float32).contiguous()
        if t_min is not None:
            t_min = t_min.contiguous()
        R, dev = rays_o.shape[0], rays_o.device
        main = torch.cuda.current_stream(dev)
        side = getattr(self, "_prefetch_stream", None)
        if side is None or side.device != dev:
            side = self._prefetch_stream = torch.cuda.Stream(dev, priority=-1)
            # pinned read-back words + completion events, recycled (three in flight at most: two held + the one being made)
            self._prefetch_ring = [(torch.empty((2,), dtype=torch.int64, pin_memory=True), torch.cuda.Event())
                                   for _ in range(4)]           # (total, "a ray has more samples than the stash keeps")
            self._prefetch_turn = 0
        key = self._march_key(rays_o, rays_d, near_plane, far_plane, render_step_size, stratified, t_min)
        binary = self.binaries[0].contiguous().view(torch.uint8)
        total_host, done = self._prefetch_ring[self._prefetch_turn % 4]
        self._prefetch_turn += 1
        # the jitter is drawn on the caller's stream (same generator, same order of draws as marching in place); the two
        # traversal kernels and the read-back are enqueued on the side stream through its raw handle
        near_planes = self._near_planes(rays_o, near_plane, t_min, render_step_size, stratified)
        work = torch.empty((3 * R + 2,), dtype=torch.int64, device=dev)      # counts [R] | packed_info [R, 2] | total | over
        counts, packed, total = work[:R], work[R:3 * R].view(R, 2), work[3 * R:]
        # the pass keeps the samples' starts as it walks past them (nsx_march_count_stash): the step that uses it copies them
        # into place instead of walking the grid a second time on its critical path
        cap = self._stash_cap(render_step_size)
        if cap * R * 4 > self.march_stash_max_bytes:
            cap = 0
        stash = torch.empty((R, cap), dtype=torch.float32, device=dev) if cap else None
        if cap:
            work[3 * R + 1:].zero_()
        side.wait_stream(main)                               # rays, grid and near planes were written on the caller's stream
        raw = C.c_void_p(side.cuda_stream)
        if cap:
            check(lib().nsx_march_count_stash(ptr(rays_o), ptr(rays_d), R, self._aabb6(), ptr(binary), self._res,
                                              ptr(near_planes), float(far_plane), float(render_step_size), ptr(counts),
                                              ptr(stash), cap, ptr(work[3 * R + 1:]), raw), "nsx_march_count_stash")
        else:
            check(lib().nsx_march_count(ptr(rays_o), ptr(rays_d), R, self._aabb6(), ptr(binary), self._res,
                                        ptr(near_planes), float(far_plane), float(render_step_size), ptr(counts), raw),
                  "nsx_march_count")
        check(lib().nsx_pack_info(ptr(counts), R, ptr(packed), ptr(total), raw), "nsx_pack_info")
        check(lib().nsx_copy_to_host_async(C.c_void_p(total_host.data_ptr()), ptr(total), 16 if cap else 8, raw),
              "nsx_copy_to_host_async")
        done.record(side)
        work.record_stream(side)                             # allocated on the caller's stream, written on the side stream
        if stash is not None:
            stash.record_stream(side)
        # (two are held at most: the pass for the step whose forward has not run yet, and the one for the step after)
        held = getattr(self, "_prefetched", None) or []
        held.append({"key": key, "rays": (rays_o, rays_d, t_min), "near_planes": near_planes, "packed": packed,
                     "total_host": total_host, "done": done, "keep": (work, binary), "stash": stash, "cap": cap})
        self._prefetched = held[-2:]
        return True

    march_stash = os.environ.get("NSX_MARCH_STASH", "1") == "1"
    march_stash_max = 2048                 # samples per ray
    march_stash_max_bytes = 1 << 28        # (a training batch of 4096 rays keeps 11.5 MB)

    def _stash_cap(self, render_step_size: float) -> int:
        """Samples per ray the prefetched counting pass keeps (0: none).  A ray of unit direction stays inside the box for
        at most its diagonal; a batch with a longer ray (directions that are not normalised) raises the pass's overflow flag
        and is marched in place."""
        if not self.march_stash:
            return 0
        a = self._aabb6()
        diag = math.sqrt(sum((float(a[3 + i]) - float(a[i])) ** 2 for i in range(3)))
        n = int(diag / float(render_step_size)) + 4
        return (n + 63) // 64 * 64 if n <= self.march_stash_max else 0

    @staticmethod
    def _stash_of(counted: Optional[dict]):
        """(stash, cap) of a prefetched pass whose every ray fits its row (read after ``done``), else None."""
        if counted is None or counted.get("stash") is None or int(counted["total_host"][1]) != 0:
            return None
        return counted["stash"], counted["cap"]

    def _take_prefetched(self, key):
        """The prefetched counting pass if it is for exactly this call (it then leaves the estimator); one that is for
        another call -- the NEXT step's, issued before this step's own sampling -- stays."""
        held = getattr(self, "_prefetched", None) or []
        for i, pre in enumerate(held):
            if pre["key"] == key:
                del held[i]
                return pre
        return None

    @torch.no_grad()
    def counted_march(self, rays_o: Tensor, rays_d: Tensor, near_plane: float, far_plane: float, render_step_size: float,
                      stratified: bool, t_min: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, int]:
        """Pass 1 of the two-pass traversal as the training-step driver needs it (engine/native_step.py): the (jittered)
        near planes [R], ``packed_info`` [R, 2] of the marched samples and their number on the host -- taken from the pass
        ``prefetch_march`` issued a step ahead when there is one for exactly these rays / arguments / grid, else run here
        (one host read-back, as in nerfacc's two-pass design).  ``rays_o`` / ``rays_d``: contiguous fp32."""
        R, dev = rays_o.shape[0], rays_o.device
        counted = None
        if getattr(self, "_prefetched", None):
            counted = self._take_prefetched(self._march_key(rays_o, rays_d, near_plane, far_plane, render_step_size,
                                                            stratified, t_min))
        self.last_march_prefetched = counted is not None
        self.last_march_stash = None                          # (stash [R, cap], cap) when the pass kept the samples' starts
        if counted is not None:
            torch.cuda.current_stream(dev).wait_event(counted["done"])
            counted["done"].synchronize()                     # long complete when issued a step ahead
            self.last_march_stash = self._stash_of(counted)
            return counted["near_planes"], counted["packed"], int(counted["total_host"][0])
        near_planes = self._near_planes(rays_o, near_plane, t_min, render_step_size, stratified)
        binary = self.binaries[0].contiguous().view(torch.uint8)
        counts = torch.empty((R,), dtype=torch.int64, device=dev)
        packed = torch.empty((R, 2), dtype=torch.int64, device=dev)
        total = torch.zeros((1,), dtype=torch.int64, device=dev)
        check(lib().nsx_march_count(ptr(rays_o), ptr(rays_d), R, self._aabb6(), ptr(binary), self._res, ptr(near_planes),
                                    float(far_plane), float(render_step_size), ptr(counts), stream()), "nsx_march_count")
        check(lib().nsx_pack_info(ptr(counts), R, ptr(packed), ptr(total), stream()), "nsx_pack_info")
        return near_planes, packed, int(total.item())

    @torch.no_grad()
    def traverse(self, rays_o: Tensor, rays_d: Tensor, near_planes: Tensor, far_plane: float, step: float,
                 want_cells: bool = False, counted: Optional[dict] = None):
        """Two-pass marching; returns (ray_indices int64 [S], t_starts, t_ends, packed_info [R,2], cells|None).
        ``counted``: pass 1 already done by ``prefetch_march`` (its near planes must be the ones passed in)."""
        rays_o = rays_o.to(torch.float32).contiguous()
        rays_d = rays_d.to(torch.float32).contiguous()
        near_planes = near_planes.to(torch.float32).contiguous()
        R = rays_o.shape[0]
        dev = rays_o.device
        binary = self.binaries[0].contiguous().view(torch.uint8)
        res = self._res
        if counted is not None:
            packed = counted["packed"]
            torch.cuda.current_stream(dev).wait_event(counted["done"])
            counted["done"].synchronize()                     # long complete when issued a step ahead
            S = int(counted["total_host"][0])
        else:
            counts = torch.empty((R,), dtype=torch.int64, device=dev)
            packed = torch.empty((R, 2), dtype=torch.int64, device=dev)
            total = torch.zeros((1,), dtype=torch.int64, device=dev)
            check(lib().nsx_march_count(ptr(rays_o), ptr(rays_d), R, self._aabb6(), ptr(binary), res, ptr(near_planes),
                                        float(far_plane), float(step), ptr(counts), stream()), "nsx_march_count")
            check(lib().nsx_pack_info(ptr(counts), R, ptr(packed), ptr(total), stream()), "nsx_pack_info")
            S = int(total.item())                       # the one host read-back (as in nerfacc's two-pass design)
        t0 = torch.empty((S,), dtype=torch.float32, device=dev)
        t1 = torch.empty((S,), dtype=torch.float32, device=dev)
        ri = torch.empty((S,), dtype=torch.int64, device=dev)
        cells = torch.empty((S,), dtype=torch.int32, device=dev) if want_cells else None
        kept = None if want_cells else self._stash_of(counted)
        if S > 0 and kept is not None:
            check(lib().nsx_march_fill_from_stash(ptr(kept[0]), kept[1], R, float(step), ptr(packed), ptr(t0), ptr(t1), ptr(ri),
                                                  stream()), "nsx_march_fill_from_stash")
        elif S > 0:
            check(lib().nsx_march_fill(ptr(rays_o), ptr(rays_d), R, self._aabb6(), ptr(binary), res,
                                       ptr(near_planes), float(far_plane), float(step), ptr(packed), ptr(t0), ptr(t1),
                                       ptr(ri), ptr(cells), stream()), "nsx_march_fill")
        return ri, t0, t1, packed, cells

    @torch.no_grad()
    def sampling(self, rays_o: Tensor, rays_d: Tensor, sigma_fn: Optional[Callable] = None,
                 alpha_fn: Optional[Callable] = None, near_plane: float = 0.0, far_plane: float = 1e10,
                 t_min: Optional[Tensor] = None, t_max: Optional[Tensor] = None, render_step_size: float = 1e-3,
                 early_stop_eps: float = 1e-4, alpha_thre: float = 0.0, stratified: bool = False,
                 cone_angle: float = 0.0, device_counts: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
        """Same contract as nerfacc 0.5.2 ``OccGridEstimator.sampling``: (ray_indices, t_starts, t_ends).

        ``device_counts`` (native extension): the number of samples that survive the visibility test stays ON THE DEVICE --
        the outputs keep the marched samples' capacity, their first ``self.last_n_kept[0]`` rows are the kept samples
        (compacted in order), and the caller runs the per-sample kernels under ``_lib.device_count(self.last_n_kept,
        capacity)``.  No host synchronisation after the marcher's own count."""
        if cone_angle != 0.0:
            raise NotImplementedError("cone_angle != 0 is not used by NeRSemble (train_nersemble.py:97)")
        if alpha_fn is not None:
            raise NotImplementedError("alpha_fn is not used by NeRSemble")
        far = float(far_plane)
        if t_max is not None:
            raise NotImplementedError("per-ray t_max is not used by NeRSemble (ray bundles carry no fars)")
        counted = None
        if getattr(self, "_prefetched", None):
            counted = self._take_prefetched(self._march_key(rays_o, rays_d, near_plane, far, render_step_size, stratified,
                                                            t_min))
        if counted is not None:
            near_planes = counted["near_planes"]
        else:
            near_planes = self._near_planes(rays_o, near_plane, t_min, render_step_size, stratified)
        self.last_keep_index, self.last_n_marched, self.last_n_kept = None, -1, None
        self.last_march_prefetched = counted is not None
        ray_indices, t_starts, t_ends, packed, _ = self.traverse(rays_o, rays_d, near_planes, far, render_step_size,
                                                                 counted=counted)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None:
            # nerfacc: alpha_thre = min(alpha_thre, occs.mean().item()); kept on the device (no host sync)
            alpha_thre = self._alpha_threshold(float(alpha_thre))
            if t_starts.shape[0] != 0:
                sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            else:
                sigmas = torch.empty((0,), device=t_starts.device)
            assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
            masks = render_visibility_from_density(t_starts, t_ends, sigmas, packed_info=packed,
                                                   early_stop_eps=early_stop_eps, alpha_thre=alpha_thre)
            n_marched = masks.shape[0]
            if device_counts and masks.is_cuda and n_marched > 0:
                # stream compaction on the device: ascending indices of the visible samples + their number
                from ._lib import device_count
                from .functional import gather_rows
                # (neither output needs clearing: rows beyond the count are never read under device_count)
                keep = torch.empty((n_marched,), dtype=torch.int64, device=masks.device)
                n_kept = torch.empty((1,), dtype=torch.int64, device=masks.device)
                scratch = torch.empty((int(lib().nsx_occ_scratch_bytes(n_marched)),), dtype=torch.uint8, device=masks.device)
                check(lib().nsx_compact_mask(ptr(masks.view(torch.uint8)), n_marched, ptr(keep), ptr(n_kept), ptr(scratch),
                                             stream()), "nsx_compact_mask")
                self.last_keep_index, self.last_n_marched, self.last_n_kept = keep, n_marched, n_kept
                with device_count(n_kept, n_marched):
                    ray_indices, t_starts, t_ends = gather_rows(keep, ray_indices, t_starts, t_ends, zero_fill=True)
                return ray_indices, t_starts, t_ends
            keep = masks.nonzero(as_tuple=True)[0]             # one host sync for all three selections
            self.last_keep_index, self.last_n_marched = keep, masks.shape[0]
            if keep.is_cuda:
                from .functional import gather_rows
                ray_indices, t_starts, t_ends = gather_rows(keep, ray_indices, t_starts, t_ends)      # one launch
            else:
                ray_indices, t_starts, t_ends = ray_indices[keep], t_starts[keep], t_ends[keep]
        return ray_indices, t_starts, t_ends

    # ---- grid update (native: csrc/occ_grid.hip) ----------------------------------------------------
    # nerfacc 0.5.2 ``update_every_n_steps`` / ``_update`` as the reference reaches them from its
    # ``update_occupancy_grid`` callback (nersemble_instant_ngp.py:184-196).  Cell selection, jitter, the EMA-max and
    # the threshold are kernels (include/nsx.h "occupancy-grid update"); only ``occ_eval_fn`` -- the model's density
    # query -- is called in between.  Randomness is counter-based (seed, step, slot), so every data-parallel rank and
    # the CPU oracle (oracle/occgrid.c) draw the same cells.
    rng_seed: int = 0            # key of the update's Philox stream (set by the model; shared by all ranks)
    n_timesteps: int = 1         # range of the per-query random timestep (``sample_times`` / ``sample_timesteps``)

    def _occ_scratch(self) -> Tensor:
        buf = getattr(self, "_occ_scratch_buf", None)
        if buf is None or buf.device != self.occs.device:
            nbytes = int(lib().nsx_occ_scratch_bytes(self.levels * self.cells_per_lvl))
            buf = self._occ_scratch_buf = torch.zeros((nbytes,), dtype=torch.uint8, device=self.occs.device)
            self._occupied_buf = torch.empty((self.cells_per_lvl,), dtype=torch.int32, device=self.occs.device)
            self._n_occ_buf = torch.zeros((1,), dtype=torch.int32, device=self.occs.device)
        return buf

    @torch.no_grad()
    def sample_cells(self, step: int, warmup: bool):
        """The density queries of the update at ``step``: (cell_ids int32 [M], positions [M,3], timesteps int32 [M],
        times [M,1] = timestep / (T - 1)).  Warm-up: every cell once; afterwards N/4 uniform draws followed by the
        occupied cells (N/4 draws from them when there are more than N/4)."""
        dev = self.occs.device
        N = self.cells_per_lvl
        scratch = self._occ_scratch()
        n_occ = 0
        if warmup:
            M = N
        else:
            binary = self.binaries[0].contiguous().view(torch.uint8)
            check(lib().nsx_occ_compact(ptr(binary), N, ptr(self._occupied_buf), ptr(self._n_occ_buf), ptr(scratch),
                                        stream()), "nsx_occ_compact")
            n_occ = int(self._n_occ_buf.item())            # the update's one read-back (nerfacc's nonzero() syncs too)
            M = N // 4 + min(N // 4, n_occ)
        cells = torch.empty((M,), dtype=torch.int32, device=dev)
        positions = torch.empty((M, 3), dtype=torch.float32, device=dev)
        timesteps = torch.empty((M,), dtype=torch.int32, device=dev)
        times = torch.empty((M, 1), dtype=torch.float32, device=dev)
        check(lib().nsx_occ_sample_cells(self._res, self._aabb6(), 1 if warmup else 0,
                                         ptr(self._occupied_buf), n_occ, int(self.rng_seed) & (2 ** 64 - 1), int(step),
                                         int(self.n_timesteps), M, ptr(cells), ptr(positions), ptr(timesteps),
                                         ptr(times), stream()), "nsx_occ_sample_cells")
        return cells, positions, timesteps, times

    @torch.no_grad()
    def apply_update(self, cell_ids: Tensor, occ: Tensor, occ_thre: float, ema_decay: float) -> None:
        """occs <- max(occs * ema_decay, occ) on ``cell_ids`` (duplicates: their maximum), binaries <- occs > min(mean,
        occ_thre).  In place: ``binaries`` / ``occs`` keep their storage (callers hold views of them)."""
        occ = occ.reshape(-1).to(torch.float32).contiguous()
        assert occ.shape[0] == cell_ids.shape[0]
        if not self.binaries.is_contiguous():
            self.binaries = self.binaries.contiguous()
        check(lib().nsx_occ_update(ptr(self.occs), ptr(self.binaries.view(torch.uint8)), self.levels * self.cells_per_lvl,
                                   ptr(cell_ids), ptr(occ), cell_ids.shape[0], float(ema_decay), float(occ_thre),
                                   ptr(self._occ_scratch()), None, stream()), "nsx_occ_update")
        # both buffers were written through raw pointers: tell torch (version-keyed caches -- the cached occs mean, a
        # prefetched march, the sampler's frustum culling -- must see the change)
        torch.autograd.graph.increment_version(self.occs)
        torch.autograd.graph.increment_version(self.binaries)
        self._occs_mean_key = None

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """nerfacc's entry point: runs ``_update`` on every n-th step; a training-time operation."""
        if not self.training:
            raise RuntimeError("OccGridEstimator.update_every_n_steps() belongs to training; in evaluation mode call "
                               "_update() yourself if the grid really has to change.")
        if step % n == 0:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256) -> None:
        cell_ids, positions, timesteps, times = self.sample_cells(step, warmup=step < warmup_steps)
        # the query's random timesteps, for an occ_eval_fn that conditions on time (the model's does)
        self.sample_timesteps, self.sample_times = timesteps, times
        try:
            occ = occ_eval_fn(positions)
        finally:
            self.sample_timesteps = self.sample_times = None
        self.apply_update(cell_ids, occ, occ_thre, ema_decay)
