"""The main pass of a training step as ONE autograd node.

``NeRSembleNGPModel.get_outputs`` + ``get_loss_dict`` + ``get_metrics_dict`` (nersemble_instant_ngp.py:280-422) on the
kept samples of a step are, in this package's modular form, nine autograd Functions (sample positions, deformation,
normalisation, HashEnsemble, mlp_base, density, mlp_head, compositing, losses) with tensor glue in between.  Each of
them is a kernel launch of a few microseconds on the device, but every Function costs ~30-50 us of host time (autograd
node construction, ``detach`` / ``contiguous`` / ``empty`` dispatches, ctypes marshalling), and once the occupancy grid
has pruned the scene the step is bound by exactly that host time.

``main_pass`` runs the same kernels in the same order with the same arguments from ONE ``torch.autograd.Function``:
forward = 7 launches (3 when the forward values of the sampler's sigma_fn pass are reused), backward = 9 launches, no
torch op in between.  The parameters enter as inputs (hash tables, the two fused MLPs, the conditioned time-code tables
that come out of the ``nn.Embedding`` lookups, the 16 deformation tensors) so gradients arrive where the optimizers look
for them; the table gradient goes to the HashEnsemble's factored sink as in the modular path.
``tests/test_training_gpu.py::test_fused_main_pass_equals_modular_path`` holds the two paths together (same loss bit
for bit, same gradients up to the order of atomics).
"""
import ctypes as C
import torch

from .. import distloss as dl
from .. import functional as F
from .._lib import check, device_count, lib, ndev, ptr, stream


class MainPassInputs:
    """Everything of the step that is data, not a differentiable input (raw device tensors + host scalars)."""
    __slots__ = ("origins", "directions", "t0", "t1", "ray_indices", "slot", "packed", "n_rays",
                 "pre_offsets", "pre_features", "pre_base", "image", "alpha_map", "depth_targets",
                 "he", "window", "field_aabb6", "deform_packed", "deform_aabb6", "deform_window7",
                 "base_hidden", "base_out_dim", "base_act", "base_w16", "head_hidden", "head_act", "head_w16", "geo_dim",
                 "background", "loss_cfg", "aux", "n_dev", "first_grid")


class _MainPass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp: MainPassInputs, tables_master, base_params, head_params, code_hash, code_deform, *deform_params):
        # inp.n_dev (optional): the number of valid samples lives on the device; the arrays have capacity rows
        with device_count(inp.n_dev, inp.t0.shape[0]):
            return _MainPass._forward(ctx, inp, tables_master, base_params, head_params, code_hash, code_deform,
                                      *deform_params)

    @staticmethod
    def _forward(ctx, inp: MainPassInputs, tables_master, base_params, head_params, code_hash, code_deform, *deform_params):
        L = lib()
        st = stream()
        dev = inp.origins.device
        S, R = inp.t0.shape[0], inp.n_rays
        he = inp.he
        H, geom = he.n_hash_encodings, he.geom
        f32, f16 = torch.float32, torch.float16
        # -- parameters in the kernels' formats (fp16 copies made once per optimizer step, tcnn.Network.half_weights)
        base_w, head_w = inp.base_w16, inp.head_w16
        code_h = code_hash.detach().contiguous()
        # what the HashEnsemble kernels see: the H grids with the batch's code rows and the window -- or, in the compact
        # first-grid phase (HashEnsemble.first_grid_phase), the contiguous copy of grid 0 with a constant code of one
        hash_slot, hash_window = inp.slot, inp.window
        comp = getattr(inp, "first_grid", None)
        if comp is not None and comp["width"] >= 2:
            he.wait_tables()
            H = comp["width"]                        # window ramp: the first H grids; codes and window as in the full layout
        elif comp is not None:
            he.wait_tables()
            H, code_h, hash_window = 1, he.first_grid_code(code_h.shape[0]), None      # (slots as in the full layout)
        code_d = code_deform.detach().contiguous()
        tables_f16 = he.half_tables() if comp is None else comp["f16"]
        # -- world positions of the samples, deformation offsets (normalised space)
        pos = torch.empty((S, 3), dtype=f32, device=dev)
        pn = torch.empty((S, 3), dtype=f32, device=dev)
        sel = torch.empty((S,), dtype=torch.uint8, device=dev)
        if inp.pre_offsets is not None:
            # offsets known (the sigma_fn pass's): positions, scene-box normalisation of (position + offset) and the
            # in-box selector in ONE launch
            offsets = inp.pre_offsets
            check(L.nsx_sample_positions(ptr(inp.origins), ptr(inp.directions), None, ptr(inp.t0), ptr(inp.t1), ptr(offsets),
                                         S, inp.field_aabb6, ptr(pos), ptr(pn), ptr(sel), ndev(S), st), "nsx_sample_positions")
        else:
            check(L.nsx_sample_positions(ptr(inp.origins), ptr(inp.directions), None, ptr(inp.t0), ptr(inp.t1), None, S, None,
                                         ptr(pos), None, None, ndev(S), st), "nsx_sample_positions")
            offsets = F.deform_fwd_rows(inp.deform_packed, pos, inp.deform_aabb6, code_d, inp.slot, inp.deform_window7)
            # -- scene-box normalisation of (position + offset), in-box selector
            check(L.nsx_sample_positions(ptr(pos), None, None, None, None, ptr(offsets), S, inp.field_aabb6, None, ptr(pn),
                                         ptr(sel), ndev(S), st), "nsx_sample_positions")
        # -- HashEnsemble, mlp_base, density
        lp = getattr(he, "level_parallel", None)
        if inp.pre_features is not None:
            feats = inp.pre_features
        elif lp is not None:
            # data-parallel run with the window open: this rank's levels for every rank's samples (engine/level_parallel.py)
            feats = lp.features(pn, code_h, hash_slot, hash_window, n_dev=inp.n_dev)
        else:
            feats = torch.empty((S, 2 * geom.n_levels), dtype=f16, device=dev)
            check(L.nsx_hash_ensemble_fwd(ptr(pn), S, ptr(tables_f16), H, C.byref(geom), ptr(code_h), code_h.stride(0),
                                          ptr(hash_slot), ptr(hash_window), ptr(feats), ndev(S), st), "nsx_hash_ensemble_fwd")
        # (level-parallel: the forward exchange whose backward this node's is -- its own, or the sigma_fn pass's)
        ctx.lp_ex = lp.last_exchange if lp is not None else None
        if inp.pre_base is not None:
            base_out = inp.pre_base
        else:
            base_out = torch.empty((S, inp.base_out_dim), dtype=f16, device=dev)
            check(L.nsx_mlp_fwd(ptr(base_w), inp.base_hidden, S, None, 0, 0, 1.0, 0.0, ptr(feats), feats.stride(0), 0,
                                feats.shape[1], inp.base_out_dim, inp.base_act, ptr(base_out), base_out.stride(0), ndev(S), st),
                  "nsx_mlp_fwd")
        density = torch.empty((S, 1), dtype=f32, device=dev)
        check(L.nsx_density_fwd(ptr(base_out), base_out.stride(0), ptr(sel), S, ptr(density), ndev(S), st), "nsx_density_fwd")
        # -- colour: mlp_head([(d + 1) / 2, geometry features]) with sigmoid, read in place
        rgb16 = torch.empty((S, 3), dtype=f16, device=dev)
        check(L.nsx_mlp_fwd(ptr(head_w), inp.head_hidden, S, ptr(inp.directions), inp.directions.stride(0), 3, 0.5, 0.5,
                            ptr(base_out), base_out.stride(0), 1, inp.geo_dim, 3, inp.head_act, ptr(rgb16), rgb16.stride(0),
                            ndev(S),
                            st), "nsx_mlp_fwd")
        # -- compositing (weights, rgb, accumulation, expected depth, rendered deformation); the colours stay fp16
        w = torch.empty((S,), dtype=f32, device=dev)
        rgb = torch.empty((R, 3), dtype=f32, device=dev)
        acc = torch.empty((R, 1), dtype=f32, device=dev)
        depth = torch.empty((R, 1), dtype=f32, device=dev)
        aux = torch.empty((R, 3), dtype=f32, device=dev)
        clip = torch.empty((2,), dtype=f32, device=dev)
        check(L.nsx_composite_fwd_h(ptr(inp.t0), ptr(inp.t1), ptr(density), ptr(rgb16), ptr(offsets), ptr(inp.packed), R,
                                    float(inp.background), ptr(clip), ptr(w), ptr(rgb), ptr(acc), ptr(depth), ptr(aux), st),
              "nsx_composite_fwd_h")
        # -- every loss term, their sum and the metrics
        use_masked, thr, l_alpha, l_depth, l_dist, l_empty, l_near, eps, max_ray = inp.loss_cfg
        per_ray = torch.empty((R, 5), dtype=f32, device=dev)
        check(L.nsx_sample_losses_fwd(ptr(w), ptr(inp.t0), ptr(inp.t1), ptr(inp.packed), R, ptr(inp.depth_targets),
                                      float(eps), int(max_ray), ptr(per_ray), st), "nsx_sample_losses_fwd")
        out = torch.empty((dl.LOSS_OUT,), dtype=f32, device=dev)
        acc1, dep1 = acc.view(-1), depth.view(-1)
        check(L.nsx_ray_losses_fwd(ptr(rgb), ptr(acc1), ptr(dep1), ptr(inp.image), ptr(inp.alpha_map),
                                   ptr(inp.depth_targets), ptr(per_ray), ptr(inp.packed), R, int(use_masked), float(thr),
                                   float(l_alpha), float(l_depth), float(l_dist), float(l_empty), float(l_near), ptr(out),
                                   st), "nsx_ray_losses_fwd")
        inp.aux = {"rgb": rgb, "accumulation": acc, "depth": depth, "weights": w, "deformation": aux,
                   "offsets": offsets, "density": density, "rgb_samples": rgb16}
        ctx.inp = inp
        ctx.save_for_backward(pos, pn, sel, feats, base_out, density, rgb16, w, rgb, acc1, dep1, clip, out, base_w, head_w,
                              code_h, code_d, tables_f16)
        ctx.shapes = (tuple(tables_master.shape), [tuple(p.shape) for p in deform_params], code_hash.shape[0])
        ctx.sink = he.grad_sink
        ctx.hash_args = (H, hash_slot, hash_window, comp is not None and comp["width"] == 1)
        ctx.code_width = code_hash.shape[1]
        # a forward whose backward will add to the sink's G (the sink counts them to know when G is complete)
        ctx.announced = ctx.sink is not None and ctx.needs_input_grad[1]
        if ctx.announced:
            ctx.sink.expect()
        return out

    @staticmethod
    def backward(ctx, g_out):
        with device_count(ctx.inp.n_dev, ctx.inp.t0.shape[0]):
            return _MainPass._backward(ctx, g_out)

    @staticmethod
    def _backward(ctx, g_out):
        (pos, pn, sel, feats, base_out, density, rgb16, w, rgb, acc1, dep1, clip, out, base_w, head_w, code_h, code_d,
         tables_f16) = ctx.saved_tensors
        inp: MainPassInputs = ctx.inp
        L = lib()
        st = stream()
        dev = pos.device
        S, R = pos.shape[0], inp.n_rays
        he = inp.he
        geom = he.geom
        H, hash_slot, hash_window, first_grid = ctx.hash_args
        f32, f16 = torch.float32, torch.float16
        use_masked, thr, l_alpha, l_depth, l_dist, l_empty, l_near, eps, max_ray = inp.loss_cfg
        g = g_out.to(f32).contiguous()
        # -- losses
        g_rgb = torch.empty((R, 3), dtype=f32, device=dev)
        g_acc = torch.empty((R,), dtype=f32, device=dev)
        g_dep = torch.empty((R,), dtype=f32, device=dev)
        g3 = torch.empty((3,), dtype=f32, device=dev)
        check(L.nsx_ray_losses_bwd(ptr(rgb), ptr(acc1), ptr(dep1), ptr(inp.image), ptr(inp.alpha_map),
                                   ptr(inp.depth_targets), R, int(use_masked), float(thr), float(l_alpha), float(l_depth),
                                   float(l_dist), float(l_empty), float(l_near), R, ptr(out), ptr(g), ptr(g_rgb), ptr(g_acc),
                                   ptr(g_dep), ptr(g3), st), "nsx_ray_losses_bwd")
        gw = torch.empty((S,), dtype=f32, device=dev)
        sums = out[dl.LOSS_SAMPLE_SUMS:dl.LOSS_SAMPLE_SUMS + 5]
        check(L.nsx_sample_losses_bwd(ptr(w), ptr(inp.t0), ptr(inp.t1), ptr(inp.packed), R, ptr(inp.depth_targets),
                                      float(eps), int(max_ray), R, ptr(sums), ptr(g3), ptr(gw), st), "nsx_sample_losses_bwd")
        # -- compositing
        # every zero-initialised buffer of the backward out of one fill (functional.zeros_many)
        n_deform = int(L.nsx_deform_param_count())
        (ds, dc16, d_head, d_base_out, d_base, gparams, gtable) = F.zeros_many(
            [((S, 1), f32), ((S, 3), f16), ((head_w.numel(),), f32), ((S, inp.base_out_dim), f16), ((base_w.numel(),), f32),
             ((n_deform,), f32), (tuple(code_d.shape), code_d.dtype)], dev)
        # the gradient of the hash features, fp32 as the hash backward reads it: every row the kernels below look at is
        # written by mlp_base's backward (rows beyond the device-side count are never read)
        dout = torch.empty((S, feats.shape[1]), dtype=f32, device=dev)
        check(L.nsx_composite_bwd_h(ptr(inp.t0), ptr(inp.t1), ptr(density), ptr(rgb16), ptr(inp.packed), R,
                                    float(inp.background), ptr(clip), ptr(acc1), ptr(dep1), ptr(gw), ptr(g_rgb), ptr(g_acc),
                                    ptr(g_dep), ptr(ds), ptr(dc16), st), "nsx_composite_bwd_h")
        # -- mlp_head: gradient of its parameters and of the geometry features (columns 1.. of base_out)
        check(L.nsx_mlp_bwd(ptr(head_w), inp.head_hidden, S, ptr(inp.directions), inp.directions.stride(0), 3, 0.5, 0.5,
                            ptr(base_out), base_out.stride(0), 1, inp.geo_dim, 3, inp.head_act, ptr(dc16), dc16.stride(0),
                            ptr(d_head), None, ptr(d_base_out), None, ndev(S), st), "nsx_mlp_bwd")
        # -- trunc_exp density: column 0 of the same gradient buffer
        check(L.nsx_density_bwd(ptr(base_out), base_out.stride(0), ptr(sel), ptr(ds), S, ptr(d_base_out), ndev(S), st),
              "nsx_density_bwd")
        # -- mlp_base
        check(L.nsx_mlp_bwd(ptr(base_w), inp.base_hidden, S, None, 0, 0, 1.0, 0.0, ptr(feats), feats.stride(0), 0,
                            feats.shape[1], inp.base_out_dim, inp.base_act, ptr(d_base_out), d_base_out.stride(0), ptr(d_base),
                            None, None, ptr(dout), ndev(S), st), "nsx_mlp_bwd")
        # -- HashEnsemble: factored table gradient into the sink, code gradient summed per code row, position gradient
        n_rows = code_h.shape[0]
        sink = ctx.sink
        need_tab = ctx.needs_input_grad[1]
        need_code = ctx.needs_input_grad[4] and not first_grid      # (the code is the constant one in that phase)
        G, dtab = None, None
        lp = getattr(he, "level_parallel", None)
        if lp is not None:
            # level-parallel exchange: dL/dfeatures travels to the levels' owners, the table gradient of the owned levels
            # stays in their gradient planes (no G here, no dense gradient), dL/dx and the code gradient come back summed
            dx, g_code_hash = lp.backward(pn, hash_slot, dout, n_dev=inp.n_dev, need_table=bool(need_tab), ex=ctx.lp_ex)
            if not need_code:
                g_code_hash = None
            return _MainPass._finish_backward(ctx, L, st, dev, S, inp, dx, sel, pos, code_d, gparams, gtable, g_code_hash,
                                              n_rows, H, None, d_base, d_head)
        if need_tab:
            if sink is not None:
                G = sink.buffer_for(code_h, hash_window, n_rows, geom.total_entries, n_samples=S)
            else:
                G = torch.zeros((n_rows, geom.total_entries, 2), dtype=f32, device=dev)
        # the code gradient leaves the kernel summed per code row (nsx_hash_ensemble_bwd_codesum): no [S, H] tensor
        g_code_hash = torch.empty((n_rows, H), dtype=f32, device=dev) if need_code else None
        dx = torch.empty((S, 3), dtype=f32, device=dev)
        G_fused = G
        if F.scatter_alone(H) and G is not None and sink is not None:
            # <= 4 grids (compact first-grid phase, static models, the first widths of the window ramp): the scatter as its own kernel.  The H = 1 instance of the fused kernel has one lane
            # per sample, so the 16 (corner, feature) items of a sample and level go out as 16 instructions of 64
            # unrelated sectors each (5.5 ms at 650 k samples); the stand-alone scatter keeps the 8-lanes-per-sample mapping
            # whose neighbouring (feature, x) items share a sector.  The gather half (dL/dx of one grid) is cheap.
            check(L.nsx_hash_ensemble_bwd_scatter(ptr(pn), S, C.byref(geom), n_rows, ptr(hash_slot), ptr(dout), ptr(G),
                                                  ptr(sink.nonfinite), 8, ndev(S), st), "nsx_hash_ensemble_bwd_scatter")
            G_fused = None
        nonfinite = ptr(sink.nonfinite) if (sink is not None and need_tab and G_fused is not None) else None
        if need_code:
            check(L.nsx_hash_ensemble_bwd_codesum(ptr(pn), S, ptr(tables_f16), H, C.byref(geom), ptr(code_h),
                                                  code_h.stride(0), n_rows, ptr(hash_slot), ptr(hash_window), ptr(dout),
                                                  ptr(G_fused), ptr(g_code_hash), ptr(F.codesum_scratch(n_rows, H, dev)),
                                                  ptr(dx), nonfinite, ndev(S), st), "nsx_hash_ensemble_bwd_codesum")
        else:
            check(L.nsx_hash_ensemble_bwd_factored(ptr(pn), S, ptr(tables_f16), H, C.byref(geom), ptr(code_h),
                                                   code_h.stride(0), n_rows, ptr(hash_slot), ptr(hash_window), ptr(dout),
                                                   ptr(G_fused), None, ptr(dx), nonfinite, ndev(S), st),
                  "nsx_hash_ensemble_bwd_factored")
        if sink is not None and need_tab and ctx.announced:
            # G is complete, and so are the gradients of the two fused MLPs (the rest of the tables' optimizer group):
            # the table optimizer may start its 12 GB pass now, beside the deformation backward below
            # (handed over only when somebody waits for them: a held reference makes autograd CLONE the two gradients
            # instead of adopting them -- two copy launches per step)
            sink.arrived(group_grads=[d_base, d_head] if sink.on_complete is not None else None)
        if need_tab and sink is None:
            dtab = torch.empty(ctx.shapes[0], dtype=f32, device=dev)
            check(L.nsx_hash_grad_expand(ptr(G), n_rows, ptr(code_h), code_h.stride(0), ptr(hash_window), H, C.byref(geom),
                                         ptr(dtab), 0, st), "nsx_hash_grad_expand")
        return _MainPass._finish_backward(ctx, L, st, dev, S, inp, dx, sel, pos, code_d, gparams, gtable, g_code_hash, n_rows,
                                          H, dtab, d_base, d_head)

    @staticmethod
    def _finish_backward(ctx, L, st, dev, S, inp, dx, sel, pos, code_d, gparams, gtable, g_code_hash, n_rows, H, dtab,
                         d_base, d_head):
        f32 = torch.float32
        # -- normalisation: gradient of the offsets
        goff = torch.empty((S, 3), dtype=f32, device=dev)
        check(L.nsx_normalise_bwd(ptr(dx), ptr(sel), S, inp.field_aabb6, ptr(goff), ndev(S), st), "nsx_normalise_bwd")
        # -- deformation field
        scratch = torch.empty(int(L.nsx_deform_scratch_bytes(S)), dtype=torch.uint8, device=dev)
        check(L.nsx_deform_bwd(ptr(inp.deform_packed), ptr(pos), S, inp.deform_aabb6, ptr(code_d), code_d.stride(0),
                               ptr(inp.slot), code_d.shape[0], inp.deform_window7, ptr(goff), ptr(scratch), ptr(gparams),
                               ptr(gtable), None, ndev(S), st), "nsx_deform_bwd")
        sizes = []
        for shp in ctx.shapes[1]:
            n = 1
            for d in shp:
                n *= d
            sizes.append(n)
        grads = [gp if len(shp) == 1 else gp.view(shp) for gp, shp in zip(torch.split(gparams, sizes), ctx.shapes[1])]
        if g_code_hash is not None and g_code_hash.shape[1] != ctx.code_width:
            full = torch.zeros((n_rows, ctx.code_width), dtype=f32, device=dev)        # (compact window-ramp layout)
            full[:, :H] = g_code_hash
            g_code_hash = full
        return (None, dtab, d_base, d_head, g_code_hash, gtable, *grads)


def main_pass(inp: MainPassInputs, tables_master, base_params, head_params, code_hash, code_deform, deform_params
              ) -> torch.Tensor:
    """Returns the ``[LOSS_OUT]`` vector of ``distloss.fused_step_losses`` (differentiable w.r.t. every parameter input);
    per-ray / per-sample forward values are left in ``inp.aux``."""
    return _MainPass.apply(inp, tables_master, base_params, head_params, code_hash, code_deform, *deform_params)
