"""End-to-end CPU restatement of ``NeRSembleNGPModel.get_outputs`` (nersemble_instant_ngp.py:280-364): occupancy-grid
marching -> [sigma_fn visibility pruning] -> sample positions -> [SE(3) deformation] -> scene-box normalisation +
selector -> HashEnsemble -> mlp_base -> trunc_exp density -> mlp_head on [(d + 1) / 2, geo features] -> render
weights -> RGB / depth / accumulation [/ deformation] over a white background.  TEST INFRASTRUCTURE ONLY (composes
the per-stage oracles of this package; nothing here is imported by the product).

Follows, in order: nersemble_volumetric_sampler.py:95-134 (nerfacc ``sampling``: per-ray near planes, jitter when
``stratified``, traversal, ``alpha_thre = min(alpha_thre, occs.mean())``, keep ``T >= eps & alpha >= thre``; sigma_fn is
None in eval mode), nersemble_instant_ngp.py:300-318 (time codes), deformation_field.py:134-166,
nersemble_nerfacto_field.py:250-301 (get_density), :303-383 (get_outputs), nersemble_instant_ngp.py:325-364
(render_weight_from_density + renderers: white background, expected depth clipped to the samples' range, rendered
deformation), with the 1-fake-sample fallback of nersemble_volumetric_sampler.py:109-115.

``render_static`` is BASELINE.json configs[0] (one timestep, one hash grid, no deformation field).
"""
import numpy as np

from . import field, hashgrid, march
from .field import normalise, sample_positions  # noqa: F401  (re-exported: tests use them from here)


def _march_and_prune(rays_o, rays_d, aabb6, binary, near, far_plane, step, sigma_fn, alpha_thre, early_stop_eps,
                     occs_mean):
    """nerfacc 0.5.2 OccGridEstimator.sampling.  Returns (ray_indices, t0, t1, info dict)."""
    ri, t0, t1, packed = march.march(rays_o, rays_d, aabb6, binary, near, far_plane, step)
    info = {"n_marched": int(ri.shape[0]), "marched_per_ray": packed[:, 1].copy()}
    if sigma_fn is not None and (alpha_thre > 0.0 or early_stop_eps > 0.0):
        thre = np.float32(min(alpha_thre, occs_mean))
        sig = sigma_fn(ri, t0, t1) if ri.shape[0] else np.zeros((0,), np.float32)
        _, T, a = march.render_weights(t0, t1, sig, packed)
        keep = (T >= np.float32(early_stop_eps)) & (a >= thre)
        info.update(sigma_marched=sig, alpha_marched=a, trans_marched=T, alpha_thre=float(thre), keep=keep,
                    marched=(ri, t0, t1))
        ri, t0, t1 = ri[keep], t0[keep], t1[keep]
    if t0.shape[0] == 0:                                     # nersemble_volumetric_sampler.py:109-115
        ri = np.zeros((1,), np.int64)
        t0 = np.ones((1,), np.float32)
        t1 = np.ones((1,), np.float32)
    return ri, t0, t1, info


def _composite(ri, t0, t1, density, rgb_s, n_rays, background, offsets=None):
    packed = march.pack_info(ri, n_rays)
    w, _, _ = march.render_weights(t0, t1, density, packed)
    acc = march.accumulate(w, None, packed)                                                            # [R, 1]
    rgb = march.accumulate(w, rgb_s, packed) + np.float32(background) * (np.float32(1.0) - acc)
    t_mid = ((t0 + t1) / np.float32(2.0))[:, None]
    depth = march.accumulate(w, t_mid, packed) / (acc + np.float32(1e-10))
    if ri.shape[0] > 0:                                                 # DepthRenderer('expected'): clip to range
        depth = np.clip(depth, t_mid.min(), t_mid.max())
    out = {"rgb": rgb, "depth": depth, "accumulation": acc, "num_samples_per_ray": packed[:, 1], "weights": w,
           "packed_info": packed}
    if offsets is not None:
        out["deformation"] = march.accumulate(w, offsets, packed)
    return out


def render_static(rays_o, rays_d, aabb, binary, tables_u16, geom, mlp_base_params, mlp_head_params,
                  near_plane=0.2, far_plane=1e3, step=0.011, background=1.0, geo_feat_dim=15, clamp_rgb=False):
    """Evaluation-mode render of the static configuration.  Returns per-ray ``rgb [R,3]``, ``depth [R,1]``,
    ``accumulation [R,1]``, ``num_samples_per_ray [R]`` and the packed per-sample ``ray_indices``, ``t_starts``,
    ``t_ends``, ``density``, ``rgb_samples``, ``weights``."""
    rays_o = np.ascontiguousarray(rays_o, dtype=np.float32)
    rays_d = np.ascontiguousarray(rays_d, dtype=np.float32)
    R = rays_o.shape[0]
    near = np.full((R,), near_plane, dtype=np.float32)
    aabb6 = np.asarray(aabb, np.float32).reshape(6)
    ri, t0, t1, _ = _march_and_prune(rays_o, rays_d, aabb6, binary, near, far_plane, step, None, 0.0, 0.0, 0.0)
    pos = sample_positions(rays_o, rays_d, ri, t0, t1)
    S = ri.shape[0]
    d = field.get_density(pos, aabb, tables_u16, 1, geom, np.ones((S, 1), dtype=np.float32), mlp_base_params,
                          geo_feat_dim)
    rgb_s = field.get_rgb(rays_d[ri], d["base"], mlp_head_params, geo_feat_dim)
    out = _composite(ri, t0, t1, d["density"], rgb_s, R, background)
    if clamp_rgb:
        out["rgb"] = np.clip(out["rgb"], 0.0, 1.0)
    out.update(ray_indices=ri, t_starts=t0, t_ends=t1, density=d["density"], rgb_samples=rgb_s)
    return out


def render_dynamic(rays_o, rays_d, times, aabb, binary, tables_u16, H, geom, mlp_base_params, mlp_head_params,
                   time_embedding, n_timesteps, deform_params=None, deform_embedding=None, window_hash=None,
                   window_deform=None, near_plane=0.2, far_plane=1e3, step=0.011, background=1.0, geo_feat_dim=15,
                   hash_disable_initial=True, hash_soft_transition=True, clamp_rgb=False,
                   training=False, near_jitter=None, alpha_thre=1e-2, early_stop_eps=0.0, occs_mean=1.0):
    """The dynamic path.  Evaluation mode (default): no jitter, no sigma_fn.  ``training=True``: per-ray near planes
    ``near_plane + near_jitter * step`` (``near_jitter`` = the U[0,1) draws of nerfacc's ``stratified`` branch, handed in
    so that both sides use the same numbers) and visibility pruning through ``field_density_fn`` on the marched
    samples, exactly the second network pass the reference runs per step.
    ``time_embedding [T, H]``, ``deform_embedding [T, 128]`` (None: the time codes themselves), ``deform_params``: flat
    fp32 vector in include/nsx.h order (None: no deformation field)."""
    rays_o = np.ascontiguousarray(rays_o, dtype=np.float32)
    rays_d = np.ascontiguousarray(rays_d, dtype=np.float32)
    times = np.asarray(times, np.float32).reshape(-1)
    R = rays_o.shape[0]
    near = np.full((R,), near_plane, dtype=np.float32)
    if training and near_jitter is not None:
        near = near + np.asarray(near_jitter, np.float32).reshape(-1) * np.float32(step)
    aabb6 = np.asarray(aabb, np.float32).reshape(6)
    kw = dict(deform_params=deform_params, deform_embedding=deform_embedding, window_hash=window_hash,
              window_deform=window_deform, hash_disable_initial=hash_disable_initial,
              hash_soft_transition=hash_soft_transition)

    def sigma_fn(ri, t0, t1):
        pos = sample_positions(rays_o, rays_d, ri, t0, t1)
        return field.field_density_fn(pos, times[ri], n_timesteps, aabb, tables_u16, H, geom, mlp_base_params,
                                      time_embedding, **kw)[0]

    ri, t0, t1, info = _march_and_prune(rays_o, rays_d, aabb6, binary, near, far_plane, step,
                                        sigma_fn if training else None, alpha_thre, early_stop_eps, occs_mean)
    pos = sample_positions(rays_o, rays_d, ri, t0, t1)
    density, d = field.field_density_fn(pos, times[ri], n_timesteps, aabb, tables_u16, H, geom, mlp_base_params,
                                        time_embedding, **kw)
    offsets = d["offsets"]
    rgb_s = field.get_rgb(rays_d[ri], d["base"], mlp_head_params, geo_feat_dim)
    out = _composite(ri, t0, t1, density, rgb_s, R, background,
                     offsets if offsets is not None else np.zeros((ri.shape[0], 3), np.float32))
    if clamp_rgb:
        out["rgb"] = np.clip(out["rgb"], 0.0, 1.0)
    out.update(ray_indices=ri, t_starts=t0, t_ends=t1, timesteps=d["timesteps"],
               offsets=offsets if offsets is not None else np.zeros((ri.shape[0], 3), np.float32),
               density=density, rgb_samples=rgb_s, sampling=info)
    return out
