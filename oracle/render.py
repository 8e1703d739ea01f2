"""End-to-end CPU restatement of the STATIC path (BASELINE.json configs[0]: one timestep, one hash grid, no
deformation field): occupancy-grid marching -> sample positions -> scene-box normalisation + selector -> HashEnsemble
(H = 1, code 1) -> mlp_base -> trunc_exp density -> mlp_head on [(d + 1) / 2, geo features] -> render weights ->
RGB / depth / accumulation over a white background.  TEST INFRASTRUCTURE ONLY (composes the per-stage oracles of this
package; nothing here is imported by the product).

Follows, in order: nersemble_volumetric_sampler.py:95-134 (nerfacc sampling, eval mode: no jitter, no sigma_fn),
nersemble_nerfacto_field.py:250-301 (get_density), :303-383 (get_outputs), nersemble_instant_ngp.py:325-364
(render_weight_from_density + renderers: white background, expected depth clipped to the samples' range).
"""
import numpy as np

from . import hashgrid, march, mlp


def sample_positions(rays_o, rays_d, ray_indices, t0, t1):
    """frustums.get_positions(): origins + directions * (starts + ends) / 2, in fp32 like the reference."""
    o = np.asarray(rays_o, dtype=np.float32)[ray_indices]
    d = np.asarray(rays_d, dtype=np.float32)[ray_indices]
    mid = ((np.asarray(t0, np.float32) + np.asarray(t1, np.float32)) / np.float32(2.0))[:, None]
    return o + d * mid


def normalise(positions, aabb):
    """SceneBox.get_normalized_positions + the (0, 1) selector that zeroes outside samples (:257, :268-269)."""
    aabb = np.asarray(aabb, dtype=np.float32).reshape(2, 3)
    pn = (positions - aabb[0]) / (aabb[1] - aabb[0])
    selector = ((pn > 0.0) & (pn < 1.0)).all(axis=-1)
    return pn * selector[:, None], selector


def render_static(rays_o, rays_d, aabb, binary, tables_u16, geom, mlp_base_params, mlp_head_params,
                  near_plane=0.2, far_plane=1e3, step=0.011, background=1.0, geo_feat_dim=15):
    """Returns a dict with per-ray ``rgb [R,3]``, ``depth [R,1]``, ``accumulation [R,1]``, ``num_samples_per_ray [R]``
    and the packed per-sample ``ray_indices``, ``t_starts``, ``t_ends``, ``density``, ``rgb_samples``, ``weights``."""
    rays_o = np.ascontiguousarray(rays_o, dtype=np.float32)
    rays_d = np.ascontiguousarray(rays_d, dtype=np.float32)
    R = rays_o.shape[0]
    near = np.full((R,), near_plane, dtype=np.float32)
    ri, t0, t1, packed = march.march(rays_o, rays_d, np.asarray(aabb, np.float32).reshape(6), binary, near,
                                     far_plane, step)
    pos = sample_positions(rays_o, rays_d, ri, t0, t1)
    pn, selector = normalise(pos, aabb)
    S = ri.shape[0]
    feats = hashgrid.ensemble_fwd(pn, tables_u16, 1, geom, np.ones((S, 1), dtype=np.float32))        # [S, 2L] fp16
    base = mlp.mlp_fwd(feats.astype(np.float32), mlp_base_params, 0, 1 + geo_feat_dim, 0)             # [S, 16] fp16
    density = np.exp(base[:, 0].astype(np.float32)) * selector.astype(np.float32)                      # trunc_exp fwd
    head_in = np.concatenate([(rays_d[ri] + np.float32(1.0)) / np.float32(2.0),
                              base[:, 1:1 + geo_feat_dim].astype(np.float32)], axis=1)                # [S, 18]
    rgb_s = mlp.mlp_fwd(head_in, mlp_head_params, 1, 3, 1).astype(np.float32)                          # sigmoid
    w, _, _ = march.render_weights(t0, t1, density, packed)
    acc = march.accumulate(w, None, packed)                                                            # [R, 1]
    rgb = march.accumulate(w, rgb_s, packed) + np.float32(background) * (np.float32(1.0) - acc)
    t_mid = ((t0 + t1) / np.float32(2.0))[:, None]
    depth = march.accumulate(w, t_mid, packed) / (acc + np.float32(1e-10))
    if S > 0:                                                           # DepthRenderer('expected'): clip to range
        depth = np.clip(depth, t_mid.min(), t_mid.max())
    return {"rgb": rgb, "depth": depth, "accumulation": acc, "num_samples_per_ray": packed[:, 1],
            "ray_indices": ri, "t_starts": t0, "t_ends": t1, "density": density, "rgb_samples": rgb_s, "weights": w}


def render_dynamic(rays_o, rays_d, times, aabb, binary, tables_u16, H, geom, mlp_base_params, mlp_head_params,
                   time_embedding, n_timesteps, deform_params=None, deform_embedding=None, window_hash=None,
                   window_deform=None, near_plane=0.2, far_plane=1e3, step=0.011, background=1.0, geo_feat_dim=15,
                   hash_disable_initial=True, hash_soft_transition=True):
    """The dynamic path in evaluation mode (no jitter, no sigma_fn): as ``render_static`` plus
      * timesteps = round(times * (T - 1)) per ray, the time code rows gathered per sample
        (nersemble_instant_ngp.py:300-318),
      * SE(3) deformation of the sample positions with the deformation time code: the NORMALISED-space offset is added
        to the WORLD-space position, as the reference does (:257-259, deformation_field.py:144,162),
      * the HashEnsemble blend with the windowed time code (hash_ensemble.py:119-158),
      * the rendered deformation = sum of weights x offsets (nersemble_deformation_renderer.py:8-29).
    ``time_embedding [T, H]``, ``deform_embedding [T, 128]`` (None: the time codes themselves), ``deform_params``: flat
    fp32 vector in include/nsx.h order (None: no deformation field)."""
    import torch
    from . import deform
    rays_o = np.ascontiguousarray(rays_o, dtype=np.float32)
    rays_d = np.ascontiguousarray(rays_d, dtype=np.float32)
    R = rays_o.shape[0]
    near = np.full((R,), near_plane, dtype=np.float32)
    aabb6 = np.asarray(aabb, np.float32).reshape(6)
    ri, t0, t1, packed = march.march(rays_o, rays_d, aabb6, binary, near, far_plane, step)
    S = ri.shape[0]
    timesteps = np.round(np.asarray(times, np.float32).reshape(-1) * np.float32(n_timesteps - 1)).astype(np.int64)
    ts = timesteps[ri]
    pos = sample_positions(rays_o, rays_d, ri, t0, t1)
    offsets = np.zeros((S, 3), dtype=np.float32)
    if deform_params is not None and S > 0:
        emb = np.asarray(deform_embedding if deform_embedding is not None else time_embedding, dtype=np.float32)
        off = deform.compute_offsets(torch.from_numpy(pos), torch.from_numpy(emb[ts]),
                                     torch.as_tensor(deform_params, dtype=torch.float32),
                                     torch.from_numpy(aabb6.reshape(2, 3)), window_deform, half=True)
        offsets = off.to(torch.float32).numpy()
    pn, selector = normalise(pos + offsets, aabb)
    codew = hashgrid.windowed_code(np.asarray(time_embedding, np.float32)[ts], H, window_hash,
                                   disable_initial=hash_disable_initial, soft_transition=hash_soft_transition)
    feats = hashgrid.ensemble_fwd(pn, tables_u16, H, geom, codew)
    base = mlp.mlp_fwd(feats.astype(np.float32), mlp_base_params, 0, 1 + geo_feat_dim, 0)
    density = np.exp(base[:, 0].astype(np.float32)) * selector.astype(np.float32)
    head_in = np.concatenate([(rays_d[ri] + np.float32(1.0)) / np.float32(2.0),
                              base[:, 1:1 + geo_feat_dim].astype(np.float32)], axis=1)
    rgb_s = mlp.mlp_fwd(head_in, mlp_head_params, 1, 3, 1).astype(np.float32)
    w, _, _ = march.render_weights(t0, t1, density, packed)
    acc = march.accumulate(w, None, packed)
    rgb = march.accumulate(w, rgb_s, packed) + np.float32(background) * (np.float32(1.0) - acc)
    t_mid = ((t0 + t1) / np.float32(2.0))[:, None]
    depth = march.accumulate(w, t_mid, packed) / (acc + np.float32(1e-10))
    if S > 0:
        depth = np.clip(depth, t_mid.min(), t_mid.max())
    return {"rgb": rgb, "depth": depth, "accumulation": acc, "deformation": march.accumulate(w, offsets, packed),
            "num_samples_per_ray": packed[:, 1], "ray_indices": ri, "t_starts": t0, "t_ends": t1, "timesteps": ts,
            "offsets": offsets, "density": density, "rgb_samples": rgb_s, "weights": w}
