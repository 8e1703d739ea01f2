"""CPU restatement of the field-level compositions of the path.  TEST INFRASTRUCTURE ONLY (composes the per-stage
oracles of this package; nothing here is imported by the product).

  get_density / get_density_bwd   NeRSembleNeRFactoField.get_density  nersemble_nerfacto_field.py:250-301 and its
                                  autograd: scene-box normalisation :257, (0,1) selector + masking :268-269,
                                  HashEnsemble :278-281, mlp_base + split :285-286, trunc_exp * selector :292-293
                                  (nerfstudio ``trunc_exp``: fwd exp(x), bwd g * exp(clamp(x, -15, 15)) -- UPSTREAM,
                                  parity unpinned)
  get_rgb                         NeRSembleNeRFactoField.get_outputs  :303-383 (Identity direction encoding of
                                  (d + 1) / 2, cat with the 15 geometry features, mlp_head + sigmoid)
  field_density_fn                NeRSembleNGPModel.field_density_fn  nersemble_instant_ngp.py:235-266: timesteps =
                                  round(times * (T - 1)), the two embedding lookups, deformation offsets in
                                  NORMALISED space added to the WORLD position (:257-259, reproduced as is), then
                                  field.density_fn (nersemble_nerfacto_field.py:228-248: dummy frustums with
                                  starts = ends = 0, i.e. the positions themselves) -> get_density
"""
import numpy as np

from . import hashgrid, mlp


def sample_positions(rays_o, rays_d, ray_indices, t0, t1):
    """frustums.get_positions(): origins + directions * (starts + ends) / 2, in fp32 like the reference."""
    o = np.asarray(rays_o, dtype=np.float32)[ray_indices]
    d = np.asarray(rays_d, dtype=np.float32)[ray_indices]
    mid = ((np.asarray(t0, np.float32) + np.asarray(t1, np.float32)))[:, None]
    return o + (d * mid) / np.float32(2.0)


def normalise(positions, aabb):
    """SceneBox.get_normalized_positions + the (0, 1) selector that zeroes outside samples (:257, :268-269)."""
    aabb = np.asarray(aabb, dtype=np.float32).reshape(2, 3)
    pn = (np.asarray(positions, np.float32) - aabb[0]) / (aabb[1] - aabb[0])
    selector = ((pn > 0.0) & (pn < 1.0)).all(axis=-1)
    return pn * selector[:, None].astype(np.float32), selector


def get_density(positions_world, aabb, tables_u16, H, geom, codew, mlp_base_params, geo_feat_dim=15):
    """positions_world [S,3] fp32 (offsets already added) -> dict(density [S] fp32, base [S,16] fp16, pn, selector,
    features [S,2L] fp16).  ``codew`` [S,H] fp32 = conditioning code x grid window (hashgrid.windowed_code)."""
    pn, selector = normalise(positions_world, aabb)
    feats = hashgrid.ensemble_fwd(pn, tables_u16, H, geom, codew)
    base = mlp.mlp_fwd(feats.astype(np.float32), mlp_base_params, 0, 1 + geo_feat_dim, 0)          # [S,16] fp16
    density = np.exp(base[:, 0].astype(np.float32)) * selector.astype(np.float32)
    return {"density": density, "base": base, "pn": pn, "selector": selector, "features": feats}


def get_density_bwd(fwd, aabb, tables_u16, H, geom, codew, mlp_base_params, g_density, g_embedding, geo_feat_dim=15,
                    want_table=True):
    """Autograd of get_density for upstream gradients g_density [S] (fp32) and g_embedding [S,15] (the fp16 geometry
    features' gradient).  Returns dict(d_params fp64, d_table fp32 tcnn layout | None, d_codew [S,H], d_positions
    [S,3]).  As in the reference's tcnn path the gradient entering mlp_base is fp16 (the kernels round it)."""
    sel = fwd["selector"].astype(np.float32)
    h0 = fwd["base"][:, 0].astype(np.float32)
    d_base = np.zeros((h0.shape[0], 1 + geo_feat_dim), dtype=np.float64)
    d_base[:, 0] = (np.asarray(g_density, np.float32) * sel * np.exp(np.clip(h0, -15.0, 15.0))).astype(np.float16)
    d_base[:, 1:] = np.asarray(g_embedding).astype(np.float16)
    d_params, d_feat = mlp.mlp_bwd(fwd["features"].astype(np.float32), mlp_base_params, 0, 1 + geo_feat_dim, 0, d_base)
    d_table, d_codew, d_pn = hashgrid.ensemble_bwd(fwd["pn"], tables_u16, H, geom, codew,
                                                   d_feat[:, :2 * geom.n_levels].astype(np.float32),
                                                   want_table=want_table)
    aabb = np.asarray(aabb, dtype=np.float32).reshape(2, 3)
    d_pos = (d_pn * sel[:, None]) / (aabb[1] - aabb[0])
    return {"d_params": d_params, "d_table": d_table, "d_codew": d_codew, "d_positions": d_pos, "d_features": d_feat}


def get_rgb(directions, base, mlp_head_params, geo_feat_dim=15):
    """mlp_head([(d + 1) / 2, geometry features]) with sigmoid output -> [S,3] fp32 (:313, :371-377)."""
    head_in = np.concatenate([(np.asarray(directions, np.float32) + np.float32(1.0)) / np.float32(2.0),
                              np.asarray(base)[:, 1:1 + geo_feat_dim].astype(np.float32)], axis=1)
    return mlp.mlp_fwd(head_in, mlp_head_params, 1, 3, 1).astype(np.float32)


def timesteps_of(times, n_timesteps):
    """round(times * (T - 1)).int() -- torch.round is half-to-even, as np.round (nersemble_instant_ngp.py:249, :303)."""
    return np.round(np.asarray(times, np.float32).reshape(-1) * np.float32(n_timesteps - 1)).astype(np.int64)


def deformation_offsets(positions_world, codes, deform_params, aabb, window_deform, half=True):
    import torch
    from . import deform
    if positions_world.shape[0] == 0:
        return np.zeros((0, 3), dtype=np.float32)
    aabb = np.asarray(aabb, np.float32).reshape(2, 3)
    off = deform.compute_offsets(torch.from_numpy(np.ascontiguousarray(positions_world, dtype=np.float32)),
                                 torch.from_numpy(np.ascontiguousarray(codes, dtype=np.float32)),
                                 torch.as_tensor(deform_params, dtype=torch.float32), torch.from_numpy(aabb),
                                 window_deform, half=half)
    return off.to(torch.float32).numpy()


def field_density_fn(positions_world, times, n_timesteps, aabb, tables_u16, H, geom, mlp_base_params, time_embedding,
                     deform_params=None, deform_embedding=None, window_hash=None, window_deform=None,
                     hash_disable_initial=True, hash_soft_transition=True, disable_occupancy_grid=False):
    """nersemble_instant_ngp.py:235-266.  Returns (density [N] fp32, dict of intermediates)."""
    pos = np.ascontiguousarray(positions_world, dtype=np.float32)
    if disable_occupancy_grid:
        return np.ones((pos.shape[0],), dtype=np.float32), {}
    ts = timesteps_of(times, n_timesteps)
    offsets = None
    if deform_params is not None:
        emb = np.asarray(deform_embedding if deform_embedding is not None else time_embedding, dtype=np.float32)
        offsets = deformation_offsets(pos, emb[ts], deform_params, aabb, window_deform)
        pos = pos + offsets                      # normalised-space offset on a world-space position (:257-259)
    codew = hashgrid.windowed_code(np.asarray(time_embedding, np.float32)[ts], H, window_hash,
                                   disable_initial=hash_disable_initial, soft_transition=hash_soft_transition)
    out = get_density(pos, aabb, tables_u16, H, geom, codew, mlp_base_params)
    out["offsets"], out["timesteps"], out["codew"] = offsets, ts, codew
    return out["density"], out
