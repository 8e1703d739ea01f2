"""Shared test helpers: seeded tcnn-layout tables and the small geometry used by fixtures."""
import numpy as np

# small geometry: levels 0-1 dense (16^3, 24^3), levels 2-15 hashed (2^15); tables are regenerated from a seed
SMALL_GEOM_KW = dict(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16, log2_hashmap_size=15)
REF_GEOM_KW = dict(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16, log2_hashmap_size=19)


def ens_layout(H: int):
    total = 2 * H
    f_enc = 8 if total >= 8 else total
    p = 4 if total >= 8 else H
    c = (total + 7) // 8
    return f_enc, p, c


def make_tcnn_tables(H: int, geom, seed: int, amplitude: float = 0.5) -> np.ndarray:
    """Seeded fp32 tables in the reference's tcnn layout [C, total, F_enc], values exactly fp16-representable."""
    f_enc, p, c = ens_layout(H)
    rng = np.random.default_rng(seed)
    t = (rng.random((c, geom.total_entries, f_enc), dtype=np.float32) * 2 - 1) * amplitude
    return t.astype(np.float16).astype(np.float32)


DEFORM_KEYS = [f"se3_field.mlp_stem.layers.{i}.{k}" for i in range(6) for k in ("weight", "bias")] + \
    ["se3_field.mlp_r.layers.0.weight", "se3_field.mlp_r.layers.0.bias",
     "se3_field.mlp_v.layers.0.weight", "se3_field.mlp_v.layers.0.bias"]


def make_deform_state_dict(seed: int, width: int = 128, code_dim: int = 128, head_scale: float = 0.05) -> dict:
    """Seeded fp32 state dict of an SE3DeformationField (reference key names, deformation_field.py:50-69) with
    nn.Linear-like magnitudes U(+-1/sqrt(fan_in)) and rotation / translation heads large enough to exercise the SE(3)
    exponential (the reference initialises them at +-1e-5, i.e. the identity).  numpy's PCG64 stream: the same
    arrays wherever this runs, so the goldens made from them need not store 127 756 weights."""
    rng = np.random.default_rng(seed)
    n_in = 45 + code_dim
    shapes = [(width, n_in), (width, width), (width, width), (width, width), (width, n_in + width), (width, width)]
    sd = {}
    for i, (o, k) in enumerate(shapes):
        bound = 1.0 / np.sqrt(k)
        sd[f"se3_field.mlp_stem.layers.{i}.weight"] = rng.uniform(-bound, bound, (o, k)).astype(np.float32)
        sd[f"se3_field.mlp_stem.layers.{i}.bias"] = rng.uniform(-bound, bound, (o,)).astype(np.float32)
    for head in ("mlp_r", "mlp_v"):
        sd[f"se3_field.{head}.layers.0.weight"] = rng.uniform(-head_scale, head_scale, (3, width)).astype(np.float32)
        sd[f"se3_field.{head}.layers.0.bias"] = rng.uniform(-head_scale, head_scale, (3,)).astype(np.float32)
    return sd


def make_smooth_tcnn_tables(H: int, geom, seed: int, amplitude: float = 0.5) -> np.ndarray:
    """Like ``make_tcnn_tables`` but "trained-like": the amplitude of level l falls with its resolution
    (amplitude * res[0] / res[l]), so the encoded field has a bounded spatial gradient at every level.  Composition
    tests feed positions that carry fp16-level noise from an earlier stage (the deformation offsets); with white-noise
    fine levels that noise would be amplified by the level's resolution and say nothing about the composition."""
    f_enc, p, c = ens_layout(H)
    rng = np.random.default_rng(seed)
    t = np.empty((c, geom.total_entries, f_enc), dtype=np.float32)
    for l in range(geom.n_levels):
        lo, hi = int(geom.offset[l]), int(geom.offset[l + 1])
        a = amplitude * float(geom.res[0]) / float(geom.res[l])
        t[:, lo:hi] = (rng.random((c, hi - lo, f_enc), dtype=np.float32) * 2 - 1) * a
    return t.astype(np.float16).astype(np.float32)


def randomise_model(model, seed: int, oracle_geom, table_amplitude: float = 0.5, code_std: float = 0.5,
                    head_scale: float = 0.2):
    """Gives a NeRSembleNGPModel non-trivial, seeded weights THROUGH ITS STATE DICT (the reference's key names and
    tcnn table layout) and returns the same weights as the numpy arrays the oracle consumes:
    dict(tables_u16 [C,total,F_enc], mlp_base, mlp_head, time_embedding, deform_embedding, deform_params, aabb)."""
    import torch
    from oracle import deform as od, mlp as omlp
    rng = np.random.default_rng(seed)
    H = model.field.hash_ensemble.n_hash_encodings
    tabs = make_smooth_tcnn_tables(H, oracle_geom, seed + 1, table_amplitude)
    sd = {f"field.hash_ensemble.hash_encodings.{c}.params": torch.from_numpy(tabs[c].reshape(-1))
          for c in range(tabs.shape[0])}

    def f16(a):
        return a.astype(np.float16).astype(np.float32)

    base = f16(rng.uniform(-1, 1, omlp.param_count(0)).astype(np.float32) * np.float32(np.sqrt(6.0 / (32 + 64))))
    head = f16(rng.uniform(-1, 1, omlp.param_count(1)).astype(np.float32) * np.float32(np.sqrt(6.0 / (64 + 64))))
    sd["field.mlp_base.params"] = torch.from_numpy(base)
    sd["field.mlp_head.params"] = torch.from_numpy(head)
    out = {"tables_u16": tabs.astype(np.float16).view(np.uint16), "mlp_base": base, "mlp_head": head,
           "time_embedding": None, "deform_embedding": None, "deform_params": None,
           "aabb": model.scene_box.aabb.detach().cpu().numpy().astype(np.float32)}
    if model.time_embedding is not None:
        T = model.time_embedding.weight.shape[0]
        te = (rng.standard_normal((T, H)) * code_std).astype(np.float32)
        sd["time_embedding.weight"] = torch.from_numpy(te)
        out["time_embedding"] = te
    if model.time_embedding_deformation is not None:
        T, Cd = model.time_embedding_deformation.weight.shape
        td = (rng.standard_normal((T, Cd)) * 0.3).astype(np.float32)
        sd["time_embedding_deformation.weight"] = torch.from_numpy(td)
        out["deform_embedding"] = td
    if model.deformation_field is not None:
        dsd = make_deform_state_dict(seed + 2, head_scale=head_scale)
        for k, v in dsd.items():
            sd["deformation_field." + k] = torch.from_numpy(v)
        out["deform_params"] = od.flat_from_state_dict(dsd).to(torch.float32).numpy()
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    return out


def export_oracle_weights(model) -> dict:
    """The model's CURRENT weights, read from ``state_dict()`` (the reference's key names, tcnn table layout), as the
    numpy arrays the oracle consumes.  Tables are rounded to fp16 exactly as the kernels' working copy is."""
    import torch
    from oracle import deform as od
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    C = model.field.hash_ensemble.n_tcnn_encodings
    total = model.field.hash_ensemble.geom.total_entries
    tabs = np.stack([sd[f"field.hash_ensemble.hash_encodings.{c}.params"].numpy().reshape(total, -1) for c in range(C)])
    out = {"tables_u16": np.ascontiguousarray(tabs.astype(np.float16)).view(np.uint16),
           "mlp_base": sd["field.mlp_base.params"].numpy().astype(np.float32),
           "mlp_head": sd["field.mlp_head.params"].numpy().astype(np.float32),
           "time_embedding": sd["time_embedding.weight"].numpy() if "time_embedding.weight" in sd else None,
           "deform_embedding": sd["time_embedding_deformation.weight"].numpy()
           if "time_embedding_deformation.weight" in sd else None,
           "deform_params": None, "aabb": model.scene_box.aabb.detach().cpu().numpy().astype(np.float32)}
    if model.deformation_field is not None:
        dsd = {k[len("deformation_field."):]: v for k, v in sd.items() if k.startswith("deformation_field.se3_field")}
        out["deform_params"] = od.flat_from_state_dict(dsd).to(torch.float32).numpy()
    return out
