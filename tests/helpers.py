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
