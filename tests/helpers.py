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
