"""numpy-facing wrappers of the C oracle for the hash grid / HashEnsemble (tcnn layout).

Layout conventions (the reference's, hash_ensemble.py:84-112):
  C = ceil(2H/8) encodings, each ``[total_entries, F_enc]`` fp16 with F_enc = 8 (or 2H if 2H < 8);
  logical grid h = c*P + p (P = 4, or H if 2H < 8), tcnn feature j = p*2 + f.
``tables`` here is a uint16 array ``[C, total_entries, F_enc]`` holding fp16 bit patterns.
"""
import ctypes as C
from math import ceil

import numpy as np

from .capi import lib, ptr, GridGeom, grid_geometry


def ens_layout(H: int):
    total = 2 * H
    f_enc = 8 if total >= 8 else total
    p = 4 if total >= 8 else H
    c = ceil(total / 8)
    return f_enc, p, c


def indices(x: np.ndarray, g: GridGeom):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, L = x.shape[0], g.n_levels
    idx = np.empty((B, L, 8), dtype=np.uint32)
    w = np.empty((B, L, 3), dtype=np.float32)
    lib().nsxo_hashgrid_indices(ptr(x), C.c_int64(B), C.byref(g), ptr(idx), ptr(w))
    return idx, w


def hashgrid_fwd(x: np.ndarray, table_u16: np.ndarray, g: GridGeom) -> np.ndarray:
    """One tcnn HashGrid encoding: table [total, F_enc] fp16 bits -> out [B, L*F_enc] fp16 (as np.float16)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    table_u16 = np.ascontiguousarray(table_u16.view(np.uint16))
    assert table_u16.shape[0] == g.total_entries
    f_enc = table_u16.shape[1]
    out = np.empty((x.shape[0], g.n_levels * f_enc), dtype=np.uint16)
    lib().nsxo_hashgrid_fwd(ptr(x), C.c_int64(x.shape[0]), ptr(table_u16), C.c_int(f_enc), C.byref(g), ptr(out))
    return out.view(np.float16)


def ensemble_fwd(x: np.ndarray, tables_u16: np.ndarray, H: int, g: GridGeom, codew: np.ndarray) -> np.ndarray:
    """Fused HashEnsemble forward; codew [B,H] fp32 = code * grid-window. Returns [B, 2L] np.float16."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    codew = np.ascontiguousarray(codew, dtype=np.float32)
    f_enc, p, c = ens_layout(H)
    tables_u16 = np.ascontiguousarray(tables_u16.view(np.uint16))
    assert tables_u16.shape == (c, g.total_entries, f_enc), tables_u16.shape
    assert codew.shape == (x.shape[0], H)
    out = np.empty((x.shape[0], g.n_levels * 2), dtype=np.uint16)
    lib().nsxo_ensemble_fwd(ptr(x), C.c_int64(x.shape[0]), ptr(tables_u16), C.c_int(H), C.byref(g),
                            ptr(codew), ptr(out))
    return out.view(np.float16)


def ensemble_fwd_fast(x: np.ndarray, tables_u16: np.ndarray, H: int, g: GridGeom, codew: np.ndarray) -> np.ndarray:
    """The CPU-baseline port of ``ensemble_fwd`` (fp32 accumulation, table-driven fp16 decode; nsx_oracle.c): what
    bench.py's ``cpu_baseline`` times.  NOT the checker."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    codew = np.ascontiguousarray(codew, dtype=np.float32)
    f_enc, p, c = ens_layout(H)
    tables_u16 = np.ascontiguousarray(tables_u16.view(np.uint16))
    assert tables_u16.shape == (c, g.total_entries, f_enc), tables_u16.shape
    assert codew.shape == (x.shape[0], H)
    out = np.empty((x.shape[0], g.n_levels * 2), dtype=np.uint16)
    lib().nsxo_ensemble_fwd_fast(ptr(x), C.c_int64(x.shape[0]), ptr(tables_u16), C.c_int(H), C.byref(g),
                                 ptr(codew), ptr(out))
    return out.view(np.float16)


def ensemble_bwd(x, tables_u16, H, g, codew, dout, want_table=True):
    """Returns (dtable fp32 [C,total,F_enc] or None, dcodew [B,H], dx [B,3])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    codew = np.ascontiguousarray(codew, dtype=np.float32)
    dout = np.ascontiguousarray(dout, dtype=np.float32)
    f_enc, p, c = ens_layout(H)
    tables_u16 = np.ascontiguousarray(tables_u16.view(np.uint16))
    B = x.shape[0]
    assert dout.shape == (B, g.n_levels * 2)
    dtable = np.zeros((c, g.total_entries, f_enc), dtype=np.float32) if want_table else None
    dcodew = np.empty((B, H), dtype=np.float32)
    dx = np.empty((B, 3), dtype=np.float32)
    lib().nsxo_ensemble_bwd(ptr(x), C.c_int64(B), ptr(tables_u16), C.c_int(H), C.byref(g), ptr(codew), ptr(dout),
                            ptr(dtable) if want_table else None, ptr(dcodew), ptr(dx))
    return dtable, dcodew, dx


def posenc_window(windows_param: float, min_bands: float, max_bands: float, n: int) -> np.ndarray:
    """Restates hash_ensemble.py:12-28 in float32 numpy."""
    bands = np.linspace(min_bands, max_bands, n, dtype=np.float32)
    xx = np.clip(np.float32(windows_param) - bands, 0, 1).astype(np.float32)
    return (0.5 * (1 - np.cos(np.float32(np.pi) * xx))).astype(np.float32)


def windowed_code(code: np.ndarray, H: int, window_hash_encodings, disable_initial=True, soft_transition=True):
    """Host glue of hash_ensemble.py:119-138 folded onto the code: returns code*window (fp32)."""
    code = np.asarray(code, dtype=np.float32).copy()
    if window_hash_encodings is None:
        return code
    w = float(window_hash_encodings)
    if w == 1 and disable_initial:
        code = np.ones_like(code)
    elif soft_transition and w < 2:
        alpha = np.float32(w - 1)
        code = alpha * code
        code[:, 0] += (1 - alpha) * 1
    win = posenc_window(w, 0, H - 1, H)
    return (code * win[None, :]).astype(np.float32)
