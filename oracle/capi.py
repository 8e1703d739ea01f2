"""ctypes loader for oracle/_build/libnsx_oracle.so (built with gcc from oracle/*.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libnsx_oracle.so")
NSXO_MAX_LEVELS = 32


class GridGeom(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32),
        ("log2_hashmap_size", C.c_int32),
        ("base_resolution", C.c_int32),
        ("per_level_scale", C.c_float),
        ("scale", C.c_float * NSXO_MAX_LEVELS),
        ("res", C.c_uint32 * NSXO_MAX_LEVELS),
        ("size", C.c_uint32 * NSXO_MAX_LEVELS),
        ("offset", C.c_uint32 * (NSXO_MAX_LEVELS + 1)),
    ]

    @property
    def total_entries(self) -> int:
        return int(self.offset[self.n_levels])


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if force or stale:
        subprocess.run(["make", "-s", "-C", _HERE, "-B", "_build/libnsx_oracle.so"], check=True)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.nsxo_f2h.restype = C.c_uint16
        _lib.nsxo_f2h.argtypes = [C.c_float]
        _lib.nsxo_h2f.restype = C.c_float
        _lib.nsxo_h2f.argtypes = [C.c_uint16]
    return _lib


def ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"], "oracle buffers must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def grid_geometry(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16,
                  log2_hashmap_size=19) -> GridGeom:
    """tcnn HashGrid geometry for the reference config (hash_ensemble.py:31-39)."""
    g = GridGeom()
    lib().nsxo_grid_geometry(C.c_int(n_levels), C.c_float(per_level_scale), C.c_int(base_resolution),
                             C.c_int(log2_hashmap_size), C.byref(g))
    return g
