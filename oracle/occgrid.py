"""numpy-facing wrappers of oracle/occgrid.c (occupancy-grid update).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from .capi import lib, ptr


def philox4x32_10(ctr, key):
    c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
    out = (C.c_uint32 * 4)()
    lib().nsxo_philox4x32_10(c, k, out)
    return [int(v) for v in out]


def sample_cells(binaries, aabb, warmup: bool, seed: int, step: int, n_timesteps: int):
    """binaries [res,res,res] bool -> (cell_ids int32 [M], positions fp32 [M,3], timesteps int32 [M], times fp32 [M])."""
    b = np.ascontiguousarray(binaries, dtype=np.uint8)
    res = b.shape[-1]
    b = b.reshape(-1)
    N = res ** 3
    assert b.shape[0] == N
    L = lib()
    L.nsxo_occ_num_slots.restype = C.c_int64
    L.nsxo_occ_sample_cells.restype = C.c_int64
    M = int(L.nsxo_occ_num_slots(C.c_int64(N), C.c_int64(int(b.sum())), C.c_int(int(warmup))))
    cells = np.empty(M, np.int32)
    pos = np.empty((M, 3), np.float32)
    ts = np.empty(M, np.int32)
    times = np.empty(M, np.float32)
    aabb = np.ascontiguousarray(aabb, dtype=np.float32).reshape(6)
    got = L.nsxo_occ_sample_cells(ptr(b), C.c_int(res), ptr(aabb), C.c_int(int(warmup)), C.c_uint64(seed),
                                  C.c_int64(step), C.c_int(n_timesteps), ptr(cells), ptr(pos), ptr(ts), ptr(times))
    assert got == M
    return cells, pos, ts, times


def update(occs, binaries, cell_ids, occ_values, ema_decay=0.95, occ_thre=0.01):
    """Returns (occs_new fp32 [N], binaries_new bool [N], threshold)."""
    o = np.array(occs, dtype=np.float32).reshape(-1).copy()
    b = np.array(binaries, dtype=np.uint8).reshape(-1).copy()
    cells = np.ascontiguousarray(cell_ids, dtype=np.int32)
    vals = np.ascontiguousarray(occ_values, dtype=np.float32).reshape(-1)
    assert cells.shape == vals.shape
    f = lib().nsxo_occ_update
    f.restype = C.c_float
    thre = f(ptr(o), ptr(b), C.c_int64(o.shape[0]), ptr(cells), ptr(vals), C.c_int64(cells.shape[0]),
             C.c_float(ema_decay), C.c_float(occ_thre))
    return o, b.astype(bool), float(thre)
