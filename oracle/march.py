"""numpy-facing wrappers of oracle/march.c (nerfacc 0.5.2 / torch_efficient_distloss restatements).
TEST INFRASTRUCTURE ONLY."""
import ctypes as C

import numpy as np

from .capi import lib, ptr


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def march(rays_o, rays_d, aabb, binary, near, far_plane, step, want_cells=False):
    """Single-level occupancy-grid traversal. binary [res,res,res] bool; near [R] per-ray near plane.
    Returns (ray_indices int64 [S], t_starts [S], t_ends [S], packed_info int64 [R,2][, cells int32 [S]])."""
    rays_o, rays_d, aabb, near = _f32(rays_o), _f32(rays_d), _f32(aabb).reshape(6), _f32(near)
    b = np.ascontiguousarray(binary, dtype=np.uint8)
    res = b.shape[0]
    assert b.shape == (res, res, res)
    R = rays_o.shape[0]
    counts = np.zeros(R, dtype=np.int64)
    L = lib()
    L.nsxo_march_count(ptr(rays_o), ptr(rays_d), C.c_int64(R), ptr(aabb), ptr(b), C.c_int(res), ptr(near),
                       C.c_float(far_plane), C.c_float(step), ptr(counts))
    starts = np.cumsum(counts) - counts
    S = int(counts.sum())
    t0 = np.empty(S, np.float32)
    t1 = np.empty(S, np.float32)
    ri = np.empty(S, np.int64)
    cells = np.empty(S, np.int32) if want_cells else None
    L.nsxo_march_fill(ptr(rays_o), ptr(rays_d), C.c_int64(R), ptr(aabb), ptr(b), C.c_int(res), ptr(near),
                      C.c_float(far_plane), C.c_float(step), ptr(starts), ptr(t0), ptr(t1), ptr(ri),
                      ptr(cells) if want_cells else None)
    packed = np.stack([starts, counts], axis=1)
    if want_cells:
        return ri, t0, t1, packed, cells
    return ri, t0, t1, packed


def pack_info(ray_indices, n_rays):
    counts = np.bincount(np.asarray(ray_indices, dtype=np.int64), minlength=n_rays).astype(np.int64)
    starts = np.cumsum(counts) - counts
    return np.stack([starts, counts], axis=1)


def render_weights(t0, t1, sigma, packed):
    t0, t1, sigma = _f32(t0), _f32(t1), _f32(sigma)
    packed = np.ascontiguousarray(packed, dtype=np.int64)
    S = t0.shape[0]
    w, T, a = np.empty(S, np.float32), np.empty(S, np.float32), np.empty(S, np.float32)
    lib().nsxo_render_weights(ptr(t0), ptr(t1), ptr(sigma), ptr(packed), C.c_int64(packed.shape[0]), ptr(w), ptr(T),
                              ptr(a))
    return w, T, a


def render_weights_bwd(t0, t1, sigma, packed, gw):
    t0, t1, sigma, gw = _f32(t0), _f32(t1), _f32(sigma), _f32(gw)
    packed = np.ascontiguousarray(packed, dtype=np.int64)
    ds = np.empty_like(sigma)
    lib().nsxo_render_weights_bwd(ptr(t0), ptr(t1), ptr(sigma), ptr(packed), C.c_int64(packed.shape[0]), ptr(gw),
                                  ptr(ds))
    return ds


def accumulate(w, values, packed):
    w = _f32(w)
    packed = np.ascontiguousarray(packed, dtype=np.int64)
    R = packed.shape[0]
    if values is None:
        out = np.empty((R, 1), np.float32)
        lib().nsxo_accumulate(ptr(w), None, C.c_int(1), ptr(packed), C.c_int64(R), ptr(out))
        return out
    values = _f32(values)
    Cc = values.shape[1]
    out = np.empty((R, Cc), np.float32)
    lib().nsxo_accumulate(ptr(w), ptr(values), C.c_int(Cc), ptr(packed), C.c_int64(R), ptr(out))
    return out


def distloss(w, m, interval, packed, n_rays, want_grad=True):
    w, m, interval = _f32(w), _f32(m), _f32(interval)
    packed = np.ascontiguousarray(packed, dtype=np.int64)
    g = np.empty_like(w) if want_grad else None
    f = lib().nsxo_distloss
    f.restype = C.c_double
    loss = f(ptr(w), ptr(m), ptr(interval), ptr(packed), C.c_int64(packed.shape[0]), C.c_int64(n_rays),
             ptr(g) if want_grad else None)
    return float(loss), g
