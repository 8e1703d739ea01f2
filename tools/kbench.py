"""Kernel micro-benchmarks (development aid): times individual libnsx kernels with HIP events."""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib, functional as F  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--log2B", type=int, default=20)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--generic", action="store_true", help="also time the generic (2H atomics per corner) backward")
    ap.add_argument("--coherent", action="store_true", help="ray-like spatially coherent samples instead of uniform")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, B = a.H, 1 << a.log2B
    g = _lib.grid_geometry()
    Hp = _lib.padded_grids(H)
    gen = torch.Generator(device=dev).manual_seed(0)
    master = (torch.rand((g.total_entries, 2, Hp), device=dev, generator=gen) - 0.5)
    f16 = master.half()
    if a.coherent:
        R = 4096
        o = torch.rand((R, 1, 3), device=dev, generator=gen)
        d = torch.nn.functional.normalize(torch.randn((R, 1, 3), device=dev, generator=gen), dim=-1)
        t = torch.linspace(0, 0.5, B // R, device=dev)[None, :, None]
        x = ((o + d * t) % 1.0).reshape(-1, 3).contiguous()
    else:
        x = torch.rand((B, 3), device=dev, generator=gen)
    T = 24
    emb = torch.randn((T, H), device=dev, generator=gen)
    ts = torch.randint(0, T, (B,), device=dev, generator=gen, dtype=torch.int32)
    dout = torch.randn((B, 32), device=dev, generator=gen)
    res = {"H": H, "B": B}
    ms = timeit(lambda: F._hash_ensemble_fwd_raw(x, f16, H, g, emb, ts, None), a.iters)
    bytes_fwd = B * (512 * H + 80)
    res["fwd_ms"] = ms
    res["fwd_GBps"] = bytes_fwd / ms / 1e6
    res["fwd_Msamples_s"] = B / ms / 1e3
    dtab = torch.zeros_like(master)
    dcode = torch.empty((B, H), device=dev)
    dx = torch.empty((B, 3), device=dev)
    import ctypes as C

    def bwd(dt=dtab):
        _lib.check(_lib.lib().nsx_hash_ensemble_bwd(_lib.ptr(x), B, _lib.ptr(f16), H, C.byref(g), _lib.ptr(emb),
                                                    emb.stride(0), _lib.ptr(ts), None, _lib.ptr(dout), _lib.ptr(dt),
                                                    _lib.ptr(dcode), _lib.ptr(dx), None, _lib.stream()))
    if a.generic:
        ms = timeit(bwd, 2, 1)
        res["bwd_generic_ms"] = ms
    ms = timeit(lambda: bwd(None), max(3, a.iters // 2))
    res["bwd_notable_ms"] = ms
    G = torch.zeros((T, g.total_entries, 2), device=dev)

    def bwdf(Gp=G):
        _lib.check(_lib.lib().nsx_hash_ensemble_bwd_factored(_lib.ptr(x), B, _lib.ptr(f16), H, C.byref(g),
                                                             _lib.ptr(emb), emb.stride(0), T, _lib.ptr(ts), None,
                                                             _lib.ptr(dout), _lib.ptr(Gp), _lib.ptr(dcode),
                                                             _lib.ptr(dx), None, None, _lib.stream()))
    res["bwd_factored_ms"] = timeit(bwdf, a.iters)
    res["bwd_factored_GBps"] = B * (1024 * H + 76) / res["bwd_factored_ms"] / 1e6
    res["expand_ms"] = timeit(lambda: _lib.check(_lib.lib().nsx_hash_grad_expand(
        _lib.ptr(G), T, _lib.ptr(emb), emb.stride(0), None, H, C.byref(g), _lib.ptr(dtab), 0, _lib.stream())), a.iters)
    res["memsetG_ms"] = timeit(lambda: G.zero_(), a.iters)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
