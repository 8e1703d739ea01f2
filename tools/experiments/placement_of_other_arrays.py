"""Experiment (round 6): engine/placement.py found the table optimizer's 12 GB pass 20 % faster on some physical placements of its
arrays than on others.  Do the other two large streams of the step care?  (a) the factored gradient G (24 planes, 1.2 GB) under
nsx_hash_ensemble_bwd_scatter -- memory-side atomics; (b) the fp16 working tables (806 MB) under nsx_hash_ensemble_fwd.  K fresh
allocations each, held at once, the same kernel on the same ray-ordered samples timed on every one."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from nersemble_amd import _lib  # noqa: E402
from nersemble_amd._lib import check, lib, ptr, stream  # noqa: E402

dev = torch.device("cuda:0")
g = _lib.grid_geometry()
total = int(g.offset[g.n_levels])
S, T, H, K = 1 << 20, 24, 32, 10
torch.manual_seed(0)
o = torch.rand(4096, 1, 3, device=dev) * 0.5 + 0.1
d = torch.nn.functional.normalize(torch.rand(4096, 1, 3, device=dev) - 0.3, dim=-1)
t = torch.arange(256, device=dev).view(1, 256, 1) * 0.0015
x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
slot = (torch.arange(S, device=dev) // 256 % T).int()
dout = torch.randn(S, 32, device=dev)
code = torch.randn(T, H, device=dev)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


Gs = [torch.zeros(T, total, 2, device=dev) for _ in range(K)]
ms = [timed(lambda G=G: check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), S, C.byref(g), T, ptr(slot), ptr(dout), ptr(G), None, 8,
                                                                   None, stream()), "scatter")) for G in Gs]
print("scatter over", K, "placements of G (ms):", [round(v, 3) for v in ms], "spread", round(max(ms) / min(ms), 3))
again = [timed(lambda G=G: check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), S, C.byref(g), T, ptr(slot), ptr(dout), ptr(G), None, 8,
                                                                      None, stream()), "scatter")) for G in Gs]
print("the same placements again:          ", [round(v, 3) for v in again])
del Gs
tabs = [(torch.rand(total, 2, H, device=dev) - 0.5).half() for _ in range(K)]
out = torch.empty(S, 32, dtype=torch.float16, device=dev)
ms = [timed(lambda tb=tb: check(lib().nsx_hash_ensemble_fwd(ptr(x), S, ptr(tb), H, C.byref(g), ptr(code), code.stride(0), ptr(slot),
                                                             None, ptr(out), None, stream()), "fwd")) for tb in tabs]
print("forward over", K, "placements of the fp16 tables (ms):", [round(v, 3) for v in ms], "spread", round(max(ms) / min(ms), 3))
xu = torch.rand(S, 3, device=dev)
ms = [timed(lambda tb=tb: check(lib().nsx_hash_ensemble_fwd(ptr(xu), S, ptr(tb), H, C.byref(g), ptr(code), code.stride(0), ptr(slot),
                                                             None, ptr(out), None, stream()), "fwd")) for tb in tabs]
print("forward, uniform samples (ms):      ", [round(v, 3) for v in ms], "spread", round(max(ms) / min(ms), 3))
