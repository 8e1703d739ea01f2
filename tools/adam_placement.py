"""Development aid: how much does the PLACEMENT of the table-Adam streams (master / exp_avg / exp_avg_sq / fp16 tables)
change the pass's duration?  (a) K sets of separate allocations held at the same time, (b) one block carved with
different inter-stream paddings."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib
from nersemble_amd._lib import check, lib, ptr, stream

dev = torch.device("cuda:0")
H, SLOTS = 32, 24
g = _lib.grid_geometry()
total = int(g.offset[g.n_levels])
n = total * 2 * H
G = torch.zeros(SLOTS * total * 2, device=dev)
G.view(-1)[::97] = 1e-3
code = torch.randn(SLOTS, H, device=dev)


def time_set(bufs, iters=6):
    master, m, v, f16 = bufs
    ts = []
    for it in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(lib().nsx_adam_hash_factored(ptr(G), SLOTS, ptr(code), code.stride(0), None, H, C.byref(g), ptr(master),
                                           ptr(m), ptr(v), ptr(f16), 5e-3, 0.9, 0.999, 1e-15, it + 1, None, None,
                                           stream()), "adam")
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    return min(ts), sorted(ts)[len(ts) // 2]


def fresh_set():
    return (torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev),
            torch.zeros(n, device=dev, dtype=torch.float16))


held = []
for k in range(6):
    bufs = fresh_set()
    held.append(bufs)
    lo, med = time_set(bufs)
    print(f"separate set {k}: min {lo:.3f} med {med:.3f}  {[hex(b.data_ptr()) for b in bufs]}", flush=True)
# the same sets again (is a set's speed a property of its pages?)
for k in (0, 3, 5):
    lo, med = time_set(held[k])
    print(f"separate set {k} again: min {lo:.3f} med {med:.3f}", flush=True)
del held
torch.cuda.empty_cache()

block = torch.empty(16 * 2 ** 30, dtype=torch.uint8, device=dev)
MB = 1 << 20
for pad in (0, 2 * MB, 6 * MB, 14 * MB, 62 * MB, 254 * MB, 1022 * MB, 4096, 64 * 1024 + 4096, 1 * MB):
    off, bufs = 0, []
    for nbytes, dt in ((n * 4, torch.float32),) * 3 + ((n * 2, torch.float16),):
        bufs.append(block[off:off + nbytes].view(dt))
        off += (nbytes + pad + 255) // 256 * 256
    for b in bufs:
        b.zero_()
    lo, med = time_set(bufs)
    print(f"one block, pad {pad / MB:9.3f} MiB: min {lo:.3f} med {med:.3f}", flush=True)
