"""Development aid: nsx_adam_hash_factored alone, with the five big streams placed in different ways."""
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


def run(name, alloc):
    master, m, v = alloc(n * 4).view(torch.float32), alloc(n * 4).view(torch.float32), alloc(n * 4).view(torch.float32)
    f16 = alloc(n * 2).view(torch.float16)
    G = alloc(SLOTS * total * 2 * 4).view(torch.float32)
    for t in (master, m, v, G):
        t.zero_()
    G.view(-1)[::97] = 1e-3
    code = torch.randn(SLOTS, H, device=dev)
    ts = []
    for it in range(6):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(lib().nsx_adam_hash_factored(ptr(G), SLOTS, ptr(code), code.stride(0), None, H, C.byref(g), ptr(master), ptr(m),
                                           ptr(v), ptr(f16), 5e-3, 0.9, 0.999, 1e-15, it + 1, None, None, stream()), "adam")
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    addrs = [hex(t.data_ptr()) for t in (master, m, v, f16, G)]
    print(f"{name:28s} min {min(ts):.3f} ms  med {sorted(ts)[3]:.3f} ms   {addrs}")


def separate(nbytes):
    return torch.empty(nbytes, dtype=torch.uint8, device=dev)


class Carve:
    def __init__(self, pad):
        self.buf = torch.empty(12 * 2 ** 30, dtype=torch.uint8, device=dev)
        self.off, self.pad = 0, pad

    def __call__(self, nbytes):
        t = self.buf[self.off:self.off + nbytes]
        self.off += (nbytes + self.pad + 255) // 256 * 256
        return t


run("separate allocations", separate)
run("one block, contiguous", Carve(0))
run("one block, +1 MiB+4 KiB pad", Carve((1 << 20) + 4096))
run("one block, +333 KiB pad", Carve(333 * 1024))
run("separate allocations (again)", separate)
