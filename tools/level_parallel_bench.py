"""Development aid: what ONE rank of a level-parallel data-parallel run (engine/level_parallel.py) computes, timed on one GPU --
the kernels of rank 0 of W ranks on W x S samples against L / W levels of the reference geometry (H = 32), beside the same
kernels of the data-parallel replica (S samples, all 16 levels) and the optimizer passes of both schemes.  No collectives:
the links' share of a step stays arithmetic; this replaces the arithmetic for the KERNELS.

    python tools/level_parallel_bench.py [--samples 100000,900000] [--world 2,4,8]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib, functional as F  # noqa: E402
from nersemble_amd._lib import check, lib, ptr, stream  # noqa: E402
from nersemble_amd.engine.level_parallel import sub_geometry  # noqa: E402


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
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
    ap.add_argument("--samples", default="100000,350000,900000")
    ap.add_argument("--world", default="2,4,8")
    ap.add_argument("--adam-only", action="store_true", help="only the optimizer passes")
    ap.add_argument("--last-rank-only", action="store_true", help="only the owner of the finest levels (profiling runs)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H, T = 32, 24
    g = _lib.grid_geometry()
    gen = torch.Generator(device=dev).manual_seed(0)
    f16 = ((torch.rand((g.total_entries, 2, H), device=dev, generator=gen) - 0.5) * 0.2).half()
    master = f16.float()
    m, v = torch.zeros_like(master), torch.zeros_like(master)
    one, zero = torch.ones((1,), device=dev), torch.zeros((1,), device=dev)
    win = torch.ones((H,), device=dev)
    out = {"H": H, "entries": g.total_entries, "note": "uniformly random positions (no ray coherence): the gathers' worst case"}

    def kernels(geom, tables, S, planes, tag):
        x = torch.rand((S, 3), device=dev, generator=gen)
        code = torch.randn((T, H), device=dev, generator=gen) * 0.5
        slot = torch.randint(0, T, (S,), device=dev, generator=gen, dtype=torch.int32)
        L = geom.n_levels
        feats = torch.empty((S, 2 * L), device=dev, dtype=torch.float16)
        dout = torch.randn((S, 2 * L), device=dev, generator=gen)
        G = torch.zeros((planes, geom.total_entries, 2), device=dev)
        dx = torch.empty((S, 3), device=dev)
        rows = torch.empty((T, H), device=dev)

        def fwd():
            check(lib().nsx_hash_ensemble_fwd(ptr(x), S, ptr(tables), H, C.byref(geom), ptr(code), code.stride(0), ptr(slot),
                                              ptr(win), ptr(feats), None, stream()), "fwd")

        def bwd():
            check(lib().nsx_hash_ensemble_bwd_codesum(ptr(x), S, ptr(tables), H, C.byref(geom), ptr(code), code.stride(0), T,
                                                      ptr(slot), ptr(win), ptr(dout), ptr(G[:T]), ptr(rows),
                                                      ptr(F.codesum_scratch(T, H, dev)), ptr(dx), None, None, stream()), "bwd")
        return {"S": S, "levels": L, f"{tag}_fwd_ms": round(timeit(fwd), 4), f"{tag}_bwd_ms": round(timeit(bwd), 4)}

    def adam(geom, e0, e1, planes, consume):
        codes = torch.randn((planes, H), device=dev, generator=gen) * 0.5
        G = torch.zeros((planes, e1 - e0, 2), device=dev)
        fn = lib().nsx_adam_hash_factored_consume if consume else lib().nsx_adam_hash_factored

        def step():
            check(fn(ptr(G), planes, ptr(codes), codes.stride(0), ptr(win), H, C.byref(geom), ptr(master[e0:e1]), ptr(m[e0:e1]),
                     ptr(v[e0:e1]), ptr(f16[e0:e1]), 0.0, 0.9, 0.999, 1e-15, 3, ptr(one), ptr(zero), stream()), "adam")
        return round(timeit(step), 4)

    out["replica"] = {} if a.last_rank_only else {"adam_full_table_24_planes_ms": adam(g, 0, g.total_entries, T, False)}
    for S in ([] if a.adam_only else [int(s) for s in a.samples.split(",")]):
        out["replica"][f"S={S}"] = kernels(g, f16, S, T, "replica")
    for W in [int(w) for w in a.world.split(",")]:
        n_own = g.n_levels // W
        res = {}
        for r in ((W - 1,) if a.last_rank_only else (0, W - 1)):   # the coarsest and the finest levels' owner
            sg = sub_geometry(g, r * n_own, n_own)
            e0, e1 = int(g.offset[r * n_own]), int(g.offset[(r + 1) * n_own])
            rr = {"levels": [r * n_own, (r + 1) * n_own], "entries": e1 - e0,
                  "adam_slice_ms": adam(sg, e0, e1, min(W * T, 192), False),
                  "adam_slice_consume_ms": adam(sg, e0, e1, min(W * T, 192), True)}
            for S in ([] if a.adam_only else [int(s) for s in a.samples.split(",")]):
                rr[f"S={S}/rank"] = kernels(sg, f16[e0:e1], W * S, min(W * T, 192), "level_parallel")
            res[f"rank{r}"] = rr
        out[f"world_{W}"] = res
    print(json.dumps(out))


if __name__ == "__main__":
    main()
