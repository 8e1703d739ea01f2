"""Development aid: the two forwards of the deformation field against each other on the same inputs -- nsx_deform_fwd (general:
per-sample code rows, 11 K-steps in the input GEMMs) and nsx_deform_fwd_rows (codes are rows of a table: the code columns
summed per row first; terms in LDS for <= 64 rows, in L2 otherwise) -- each timed alone at S samples, outputs compared.

    python tools/deform_fwd_ab.py [--S 1048576] [--iters 20] [--rows 24,1,100,475]

(Rounds 3-4 A/B-ed kernel VARIANTS of nsx_deform_fwd through an environment variable read inside the entry point -- two-block,
pinned LDS reads, anti-phase waves, timing probes with wrong results on purpose.  All measured equal or slower, DESIGN.md 7b;
their source is in the history at commit 010eae6 and no longer in libnsx.so.)
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOP_PER_SAMPLE = 253_952          # the general kernel's (what the reference's eight Linear layers spend)
PEAK_TFLOPS = 2500.0


def timeit(fn, iters):
    for _ in range(3):
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
    ap.add_argument("--S", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rows", default="24,1,64,100,475")
    a = ap.parse_args()
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(dev)
    with torch.no_grad():
        for p in df.parameters():
            if p.requires_grad and p.dim() == 2:
                p.mul_(3.0)                               # (non-trivial offsets)
    packed, aabb6, w7 = df.packed_params(), df._aabb6(), F.deform_window7(3.5)
    S = a.S
    g = torch.Generator().manual_seed(S)
    pos = (torch.rand(S, 3, generator=g) * (aabb[1] - aabb[0]) + aabb[0]).to(dev)
    line = {}
    for T in [int(t) for t in a.rows.split(",")]:
        table = (torch.randn(T, 128, generator=g) * 0.3).to(dev)
        slot = torch.randint(0, T, (S,), generator=g, dtype=torch.int32).to(dev)
        want, got = torch.empty((S, 3), device=dev), torch.empty((S, 3), device=dev)
        terms = torch.empty((int(lib().nsx_deform_terms_floats(T)),), device=dev)

        def general():
            check(lib().nsx_deform_fwd(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0), ptr(slot), w7, ptr(want),
                                       None, stream()), "nsx_deform_fwd")

        def rows():
            check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0), ptr(slot), T, w7,
                                            ptr(got), ptr(terms), None, stream()), "nsx_deform_fwd_rows")

        ms_g, ms_r = timeit(general, a.iters), timeit(rows, a.iters)
        frac = lambda ms: round(S * FLOP_PER_SAMPLE / (ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4)
        line[f"rows_{T}"] = {"general_ms": round(ms_g, 4), "rows_ms": round(ms_r, 4), "general_frac_of_mfma_peak": frac(ms_g),
                             "rows_frac_of_mfma_peak_in_general_flops": frac(ms_r),
                             "terms_in": "LDS" if T <= 64 else "L2",
                             "max_abs_diff": float((got - want).abs().max()), "abs_max": float(want.abs().max()),
                             "finite": bool(torch.isfinite(got).all())}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
