"""Development aid: the MFMA kernels alone on S samples (default 2^20): deform_fwd, deform_bwd (chain + wgrad + finish),
mlp_fwd / mlp_bwd for mlp_base (32 -> 64 -> 16) and mlp_head (18 -> 64 -> 64 -> 3).  Prints one JSON line with the
HIP-event time and the fraction of the dense fp16 MFMA peak of each; `--iters 1 --warmup 1` is what the counter passes
of tools/sq_counters.sh run.

    python tools/mfma_bench.py [--S N] [--iters K] [--only deform_fwd,mlp_bwd_head,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import functional as F  # noqa: E402
from nersemble_amd._lib import check, lib, ptr, stream  # noqa: E402
from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig  # noqa: E402

PEAK_TFLOPS = 2500.0          # dense fp16 MFMA peak of one MI355X (MI355X_MICROARCH.md)
# backward kernels recompute the forward: forward + dX chain + weight gradients = 3 x the forward's FLOPs (bench.py's model)
FLOP = {"deform_fwd": 253_952, "deform_fwd_general": 253_952, "deform_bwd": 3 * 253_952, "mlp_fwd_base": 6_144, "mlp_bwd_base": 3 * 6_144,
        "mlp_fwd_head": 14_336, "mlp_bwd_head": 3 * 14_336}


def timeit(fn, iters, warmup):
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
    ap.add_argument("--S", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--only", type=str, default="")
    a = ap.parse_args()
    only = set(x for x in a.only.split(",") if x)
    dev = torch.device("cuda:0")
    S = a.S
    torch.manual_seed(0)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(dev)
    pos = (torch.rand(S, 3) * (aabb[1] - aabb[0]) + aabb[0]).to(dev)
    table = (torch.randn(24, 128) * 0.3).to(dev).requires_grad_(True)
    slot = torch.randint(0, 24, (S,), dtype=torch.int32, device=dev)
    g = torch.randn(S, 3, device=dev)
    feats = torch.randn(S, 32, device=dev).half()
    base_out = torch.randn(S, 16, device=dev).half()
    dirs = torch.randn(S, 3, device=dev)
    fns = {}

    def deform_fwd():
        with torch.no_grad():
            df.compute_offsets(pos, table, 3.5, code_index=slot)

    off = df.compute_offsets(pos, table, 3.5, code_index=slot)

    def deform_bwd():
        off.backward(g, retain_graph=True)

    codes_per_sample = table.detach()[slot.long()].contiguous()

    def deform_fwd_general():             # the per-sample-code operator: nsx_deform_fwd (11 K-steps in the input GEMMs)
        with torch.no_grad():
            df.compute_offsets(pos, codes_per_sample, 3.5)

    # deform_fwd: codes are rows of a table -> nsx_deform_fwd_rows (the model's route; priced in the general kernel's FLOPs)
    fns["deform_fwd"], fns["deform_fwd_general"], fns["deform_bwd"] = deform_fwd, deform_fwd_general, deform_bwd
    for nh, name in ((0, "base"), (1, "head")):
        w = (torch.randn(F.mlp_param_count(nh), device=dev) * 0.1).half()
        dW = torch.zeros(w.numel(), device=dev)
        if nh == 0:
            out = torch.empty(S, 16, device=dev).half()
            dout = torch.randn(S, 16, device=dev).half()
            db32 = torch.empty(S, 32, device=dev)
            fns["mlp_fwd_base"] = lambda w=w, out=out: check(lib().nsx_mlp_fwd(
                ptr(w), 0, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, 0, ptr(out), 16, None, stream()), "f")
            fns["mlp_bwd_base"] = lambda w=w, dout=dout, dW=dW, db32=db32: check(lib().nsx_mlp_bwd(
                ptr(w), 0, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, 0, ptr(dout), 16, ptr(dW), None, None,
                ptr(db32), None, stream()), "b")
        else:
            out3 = torch.empty(S, 3, device=dev).half()
            dout3 = torch.randn(S, 3, device=dev).half()
            dbo = torch.zeros(S, 16, device=dev).half()
            fns["mlp_fwd_head"] = lambda w=w, out3=out3: check(lib().nsx_mlp_fwd(
                ptr(w), 1, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, 1, ptr(out3), 3, None, stream()), "f")
            fns["mlp_bwd_head"] = lambda w=w, dout3=dout3, dW=dW, dbo=dbo: check(lib().nsx_mlp_bwd(
                ptr(w), 1, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, 1, ptr(dout3), 3, ptr(dW), None,
                ptr(dbo), None, None, stream()), "h")
    res = {"S": S, "peak_tflops": PEAK_TFLOPS}
    for name, fn in fns.items():
        if only and name not in only:
            continue
        ms = timeit(fn, a.iters, a.warmup)
        tf = FLOP[name] * S / ms / 1e9
        res[name] = {"ms": round(ms, 4), "tflops": round(tf, 1), "frac_of_mfma_peak": round(tf / PEAK_TFLOPS, 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
