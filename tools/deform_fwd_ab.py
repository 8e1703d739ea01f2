"""Development aid: the variants of nsx_deform_fwd (NSX_DEFORM_FWD, read once per process) against each other -- each in its
own process on the same inputs: outputs must be bit-identical to variant 1, the kernel is timed alone at S samples.

    python tools/deform_fwd_ab.py [--variants 1,4,5] [--S 1048576] [--iters 20]          (driver)
    NSX_DEFORM_FWD=4 python tools/deform_fwd_ab.py --worker out.pt                          (one variant)
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOP_PER_SAMPLE = 253_952
PEAK_TFLOPS = 2500.0


def worker(path, S, iters):
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(dev)
    with torch.no_grad():
        for p in df.parameters():
            if p.requires_grad and p.dim() == 2:
                p.mul_(3.0)                               # (non-trivial offsets)
    res = {}
    for n in (100_003, 257, S):                           # ragged tails, fewer tiles than waves, the timed size
        g = torch.Generator().manual_seed(n)
        pos = (torch.rand(n, 3, generator=g) * (aabb[1] - aabb[0]) + aabb[0]).to(dev)
        table = (torch.randn(24, 128, generator=g) * 0.3).to(dev)
        slot = torch.randint(0, 24, (n,), generator=g, dtype=torch.int32).to(dev)
        with torch.no_grad():
            res[n] = df.compute_offsets(pos, table, 3.5, code_index=slot).cpu()
    with torch.no_grad():
        for _ in range(3):
            df.compute_offsets(pos, table, 3.5, code_index=slot)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            df.compute_offsets(pos, table, 3.5, code_index=slot)
        e.record()
        torch.cuda.synchronize()
    torch.save({"out": res, "ms": s.elapsed_time(e) / iters}, path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="1,4,5")
    ap.add_argument("--S", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--worker", default="")
    a = ap.parse_args()
    if a.worker:
        worker(a.worker, a.S, a.iters)
        return
    tmp = tempfile.mkdtemp()
    got = {}
    for v in a.variants.split(","):
        path = os.path.join(tmp, f"v{v}.pt")
        env = dict(os.environ, NSX_DEFORM_FWD="1", NSX_DEFORM_FWD_TERMS="1") if v == "T" else dict(os.environ, NSX_DEFORM_FWD=v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", path, "--S", str(a.S), "--iters",
                            str(a.iters)], env=env, capture_output=True, text=True, timeout=600)
        if r.returncode != 0 or not os.path.exists(path):
            got[v] = {"error": r.stderr[-1500:]}
            continue
        got[v] = torch.load(path)
    ref = got.get("1")
    line = {}
    for v, d in got.items():
        if "error" in d:
            line[v] = d
            continue
        same = None
        if ref is not None and "out" in ref:
            same = all(torch.equal(d["out"][n], ref["out"][n]) for n in ref["out"])
        finite = all(bool(torch.isfinite(t).all()) for t in d["out"].values())
        dmax = None
        if ref is not None and "out" in ref:
            dmax = float(max((d["out"][n] - ref["out"][n]).abs().max() for n in ref["out"]))
        line[v] = {"ms": round(d["ms"], 4), "frac_of_mfma_peak": round(a.S * FLOP_PER_SAMPLE / (d["ms"] * 1e-3) / 1e12
                                                                      / PEAK_TFLOPS, 4),
                   "bit_identical_to_variant_1": same, "max_abs_diff_to_variant_1": dmax, "finite": finite,
                   "abs_max": float(max(t.abs().max() for t in d["out"].values()))}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
