"""Development aid: time the two ray-marching passes alone on the benchmark's rays (4096 rays of the synthetic rig,
P30 box, step 0.011) for a fully occupied grid (the warm-up window) and a pruned one (ellipsoid shell)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import nerfacc  # noqa: E402
from nersemble_amd._lib import check, lib, ptr, stream  # noqa: E402
from nersemble_amd.data.synthetic import SyntheticNeRSembleData  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--tag", default="")
ap.add_argument("--contend", action="store_true", help="run the table Adam pass on a side stream meanwhile")
a = ap.parse_args()
dev = torch.device("cuda:0")
box = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
data = SyntheticNeRSembleData(box, n_timesteps=30, n_rays=4096, device=dev)
bundle, _ = data.next_train(0)
grid = nerfacc.OccGridEstimator(box.reshape(-1), resolution=128, levels=1).to(dev)
o, d = bundle.origins.contiguous(), bundle.directions.contiguous()
R = o.shape[0]
torch.manual_seed(0)
near = torch.full((R,), 0.2, device=dev) + torch.rand(R, device=dev) * 0.011
res = 128
c = (torch.stack(torch.meshgrid(*[torch.arange(res, device=dev)] * 3, indexing="ij"), -1).float() + 0.5) / res
world = c * (box[1] - box[0]).to(dev) + box[0].to(dev)
ell = (((world - data.center) / (data.semi_axes * 1.15)) ** 2).sum(-1) <= 1.0


side = torch.cuda.Stream()
if a.contend:
    N = 400 * 1000 * 1000
    hog = [torch.zeros(N, device=dev) for _ in range(4)]          # grad, master, exp_avg, exp_avg_sq
    hog16 = torch.zeros(N, device=dev, dtype=torch.float16)


def hog_once():
    """One 403 M-parameter Adam pass (~2 ms of 5.5 TB/s streaming) on the side stream."""
    check(lib().nsx_adam_dense(ptr(hog[0]), N, ptr(hog[1]), ptr(hog[2]), ptr(hog[3]), ptr(hog16), 1e-3, 0.9, 0.999,
                               1e-15, 1, None, None, side.cuda_stream), "adam")


def timed(fn):
    """Average duration of one call; with --contend every call starts while an Adam pass streams on the side."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    total = 0.0
    for _ in range(a.iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if a.contend:
            hog_once()
            # let the hog get going before the timed kernel is queued
            torch.cuda._sleep(200000)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        total += s.elapsed_time(e)
    return total / a.iters


for name, occ in (("full", torch.ones_like(ell)), ("pruned", ell)):
    grid.binaries[0] = occ
    binary = grid.binaries[0].contiguous().view(torch.uint8)
    counts = torch.empty((R,), dtype=torch.int64, device=dev)
    packed = torch.empty((R, 2), dtype=torch.int64, device=dev)
    total = torch.zeros((1,), dtype=torch.int64, device=dev)

    def count():
        check(lib().nsx_march_count(ptr(o), ptr(d), R, grid._aabb_host, ptr(binary), res, ptr(near), 1e3, 0.011,
                                    ptr(counts), stream()), "count")
    count()
    check(lib().nsx_pack_info(ptr(counts), R, ptr(packed), ptr(total), stream()), "pack")
    S = int(total.item())
    t0 = torch.empty((S,), device=dev)
    t1 = torch.empty((S,), device=dev)
    ri = torch.empty((S,), dtype=torch.int64, device=dev)

    def fill():
        check(lib().nsx_march_fill(ptr(o), ptr(d), R, grid._aabb_host, ptr(binary), res, ptr(near), 1e3, 0.011,
                                   ptr(packed), ptr(t0), ptr(t1), ptr(ri), None, stream()), "fill")
    print(f"{a.tag} {name}: S={S} max/ray={int(counts.max())} count {timed(count) * 1e3:.1f} us  "
          f"fill {timed(fill) * 1e3:.1f} us", flush=True)
