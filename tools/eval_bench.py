"""Development aid: render one evaluation image (one timestep) with and without the pre-blended eval grid.
``--price``: per C-ABI entry point, the milliseconds one pre-blended image spends in it (HIP events around every native call
of the render, ``_lib.KernelProfiler``) beside the image's wall time -- the table DESIGN.md §5 prices the eval image with."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib  # noqa: E402
from nersemble_amd.workloads import build_workload  # noqa: E402

torch.manual_seed(0)
trainer, data, info = build_workload("p030_h32", device="cuda:0")
for s in range(20):
    trainer.train_iteration(s, *data.next_train(s))
model = trainer.model
model.eval()
bundle, batch, (h, w) = data.eval_image_rays(cam=2, timestep=5, downscale=4)
n = bundle.origins.shape[0]
chunk = 32768


def render():
    outs, samples = [], 0
    with torch.no_grad():
        for i in range(0, n, chunk):
            sl = slice(i, min(i + chunk, n))
            from nersemble_amd.rays import RayBundle
            b = RayBundle(origins=bundle.origins[sl], directions=bundle.directions[sl], pixel_area=bundle.pixel_area[sl],
                          camera_indices=bundle.camera_indices[sl], times=bundle.times[sl])
            o = model(b)
            outs.append(o["rgb"])
            samples += o["num_samples_per_ray"].sum()
    return torch.cat(outs), int(samples)


for fast, fused in ((False, True), (True, False), (True, True), (True, False), (True, True)):
    model.eval_preblend = fast
    model.field.fused_eval_density = fused         # (only read on the pre-blended route)
    render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img, samples = render()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"preblend={fast} fused_density={fused}: {h}x{w} = {n} rays, {samples} samples, {dt * 1e3:.1f} ms, {samples / dt / 1e6:.1f} M samples/s, "
          f"psnr {float(10 * torch.log10(1 / ((img - batch['image']) ** 2).mean())):.2f}")

if "--price" in sys.argv:
    model.eval_preblend = True
    prof = _lib.profiler
    prof.watch, prof.alias = None, {}
    prof.prewarm(4096)
    prof.reset()
    prof.enabled = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    img, samples = render()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof.enabled = False
    rows = sorted(prof.summary().items(), key=lambda kv: -kv[1]["total_ms"])
    native = sum(v["total_ms"] for _, v in rows)
    print(json.dumps({"image": f"{h}x{w}", "rays": n, "samples": samples, "wall_ms_with_events": round(dt * 1e3, 2),
                      "native_calls_ms": round(native, 2),
                      "per_entry_point": {k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 3),
                                              "avg_ms": round(v["avg_ms"], 4)} for k, v in rows}}))
