"""Round 6: the trainer's default configuration (compact first-grid phase with two gradient planes, the counting pass keeping its
samples) against the same run with both late changes switched off -- 4000 steps of p030_h32 each, PSNR of the training batches
averaged over the last 200 steps and an evaluation image at the end.  Same seeds; the runs differ by the order of their gradient
atomics, as two runs of either configuration do.

    python tools/soak_r06.py [--steps 4000]
"""
import argparse
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(steps):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)
    trainer, data, info = build_workload("p030_h32", device="cuda:0")
    psnr, t0 = [], time.time()
    for step in range(steps):
        nxt = data.next_train(step + 1)
        loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step), next_ray_bundle=nxt[0])
        if step >= steps - 200:
            psnr.append(metrics["psnr"])
    torch.cuda.synchronize()
    dt = time.time() - t0
    trainer.flush_scheduler_step()
    model = trainer.model
    model.eval()
    bundle, batch, (h, w) = data.eval_image_rays(cam=2, timestep=5, downscale=4)
    from nersemble_amd.rays import RayBundle
    outs = []
    with torch.no_grad():
        for i in range(0, bundle.origins.shape[0], 32768):
            sl = slice(i, i + 32768)
            outs.append(model(RayBundle(origins=bundle.origins[sl], directions=bundle.directions[sl], pixel_area=bundle.pixel_area[sl],
                                        camera_indices=bundle.camera_indices[sl], times=bundle.times[sl]))["rgb"])
    img = torch.cat(outs)
    ev = float(10 * torch.log10(1 / ((img - batch["image"]) ** 2).mean()))
    he = model.field.hash_ensemble
    print(f"planes={os.environ.get('NSX_FIRST_GRID_PLANES', 'default (2)')} stash={os.environ.get('NSX_MARCH_STASH', 'default (1)')}: "
          f"{steps} steps in {dt:.1f} s ({dt / steps * 1e3:.3f} ms/step incl. the loader), train PSNR (last 200 steps) "
          f"{float(torch.stack(psnr).mean()):.3f}, evaluation image PSNR {ev:.3f}, loss {loss.item():.5f}", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        one(a.steps)
    else:
        for env in ({}, {"NSX_FIRST_GRID_PLANES": "0", "NSX_MARCH_STASH": "0"}, {}, {"NSX_FIRST_GRID_PLANES": "0", "NSX_MARCH_STASH": "0"}):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--steps", str(a.steps)], env=dict(os.environ, **env), check=False)
