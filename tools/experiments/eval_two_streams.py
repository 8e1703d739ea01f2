"""Experiment: chunks of one evaluation image on two streams (chunk 0 on the main stream first: it builds the per-image caches)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from nersemble_amd.workloads import build_workload
from nersemble_amd.rays import RayBundle
torch.manual_seed(0)
trainer, data, info = build_workload("p030_h32", device="cuda:0")
for s in range(20):
    trainer.train_iteration(s, *data.next_train(s))
model = trainer.model
model.eval()
bundle, batch, (h, w) = data.eval_image_rays(cam=2, timestep=5, downscale=4)
n = bundle.origins.shape[0]
chunk = 32768
side = [torch.cuda.Stream(), torch.cuda.Stream()]

def piece(i):
    sl = slice(i, min(i + chunk, n))
    return RayBundle(origins=bundle.origins[sl], directions=bundle.directions[sl], pixel_area=bundle.pixel_area[sl],
                     camera_indices=bundle.camera_indices[sl], times=bundle.times[sl])

def render(n_streams):
    outs = []
    main = torch.cuda.current_stream()
    with torch.no_grad():
        for k, i in enumerate(range(0, n, chunk)):
            if n_streams == 1 or k == 0:
                outs.append(model(piece(i))["rgb"])
                if n_streams > 1:
                    for s in side[:n_streams]:
                        s.wait_stream(main)
            else:
                s = side[(k - 1) % n_streams]
                with torch.cuda.stream(s):
                    o = model(piece(i))["rgb"]
                o.record_stream(main)
                outs.append(o)
        if n_streams > 1:
            for s in side[:n_streams]:
                main.wait_stream(s)
    return torch.cat(outs)

ref = None
for ns in (1, 2, 1, 2, 1, 2):
    render(ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); img = render(ns); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if ref is None: ref = img
    print(f"streams={ns}: {dt*1e3:.2f} ms, equal to sequential: {bool(torch.equal(img, ref))}, max diff {float((img-ref).abs().max()):.2e}")
