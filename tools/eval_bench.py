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
    # what bounds the dominant kernel: not HBM bytes -- 572 algorithmic bytes per sample at the measured rate
    # are ~1.8 TB/s -- and not the matrix cores (8 MFMAs per 32 samples).  The counter passes of this very command
    # (profiles/pmc/r06_density_fused_counters.json: L2 hit rate 0.41, 13.6 L2 misses per sample, waves parked on s_waitcnt 66 % of
    # their cycles) read like a latency bound; five experiments (profiles/r06_eval_lookup_bound_experiments.txt)
    # say otherwise: more waves per SIMD, merged 8-byte reads, a level-major order that keeps each level in L2 and the pair's
    # lanes side by side all leave the time where it is, and ONE level costs 17-23 us per 2^20 samples whether its table is
    # 16 KB or 2 MB.  The cost is per dword gathered: 128 per sample at ~0.6 per clock and CU.
    dens = dict(rows).get("nsx_density_fused_fwd")
    roofline = None
    if dens is not None:
        pmc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc",
                           "r06_density_fused_counters.json")
        miss_per_sample = l2_hit = wait = None
        if os.path.exists(pmc):
            c = json.load(open(pmc))
            l2_hit, wait = c["derived"].get("l2_hit_rate"), c["derived"].get("SQ_WAIT_ANY_frac_of_wave_cycles")
            miss_per_sample = c["counters"]["TCC_MISS_sum"]["sum_last_render"] / 25881293.0
        roofline = {"kernel": "nsx_density_fused_fwd", "bound": "the CU's gather rate: 128 divergent dword reads per sample at ~0.6 per clock and CU (DESIGN.md 7b)",
                    "ms_per_2^20_samples": round(dens["total_ms"] / samples * 2 ** 20, 4),
                    "achieved": round(samples * 64 / (dens["total_ms"] * 1e-3) / 1e9, 1), "unit": "G line requests/s (16 levels x 4 "
                    "(y, z) corner pairs per sample)",
                    "dword_reads_per_clock_per_cu": round(samples * 128 / (dens["total_ms"] * 1e-3) / 256 / 2.4e9, 3),
                    "peak": None, "frac": None,
                    "hbm_view": {"algorithmic_bytes_per_sample": 572, "achieved_GBps": round(samples * 572 / (dens["total_ms"] * 1e-3) / 1e9, 1),
                                 "frac_of_8_TBps": round(samples * 572 / (dens["total_ms"] * 1e-3) / 8e12, 3)},
                    "counters": {"l2_hit_rate": l2_hit, "l2_misses_per_sample": miss_per_sample,
                                 "wave_cycles_waiting_on_memory": wait, "source": "profiles/pmc/r06_density_fused_counters.json"}}
    print(json.dumps({"image": f"{h}x{w}", "rays": n, "samples": samples, "wall_ms_with_events": round(dt * 1e3, 2),
                      "native_calls_ms": round(native, 2), "roofline": roofline,
                      "per_entry_point": {k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 3),
                                              "avg_ms": round(v["avg_ms"], 4)} for k, v in rows}}))
