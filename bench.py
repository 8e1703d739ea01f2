"""bench.py -- ray-samples/sec of the NeRSemble training hot path on MI355X (BASELINE.json metric).

One "step" = one full training iteration over one synthetic batch of 4096 rays: occupancy callback (amortised,
every 16 steps) -> ray marching -> sigma_fn density pass (deformation + HashEnsemble + mlp_base, no grad) ->
deformation -> HashEnsemble -> mlp_base -> mlp_head -> weights / compositing -> losses (incl. distortion loss)
-> backward through all of it -> GradScaler + Adam on every parameter group.  Nothing is skipped or cached.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `value` = ray samples processed by all ranks / max-over-ranks wall time.
`roofline` is measured live with HIP events on the kernels' stream over the timed region for the dominant
kernel; `cpu_baseline` times the CPU oracle ("port") of the fused HashEnsemble forward on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)


def algorithmic_bytes(name: str, units: int, H: int) -> float:
    """per-sample algorithmic bytes of the hash kernels."""
    if name == "nsx_hash_ensemble_fwd":
        return units * (512.0 * H + 80.0)
    if name in ("nsx_hash_ensemble_bwd_factored", "nsx_hash_ensemble_bwd"):
        return units * (1024.0 * H + 76.0)
    return 0.0


def cpu_baseline(H: int, seconds_budget: float = 15.0):
    """CPU oracle (C restatement, OpenMP on all host cores) of the fused HashEnsemble forward on a bounded
    sample of the same workload: reference geometry, H grids, uniformly random positions."""
    import numpy as np
    import oracle
    from oracle import hashgrid as ohg
    g = oracle.grid_geometry()
    rng = np.random.default_rng(0)
    f_enc, p, c = ohg.ens_layout(H)
    tabs = rng.integers(0, 2 ** 16, size=(c, g.total_entries, f_enc), dtype=np.uint16) & np.uint16(0x3BFF)
    B = 1 << 14
    x = rng.random((B, 3), dtype=np.float32)
    code = rng.standard_normal((B, H)).astype(np.float32)
    ohg.ensemble_fwd(x[:256], tabs, H, g, code[:256])           # warm up / page in
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds_budget:
        ohg.ensemble_fwd(x, tabs, H, g, code)
        n += B
    dt = time.time() - t0
    return {"value": n / dt, "unit": "ray-samples/s (HashEnsemble forward only)", "cores": os.cpu_count(),
            "kind": "port", "sample": f"{n} samples of the H={H} fused HashEnsemble forward (oracle/nsx_oracle.c, "
                                      f"OpenMP), reference geometry 16 levels x 2^19, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="p030_h32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from nersemble_amd import _lib
    from nersemble_amd.workloads import build_workload, WORKLOADS
    torch.manual_seed(19980801)            # identical initial weights on every rank
    trainer, data, info = build_workload(a.workload, device=dev, rank=rank, world_size=world)
    H = WORKLOADS[a.workload]["H"]

    # synthetic inputs are generated up front: they are resident in HBM when the timed region starts
    batches = [data.next_train(s) for s in range(a.warmup + a.steps)]
    torch.cuda.synchronize()

    def run(n_steps, first_step):
        samples = 0
        for s in range(first_step, first_step + n_steps):
            bundle, batch = batches[s]
            loss, loss_dict, metrics = trainer.train_iteration(s, bundle, batch)
            samples += metrics["num_samples_per_batch"]
        return samples, loss, metrics

    run(a.warmup, 0)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.profiler.reset()
    _lib.profiler.enabled = True
    t0 = time.perf_counter()
    samples, loss, metrics = run(a.steps, a.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profiler.enabled = False
    samples = int(samples)

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    n = torch.tensor([samples], device=dev, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    dt_max, total_samples = float(t.item()), int(n.item())

    if rank == 0:
        prof = _lib.profiler.summary()
        hash_kernels = {k: v for k, v in prof.items() if algorithmic_bytes(k, 1, H) > 0}
        dom_name = max(hash_kernels, key=lambda k: hash_kernels[k]["total_ms"]) if hash_kernels else None
        roofline = None
        if dom_name:
            d = hash_kernels[dom_name]
            per_launch_bytes = algorithmic_bytes(dom_name, d["units"], H) / d["calls"]
            achieved = per_launch_bytes / (d["avg_ms"] * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                        "avg_launch_ms": round(d["avg_ms"], 4), "launches": d["calls"],
                        "algorithmic_bytes_per_launch": per_launch_bytes}
        kernels = {k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 3), "avg_ms": round(v["avg_ms"], 4)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        out = {
            "metric": "ray-samples/sec training, 4096 rays x 2^20 samples", "value": total_samples / dt_max,
            "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": a.workload, "participant": info["participant"], "n_hash_encodings": H,
                       "rays_per_gpu": info["rays"], "max_n_samples_per_batch": "2^20",
                       "samples_per_step_per_gpu": samples / a.steps, "n_timesteps": info["n_timesteps"],
                       "parallelism": f"dp{world}", "params": info["params"]},
            "rays_per_sec": world * info["rays"] * a.steps / dt_max,
            "psnr_last": float(metrics["psnr"].detach()), "loss_last": float(loss.detach()),
            "roofline": roofline, "native_kernel_ms": kernels,
            "native_ms_per_step": sum(v["total_ms"] for v in prof.values()) / a.steps,
        }
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
