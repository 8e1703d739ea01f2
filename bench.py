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


MFMA_PEAK_TFLOPS = 2500.0       # dense fp16 MFMA peak (MI355X_MICROARCH.md)
DEFORM_FWD_FLOPS = 253952.0     # 2 * 126 976 MAC per sample


def kernel_model(name: str, ints, H: int, total_entries: int):
    """(bound, work per launch) for the kernels with a stated algorithmic cost.
    `ints` = the integer arguments of the C-ABI call as recorded by the profiler."""
    if name == "nsx_hash_ensemble_fwd":                       # (B, H, code_stride)
        return "hbm", ints[0] * (512.0 * H + 80.0)
    if name == "nsx_hash_ensemble_bwd_factored":              # (B, H, code_stride, n_slots)
        # table gather for dL/dcode and dL/dx (512 H) + read-modify-write of G (128 corners x 2 floats x 2) + fp16
        # dout (64) + dcode (4 H) + x, dx, slot (28): the factored gradient moves FEWER bytes than SURVEY's dense
        # count 1024 H + 76 -- the kernel is priced against what it has to move
        return "hbm", ints[0] * (512.0 * H + 2048.0 + 64.0 + 4.0 * H + 28.0)
    if name == "nsx_hash_ensemble_bwd":
        return "hbm", ints[0] * (1024.0 * H + 76.0)
    if name == "nsx_adam_hash_factored":                      # (n_slots, code_stride, H, step)
        Hp = 1
        while Hp < H:
            Hp *= 2
        params = total_entries * 2.0 * Hp
        # master, m, v read + written (24 B) + fp16 copy written (2 B) per parameter + G read once
        return "hbm", params * 26.0 + ints[0] * total_entries * 8.0
    if name == "nsx_adam_dense_f16grad":                      # (n, step): the rank's shard in data-parallel runs
        return "hbm", ints[0] * 28.0                          # fp16 gradient + master / m / v read + written + fp16 copy
    if name == "nsx_deform_fwd":                              # (S, code_stride)
        return "mfma", ints[0] * DEFORM_FWD_FLOPS
    if name == "nsx_deform_bwd":                              # recompute fwd + dX chain + weight gradients = 3x fwd
        return "mfma", ints[0] * DEFORM_FWD_FLOPS * 3.0
    return None, 0.0


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
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                                                      "multi-rank control flow where RCCL cannot run)")
    ap.add_argument("--ranks-share-gpu0", action="store_true",
                    help="testing aid: every rank uses cuda:0 (several ranks on a 1-GPU box, with --backend gloo)")
    ap.add_argument("--reserve-gb", type=float, default=24.0,
                    help="allocator warm-up: device memory handed to torch's caching allocator before the first step")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not record HIP events around the native calls (no roofline block; measures their overhead)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.ranks_share_gpu0:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=a.backend)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = f"cuda:{local_rank}"
    torch.cuda.set_device(local_rank)

    from nersemble_amd import _lib
    from nersemble_amd.workloads import build_workload, WORKLOADS
    # Allocator warm-up.  Every sample-sized tensor (3.7 KB of scratch per sample for the deformation backward alone)
    # is re-requested each step with a different size; whenever the marcher reaches a new maximum the caching allocator
    # has to go to hipMalloc, which on a fresh box costs ~100 ms per step it happens in.  One block allocated and
    # released here stays in the allocator's cache and is carved up instead (288 GB of HBM: the reserve is free).
    # (After the model is built: the trainer picks the physical placement of the table-optimizer streams from fresh
    # allocations and returns the losers with empty_cache(), engine/placement.py -- arrays carved out of one big
    # cached block are always at the slow end of the placement spread.)
    torch.manual_seed(19980801)            # identical initial weights on every rank
    trainer, data, info = build_workload(a.workload, device=dev, rank=rank, world_size=world)
    if a.reserve_gb > 0:
        reserve = torch.empty(int(a.reserve_gb * 2 ** 30), dtype=torch.uint8, device=dev)
        del reserve
    H = WORKLOADS[a.workload]["H"]

    # synthetic inputs are generated up front: they are resident in HBM when the timed region starts
    batches = [data.next_train(s) for s in range(a.warmup + a.steps)]
    torch.cuda.synchronize()

    step_marks = []                          # (event at step start, device-side sample count) per timed step

    def run(n_steps, first_step, mark=False):
        samples = 0
        for s in range(first_step, first_step + n_steps):
            if mark:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
            bundle, batch = batches[s]
            loss, loss_dict, metrics = trainer.train_iteration(s, bundle, batch)
            samples += metrics["num_samples_per_batch"]
            if mark:
                step_marks.append((ev, metrics["num_samples_per_batch"]))
        return samples, loss, metrics

    run(a.warmup, 0)
    # Python's cyclic garbage collector pauses the host for tens of ms whenever its generation-2 threshold trips
    # (sporadic 70-100 ms steps); training loops collect at controlled points instead (NeRSembleTrainer.gc_every)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    # HIP events only around the calls that are priced against a roofline (+ the other large kernels), allocated
    # before the timed region
    _lib.profiler.watch = {"nsx_hash_ensemble_fwd", "nsx_hash_ensemble_bwd_factored", "nsx_hash_ensemble_bwd",
                           "nsx_adam_hash_factored", "nsx_adam_dense", "nsx_deform_fwd", "nsx_deform_bwd",
                           "nsx_mlp_fwd", "nsx_mlp_bwd", "nsx_check_finite", "nsx_march_count", "nsx_march_fill",
                           "nsx_hash_grad_expand", "nsx_hash_grad_expand_f16", "nsx_adam_dense_f16grad",
                           "nsx_check_finite_f16"}
    if not a.no_kernel_events:
        _lib.profiler.prewarm(2 * 16 * a.steps + 64)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.profiler.reset()
    _lib.profiler.enabled = not a.no_kernel_events
    t0 = time.perf_counter()
    samples, loss, metrics = run(a.steps, a.warmup, mark=True)
    end_mark = torch.cuda.Event(enable_timing=True)
    end_mark.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trainer.flush_scheduler_step()
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
        total_entries = trainer.model.field.hash_ensemble.geom.total_entries
        work = {}
        for name, st, en, ints in _lib.profiler.records:
            bound, w = kernel_model(name, ints, H, total_entries)
            if bound:
                d = work.setdefault(name, {"bound": bound, "work": 0.0})
                d["work"] += w
        rooflines = {}
        for name, d in work.items():
            p = prof[name]
            per_launch = d["work"] / p["calls"]
            if d["bound"] == "hbm":
                ach = per_launch / (p["avg_ms"] * 1e-3) / 1e9
                rooflines[name] = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None,
                                   "avg_launch_ms": round(p["avg_ms"], 4), "launches": p["calls"],
                                   "algorithmic_bytes_per_launch": per_launch}
            else:
                ach = per_launch / (p["avg_ms"] * 1e-3) / 1e12
                rooflines[name] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                                   "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                                   "avg_launch_ms": round(p["avg_ms"], 4), "launches": p["calls"],
                                   "algorithmic_flops_per_launch": per_launch}
        # the dominant kernel = the modelled kernel with the largest total time in the timed region
        dom_name = max(rooflines, key=lambda k: prof[k]["total_ms"]) if rooflines else None
        roofline = rooflines.get(dom_name)
        pmc_path = os.path.join(ROOT, "profiles", "pmc", "latest.json")
        # the committed counter passes were taken on the default workload at one rank: their per-launch bytes say
        # nothing about another table size / sample count, so any other run reports traffic = null
        if roofline and os.path.exists(pmc_path) and a.workload == "p030_h32" and world == 1:
            try:
                pmc = json.load(open(pmc_path)).get("per_launch_hbm_bytes", {})
                if dom_name in pmc:
                    roofline["traffic"] = pmc[dom_name]       # from the committed rocprofv3 --pmc pass of this command
            except Exception:
                pass
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
            "roofline": roofline, "rooflines": rooflines, "native_kernel_ms": kernels,
            "native_ms_per_step": sum(v["total_ms"] for v in prof.values()) / a.steps,
            # device-timeline duration and marched samples of every timed step (the occupancy grid refines over the
            # first steps, so the sample count -- and with it the step time -- falls during the run)
            "per_step": [{"ms": round(step_marks[i][0].elapsed_time(
                              step_marks[i + 1][0] if i + 1 < len(step_marks) else end_mark), 3),
                          "samples": int(step_marks[i][1])} for i in range(len(step_marks))],
        }
        if trainer.placement_report is not None:
            out["table_placement"] = trainer.placement_report       # one-off, before the warm-up (engine/placement.py)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
