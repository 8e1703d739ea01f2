"""bench.py -- ray-samples/sec of the NeRSemble training hot path on MI355X (BASELINE.json metric).

One "step" = one full training iteration over one synthetic batch of 4096 rays: occupancy callback (amortised,
every 16 steps) -> ray marching -> sigma_fn density pass (deformation + HashEnsemble + mlp_base, no grad) ->
deformation -> HashEnsemble -> mlp_base -> mlp_head -> weights / compositing -> losses (incl. distortion loss)
-> backward through all of it -> GradScaler + Adam on every parameter group.  Nothing is skipped or cached.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line on rank 0.  `value` = ray samples processed by all ranks / max-over-ranks wall time.
`roofline` is measured live with HIP events on the kernels' stream over the timed region for the dominant
kernel; `kernels_alone` times the priced kernels one at a time on 2^20 uniformly random samples right after the timed
region (no co-running stream, no cache-friendly ray order); `cpu_baseline` times the PyTorch-CPU restatement of the
encoder path (oracle/torch_cpu.py) on the host cores.

Which steps are timed matters: the samples a 4096-ray batch keeps fall from ~2^20 at the start of training to ~10^5
once the model has learnt where the volume is empty (the occupancy grid prunes the marcher, the sigma_fn visibility test
prunes what it marched), and with them the step time, while the table optimizer's 12 GB pass per step stays.  The timed
region is the W warm-up + K steps the caller asks for, from a fresh model (as in round 1).  Two stationary readings are
added to the same line: `steady_state` -- training continues to step `--steady-after` (600) and 100 more steps are
timed there -- and, on request, `--preroll N` moves the whole timed region behind N untimed steps.  `--grid frozen`
keeps the occupancy grid of the end of the warm-up (its update still runs with all its work, the result is not adopted);
measured, that alone does not hold the sample count (the visibility pruning follows the model, not the grid).
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


SPLIT_SCATTER = True            # set in main() from the trainer's gradient sink: the factored backward runs as two kernels


MLP_BASE_FLOPS = 6_144.0
MLP_HEAD_FLOPS = 14_336.0


def kernel_model(name: str, ints, H: int, total_entries: int):
    """(bound, work per launch) for the kernels with a stated algorithmic cost.
    `ints` = the integer arguments of the C-ABI call as recorded by the profiler."""
    if name == "nsx_hash_ensemble_bwd_codesum":               # the factored backward + in-kernel code-gradient sums
        name = "nsx_hash_ensemble_bwd_factored"               # (same byte model; the [B, H] dcode tensor it is priced
        #                                                        with no longer exists -- the model stays the harder one)
    if name in ("nsx_hash_ensemble_fwd", "nsx_hash_ensemble_bwd_factored"):
        H = ints[1] if len(ints) > 1 else H                   # (the compact first-grid phase calls these with H = 1)
    if name == "nsx_hash_ensemble_fwd":                       # (B, H, code_stride)
        return "hbm", ints[0] * (512.0 * H + 80.0)
    if name == "nsx_hash_ensemble_bwd_factored":              # (B, H, code_stride, n_slots)
        # table gather for dL/dcode and dL/dx (512 H) + read-modify-write of G (128 corners x 2 floats x 2) + fp16
        # dout (64) + dcode (4 H) + x, dx, slot (28): the factored gradient moves FEWER bytes than SURVEY's dense
        # count 1024 H + 76 -- the kernel is priced against what it has to move
        if SPLIT_SCATTER:                                     # gather half only: table reads + dout + dcode + x, dx, slot
            return "hbm", ints[0] * (512.0 * H + 64.0 + 4.0 * H + 28.0)
        return "hbm", ints[0] * (512.0 * H + 2048.0 + 64.0 + 4.0 * H + 28.0)
    if name == "nsx_hash_ensemble_bwd_scatter":               # (B, n_slots, blocks_per_cu): read-modify-write of G + dout
        # + x + slot.  Priced against HBM for uniformity; what bounds it is the rate of memory-side fp32 atomics
        # (measured ~14-20 G 32-B sectors/s; 64 sector requests per sample without duplicate merging)
        return "hbm", ints[0] * (2048.0 + 128.0 + 12.0 + 4.0)
    if name == "nsx_hash_ensemble_bwd":
        return "hbm", ints[0] * (1024.0 * H + 76.0)
    if name == "nsx_adam_hash_factored":                      # (n_slots, code_stride, H, step)
        H = ints[2] if len(ints) > 2 else H
        Hp = 1
        while Hp < H:
            Hp *= 2
        params = total_entries * 2.0 * Hp
        # master, m, v read + written (24 B) + fp16 copy written (2 B) per parameter + G read once
        return "hbm", params * 26.0 + ints[0] * total_entries * 8.0
    if name == "nsx_adam_dense_f16grad":                      # (n, step): the rank's shard in data-parallel runs
        return "hbm", ints[0] * 28.0                          # fp16 gradient + master / m / v read + written + fp16 copy
    if name in ("nsx_deform_fwd", "nsx_deform_fwd_rows"):     # (S, code_stride[, n_rows])
        # (the table-indexed forward spends 192 of the 256 MFMAs: priced in the FLOPs of the reference's eight Linear
        # layers, the work it replaces)
        return "mfma", ints[0] * DEFORM_FWD_FLOPS
    if name == "nsx_deform_bwd":                              # recompute fwd + dX chain + weight gradients = 3x fwd
        return "mfma", ints[0] * DEFORM_FWD_FLOPS * 3.0
    if name in ("nsx_mlp_fwd", "nsx_mlp_bwd") and len(ints) > 1:     # (n_hidden_mats, B, ...): SURVEY 8(d): mlp_base
        # 2 (32 64 + 64 16) = 6 144 FLOP per sample, mlp_head (padded) 2 (32 64 + 64 64 + 64 16) = 14 336; backward =
        # recompute + dX + dW = 3x.  Memory / latency bound (<= 16 MFMAs per 32 samples): priced so that nobody has to guess
        flops = MLP_HEAD_FLOPS if ints[0] == 1 else MLP_BASE_FLOPS
        return "mfma", ints[1] * flops * (3.0 if name == "nsx_mlp_bwd" else 1.0)
    return None, 0.0


def cpu_baseline(H: int, seconds_budget: float = 30.0):
    """The encoder path on the host cores, TWO restatements, each bounded in time and each reported under its own name:
      * ``value`` / ``unit`` / ``cores``: the C port of the fused HashEnsemble forward (oracle/nsx_oracle.c::
        nsxo_ensemble_fwd_fast: the oracle's forward with fp32 accumulation and table-driven fp16 decode, OpenMP over ALL
        cores; held to the checker in tests/test_oracle_hash.py) -- the faster of the two on every box so far, and the one
        whose unit (HashEnsemble forward alone) is what `value` means;
      * ``torch_cpu``: SURVEY.md 8(d)'s PyTorch-CPU encoder (oracle/torch_cpu.py: HashEnsemble forward + mlp_base as gathers
        + einsum, held to the C oracle in tests/test_oracle_hash.py) swept over S = 2^16, 2^18, 2^20 uniformly random samples
        at the reference geometry as far as the budget allows, at the intra-op thread count a calibration picks (torch's
        CPU gathers do not scale to hundreds of threads), AND once on all cores.
    The reference has no CPU encoder of its own (tinycudann is CUDA-only): kind = "port".  A baseline, not a target."""
    import numpy as np
    import oracle
    from oracle import hashgrid as ohg, torch_cpu
    cores = os.cpu_count() or 1
    g = oracle.grid_geometry()
    sweep, threads = torch_cpu.time_encoder_sweep(H, g, budget_s=0.6 * seconds_budget, max_threads=cores)
    best = max(sweep, key=lambda r: r["samples_per_s"])
    all_cores = getattr(torch_cpu.time_encoder_sweep, "all_cores", None)
    rng = np.random.default_rng(0)
    f_enc, p, c = ohg.ens_layout(H)
    tabs = rng.integers(0, 2 ** 16, size=(c, g.total_entries, f_enc), dtype=np.uint16) & np.uint16(0x3BFF)
    B = 1 << 16
    x = rng.random((B, 3), dtype=np.float32)
    code = rng.standard_normal((B, H)).astype(np.float32)
    # (the baseline port of the oracle's forward: fp32 accumulation, table-driven fp16 decode -- oracle/nsx_oracle.c;
    # the double-precision checker itself is ~10x slower and is not what a CPU implementation would look like)
    ohg.ensemble_fwd_fast(x[:256], tabs, H, g, code[:256])      # warm up / page in
    t0, n = time.time(), 0
    while time.time() - t0 < 0.3 * seconds_budget:
        ohg.ensemble_fwd_fast(x, tabs, H, g, code)
        n += B
    c_rate = n / (time.time() - t0)
    return {"value": c_rate, "unit": "ray-samples/s (HashEnsemble forward on the host: C port of the oracle, OpenMP)",
            "cores": cores, "kind": "port",
            "sample": f"H={H}, 16 levels x 2^19, {n} uniformly random samples in batches of {B}, fp32 accumulation, "
                      f"table-driven fp16 decode, {cores} OpenMP threads",
            "torch_cpu": {"value": best["samples_per_s"],
                          "unit": "ray-samples/s (HashEnsemble forward + mlp_base, PyTorch on the host: oracle/torch_cpu.py)",
                          "threads": threads, "cores_available": cores, "sweep": sweep, "all_cores": all_cores,
                          "note": "`threads` = the intra-op thread count a 4096-sample calibration picked; `all_cores` = the "
                                  "same encoder on every core (about 3 s of samples)"}}


def compute_rooflines(prof, records, tags, kept, H: int, total_entries: int, side_stream: bool, pmc_state=None):
    """(roofline of the dominant priced kernel, all rooflines) from the profiler's records of a timed region.
    ``kept``: per step (the records' tags) the rows a launch under a device-side sample count PROCESSED -- such a call is
    made with the marched capacity and the kernel reads the kept count when it runs: the step's num_samples_per_batch,
    known to the host after the region.  ``pmc_state``: which committed counter pass belongs to this state of the run
    ("headline": profiles/pmc/latest.json; "steady_full" / "steady_open_window" / "steady_compact":
    profiles/pmc/r05_<state>.json) -- counters need passes of their own, `traffic` is theirs, never this run's."""
    work = {}
    for (name, st, en, ints), (tag, counted) in zip(records, tags):
        if counted and tag is not None and 0 <= tag < len(kept):
            ints = list(ints)
            ints[1 if name.startswith("nsx_mlp_") else 0] = kept[tag]      # (the MLP calls lead with n_hidden_mats)
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
            if ach > HBM_PEAK_GBPS:
                # ray-ordered samples: the coarse levels are served from L2 / Infinity Cache, so the algorithmic
                # byte count is not HBM traffic here -- not a roofline fraction (see kernels_alone for one)
                rooflines[name]["frac"] = None
                rooflines[name]["note"] = ("algorithmic bytes per second exceed the HBM peak: cache hits on "
                                           "ray-coherent samples; the HBM-roofline fraction of this kernel is "
                                           "kernels_alone's (uniform samples)")
        else:
            ach = per_launch / (p["avg_ms"] * 1e-3) / 1e12
            rooflines[name] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                               "avg_launch_ms": round(p["avg_ms"], 4), "launches": p["calls"],
                               "algorithmic_flops_per_launch": per_launch}
            if name in ("nsx_deform_fwd", "nsx_deform_fwd_rows") and side_stream:
                # most of its launches (the sigma_fn pass of the NEXT step) run beside the table optimizer's 12 GB
                # pass on the other stream, 3-5x slower than alone and off the critical path: a duration measured
                # while co-running says nothing about the kernel
                rooflines[name]["note"] = ("launched beside the table optimizer's stream (off the critical path): "
                                           "this duration is a co-scheduling figure; the kernel's own fraction is "
                                           "kernels_alone's")
    # the dominant kernel = the modelled kernel with the largest total time in the timed region; kernels whose totals
    # are within 5 % of it are named beside it (in the benchmark window the hash backward is as large as the optimizer)
    priced = [k for k in rooflines if rooflines[k]["frac"] is not None]
    dom_name = max(priced, key=lambda k: prof[k]["total_ms"]) if priced else None
    roofline = dict(rooflines[dom_name]) if dom_name else None
    if roofline:
        top = prof[dom_name]["total_ms"]
        roofline["total_ms_in_region"] = round(top, 3)
        roofline["co_dominant"] = [
            {"kernel": k, "total_ms_in_region": round(prof[k]["total_ms"], 3), "frac": rooflines[k]["frac"],
             "avg_launch_ms": rooflines[k]["avg_launch_ms"], "bound": rooflines[k]["bound"]}
            for k in sorted(rooflines, key=lambda k: -prof[k]["total_ms"])
            if k != dom_name and prof[k]["total_ms"] >= 0.95 * top]
    pmc_path = {"headline": os.path.join(ROOT, "profiles", "pmc", "latest.json")}.get(
        pmc_state, os.path.join(ROOT, "profiles", "pmc", f"r05_{pmc_state}.json") if pmc_state else None)
    if roofline and pmc_path and os.path.exists(pmc_path):
        try:
            pmc_doc = json.load(open(pmc_path))
            pmc = pmc_doc.get("per_launch_hbm_bytes", {})
            for r in [roofline] + roofline["co_dominant"]:
                if r["kernel"] in pmc:
                    # NOT a counter of this run: the committed rocprofv3 --pmc passes of this command in this state
                    # (counters need their own passes; gpurun refuses --pmc together with tracing)
                    r["traffic"] = pmc[r["kernel"]]
                    r["traffic_source"] = os.path.relpath(pmc_path, ROOT) + " <- " + str(pmc_doc.get("source", "?"))
        except Exception:
            pass
    return roofline, rooflines


def steady_state(trainer, data, first_step: int, settle_at: int, rays: int, n_timed: int = 100, datamanager=None,
                 H: int = 0, pmc_state=None, n_profiled: int = 20):
    """Continues the run to step `settle_at` (untimed), then times `n_timed` steps: by then the occupancy grid and the
    visibility pruning have settled and every step sees about the same number of samples.  `datamanager`: draw every
    batch inside the timed loop through its `next_train` instead of pre-generating them.
    `H` > 0: `n_profiled` MORE steps follow with HIP events around every priced native call (not inside the timed steps:
    the events cost ~0.08 ms per step) -- the regime's own `roofline` / `rooflines`, with `traffic` from the committed
    counter pass of this state (`pmc_state`)."""
    import gc
    src = datamanager if datamanager is not None else data
    step = first_step
    while step < settle_at:
        trainer.train_iteration(step, *src.next_train(step))
        step += 1
    batches = [data.next_train(step + i) for i in range(n_timed + 1)] if datamanager is None else None
    nxt = datamanager.next_train(step) if datamanager is not None else None
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    lp_opt = trainer.optimizers.get(trainer.group_of_tables())
    lp_opt = lp_opt if isinstance(lp_opt, LevelParallelTableAdam) else None
    if lp_opt is not None:
        lp_opt.comm_report()
        lp_opt.timing = lp_opt.lp.timing = True
    gc.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    samples, counts = 0, []
    for i in range(n_timed):
        if datamanager is None:
            cur, ahead = batches[i], batches[i + 1]
        else:
            cur = nxt
            ahead = nxt = datamanager.next_train(step + i + 1)
        _, _, metrics = trainer.train_iteration(step + i, *cur, next_ray_bundle=ahead[0])
        counts.append(metrics["num_samples_per_batch"])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trainer.flush_scheduler_step()
    counts = [int(c) for c in counts]
    samples = sum(counts)
    out = {"from_step": step, "steps": n_timed, "ms_per_step": dt / n_timed * 1e3, "value": samples / dt,
           "unit": "ray-samples/s", "rays_per_sec": rays * n_timed / dt, "samples_per_step_min_max": [min(counts), max(counts)],
           "psnr": float(metrics["psnr"].detach())}
    if lp_opt is not None:
        lp_opt.timing = lp_opt.lp.timing = False
        out["_comm"] = lp_opt.comm_report()
        # host issue time: the same steps issued into an EMPTY queue (a device synchronisation in front of every step), the
        # median of the host's time inside train_iteration -- what the host needs per step when it never waits for the device
        host = []
        more = [data.next_train(step + n_timed + i) for i in range(65)]
        for i in range(64):
            torch.cuda.synchronize()
            th = time.perf_counter()
            trainer.train_iteration(step + n_timed + i, *more[i], next_ray_bundle=more[i + 1][0])
            if (step + n_timed + i) % 16 != 0:
                host.append(time.perf_counter() - th)
        torch.cuda.synchronize()
        trainer.flush_scheduler_step()
        host.sort()
        out["host_issue_ms_per_step"] = host[len(host) // 2] * 1e3
        out["host_issue_ms_per_step_min"] = host[0] * 1e3
        step += 64
    if H > 0 and datamanager is None and n_profiled > 0:
        from nersemble_amd import _lib
        prof = _lib.profiler
        more = [data.next_train(step + n_timed + i) for i in range(n_profiled + 1)]
        prof.prewarm(2 * 16 * n_profiled + 64)
        torch.cuda.synchronize()
        prof.reset()
        prof.enabled = True
        kept = []
        for i in range(n_profiled):
            prof.tag = i
            _, _, m = trainer.train_iteration(step + n_timed + i, *more[i], next_ray_bundle=more[i + 1][0])
            kept.append(m["num_samples_per_batch"])
        torch.cuda.synchronize()
        prof.enabled, prof.tag = False, None
        trainer.flush_scheduler_step()
        prof.collect_native()
        kept = [int(c) for c in kept]
        total_entries = trainer.model.field.hash_ensemble.geom.total_entries
        summary = prof.summary()
        roofline, rooflines = compute_rooflines(summary, prof.records, prof.tags, kept, H, total_entries,
                                                trainer._opt_stream is not None, pmc_state)
        out["roofline"] = roofline
        out["rooflines"] = {k: {"frac": v["frac"], "avg_launch_ms": v["avg_launch_ms"], "bound": v["bound"],
                                "total_ms": round(summary[k]["total_ms"], 3)} for k, v in rooflines.items()}
        out["profiled_steps"] = n_profiled
        prof.reset()
    return out


def steady_state_ranks(trainer, data, first_step: int, settle_at: int, rays: int, world: int, dev, backend: str,
                       n_timed: int = 100):
    """``steady_state`` for N > 1 ranks (every rank calls it: the steps are collective): continues to step `settle_at`, then
    times `n_timed` steps between barriers, max over ranks; with the level-parallel exchange the `comm` block of exactly
    these steps (bytes arriving per rank and step, the job's samples per step) -- the regime a run lives in."""
    import gc
    import torch.distributed as dist
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    step = first_step
    while step < settle_at:
        trainer.train_iteration(step, *data.next_train(step))
        step += 1
    batches = [data.next_train(step + i) for i in range(n_timed + 1)]
    table_opt = trainer.optimizers.get(trainer.group_of_tables())
    lp = isinstance(table_opt, LevelParallelTableAdam)
    if lp:
        table_opt.comm_report()
        table_opt.timing = True
    gc.collect()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    counts = []
    for i in range(n_timed):
        _, _, metrics = trainer.train_iteration(step + i, *batches[i], next_ray_bundle=batches[i + 1][0])
        counts.append(metrics["num_samples_per_batch"])
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    trainer.flush_scheduler_step()
    counts = [int(c) for c in counts]
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    n = torch.tensor([sum(counts)], device=dev, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    out = {"from_step": step, "steps": n_timed, "ms_per_step": float(t.item()) / n_timed * 1e3,
           "value": int(n.item()) / float(t.item()), "unit": "ray-samples/s", "rays_per_sec": world * rays * n_timed / float(t.item()),
           "samples_per_step_min_max_rank0": [min(counts), max(counts)], "psnr_rank0": float(metrics["psnr"].detach())}
    if lp:
        table_opt.timing = False
        comm = table_opt.comm_report()
        comm["backend"] = backend
        comm["bar_bytes_per_rank"] = 2 * 64 * (comm["samples_fwd_per_step"] + comm["samples_bwd_per_step"]) / 2 * 1.05
        out["comm"] = comm
    return out


def first_grid_phase_block(a):
    """The same command with `--compact-first-grid` in a process of its own (fresh allocator, own placement calibration):
    what the steps of this benchmark cost in the compact first-grid phase -- the DEFAULT of `NeRSembleTrainer` since round 3
    (this benchmark switches it off for its headline, which prices the full 32-grid layout).  The coarse-to-fine window
    keeps one hash grid on for the first 40 000 steps of the default schedule (train_nersemble.py:77-78), so every step
    this benchmark runs is in that phase; the other 31 grids have zero blend weight, zero gradient and zero Adam moments
    there, and a contiguous copy of grid 0 trained with the H = 1 kernels gives the same results
    (tests/test_training_gpu.py::test_compact_first_grid_phase_is_the_same_training).  Reported BESIDE the headline, which
    prices a step in the full 32-grid layout -- what 87 % of a 300 000-step run costs."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--compact-first-grid", "--no-cpu-baseline", "--no-kernels-alone",
           "--steps", str(a.steps), "--warmup", str(a.warmup), "--workload", a.workload,
           "--steady-after", str(a.steady_after), "--reserve-gb", str(a.reserve_gb)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as exc:                                     # the block is a bonus: never lose the headline over it
        return {"error": repr(exc)[:300]}
    keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "rays_per_sec", "psnr_last",
                                  "samples_per_step_min_max", "steady_state")}
    keep["native_kernel_avg_ms"] = {k: v["avg_ms"] for k, v in (d.get("native_kernel_ms") or {}).items()
                                    if v["avg_ms"] >= 0.05}
    keep["command"] = " ".join(cmd[1:])
    return keep


def open_window_block(a):
    """The same command with the coarse-to-fine window OPEN (`--window-hash 0 1`: every hash grid on from step 1), in a
    process of its own.  The reference keeps `window_hash_encodings == 1` only for steps 0 ... 40 000 of 300 001
    (train_nersemble.py:77-78, hash_ensemble.py:121-123); from step 80 000 on all H grids are blended with the trained
    time codes, the code gradient is needed (ens_bwd<DCODE = true> with the per-row sums in the kernel) and
    `time_embedding` is stepped.  This block prices that 73-87 % of the schedule; the headline stays the default
    schedule's first steps (BASELINE's configuration from a fresh model)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--window-hash", "0", "1", "--no-cpu-baseline", "--no-first-grid-phase",
           "--no-open-window", "--no-kernels-alone", "--steps", str(a.steps), "--warmup", str(a.warmup), "--workload", a.workload,
           "--steady-after", str(a.steady_after), "--reserve-gb", str(a.reserve_gb)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as exc:                                     # the block is a bonus: never lose the headline over it
        return {"error": repr(exc)[:300]}
    keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "rays_per_sec", "psnr_last",
                                  "samples_per_step_min_max", "steady_state", "rooflines")}
    keep["native_kernel_avg_ms"] = {k: v["avg_ms"] for k, v in (d.get("native_kernel_ms") or {}).items()
                                    if v["avg_ms"] >= 0.05}
    keep["command"] = " ".join(cmd[1:])
    return keep


def build_datamanager(data, dev, n_timesteps_cached: int = 20):
    """``NeRSembleVanillaDataManager`` (datamanager/nersemble_datamanager.py:15-118 with the values of
    train_nersemble.py:172-179: 4096 rays from a cache of 24 images, redrawn every 20 iterations) over an in-memory
    dataset of the synthetic rig's 12 training cameras x `n_timesteps_cached` timesteps at full resolution (1100 x 1604;
    7 GB of fp32 images resident on the device -- the dataset's file decoding is out of scope, the cache refresh is a
    24-image stack of resident tensors).  Every image is rendered before anything is timed."""
    from nersemble_amd.data.datamanager import NeRSembleVanillaDataManager, NeRSembleVanillaDataManagerConfig
    T = data.n_timesteps
    stride = max(T // n_timesteps_cached, 1)
    images = [(int(c), t) for t in range(0, T, stride) for c in data.train_cams.tolist()]
    ds = data.image_dataset(images, downscale=1)
    for i in range(len(ds)):
        ds[i]                                                     # (rendered and cached now, not in the timed region)
    gen = torch.Generator().manual_seed(19980801)
    return NeRSembleVanillaDataManager(NeRSembleVanillaDataManagerConfig(train_num_rays_per_batch=data.n_rays), ds,
                                       device=dev, generator=gen), len(images)


def with_datamanager_block(a):
    """The same command with `--with-datamanager` in a process of its own: what the step costs when
    `datamanager.next_train(step)` runs inside it, as in the reference's clocked `train_iteration`
    (nersemble_trainer.py:41-68,169-206 -> nersemble_datamanager.py:76-81)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--with-datamanager", "--no-cpu-baseline", "--no-first-grid-phase",
           "--no-open-window", "--no-kernels-alone", "--no-with-datamanager", "--steps", str(a.steps), "--warmup",
           str(a.warmup), "--workload", a.workload, "--steady-after", str(a.steady_after), "--reserve-gb", str(a.reserve_gb)]
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
    except Exception as exc:                                     # the block is a bonus: never lose the headline over it
        return {"error": repr(exc)[:300]}
    keep = {k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "warmup", "rays_per_sec", "psnr_last",
                                  "samples_per_step_min_max", "steady_state", "datamanager")}
    keep["command"] = " ".join(cmd[1:])
    return keep


def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script through torch.distributed.run on this
    node (127.0.0.1, a free port) and hand their output through.  The driver's own launch line sets WORLD_SIZE and never
    gets here."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def kernels_alone(trainer, H: int, log2_s: int = 20, iters: int = 10):
    """The priced kernels ONE AT A TIME on S = 2^20 uniformly random samples (seed 0) with the run's own tables and
    weights: no optimizer stream beside them, no ray-coherent sample order that would serve the coarse levels from L2 /
    Infinity Cache.  These are the numbers to hold against BASELINE.json's ">= 70 % of the HBM roofline on the 32-grid
    HashEnsemble"; `rooflines` above are the same kernels as they run inside the step."""
    import ctypes as C
    from nersemble_amd import _lib, functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    model = trainer.model
    he = model.field.hash_ensemble
    dev = he.tables.device
    g, S, T = he.geom, 1 << log2_s, 24
    gen = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand((S, 3), device=dev, generator=gen)
    code = torch.randn((T, H), device=dev, generator=gen) * 0.5
    slot = torch.randint(0, T, (S,), device=dev, generator=gen, dtype=torch.int32)
    dout = torch.randn((S, 2 * g.n_levels), device=dev, generator=gen)
    f16 = he.half_tables()
    torch.cuda.synchronize()

    def timeit(fn, n=iters, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n

    out = {}

    def entry(name, ms, bound, work):
        if bound == "hbm":
            ach = work / (ms * 1e-3) / 1e9
            out[name] = {"bound": "hbm", "ms": round(ms, 4), "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "algorithmic_bytes": work,
                         "samples_per_s": S / (ms * 1e-3)}
        else:
            ach = work / (ms * 1e-3) / 1e12
            out[name] = {"bound": "mfma", "ms": round(ms, 4), "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "algorithmic_flops": work,
                         "samples_per_s": S / (ms * 1e-3)}

    entry("nsx_hash_ensemble_fwd", timeit(lambda: F._hash_ensemble_fwd_raw(x, f16, H, g, code, slot, None)), "hbm",
          kernel_model("nsx_hash_ensemble_fwd", [S], H, g.total_entries)[1])
    G = torch.zeros((T, g.total_entries, 2), device=dev)
    dcode = torch.empty((S, H), device=dev)
    dx = torch.empty((S, 3), device=dev)

    def bwd():
        check(lib().nsx_hash_ensemble_bwd_factored(ptr(x), S, ptr(f16), H, C.byref(g), ptr(code), code.stride(0), T,
                                                   ptr(slot), None, ptr(dout), ptr(G), ptr(dcode), ptr(dx), None,
                                                   None,
                                                   stream()), "nsx_hash_ensemble_bwd_factored")
    def gather():
        check(lib().nsx_hash_ensemble_bwd_factored(ptr(x), S, ptr(f16), H, C.byref(g), ptr(code), code.stride(0), T,
                                                   ptr(slot), None, ptr(dout), None, ptr(dcode), ptr(dx), None,
                                                   None,
                                                   stream()), "nsx_hash_ensemble_bwd_factored")

    def scatter():
        check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), S, C.byref(g), T, ptr(slot), ptr(dout), ptr(G), None, 8,
                                                  None,
                                                  stream()), "nsx_hash_ensemble_bwd_scatter")
    rows = torch.empty((T, H), device=dev)
    win = torch.ones((H,), device=dev)

    def bwd_codesum():
        check(lib().nsx_hash_ensemble_bwd_codesum(ptr(x), S, ptr(f16), H, C.byref(g), ptr(code), code.stride(0), T,
                                                  ptr(slot), ptr(win), ptr(dout), ptr(G), ptr(rows),
                                                  ptr(F.codesum_scratch(T, H, dev)), ptr(dx), None, None, stream()),
              "nsx_hash_ensemble_bwd_codesum")

    def bwd_nocode():
        check(lib().nsx_hash_ensemble_bwd_factored(ptr(x), S, ptr(f16), H, C.byref(g), ptr(code), code.stride(0), T,
                                                   ptr(slot), None, ptr(dout), ptr(G), None, ptr(dx), None,
                                                   None,
                                                   stream()), "nsx_hash_ensemble_bwd_factored")
    entry("nsx_hash_ensemble_bwd_factored (fused gather + scatter)", timeit(bwd), "hbm",
          S * (512.0 * H + 2048.0 + 64.0 + 4.0 * H + 28.0))
    # window open: code gradient summed per code row in the kernel (no [S, H] tensor); window closed: no code gradient
    entry("nsx_hash_ensemble_bwd_codesum (fused, code-gradient sums in the kernel)", timeit(bwd_codesum), "hbm",
          S * (512.0 * H + 2048.0 + 64.0 + 28.0))
    entry("nsx_hash_ensemble_bwd_factored (fused, no code gradient)", timeit(bwd_nocode), "hbm",
          S * (512.0 * H + 2048.0 + 64.0 + 28.0))
    entry("nsx_hash_ensemble_bwd_factored (gather half)", timeit(gather), "hbm", S * (512.0 * H + 64.0 + 4.0 * H + 28.0))
    entry("nsx_hash_ensemble_bwd_scatter", timeit(scatter), "hbm", S * (2048.0 + 128.0 + 12.0 + 4.0))
    out["nsx_hash_ensemble_bwd_scatter"]["note"] = ("bound by memory-side fp32 atomics, not by bandwidth: uniform samples "
                                                    "share no cells, 64 sector requests per sample")
    # table Adam on the run's own state, value-preserving (lr 0 keeps master / working copy; the moments only decay)
    opt = trainer.optimizers.get(trainer.group_of_tables())
    from nersemble_amd.engine.hash_adam import HashTableAdam
    if isinstance(opt, HashTableAdam):
        he.wait_tables()
        st = opt._state()
        G.zero_()
        one = torch.ones((1,), device=dev)
        zero = torch.zeros((1,), device=dev)

        def adam():
            check(lib().nsx_adam_hash_factored(ptr(G), T, ptr(code), code.stride(0), None, H, C.byref(g), ptr(he.tables.data),
                                               ptr(st["exp_avg"]), ptr(st["exp_avg_sq"]), ptr(f16), 0.0, 0.9, 0.999, 1e-15,
                                               max(int(st["step"]), 1), ptr(one), ptr(zero), stream()),
                  "nsx_adam_hash_factored")
        entry("nsx_adam_hash_factored", timeit(adam), "hbm",
              kernel_model("nsx_adam_hash_factored", [T], H, g.total_entries)[1])
    del G
    df = model.deformation_field
    if df is not None:
        box = model.scene_box.aabb.to(dev)
        pos = x * (box[1] - box[0]) + box[0]
        dcode_t = torch.randn((T, 128), device=dev, generator=gen) * 0.1
        packed = df.packed_params()
        aabb6 = df._aabb6()
        w7 = F.deform_window7(3.5)
        off = torch.empty((S, 3), device=dev)

        def dfwd():
            check(lib().nsx_deform_fwd(ptr(packed), ptr(pos), S, aabb6, ptr(dcode_t), dcode_t.stride(0), ptr(slot), w7,
                                       ptr(off), None, stream()), "nsx_deform_fwd")
        entry("nsx_deform_fwd", timeit(dfwd), "mfma", S * DEFORM_FWD_FLOPS)
        out["nsx_deform_fwd"]["note"] = "the per-sample-code operator (11 K-steps in the input GEMMs); the model's forwards take nsx_deform_fwd_rows"
        terms = torch.empty((int(lib().nsx_deform_terms_floats(T)),), device=dev)

        def dfwd_rows():
            check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(dcode_t), dcode_t.stride(0), ptr(slot), T, w7,
                                            ptr(off), ptr(terms), None, stream()), "nsx_deform_fwd_rows")
        entry("nsx_deform_fwd_rows", timeit(dfwd_rows), "mfma", S * DEFORM_FWD_FLOPS)
        out["nsx_deform_fwd_rows"]["note"] = ("codes are rows of a table (every forward of the model): code columns summed per "
                                              "row first, 192 of 256 MFMAs per tile; priced in the FLOPs of the reference's "
                                              "eight Linear layers")
        goff = torch.randn((S, 3), device=dev, generator=gen)
        gparams = torch.zeros(int(lib().nsx_deform_param_count()), device=dev)
        gtable = torch.zeros_like(dcode_t)
        scratch = torch.empty(int(lib().nsx_deform_scratch_bytes(S)), dtype=torch.uint8, device=dev)

        def dbwd():
            check(lib().nsx_deform_bwd(ptr(packed), ptr(pos), S, aabb6, ptr(dcode_t), dcode_t.stride(0), ptr(slot), T, w7,
                                       ptr(goff), ptr(scratch), ptr(gparams), ptr(gtable), None, None, stream()),
                  "nsx_deform_bwd")
        entry("nsx_deform_bwd", timeit(dbwd, n=max(3, iters // 2)), "mfma", S * DEFORM_FWD_FLOPS * 3.0)
    # the two fused MLPs (tcnn FullyFusedMLP equivalents): <= 16 MFMAs per 32 samples -- bound by their 100-170 B of
    # activations per sample and by launch ramps, priced against the matrix cores all the same
    field = model.field
    feats = torch.randn((S, 32), device=dev, generator=gen).half()
    base_out = torch.randn((S, 16), device=dev, generator=gen).half()
    dirs = torch.nn.functional.normalize(torch.randn((S, 3), device=dev, generator=gen), dim=-1)
    for net, tag, flops in ((field.mlp_base, "mlp_base", MLP_BASE_FLOPS), (field.mlp_head, "mlp_head", MLP_HEAD_FLOPS)):
        w16 = net.half_weights()
        nh = net.n_hidden_mats
        dW = torch.zeros((w16.numel(),), device=dev)
        if tag == "mlp_base":
            o16 = torch.empty((S, 16), device=dev, dtype=torch.float16)
            d16 = torch.randn((S, 16), device=dev, generator=gen).half()
            d32 = torch.empty((S, 32), device=dev)

            def mfwd(w16=w16, nh=nh, o16=o16):
                check(lib().nsx_mlp_fwd(ptr(w16), nh, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, net.out_act,
                                        ptr(o16), 16, None, stream()), "nsx_mlp_fwd")

            def mbwd(w16=w16, nh=nh, d16=d16, dW=dW, d32=d32):
                check(lib().nsx_mlp_bwd(ptr(w16), nh, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, net.out_act,
                                        ptr(d16), 16, ptr(dW), None, None, ptr(d32), None, stream()), "nsx_mlp_bwd")
        else:
            o3 = torch.empty((S, 3), device=dev, dtype=torch.float16)
            d3 = torch.randn((S, 3), device=dev, generator=gen).half()
            dbo = torch.zeros((S, 16), device=dev, dtype=torch.float16)

            def mfwd(w16=w16, nh=nh, o3=o3, act=net.out_act):
                check(lib().nsx_mlp_fwd(ptr(w16), nh, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, act,
                                        ptr(o3), 3, None, stream()), "nsx_mlp_fwd")

            def mbwd(w16=w16, nh=nh, d3=d3, dW=dW, dbo=dbo, act=net.out_act):
                check(lib().nsx_mlp_bwd(ptr(w16), nh, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, act,
                                        ptr(d3), 3, ptr(dW), None, ptr(dbo), None, None, stream()), "nsx_mlp_bwd")
        entry(f"nsx_mlp_fwd ({tag})", timeit(mfwd), "mfma", S * flops)
        entry(f"nsx_mlp_bwd ({tag})", timeit(mbwd), "mfma", S * flops * 3.0)
    return {"samples": S, "sampling": "uniform random positions in the scene box, 24 time-code slots, seed 0",
            "kernels": out}


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
    ap.add_argument("--grid", choices=("frozen", "live"), default="live",
                    help="frozen: the occupancy-grid state at the end of the warm-up is kept for the timed region (the "
                         "update still runs with all its work, its result is not adopted); live: the grid evolves")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 4096 rays per rank; strong: the 4096-ray batch is sliced 4096/N rays per rank with global "
                         "loss normalisers (SURVEY.md 8e)")
    ap.add_argument("--compact-first-grid", action="store_true",
                    help="NeRSembleTrainer(compact_first_grid=True), the trainer's own default -- the headline of this "
                         "benchmark runs WITHOUT it: while the coarse-to-fine window keeps one hash grid on "
                         "(the first 40 000 steps of the default schedule -- all of this benchmark's steps) train a "
                         "contiguous copy of that grid with the H = 1 kernels.  Same results; NOT the headline, which "
                         "prices a step in the full 32-grid layout.  The default run reports it as `first_grid_phase`")
    ap.add_argument("--no-first-grid-phase", action="store_true", help="skip the `first_grid_phase` block")
    ap.add_argument("--window-hash", type=int, nargs=2, default=None, metavar=("BEGIN", "END"),
                    help="steps of the coarse-to-fine schedule of the hash grids instead of the workload's (40000 80000, "
                         "train_nersemble.py:77-78); `0 1` = every grid on from step 1: the state of a run after END")
    ap.add_argument("--window-open", action="store_true", help="shorthand for --window-hash 0 1")
    ap.add_argument("--table-parallel", choices=("auto", "level", "shard"), default="auto",
                    help="N > 1: how the hash tables' step is shared -- shard: fp16 reduce-scatter / shard Adam / all-gather "
                         "(engine/sharded_adam.py); level: rank r owns levels [r L / N, (r + 1) L / N), samples travel instead "
                         "of parameters (engine/level_parallel.py); auto: shard while the coarse-to-fine window is below "
                         "H / 2, level from then on")
    ap.add_argument("--no-open-window", action="store_true", help="skip the `open_window` block")
    ap.add_argument("--sharded-one-rank", action="store_true",
                    help="N = 1 only: the table step of a data-parallel rank (ShardedTableAdam: fp16 expansion, reduce-scatter "
                         "and all-gather through RCCL on a ONE-rank group, Adam on the shard) instead of the fused "
                         "single-GPU pass -- what a rank of an N-GPU job computes per step, without the links")
    ap.add_argument("--level-parallel-one-rank", type=int, default=0, metavar="N",
                    help="N = 1 process only: train as rank --rank of an N-rank LEVEL-PARALLEL job whose other ranks are "
                         "replicas of this one (engine/level_parallel.py, emulate): the level partition, the sample "
                         "exchange's payloads and per-source-rank launches, N x 24 gradient planes, and every collective of "
                         "such a rank through RCCL on a ONE-rank group (+ the gloo side group).  What a rank of an N-GPU "
                         "job computes and issues per step with the window open, without the links (NOT the headline; the "
                         "numbers it trains on are not the model's).  Implies --window-hash 0 1")
    ap.add_argument("--rank", type=int, default=-1, help="the emulated rank of --level-parallel-one-rank (default N - 1: the "
                                                         "finest levels' owner, the slowest rank)")
    ap.add_argument("--lp-launch-per-source", action="store_true",
                    help="level-parallel runs: one HashEnsemble launch per source rank (NSX_OPT_LP_ONE_LAUNCH = 0) instead of "
                         "all source ranks in one launch -- the A/B switch of that change")
    ap.add_argument("--lp-torch-collectives", action="store_true",
                    help="level-parallel runs: the exchange's collectives through torch.distributed (five calls per step with "
                         "Python in between) instead of the library's own RCCL communicator (one C call per direction) -- "
                         "the A/B switch of that change")
    ap.add_argument("--bucket-tail-copies", action="store_true",
                    help="data-parallel / level-parallel runs: the tail of the small gradients' bucket by one copy per piece "
                         "(torch) instead of nsx_bucket_pack / _unpack -- the A/B switch of that change")
    ap.add_argument("--no-kernels-alone", action="store_true", help="skip the stand-alone kernel timings after the run")
    ap.add_argument("--with-datamanager", action="store_true",
                    help="draw every batch INSIDE the timed loop through NeRSembleVanillaDataManager.next_train (24-image "
                         "cache refreshed every 20 iterations, device pixel sampler, nsx_generate_rays), as the reference's "
                         "TRAIN_RAYS_PER_SEC clocks it (nersemble_trainer.py:41-68); default: batches pre-generated")
    ap.add_argument("--no-with-datamanager", action="store_true", help="skip the `with_datamanager` block")
    ap.add_argument("--preroll", type=int, default=0,
                    help="untimed training steps BEFORE the warm-up (e.g. 600: the occupancy grid and the visibility "
                         "pruning have settled, the timed window is stationary)")
    ap.add_argument("--steady-after", type=int, default=600,
                    help="after the timed region, training continues to this step and 100 more steps are timed as the "
                         "`steady_state` block (0: skip)")
    a = ap.parse_args()
    if (a.window_open or a.level_parallel_one_rank) and a.window_hash is None:
        a.window_hash = [0, 1]
    lp_emulation = None
    if a.level_parallel_one_rank:
        lp_emulation = (a.level_parallel_one_rank, a.rank if a.rank >= 0 else a.level_parallel_one_rank - 1)
        a.no_first_grid_phase = a.no_open_window = a.no_with_datamanager = True
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))
    if a.bucket_tail_copies:
        from nersemble_amd.engine import parallel as _par
        _par._NATIVE_PIECES = 0
    if a.lp_torch_collectives:
        from nersemble_amd.engine import level_parallel as _lpm
        _lpm.NATIVE_COLLECTIVES = False
    if a.lp_launch_per_source:
        from nersemble_amd import _lib
        _lib.check(_lib.lib().nsx_set_option(_lib.NSX_OPT_LP_ONE_LAUNCH, 0), "nsx_set_option")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.ranks_share_gpu0:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")     # (persistent buffers: no per-call recordStream)
        torch.cuda.set_device(local_rank)
        if a.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend=a.backend)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    if a.sharded_one_rank or lp_emulation:
        assert world == 1, "--sharded-one-rank / --level-parallel-one-rank are single-process measurements"
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s_.getsockname()[1])
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")     # (persistent buffers: no per-call recordStream)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=a.backend, rank=0, world_size=1,
                                **({"device_id": torch.device(f"cuda:{local_rank}")} if a.backend == "nccl" else {}))
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
    n_rays = None
    if a.scaling == "strong" and world > 1:
        n_rays = WORKLOADS[a.workload]["rays"] // world
    trainer, data, info = build_workload(a.workload, device=dev, rank=rank, world_size=world, n_rays=n_rays,
                                         global_loss_normalisers=(a.scaling == "strong" and world > 1),
                                         compact_first_grid=a.compact_first_grid,
                                         window_hash=tuple(a.window_hash) if a.window_hash else None,
                                         table_parallel=a.table_parallel, level_parallel_emulation=lp_emulation,
                                         **({"sharded_table_adam": True} if a.sharded_one_rank else {}))
    if a.reserve_gb > 0:
        reserve = torch.empty(int(a.reserve_gb * 2 ** 30), dtype=torch.uint8, device=dev)
        del reserve
    H = WORKLOADS[a.workload]["H"]
    if os.environ.get("NSX_EARLY_TABLE_STEP") == "1":            # experiment knobs (engine/trainer.py)
        trainer.early_table_step = True
    if os.environ.get("NSX_PREFETCH_MARCH") == "0":
        trainer.prefetch_march = False
    if os.environ.get("NSX_NATIVE_STEP") == "0":                  # A/B: every native call issued from Python (round 3's path)
        trainer.model.native_step = False
    table_opt = trainer.optimizers.get(trainer.group_of_tables())

    for s in range(a.preroll):                                   # optional: start the measurement from a settled state
        trainer.train_iteration(s, *data.next_train(s))
    # synthetic inputs are generated up front: they are resident in HBM when the timed region starts
    # (one more than is trained on: like a loader, the loop knows the next batch, and the trainer starts the counting pass
    # of its ray marching one step ahead -- every timed step issues exactly one such pass)
    dm, dm_stats = None, {"calls": 0, "host_s": 0.0, "each": []}
    if a.with_datamanager:
        dm, n_dm_images = build_datamanager(data, dev)
        batches = None
    else:
        batches = [data.next_train(a.preroll + s) for s in range(a.warmup + a.steps + 1)]
    torch.cuda.synchronize()

    def fetch(s):
        """The batch of step `s`: pre-generated, or drawn NOW through the datamanager (host time booked)."""
        if dm is None:
            return batches[s]
        t_ = time.perf_counter()
        out_ = dm.next_train(a.preroll + s)
        dm_stats["host_s"] += time.perf_counter() - t_
        dm_stats["each"].append(time.perf_counter() - t_)
        dm_stats["calls"] += 1
        return out_

    step_marks = []                          # (event at step start, device-side sample count) per timed step
    held = {"next": fetch(0)}

    def run(n_steps, first_step, mark=False):
        samples = 0
        for s in range(first_step, first_step + n_steps):
            if mark:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                _lib.profiler.tag = len(step_marks)
            bundle, batch = held["next"]
            held["next"] = fetch(s + 1)      # (like a loader, the loop knows its next batch: one next_train per step)
            loss, loss_dict, metrics = trainer.train_iteration(a.preroll + s, bundle, batch,
                                                               next_ray_bundle=held["next"][0])
            samples += metrics["num_samples_per_batch"]
            if mark:
                step_marks.append((ev, metrics["num_samples_per_batch"]))
        return samples, loss, metrics

    run(a.warmup, 0)
    if a.grid == "frozen":
        grid = trainer.model.occupancy_grid
        adopt = grid.apply_update

        def run_update_without_adopting(cell_ids, occ, occ_thre, ema_decay):
            held = (grid.occs.clone(), grid.binaries.clone())
            adopt(cell_ids, occ, occ_thre, ema_decay)           # the full update: scatter-max, EMA, mean, threshold
            grid.occs.copy_(held[0])
            grid.binaries.copy_(held[1])
            grid._occs_mean_key = None

        grid.apply_update = run_update_without_adopting
    # Python's cyclic garbage collector pauses the host for tens of ms whenever its generation-2 threshold trips
    # (sporadic 70-100 ms steps); training loops collect at controlled points instead (NeRSembleTrainer.gc_every)
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    # HIP events only around the calls that are priced against a roofline (+ the other large kernels), allocated
    # before the timed region
    global SPLIT_SCATTER
    sink = trainer.model.field.hash_ensemble.grad_sink
    SPLIT_SCATTER = bool(sink is not None and sink.split_scatter)
    _lib.profiler.watch = {"nsx_hash_ensemble_fwd", "nsx_hash_ensemble_bwd_factored", "nsx_hash_ensemble_bwd",
                           "nsx_hash_ensemble_bwd_codesum",
                           "nsx_hash_ensemble_bwd_scatter",
                           "nsx_adam_hash_factored", "nsx_adam_hash_factored_consume", "nsx_adam_dense",
                           "nsx_deform_fwd", "nsx_deform_fwd_rows", "nsx_deform_bwd",
                           "nsx_mlp_fwd", "nsx_mlp_bwd", "nsx_check_finite", "nsx_march_fill", "nsx_march_fill_from_stash",
                           "nsx_hash_grad_expand", "nsx_hash_grad_expand_f16", "nsx_adam_dense_f16grad",
                           "nsx_check_finite_f16", "nsx_lp_fwd_run", "nsx_lp_bwd_run",
                           # (the same two with their collectives, issued by the library: csrc/comm.hip)
                           "nsx_lp_forward", "nsx_lp_backward"}
    # (the variant of the table optimizer that also clears the gradient pieces it reads is priced as the optimizer pass)
    _lib.profiler.alias = {"nsx_adam_hash_factored_consume": "nsx_adam_hash_factored"}
    if not a.no_kernel_events:
        _lib.profiler.prewarm(2 * 16 * a.steps + 64)
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    table_opt = trainer.optimizers.get(trainer.group_of_tables())     # (the warm-up may have switched the exchange)
    if isinstance(table_opt, (ShardedTableAdam, LevelParallelTableAdam)):
        table_opt.timing = True                # HIP events around expand / reduce-scatter / shard Adam / all-gather
        table_opt._events = []
        if isinstance(table_opt, LevelParallelTableAdam):
            table_opt.comm_report()            # (zeroes the byte / sample counters of the warm-up)
            table_opt.timing = True
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    _lib.profiler.reset()
    _lib.profiler.enabled = not a.no_kernel_events
    dm_stats["calls"], dm_stats["host_s"], dm_stats["each"] = 0, 0.0, []
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
    dm_timed = dict(dm_stats)

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    n = torch.tensor([samples], device=dev, dtype=torch.int64)
    dt_all = [dt]
    comm = None
    if world > 1 or a.sharded_one_rank or lp_emulation:
        every = torch.zeros((world,), device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, t)
        dt_all = every.tolist()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        table_opt = trainer.optimizers.get(trainer.group_of_tables())
        if isinstance(table_opt, LevelParallelTableAdam):
            table_opt.timing = False
            comm = table_opt.comm_report()
            comm["backend"] = a.backend
            comm["rccl_ranks"] = world if a.backend == "nccl" else 0
            if lp_emulation:
                comm["emulated_world_size"] = lp_emulation[0]
            comm["ms_per_step_per_rank_min_max"] = [min(dt_all) / a.steps * 1e3, max(dt_all) / a.steps * 1e3]
            # the bar of the round-4 review: the exchange must scale with the samples (2 x 64 B per sample of the JOB and
            # pass for features out / gradients back), not with the 806 MB of parameters
            job_samples = max(comm["samples_bwd_per_step"], 1.0)
            comm["bytes_per_job_sample"] = comm["bytes_per_rank"] / job_samples
            comm["bar_bytes_per_rank"] = 2 * 64 * (comm["samples_fwd_per_step"] + comm["samples_bwd_per_step"]) / 2 * 1.05
            comm["note"] = ("level-parallel exchange (engine/level_parallel.py): bytes ARRIVING at this rank per step over all "
                            "its collectives -- positions + code slots all-gathered, feature columns / dL/dfeatures / "
                            "dL/dx by all-to-all (counted as all-to-all also on the gloo stand-in, which gathers) -- and "
                            "the job's samples that went through this rank's kernels; no table gradient and no table "
                            "values travel")
        elif isinstance(table_opt, ShardedTableAdam):
            table_opt.timing = False
            comm = table_opt.comm_report()                      # this rank's (rank 0 prints)
            comm["backend"] = a.backend
            comm["rccl_ranks"] = world if a.backend == "nccl" else 0
            comm["ms_per_step_per_rank_min_max"] = [min(dt_all) / a.steps * 1e3, max(dt_all) / a.steps * 1e3]
            comm["note"] = ("HIP events on the streams the phases run on, rank 0, mean over the timed steps; "
                            "reduce_scatter_ms spans the first piece's issue to the last piece's completion (the "
                            "expansion of the later pieces runs inside it), reduce_scatter_exposed_ms is what the main "
                            "stream waited for it before the inf check; shard Adam + all-gather run on the optimizer "
                            "stream beside the step's tail and the next step's marching")
    dt_max, total_samples = float(t.item()), int(n.item())
    steady_ranks = None
    if world > 1 and a.steady_after > 0:
        steady_ranks = steady_state_ranks(trainer, data, a.preroll + a.warmup + a.steps, a.steady_after, info["rays"], world,
                                          dev, a.backend)

    if rank == 0:
        _lib.profiler.collect_native()          # the kernel calls the native step drivers made (their own HIP events)
        prof = _lib.profiler.summary()
        total_entries = trainer.model.field.hash_ensemble.geom.total_entries
        kept = [int(c) for _, c in step_marks]
        pmc_state = "headline" if (a.workload == "p030_h32" and world == 1 and a.preroll == 0 and a.window_hash is None
                                   and not a.compact_first_grid) else None
        roofline, rooflines = compute_rooflines(prof, _lib.profiler.records, _lib.profiler.tags, kept, H, total_entries,
                                                trainer._opt_stream is not None, pmc_state)
        kernels = {k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 3), "avg_ms": round(v["avg_ms"], 4)}
                   for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])}
        out = {
            "metric": "ray-samples/sec training, 4096 rays x 2^20 samples", "value": total_samples / dt_max,
            "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt_max / a.steps * 1e3, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": a.workload, "participant": info["participant"], "n_hash_encodings": H,
                       "rays_per_gpu": info["rays"], "max_n_samples_per_batch": "2^20",
                       "samples_per_step_per_gpu": samples / a.steps, "n_timesteps": info["n_timesteps"],
                       "parallelism": f"dp{world}", "params": info["params"],
                       "occupancy_grid": "state at the end of the warm-up kept for the timed region; the update runs on "
                                         "schedule, its result is not adopted" if a.grid == "frozen" else "live",
                       "rccl_ranks": world if (world > 1 and a.backend == "nccl") else 0,
                       "table_step": ("one rank of a data-parallel job: ShardedTableAdam on a one-rank group (NOT the headline)"
                                      if a.sharded_one_rank else
                                      f"rank {lp_emulation[1]} of {lp_emulation[0]} of a level-parallel job, the other ranks "
                                      f"emulated as replicas, collectives on a one-rank group (NOT the headline)"
                                      if lp_emulation else type(table_opt).__name__),
                       "early_table_step": bool(trainer.early_table_step),
                       "march_count_one_step_ahead": bool(trainer.prefetch_march),
                       "table_adam_consumes_gradient": bool(getattr(table_opt, "consume_gradient", False)),
                       "compact_first_grid": bool(trainer.model.field.hash_ensemble.compact_first_grid),
                       "native_step_drivers": bool(trainer.model.native_step and trainer.model._native is not None),
                       "window_hash_schedule": list(a.window_hash) if a.window_hash else
                       [trainer.model.config.window_hash_encodings_begin, trainer.model.config.window_hash_encodings_end]},
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
        if comm is not None:
            out["comm"] = comm
        if dm is not None:
            out["datamanager"] = {
                "next_train_host_us_per_step": dm_timed["host_s"] / max(dm_timed["calls"], 1) * 1e6,
                "next_train_host_us_median": sorted(dm_timed["each"])[len(dm_timed["each"]) // 2] * 1e6
                if dm_timed["each"] else None,
                "next_train_host_us_max": max(dm_timed["each"]) * 1e6 if dm_timed["each"] else None,
                "max_is": "the call that redraws the 24-image cache (one in 20: a 0.7 GB stack of resident tensors)",
                "next_train_calls_in_timed_region": dm_timed["calls"], "dataset_images": n_dm_images,
                "image_cache": "24 images, redrawn every 20 iterations (train_nersemble.py:174-175)",
                "note": "next_train(step) is called inside the timed loop, one call per step, for the NEXT step's batch "
                        "(its ray bundle is handed to train_iteration as next_ray_bundle)"}
        if a.steady_after > 0 and world == 1:
            state = None
            if a.workload == "p030_h32" and a.preroll == 0:
                state = ("steady_compact" if a.compact_first_grid else
                         "steady_open_window" if (a.window_hash and tuple(a.window_hash) == (0, 1)) else
                         "steady_full" if a.window_hash is None else None)
            if lp_emulation:
                # the emulated rank of a TRAINED model: a single-GPU run to the steady state (the real model: its occupancy
                # grid and visibility pruning decide how many samples a step marches and keeps), which then goes on as the
                # emulated rank with frozen parameters and the true feature columns (engine/trainer.py,
                # become_emulated_level_parallel_rank) -- every kernel and collective of such a rank at a trained model's
                # sample counts.  (The timed region above is the same rank from a FRESH model: dense steps need no model.)
                del trainer
                torch.cuda.empty_cache()
                torch.manual_seed(19980801)
                trainer, data, info = build_workload(a.workload, device=dev, window_hash=tuple(a.window_hash),
                                                     compact_first_grid=False)
                for s_ in range(a.steady_after):
                    trainer.train_iteration(s_, *data.next_train(s_))
                trainer.flush_scheduler_step()
                trainer.become_emulated_level_parallel_rank(*lp_emulation)
                out["steady_state"] = steady_state(trainer, data, a.steady_after, a.steady_after, info["rays"],
                                                   H=0 if a.no_kernel_events else H)
                out["steady_state"]["comm"] = out["steady_state"].pop("_comm", None)
                out["steady_state"]["protocol"] = (
                    f"{a.steady_after} single-GPU steps (window open), then rank {lp_emulation[1]} of {lp_emulation[0]} "
                    f"emulated with frozen parameters; the replicas' feature columns replaced by one full-geometry forward "
                    f"per pass (comm.shadow_fwd_ms: extra work of the emulation, inside ms_per_step)")
            else:
                out["steady_state"] = steady_state(trainer, data, a.preroll + a.warmup + a.steps, a.steady_after,
                                                   info["rays"], datamanager=dm, H=0 if a.no_kernel_events else H,
                                                   pmc_state=state)
        if steady_ranks is not None:
            out["steady_state"] = steady_ranks
        spp = [p["samples"] for p in out["per_step"]]
        out["samples_per_step_min_max"] = [min(spp), max(spp)] if spp else None
        if len(out["per_step"]) > 40:                            # long runs: every k-th step is enough to see the trend
            out["per_step"] = out["per_step"][::len(out["per_step"]) // 40]
        if not a.no_kernels_alone and world == 1:
            out["kernels_alone"] = kernels_alone(trainer, H)
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(H)
        if world == 1 and not a.compact_first_grid and not a.no_first_grid_phase and a.preroll == 0 \
                and a.window_hash is None:
            out["first_grid_phase"] = first_grid_phase_block(a)
        if world == 1 and not a.compact_first_grid and not a.no_open_window and a.preroll == 0 and a.window_hash is None:
            out["open_window"] = open_window_block(a)
        if world == 1 and not a.compact_first_grid and not a.no_with_datamanager and not a.with_datamanager \
                and a.preroll == 0 and a.window_hash is None:
            out["with_datamanager"] = with_datamanager_block(a)
        print(json.dumps(out))
    if world > 1 or a.sharded_one_rank or lp_emulation:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
