"""Host time of ONE steady-state step issued into an empty queue, by section (median over 60 steps).

The compact first-grid phase (steps 0 ... 40 000, the trainer's default) has ~1.9 ms of device work per step in steady state --
about what the host needs to issue a step -- so there the host's sections are the step's critical path.

    python tools/host_sections.py [--full-layout] [--window-open]
"""
import argparse
import collections
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="p030_h32")
    ap.add_argument("--full-layout", action="store_true")
    ap.add_argument("--window-open", action="store_true")
    ap.add_argument("--settle-at", type=int, default=600)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--level-parallel-one-rank", type=int, default=0, metavar="N",
                    help="after settling: go on as the finest levels' owner of an N-rank level-parallel job (emulated, frozen)")
    ap.add_argument("--sharded-one-rank", action="store_true",
                    help="the table step of a data-parallel rank (ShardedTableAdam, collectives through RCCL on a one-rank group)")
    ap.add_argument("--fine", action="store_true",
                    help="also time the stream / event / collective / native calls (class-level wrappers: they see the autograd "
                         "engine's thread too, which cProfile does not); nested sections count twice")
    ap.add_argument("--cprofile", default=None, metavar="FILE",
                    help="instead of the section timers: cProfile over the steps, the 70 heaviest functions (own and cumulative "
                         "time, per step) written to FILE")
    a = ap.parse_args()
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)
    if a.level_parallel_one_rank or a.sharded_one_rank:
        import socket
        import torch.distributed as dist
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s_.getsockname()[1])
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")     # (persistent buffers: no per-call recordStream)
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        if a.level_parallel_one_rank:
            a.window_open, a.full_layout = True, True
    trainer, data, info = build_workload(a.workload, device="cuda:0", compact_first_grid=not a.full_layout,
                                         window_hash=(0, 1) if a.window_open else None,
                                         **({"sharded_table_adam": True} if a.sharded_one_rank else {}))
    reserve = torch.empty(24 * 2 ** 30, dtype=torch.uint8, device="cuda:0")
    del reserve
    step = 0
    while step < a.settle_at:
        trainer.train_iteration(step, *data.next_train(step))
        step += 1
    if a.level_parallel_one_rank:
        trainer.flush_scheduler_step()
        trainer.become_emulated_level_parallel_rank(a.level_parallel_one_rank, a.level_parallel_one_rank - 1)
        for _ in range(8):
            trainer.train_iteration(step, *data.next_train(step))
            step += 1
    acc = collections.defaultdict(list)
    cur = collections.defaultdict(float)

    def timed(name, fn):
        def wrapper(*args, **kw):
            t0 = time.perf_counter()
            try:
                return fn(*args, **kw)
            finally:
                cur[name] += time.perf_counter() - t0
        return wrapper

    model = trainer.model
    if a.cprofile:
        import cProfile
        import io
        import pstats
        n = a.steps
        batches = [data.next_train(step + i) for i in range(n + 1)]
        gc.collect(); gc.freeze(); gc.disable()
        prof = cProfile.Profile()
        for i in range(n):
            torch.cuda.synchronize()
            prof.enable()
            trainer.train_iteration(step + i, *batches[i], next_ray_bundle=batches[i + 1][0])
            prof.disable()
        trainer.flush_scheduler_step()
        with open(a.cprofile, "w") as f:
            for key in ("cumulative", "tottime"):
                buf = io.StringIO()
                pstats.Stats(prof, stream=buf).strip_dirs().sort_stats(key).print_stats(70)
                f.write(f"==== by {key}, totals over {n} steps (divide by {n}) ====\n" + buf.getvalue())
        return
    model.prefetch_sampling = timed("prefetch_sampling", model.prefetch_sampling)
    model.fused_train_forward = timed("forward (all)", model.fused_train_forward)
    model.sampler.forward = timed("  sampler", model.sampler.forward)
    model.field_density_fn = timed("    sigma_fn density", model.field_density_fn)
    trainer._optimizer_step_all = timed("optimizer (all)", trainer._optimizer_step_all)
    trainer._all_reduce_grads = timed("all-reduce of the small gradients", trainer._all_reduce_grads)
    lp = getattr(model.field.hash_ensemble, "level_parallel", None)
    if lp is not None:
        lp.exchange_sizes = timed("  lp: size exchange (host, gloo)", lp.exchange_sizes)
        lp.features = timed("  lp: forward exchange", lp.features)
        lp.backward = timed("  lp: backward exchange", lp.backward)
        lp._all_gather = timed("    lp: all_gather call", lp._all_gather)
        lp._all_to_all = timed("    lp: all_to_all calls", lp._all_to_all)
    if a.sharded_one_rank:
        sopt = trainer.optimizers[trainer.group_of_tables()]
        sopt.ensure_reduce_started = timed("  dp: ensure_reduce_started", sopt.ensure_reduce_started)
        sopt._start_reduce = timed("    dp: _start_reduce (expansion + reduce-scatter issue)", sopt._start_reduce)
        sopt.check_finite = timed("  dp: check_finite", sopt.check_finite)
        sopt.step = timed("  dp: step (shard Adam + all-gather issue)", sopt.step)
    if a.fine:
        import torch.distributed as dist
        from nersemble_amd.engine import native_step as ns
        counts = collections.defaultdict(int)

        def timed_n(name, fn):
            inner = timed(name, fn)

            def w(*args, **kw):
                counts[name] += 1
                return inner(*args, **kw)
            return w
        for cls, names in ((torch.cuda.Stream, ("wait_stream", "wait_event", "record_event")),
                           (torch.cuda.Event, ("wait", "record"))):
            for nm in names:
                setattr(cls, nm, timed_n(f"      {cls.__name__}.{nm}", getattr(cls, nm)))
        torch.Tensor.record_stream = timed_n("      Tensor.record_stream", torch.Tensor.record_stream)
        for nm in ("reduce_scatter_tensor", "all_gather_into_tensor", "all_reduce", "all_to_all_single"):
            setattr(dist, nm, timed_n(f"      dist.{nm}", getattr(dist, nm)))
        ns._NativeMain.forward = staticmethod(timed_n("    _NativeMain.forward", ns._NativeMain.forward))
        ns._NativeMain.backward = staticmethod(timed_n("    _NativeMain.backward", ns._NativeMain.backward))
        sink = model.field.hash_ensemble.grad_sink
        if sink is not None:
            if sink.on_complete is not None:
                sink.on_complete = timed_n("      sink.on_complete", sink.on_complete)
            sink.buffer_for = timed_n("      sink.buffer_for", sink.buffer_for)
        if a.sharded_one_rank:
            ops = sopt.ops
            for nm in ("expand_f16_bucket_width", "adam_f16grad", "check_finite_f16", "unpack_width"):
                setattr(ops, nm, timed_n(f"      ops.{nm}", getattr(ops, nm)))
            sopt._expand_and_reduce = timed_n("      dp: _expand_and_reduce", sopt._expand_and_reduce)
            sopt._step_now = timed_n("      dp: _step_now", sopt._step_now)
        real_stream_ctx = torch.cuda.stream
    trainer.flush_scheduler_step = timed("  flush_scheduler_step", trainer.flush_scheduler_step)
    trainer._defer_scheduler_step = timed("defer_scheduler_step", trainer._defer_scheduler_step)
    for cb in trainer.callbacks:
        cb.run = timed("callbacks", cb.run)
    for opt in trainer.optimizers.values():
        opt.zero_grad = timed("zero_grad", opt.zero_grad)
    real_backward = torch.autograd.backward
    torch.autograd.backward = timed("backward", real_backward)

    n = a.steps
    batches = [data.next_train(step + i) for i in range(n + 1)]
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(n):
        torch.cuda.synchronize()
        cur.clear()
        t0 = time.perf_counter()
        trainer.train_iteration(step + i, *batches[i], next_ray_bundle=batches[i + 1][0])
        cur["TOTAL"] = time.perf_counter() - t0
        if (step + i) % 16 != 0:                    # (occupancy-update steps are a different animal)
            for k, v in cur.items():
                acc[k].append(v)
    torch.autograd.backward = real_backward
    trainer.flush_scheduler_step()
    print(f"host time per step issued into an empty queue, median of {len(acc['TOTAL'])} steps "
          f"({'full layout' if a.full_layout else 'compact first-grid phase'}{', window open' if a.window_open else ''})")
    for k, v in sorted(acc.items(), key=lambda kv: -sorted(kv[1])[len(kv[1]) // 2]):
        v = sorted(v)
        print(f"{k:34s} {v[len(v) // 2] * 1e6:8.0f} us   (min {v[0] * 1e6:.0f})")


if __name__ == "__main__":
    main()
