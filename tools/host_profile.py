"""Where the HOST spends a steady-state step: cProfile over 100 steps from step 600 of the bench workload.

The steady-state step is ~75 kernels in ~3.4 ms; the device timeline (profiles/r03_timeline_steady_state*.txt) shows idle
gaps in front of the first kernels of the backward, i.e. the host has no lead there.  This prints (a) the step time with the
device drained at the end, (b) the time the host needs to ISSUE the same steps, (c) the profile of the issuing code.

    python tools/host_profile.py [--window-open] [--steps 100] > gpurun_out/host_profile.txt
"""
import argparse
import cProfile
import gc
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="p030_h32")
    ap.add_argument("--window-open", action="store_true")
    ap.add_argument("--settle-at", type=int, default=600)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--plain", action="store_true", help="only run the timed loop (for use under rocprofv3)")
    ap.add_argument("--compact", action="store_true", help="the trainer's default: compact first-grid phase")
    ap.add_argument("--datamanager", action="store_true",
                    help="draw the batches inside the loop through NeRSembleVanillaDataManager.next_train (bench.py --with-datamanager)")
    ap.add_argument("--sharded-one-rank", action="store_true",
                    help="the table step of a data-parallel rank (ShardedTableAdam, collectives through RCCL on a one-rank group)")
    ap.add_argument("--level-parallel-one-rank", type=int, default=0, metavar="N",
                    help="after settling: go on as rank --rank of an N-rank level-parallel job (emulated, frozen; window open)")
    ap.add_argument("--rank", type=int, default=None)
    a = ap.parse_args()
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)
    if a.level_parallel_one_rank:
        a.window_open, a.compact = True, False
    if a.sharded_one_rank or a.level_parallel_one_rank:
        import socket
        import torch.distributed as dist
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(s_.getsockname()[1])
        s_.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
        torch.cuda.set_device(0)
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    trainer, data, info = build_workload(a.workload, device="cuda:0", compact_first_grid=a.compact,
                                         window_hash=(0, 1) if a.window_open else None,
                                         **({"sharded_table_adam": True} if a.sharded_one_rank else {}))
    dm = None
    if a.datamanager:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import build_datamanager
        dm, _ = build_datamanager(data, "cuda:0")
    reserve = torch.empty(24 * 2 ** 30, dtype=torch.uint8, device="cuda:0")
    del reserve
    step = 0
    while step < a.settle_at:
        trainer.train_iteration(step, *data.next_train(step))
        step += 1
    if a.level_parallel_one_rank:
        trainer.flush_scheduler_step()
        N = a.level_parallel_one_rank
        trainer.become_emulated_level_parallel_rank(N, N - 1 if a.rank is None else a.rank)
        for _ in range(8):
            trainer.train_iteration(step, *data.next_train(step))
            step += 1
    n = a.steps
    batches = [data.next_train(step + i) for i in range(3 * n + 1)] if dm is None else None
    gc.collect()
    gc.freeze()
    gc.disable()
    held = {"next": dm.next_train(step) if dm is not None else None}

    def loop(first):
        for i in range(first, first + n):
            if dm is None:
                cur, ahead = batches[i], batches[i + 1]
            else:
                cur = held["next"]
                ahead = held["next"] = dm.next_train(step + i + 1)
            trainer.train_iteration(step + i, *cur, next_ray_bundle=ahead[0])

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(0)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"steps {n}: {t_all / n * 1e3:.3f} ms/step with the device drained, {t_issue / n * 1e3:.3f} ms/step until the "
          f"host had issued them (the difference is the host's lead at the end of the loop)")
    if a.plain:
        trainer.flush_scheduler_step()
        return

    prof = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prof.enable()
    loop(n)
    prof.disable()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"under cProfile: {t_all / n * 1e3:.3f} ms/step drained, {t_issue / n * 1e3:.3f} ms/step issued")
    for key, rows in (("tottime", 60), ("cumulative", 70)):
        s = io.StringIO()
        pstats.Stats(prof, stream=s).strip_dirs().sort_stats(key).print_stats(rows)
        print(s.getvalue())

    # the same steps with the profiler off again: what the host-side cost of a step is when nothing throttles it is not
    # observable directly (the queue applies back-pressure), but a step whose device work is short shows it: time the
    # issue of ONE step after a drain, many times
    lat = []
    for i in range(2 * n, 2 * n + min(n, 50)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        trainer.train_iteration(step + i, *batches[i], next_ray_bundle=batches[i + 1][0])
        lat.append(time.perf_counter() - t0)
    lat.sort()
    print(f"host time of one step issued into an EMPTY queue: median {lat[len(lat) // 2] * 1e3:.3f} ms, "
          f"min {lat[0] * 1e3:.3f} ms, max {lat[-1] * 1e3:.3f} ms")
    trainer.flush_scheduler_step()


if __name__ == "__main__":
    main()
