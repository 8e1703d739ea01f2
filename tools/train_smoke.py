"""Development aid: run a few training iterations of a small / full config on the GPU and print losses."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd.workloads import build_workload  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="p030_h16")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(19980801)
    trainer, data, info = build_workload(a.workload, device="cuda:0", small=a.small)
    print(info)
    t0 = time.time()
    for step in range(a.steps):
        bundle, batch = data.next_train(step)
        loss, loss_dict, metrics = trainer.train_iteration(step, bundle, batch)
        if step % max(1, a.steps // 10) == 0 or step == a.steps - 1:
            torch.cuda.synchronize()
            print(step, f"loss={loss.item():.5f}", {k: round(v.item(), 6) for k, v in loss_dict.items()},
                  f"psnr={metrics['psnr'].item():.2f}", f"samples={int(metrics['num_samples_per_batch'])}",
                  f"t={time.time() - t0:.2f}s", flush=True)


if __name__ == "__main__":
    main()
