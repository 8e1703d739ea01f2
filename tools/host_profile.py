"""Host-side profile of the training step (which Python frames / torch ops the CPU spends its time in).
    python tools/host_profile.py [workload] [steps] [warmup_steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd.workloads import build_workload  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "p030_h32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
trainer, data, info = build_workload(name, device="cuda:0")
batches = [data.next_train(s) for s in range(warm + 2 * steps)]
for s in range(warm):
    trainer.train_iteration(s, *batches[s])
torch.cuda.synchronize()

t0 = time.perf_counter()
for s in range(warm, warm + steps):
    trainer.train_iteration(s, *batches[s])
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step")

pr = cProfile.Profile()
pr.enable()
for s in range(warm + steps, warm + 2 * steps):
    trainer.train_iteration(s, *batches[s])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(60)
st.sort_stats("tottime").print_stats(45)
