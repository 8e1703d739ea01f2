"""Census of the torch (non-native) ops one training step dispatches, grouped by the repo line that issued them.
    python tools/op_census.py [workload] [step]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nersemble_amd.workloads import build_workload  # noqa: E402


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.counts = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if fr.filename.startswith(ROOT) and "tools/op_census" not in fr.filename:
                site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}"
                break
        self.counts[(site, str(func).replace("aten.", ""))] += 1
        return func(*args, **(kwargs or {}))


name = sys.argv[1] if len(sys.argv) > 1 else "p030_h32"
at = int(sys.argv[2]) if len(sys.argv) > 2 else 21
torch.manual_seed(0)
trainer, data, info = build_workload(name, device="cuda:0")
batches = [data.next_train(s) for s in range(at + 1)]
for s in range(at):
    trainer.train_iteration(s, *batches[s])
torch.cuda.synchronize()
c = Census()
with c:
    trainer.train_iteration(at, *batches[at])
torch.cuda.synchronize()
by_site = collections.Counter()
for (site, op), n in c.counts.items():
    by_site[site] += n
print("total dispatched ops:", sum(c.counts.values()))
for site, n in by_site.most_common(80):
    ops = ", ".join(f"{op}x{k}" for (s2, op), k in c.counts.items() if s2 == site)
    print(f"{n:4d}  {site:60s} {ops[:150]}")
