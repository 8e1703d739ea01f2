"""Physical placement of the hash-table optimizer streams in HBM.

The table Adam pass streams four large arrays (fp32 master, exp_avg, exp_avg_sq: 1.6 GB each at H = 32; fp16 working
tables: 0.8 GB) and is HBM-bound.  Measured on MI355X (``tools/adam_placement.py``, profiles/README.md): the duration of
the SAME kernel over the SAME amount of data depends on which physical pages the four arrays were given -- 1.89 ms to
2.26 ms across six sets of separate allocations inside one process, reproducibly per set (re-timing a set gives its
number again), while one big block carved into four is always at the slow end (2.2 ms) whatever the padding between
the arrays.  The driver hands out physical memory in large fragments whose spread over the HBM stacks / channels
differs from allocation to allocation; virtual addresses say nothing about it.

With 288 GB of HBM the cheap answer is to look: before training starts, allocate a few candidate sets, time the real
Adam kernel on each (with a zero gradient, zero moments and lr = 0 the pass is value-preserving: the master keeps its
values, the moments stay 0, the fp16 copy is rewritten from the master), keep the fastest set and give the rest back.
One-off cost: ~K x 6.4 GB of transient memory and a few tens of milliseconds.

No counterpart in the reference (torch.optim.Adam over tcnn's parameter tensor, train_nersemble.py:243-246); results
are unchanged -- only where the arrays live.
"""
import ctypes as C
from typing import Dict, List, Optional

import torch

from .._lib import check, lib, ptr, stream

MIN_PARAMS = 32 * 1000 * 1000        # below this the pass is too short for placement to matter (tests, H = 1)


def _time_adam(he, master, m, v, f16, G, code, iters: int) -> float:
    """Median duration (ms) of the factored table Adam pass over one candidate set; value-preserving (see above)."""
    one = torch.ones((1,), dtype=torch.float32, device=master.device)
    zero = torch.zeros((1,), dtype=torch.float32, device=master.device)
    times = []
    for it in range(iters + 1):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(lib().nsx_adam_hash_factored(ptr(G), 1, ptr(code), code.stride(0), None, he.n_hash_encodings,
                                           C.byref(he.geom), ptr(master), ptr(m), ptr(v), ptr(f16), 0.0, 0.9, 0.999,
                                           1e-15, 1, ptr(one), ptr(zero), stream()), "nsx_adam_hash_factored")
        e.record()
        e.synchronize()
        if it > 0:                                       # first run = warm-up (page faults, TLB)
            times.append(s.elapsed_time(e))
    return sorted(times)[len(times) // 2]


DEFAULT_CANDIDATES = 14
# Round 6, last session: over four benchmark runs of the round (24 candidates) a quarter of the placements were "fast" (1.76-1.93 ms
# for the 12 GB pass) and the rest 2.03-2.19; with six candidates a run draws no fast one with probability 0.75^6 = 18 % -- the
# final check's box did (best of six: 2.05 ms, headline 7.49 instead of 7.2-7.3 ms per step).  Fourteen candidates: 2 %.  They are
# held at once (a freed set would be handed out again): 14 x 5.6 GB transient of 288 GB, ~0.2 s once per run.


@torch.no_grad()
def calibrate_table_placement(he, optimizer, candidates: int = DEFAULT_CANDIDATES, iters: int = 3,
                              min_params: int = MIN_PARAMS) -> Optional[Dict[str, object]]:
    """Re-homes ``he.tables`` (fp32 master), ``he.tables_f16`` and the Adam moments of ``optimizer`` (a
    ``HashTableAdam`` that has not stepped yet) into the fastest of ``candidates`` placements.  Returns a report
    (``candidate_ms``, ``chosen``, ``chosen_ms``) or ``None`` when nothing was done (CPU tensors, small tables, not
    enough free memory, optimizer already holds state)."""
    p = he.tables
    n = p.numel()
    if not p.is_cuda or n < min_params or candidates < 2 or len(optimizer.state.get(p, {})) != 0:
        return None
    dev = p.device
    free, _ = torch.cuda.mem_get_info(dev)
    per_set = n * 14                                     # 3 x fp32 + fp16
    candidates = int(min(candidates, 1 + (free * 0.5) // per_set))
    if candidates < 2:
        return None
    he.wait_tables()
    f16_live = he.half_tables()
    G = torch.zeros((1, he.geom.total_entries, 2), dtype=torch.float32, device=dev)
    code = torch.zeros((1, he.n_hash_encodings), dtype=torch.float32, device=dev)

    sets: List[tuple] = []
    report: List[float] = []
    for k in range(candidates):
        if k == 0:                                       # the placement the model already has
            master, f16 = p.data, f16_live
        else:
            master = torch.zeros_like(p.data)
            f16 = torch.zeros_like(f16_live)
        m, v = torch.zeros_like(p.data), torch.zeros_like(p.data)
        sets.append((master, m, v, f16))
        report.append(_time_adam(he, master, m, v, f16, G, code, iters))
    best = min(range(candidates), key=lambda k: report[k])
    master, m, v, f16 = sets[best]
    if best != 0:
        master.copy_(p.data)
        f16.copy_(f16_live)
        p.data = master
        he.tables_f16 = f16
    he.mark_half_synced()
    optimizer.state[p] = {"step": 0, "exp_avg": m, "exp_avg_sq": v}
    del sets, G, code, f16_live
    torch.cuda.empty_cache()                             # hand the losing candidates back to the driver
    return {"candidate_ms": [round(t, 4) for t in report], "chosen": best, "chosen_ms": round(report[best], 4)}
