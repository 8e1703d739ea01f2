"""Experiment (round 6): is the evaluation lookup in the 25 MB pre-blended grid bound by L2 misses that a level-by-level order
would avoid?  nsx_hashgrid_fwd (F = 2) on 2^20 ray-coherent samples: ONE call over all 16 levels (every sample touches 25 MB
of table) against 16 calls of one level each (2 MB or less of table per call: resident in an XCD's 4 MB L2)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle
from nersemble_amd import functional as F
from nersemble_amd.engine.level_parallel import sub_geometry_levels

dev = "cuda:0"
g = F.geom_from_oracle(oracle.grid_geometry()) if hasattr(F, "geom_from_oracle") else None
if g is None:
    from nersemble_amd.workloads import build_workload
    trainer, data, _ = build_workload("p030_h32", device=dev)
    g = trainer.model.field.hash_ensemble.geom
total = int(g.offset[g.n_levels])
torch.manual_seed(0)
table = (torch.rand((total, 2), device=dev) - 0.5).to(torch.float16)
S = 1 << 20
for name in ("ray-coherent", "uniform"):
    if name == "uniform":
        x = torch.rand((S, 3), device=dev)
    else:
        R, n = 4096, S // 4096
        o = torch.rand((R, 1, 3), device=dev) * 0.2 + 0.05
        d = torch.nn.functional.normalize(torch.rand((R, 1, 3), device=dev) + 0.2, dim=-1)
        t = torch.linspace(0, 0.7, n, device=dev).view(1, n, 1)
        x = (o + d * t).clamp(0, 0.999).reshape(-1, 3).contiguous()
    subs = [(sub_geometry_levels(g, [l]), table[int(g.offset[l]):int(g.offset[l + 1])].contiguous()) for l in range(g.n_levels)]

    def all_levels():
        return F.hashgrid_fwd_f16(x, table, 2, g)

    def level_by_level():
        return [F.hashgrid_fwd_f16(x, tb, 2, sg) for sg, tb in subs]

    ref = all_levels()
    parts = torch.cat(level_by_level(), dim=1)
    same = bool(torch.equal(ref, parts))
    for fn, label in ((all_levels, "one call, 16 levels"), (level_by_level, "16 calls, one level each")):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name:13s} {label:26s} {(time.perf_counter() - t0) * 100:.3f} ms per 2^20 samples   (same values: {same})")
    per = []
    for l, (sg, tb) in enumerate(subs):
        F.hashgrid_fwd_f16(x, tb, 2, sg); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            F.hashgrid_fwd_f16(x, tb, 2, sg)
        torch.cuda.synchronize()
        per.append(round((time.perf_counter() - t0) * 100, 4))
    print(f"{name:13s} per level (ms): {per}; table KB per level: {[int(tb.numel() * 2 // 1024) for _, tb in subs]}")
