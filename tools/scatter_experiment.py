"""Development aid: nsx_hash_ensemble_bwd_scatter alone on 2^20 uniform / ray-ordered samples (16 gradient planes).
Used with throw-away kernel variants to find what bounds the scatter: half the lane operations on the same
sectors (one feature only) leave the uniform case unchanged (3.74 vs 3.76 ms) -> the bound is the number of 32-byte
sector requests, not the number of float atomics; packing the feature pair into one fp16x2 atomic would not help."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib
from nersemble_amd._lib import check, lib, ptr, stream
dev = torch.device("cuda:0")
g = _lib.grid_geometry()
total = int(g.offset[g.n_levels])
S, T = 1 << 20, 16
torch.manual_seed(0)
for mode in ("uniform", "rays"):
    if mode == "uniform":
        x = torch.rand(S, 3, device=dev)
    else:   # 4096 rays x 256 consecutive samples
        o = torch.rand(4096, 1, 3, device=dev) * 0.5 + 0.1
        d = torch.nn.functional.normalize(torch.rand(4096, 1, 3, device=dev) - 0.3, dim=-1)
        t = torch.arange(256, device=dev).view(1, 256, 1) * 0.0015
        x = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous()
    slot = (torch.arange(S, device=dev) // 256 % T).int()
    dout = torch.randn(S, 32, device=dev)
    G = torch.zeros(T, total, 2, device=dev)
    def run():
        check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), S, C.byref(g), T, ptr(slot), ptr(dout), ptr(G), None, 8, None, stream()), "s")
    for _ in range(2): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    print(os.environ.get("NSX_SCATTER_EXPERIMENT", "0"), mode, round(s.elapsed_time(e) / 5, 3), "ms")
