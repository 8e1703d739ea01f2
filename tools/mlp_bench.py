"""Development aid: nsx_mlp_fwd / nsx_mlp_bwd alone (mlp_base: no hidden matrix, 32 -> 64 -> 16; mlp_head: one hidden matrix,
18 -> 64 -> 64 -> 3) on S samples.   python tools/mlp_bench.py [S] [--bwd-half-blocks N] [--bwd0-half-blocks N]"""
import argparse, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd import _lib, functional as F
from nersemble_amd._lib import check, lib, ptr, stream
ap = argparse.ArgumentParser()
ap.add_argument("S", nargs="?", type=int, default=1 << 20)
ap.add_argument("--bwd-half-blocks", type=int, default=0)
ap.add_argument("--bwd0-half-blocks", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
S = a.S
if a.bwd_half_blocks:
    check(lib().nsx_set_option(_lib.NSX_OPT_MLP_BWD_HALF_BLOCKS_PER_CU, a.bwd_half_blocks), "opt")
if a.bwd0_half_blocks:
    check(lib().nsx_set_option(_lib.NSX_OPT_MLP_BWD0_HALF_BLOCKS_PER_CU, a.bwd0_half_blocks), "opt")
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
feats = torch.randn(S, 32, device=dev).half()
base_out = torch.randn(S, 16, device=dev).half()
dirs = torch.randn(S, 3, device=dev)
line = {"S": S}
for nh, name in ((0, "base"), (1, "head")):
    w = (torch.randn(F.mlp_param_count(nh), device=dev) * 0.1).half()
    dW = torch.zeros(w.numel(), device=dev)
    if nh == 0:
        out = torch.empty(S, 16, device=dev).half()
        dout = torch.randn(S, 16, device=dev).half(); db32 = torch.empty(S, 32, device=dev)
        fwd = lambda: check(lib().nsx_mlp_fwd(ptr(w), 0, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, 0, ptr(out), 16, None, stream()), "f")
        bwd = lambda: check(lib().nsx_mlp_bwd(ptr(w), 0, S, None, 0, 0, 1.0, 0.0, ptr(feats), 32, 0, 32, 16, 0, ptr(dout), 16, ptr(dW), None, None, ptr(db32), None, stream()), "b")
    else:
        out = torch.empty(S, 3, device=dev).half()
        dout = torch.randn(S, 3, device=dev).half(); dbo = torch.zeros(S, 16, device=dev).half()
        fwd = lambda: check(lib().nsx_mlp_fwd(ptr(w), 1, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, 1, ptr(out), 3, None, stream()), "f")
        bwd = lambda: check(lib().nsx_mlp_bwd(ptr(w), 1, S, ptr(dirs), 3, 3, 0.5, 0.5, ptr(base_out), 16, 1, 15, 3, 1, ptr(dout), 3, ptr(dW), None, ptr(dbo), None, None, stream()), "h")
    line[name] = {"fwd_ms": round(timeit(fwd), 4), "bwd_ms": round(timeit(bwd), 4)}
print(json.dumps(line))
