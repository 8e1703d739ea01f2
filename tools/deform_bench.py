"""Development aid: time / profile the fused deformation kernels alone (S samples, 24 code slots)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig

ap = argparse.ArgumentParser(); ap.add_argument("--S", type=int, default=1 << 20); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda:0")
aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
torch.manual_seed(0)
df = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(dev)
pos = (torch.rand(a.S, 3) * (aabb[1] - aabb[0]) + aabb[0]).to(dev)
table = (torch.randn(24, 128) * 0.3).to(dev).requires_grad_(True)
slot = torch.randint(0, 24, (a.S,), dtype=torch.int32, device=dev)
g = torch.randn(a.S, 3, device=dev)
def fwd():
    with torch.no_grad():
        return df.compute_offsets(pos, table, 3.5, code_index=slot)
def fb():
    off = df.compute_offsets(pos, table, 3.5, code_index=slot)
    off.backward(g)
for f, name in ((fwd, "fwd"), (fb, "fwd+bwd")):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters): f()
    e.record(); torch.cuda.synchronize()
    print(name, s.elapsed_time(e) / a.iters, "ms")
