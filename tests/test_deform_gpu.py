"""GPU parity: fused SE(3) deformation kernel (PE + 6x128 MLP + heads + exp map + warp) vs the CPU oracle that
emulates the reference's fp16-autocast numerics, forward and backward."""
import numpy as np
import pytest
import torch

from oracle import deform as od

pytestmark = pytest.mark.gpu
AABB = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])


def _field(seed=0, head_scale=2e3):
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    torch.manual_seed(seed)
    df = SE3DeformationField(AABB.clone(), SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6,
                                                                     mlp_layer_width=128))
    with torch.no_grad():           # the reference initialises the heads ~0 (identity); make the SE(3) part non-trivial
        df.se3_field.mlp_r.layers[-1].weight.mul_(head_scale)
        df.se3_field.mlp_v.layers[-1].weight.mul_(head_scale)
        df.se3_field.mlp_r.layers[-1].bias.add_(0.01)
    return df


@pytest.mark.parametrize("S", [1, 31, 32, 33, 1000])
@pytest.mark.parametrize("window", [None, 0.0, 2.75, 7.0])
def test_deform_forward_per_sample_codes(S, window, cuda):
    df = _field()
    g = torch.Generator().manual_seed(S)
    pos = torch.rand(S, 3, generator=g) * (AABB[1] - AABB[0]) + AABB[0]
    codes = torch.randn(S, 128, generator=g) * 0.3
    want = od.compute_offsets(pos, codes, df.flat_params().detach(), AABB, window, half=True).float()
    dfc = df.to(cuda)
    with torch.no_grad():
        got = dfc.compute_offsets(pos.to(cuda), codes.to(cuda), window).cpu()
    # fp16 activations: a flipped rounding in a hidden unit perturbs the screw axis by ~1e-3 relative
    tol = 3e-3 * want.abs().max().item() + 2e-5
    assert (got - want).abs().max().item() <= tol, ((got - want).abs().max().item(), tol)


# gradients vs the oracle whose backward rounds dZ to fp16 where the kernel does: what is left are fp16 rounding-boundary
# flips of single dZ elements (a value computed in fp32 here, in fp64 there, landing on different sides of a tie) and the
# order of the fp32 sums
DEFORM_BWD_TOL = 2e-3


def test_deform_forward_code_table_equals_gather(cuda):
    df = _field(1).to(cuda)
    S, T = 2049, 37
    g = torch.Generator().manual_seed(3)
    pos = (torch.rand(S, 3, generator=g) * (AABB[1] - AABB[0]) + AABB[0]).to(cuda)
    table = (torch.randn(T, 128, generator=g) * 0.3).to(cuda)
    slot = torch.randint(0, T, (S,), generator=g).to(cuda)
    with torch.no_grad():
        a = df.compute_offsets(pos, table, 3.5, code_index=slot)
        b = df.compute_offsets(pos, table[slot], 3.5)
    # (round 5: the table route sums the code columns per row first -- nsx_deform_fwd_rows, same products in another fp32
    # order; the per-sample-code operator keeps the general kernel)
    assert (a - b).abs().max().item() <= 3e-3 * b.abs().max().item() + 2e-5


@pytest.mark.parametrize("S,T", [(1, 1), (33, 1), (2049, 24), (5000, 48), (700, 64), (700, 65), (3000, 475)])
@pytest.mark.parametrize("window", [None, 2.75])
def test_deform_forward_through_the_slot_terms(S, T, window, cuda):
    """nsx_deform_fwd_rows (the code columns k >= 48 of the two input layers summed per code row first, 3 of 11 K-steps left
    in the input GEMMs) against nsx_deform_fwd on the same table: the same products in another fp32 order -- equal up to
    the fp16 rounding of a hidden unit.  Tables of <= 64 rows keep the terms in LDS, larger ones (the occupancy update draws
    from every timestep of the dataset) read them from global memory: the same numbers in the same order, so a sample's
    offsets depend on its row's VALUES only -- checked by running the same samples against a compacted table."""
    import ctypes as C
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    df = _field(2).to(cuda)
    g = torch.Generator().manual_seed(S + T)
    pos = (torch.rand(S, 3, generator=g) * (AABB[1] - AABB[0]) + AABB[0]).to(cuda)
    table = (torch.randn(T, 128, generator=g) * 0.3).to(cuda)
    slot = torch.randint(0, T, (S,), generator=g, dtype=torch.int32).to(cuda)
    packed, aabb6, w7 = df.packed_params(), df._aabb6(), F.deform_window7(window)
    want = torch.empty((S, 3), device=cuda)
    check(lib().nsx_deform_fwd(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0), ptr(slot), w7, ptr(want), None,
                               stream()), "nsx_deform_fwd")
    got = torch.full((S, 3), 7.0, device=cuda)
    terms = torch.empty((int(lib().nsx_deform_terms_floats(T)),), device=cuda)
    assert terms.numel() == T * 256
    check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0), ptr(slot), T, w7, ptr(got),
                                    ptr(terms), None, stream()), "nsx_deform_fwd_rows")
    tol = 3e-3 * want.abs().max().item() + 2e-5
    assert (got - want).abs().max().item() <= tol, ((got - want).abs().max().item(), tol)
    # and against the oracle, as the general kernel is held
    ref = od.compute_offsets(pos.cpu(), table.cpu()[slot.cpu().long()], df.flat_params().detach().cpu(), AABB, window,
                             half=True).float()
    assert (got.cpu() - ref).abs().max().item() <= 3e-3 * ref.abs().max().item() + 2e-5
    # the same samples against the table compacted to the rows they use (<= 24 of them -> terms in LDS): bit for bit
    few = min(T, 24)
    slot_few = (slot % few).contiguous()
    sub = torch.randperm(T, generator=g)[:few].to(cuda)             # where the `few` rows sit in the big table
    got_big = torch.empty((S, 3), device=cuda)
    check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0),
                                    ptr(sub[slot_few.long()].to(torch.int32).contiguous()), T, w7, ptr(got_big), ptr(terms),
                                    None, stream()), "nsx_deform_fwd_rows")
    small = table[sub].contiguous()
    terms_small = torch.empty((int(lib().nsx_deform_terms_floats(few)),), device=cuda)
    got_small = torch.empty((S, 3), device=cuda)
    check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(small), small.stride(0), ptr(slot_few), few, w7,
                                    ptr(got_small), ptr(terms_small), None, stream()), "nsx_deform_fwd_rows")
    assert torch.equal(got_big, got_small)
    if T == 1:
        # a one-row table without slots: every sample takes row 0
        got_one = torch.empty((S, 3), device=cuda)
        check(lib().nsx_deform_fwd_rows(ptr(packed), ptr(pos), S, aabb6, ptr(table), table.stride(0), None, 1, w7,
                                        ptr(got_one), ptr(terms), None, stream()), "nsx_deform_fwd_rows")
        assert torch.equal(got_one, got)
        with torch.no_grad():
            assert torch.equal(df.compute_offsets(pos, table, window), got)


def test_deform_rows_entry_rejects_a_missing_slot_array(cuda):
    from nersemble_amd import functional as F
    from nersemble_amd._lib import lib, ptr, stream
    df = _field(2).to(cuda)
    pos = torch.rand(8, 3, device=cuda)
    table = torch.zeros(3, 128, device=cuda)
    out, terms = torch.empty(8, 3, device=cuda), torch.empty(3 * 256, device=cuda)
    rc = lib().nsx_deform_fwd_rows(ptr(df.packed_params()), ptr(pos), 8, df._aabb6(), ptr(table), 128, None, 3, None,
                                   ptr(out), ptr(terms), None, stream())
    assert rc != 0 and b"code_slot" in lib().nsx_last_error()
    rc = lib().nsx_deform_fwd_rows(ptr(df.packed_params()), ptr(pos), 8, df._aabb6(), ptr(table), 128, None, 1, None,
                                   ptr(out), None, None, stream())
    assert rc != 0                                                   # (the terms scratch is required)


def test_deform_nan_fallback_and_identity_init(cuda):
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    torch.manual_seed(0)
    df = SE3DeformationField(AABB.clone(), SE3DeformationFieldConfig(warp_code_dim=128)).to(cuda)
    pos = (torch.rand(100, 3) * (AABB[1] - AABB[0]) + AABB[0]).to(cuda)
    codes = torch.randn(100, 128, device=cuda) * 0.01
    with torch.no_grad():
        off = df.compute_offsets(pos, codes, None)
    assert off.abs().max().item() < 1e-3                 # reference init ~ identity transform
    codes[3, 5] = float("nan")                           # NaN deformation -> original point kept (offset 0)
    with torch.no_grad():
        off = df.compute_offsets(pos, codes, None)
    assert torch.all(off[3] == 0) and torch.isfinite(off).all()


@pytest.mark.parametrize("S,T", [(700, 9), (64, 1), (3000, 70)])
def test_deform_backward(S, T, cuda):
    df = _field(2)
    g = torch.Generator().manual_seed(S)
    pos = torch.rand(S, 3, generator=g) * (AABB[1] - AABB[0]) + AABB[0]
    table = torch.randn(T, 128, generator=g) * 0.3
    slot = torch.randint(0, T, (S,), generator=g)
    goff = torch.randn(S, 3, generator=g)
    # oracle gradients: autograd through the fp16-rounded forward (the ReLU masks are those of the autocast forward the
    # kernel reproduces) with dZ rounded to fp16 at every Linear output on the way back, where the kernel rounds
    # (oracle/deform.py round_grads; the same chain without the roundings is pinned to the reference's autograd)
    flat = df.flat_params().detach().double().requires_grad_(True)
    tab64 = table.double().requires_grad_(True)
    # (code TABLE path: everything that touches the code columns is formed through the slot from fp32 per-slot sums of
    # dZ0 / dZ4 -- the gradient that reaches the codes is never rounded per sample: round_code_grad=False)
    off = od.compute_offsets(pos, tab64[slot], flat, AABB, 2.75, half=True, dtype=torch.float64, round_grads=True,
                             round_code_grad=False)
    off.backward(goff.double())
    dfc = df.to(cuda)
    tabc = table.to(cuda).requires_grad_(True)
    offc = dfc.compute_offsets(pos.to(cuda), tabc, 2.75, code_index=slot.to(cuda))
    offc.backward(goff.to(cuda))
    got = dfc.flat_params()          # same ordering; gradients live on the module parameters
    gflat = torch.cat([p.grad.reshape(-1) for p in _ordered_params(dfc)]).cpu().double()
    lay, _ = od.flat_layout()
    for name, (o, shp) in lay.items():
        n = int(np.prod(shp))
        a, b = gflat[o:o + n], flat.grad[o:o + n]
        denom = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= DEFORM_BWD_TOL * denom, (name, (a - b).abs().max().item() / denom)
    gt, wt = tabc.grad.cpu().double(), tab64.grad
    assert (gt - wt).abs().max().item() <= DEFORM_BWD_TOL * wt.abs().max().item()


def _ordered_params(df):
    L = df.se3_field
    out = []
    for lyr in L.mlp_stem.layers:
        out += [lyr.weight, lyr.bias]
    out += [L.mlp_r.layers[0].weight, L.mlp_r.layers[0].bias, L.mlp_v.layers[0].weight, L.mlp_v.layers[0].bias]
    return out


def test_deform_backward_per_sample_codes(cuda):
    df = _field(3)
    S = 300
    g = torch.Generator().manual_seed(1)
    pos = torch.rand(S, 3, generator=g) * (AABB[1] - AABB[0]) + AABB[0]
    codes = torch.randn(S, 128, generator=g) * 0.3
    goff = torch.randn(S, 3, generator=g)
    c64 = codes.double().requires_grad_(True)
    # (the per-sample code gradient leaves the kernel from its fp32 accumulators: round_code_grad=False)
    off = od.compute_offsets(pos, c64, df.flat_params().detach().double(), AABB, None, half=True, dtype=torch.float64,
                             round_grads=True, round_code_grad=False)
    off.backward(goff.double())
    dfc = df.to(cuda)
    cc = codes.to(cuda).requires_grad_(True)
    dfc.compute_offsets(pos.to(cuda), cc, None).backward(goff.to(cuda))
    assert (cc.grad.cpu().double() - c64.grad).abs().max().item() <= DEFORM_BWD_TOL * c64.grad.abs().max().item()


def test_deform_kernels_are_race_free(cuda):
    """The LDS-DMA pipelines (weight stages in the chain kernels, the operand ring of the weight-gradient kernel) are
    ordered by counted waits and barriers only: a mis-ordered stage would show up as run-to-run differences.  The forward
    and the per-sample code gradient involve no atomics and must be bit-identical across runs; the atomically
    accumulated gradients may differ in summation order only."""
    df = _field(5).to(cuda)
    S = 40_000
    g = torch.Generator(device=cuda).manual_seed(9)
    pos = torch.rand((S, 3), device=cuda, generator=g) * (AABB[1] - AABB[0]).to(cuda) + AABB[0].to(cuda)
    codes = torch.randn((S, 128), device=cuda, generator=g) * 0.3
    goff = torch.randn((S, 3), device=cuda, generator=g)
    table = torch.randn((24, 128), device=cuda, generator=g) * 0.3
    slot = torch.randint(0, 24, (S,), device=cuda, generator=g, dtype=torch.int32)
    ref_off = ref_gc = ref_gw = ref_gt = None
    for it in range(6):
        for p in df.parameters():
            p.grad = None
        c = codes.clone().requires_grad_(True)
        off = df.compute_offsets(pos, c, 2.0)
        off.backward(goff)
        t = table.clone().requires_grad_(True)
        df.compute_offsets(pos, t, 2.0, code_index=slot).backward(goff)
        gw = torch.cat([p.grad.reshape(-1) for p in _ordered_params(df)])
        if it == 0:
            ref_off, ref_gc, ref_gw, ref_gt = off.detach().clone(), c.grad.clone(), gw.clone(), t.grad.clone()
        else:
            assert torch.equal(off.detach(), ref_off)
            assert torch.equal(c.grad, ref_gc)
            assert (gw - ref_gw).abs().max().item() <= 1e-4 * ref_gw.abs().max().item()
            assert (t.grad - ref_gt).abs().max().item() <= 1e-4 * ref_gt.abs().max().item()


def test_pack_from_separate_tensors_equals_pack_of_the_flat_vector(cuda):
    """nsx_deform_pack_tensors (16 nn.Linear tensors where they live) == nsx_deform_pack (their concatenation)."""
    from nersemble_amd import functional as F
    df = _field(5).to(cuda)
    a = F.deform_pack(df.flat_params())
    b = F.deform_pack_tensors(df.ordered_params())
    assert torch.equal(a, b)
