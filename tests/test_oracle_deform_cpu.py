"""CPU: oracle/deform.py (the checker of csrc/deform.hip) against outputs of the REFERENCE's own
``SE3DeformationField`` (deformation_field.py:119-166, run in fp32 on CPU by tests/golden/make_golden.py).

This closes the chain  HIP kernel <-> oracle/deform.py (tests/test_deform_gpu.py)  <->  reference module (here):
  * the tiny W = 32 / code 8 module of deformation.npz (weights in the fixture),
  * the training-size W = 128 / code 128 module of deformation_full.npz (weights regenerated from the seed with
    tests/helpers.make_deform_state_dict), forward for four window values and the autograd gradients w.r.t. the warp
    codes and all 127 750 parameters.
The oracle runs in its un-rounded mode (half=False; the fp16 mode adds roundings to the same code path)."""
import numpy as np
import pytest
import torch

from oracle import deform as od
from tests.helpers import DEFORM_KEYS, make_deform_state_dict

AABB = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])


def _windows(arr):
    return [None if np.isnan(w) else float(w) for w in arr]


def test_oracle_offsets_match_reference_tiny_module(golden_dir):
    z = np.load(f"{golden_dir}/deformation.npz")
    sd = {k[len("df_sd_"):]: z[k] for k in z.files if k.startswith("df_sd_")}
    flat = od.flat_from_state_dict(sd)
    assert flat.numel() == od.flat_layout(32, 8)[1]
    for i, w in enumerate(_windows(z["df_windows"])):
        got = od.compute_offsets(torch.from_numpy(z["df_pos"]), torch.from_numpy(z["df_code"]), flat,
                                 torch.from_numpy(z["df_sd_aabb"]), w, half=False).numpy()
        assert np.abs(got - z[f"df_off_{i}"]).max() <= 5e-6, (i, np.abs(got - z[f"df_off_{i}"]).max())


def test_oracle_offsets_match_reference_training_size(golden_dir):
    z = np.load(f"{golden_dir}/deformation_full.npz")
    flat = od.flat_from_state_dict(make_deform_state_dict(int(z["dff_seed"][0])))
    assert flat.numel() == od.flat_layout()[1] == 127750
    pos, code = torch.from_numpy(z["dff_pos"]), torch.from_numpy(z["dff_code"])
    for i, w in enumerate(_windows(z["dff_windows"])):
        got = od.compute_offsets(pos, code, flat, AABB, w, half=False).numpy()
        want = z[f"dff_off_{i}"]
        assert np.abs(want).max() > 1e-2                         # the SE(3) part is exercised, not the identity
        assert np.abs(got - want).max() <= 5e-6, (i, np.abs(got - want).max())


def test_oracle_gradients_match_reference_autograd(golden_dir):
    z = np.load(f"{golden_dir}/deformation_full.npz")
    flat = od.flat_from_state_dict(make_deform_state_dict(int(z["dff_seed"][0]))).requires_grad_(True)
    code = torch.from_numpy(z["dff_code"]).double().requires_grad_(True)
    off = od.compute_offsets(torch.from_numpy(z["dff_pos"]), code, flat, AABB, 2.75, half=False)
    (off * torch.from_numpy(z["dff_gw"]).double()).sum().backward()
    gc, gp = code.grad.numpy(), flat.grad.numpy()
    assert np.abs(gc - z["dff_gcode"]).max() <= 2e-5 * np.abs(z["dff_gcode"]).max()
    assert np.abs(gp - z["dff_gparams"]).max() <= 2e-5 * np.abs(z["dff_gparams"]).max()
    # every parameter tensor receives a gradient (the skip connection feeds W4 from both halves)
    lay, _ = od.flat_layout()
    for name, (o, shp) in lay.items():
        assert np.abs(z["dff_gparams"][o:o + int(np.prod(shp))]).max() > 0, name


def test_mirror_module_matches_reference_training_size(golden_dir):
    """The product's torch mirror (CPU branch of SE3DeformationField.compute_offsets) on the same fixture."""
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    z = np.load(f"{golden_dir}/deformation_full.npz")
    df = SE3DeformationField(AABB.clone(), SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6,
                                                                      mlp_layer_width=128), max_n_samples_per_batch=29)
    sd = {k: torch.from_numpy(v) for k, v in make_deform_state_dict(int(z["dff_seed"][0])).items()}
    sd["aabb"] = AABB
    df.load_state_dict(sd)
    assert [k for k in DEFORM_KEYS] == [k for k in df.state_dict() if k != "aabb"]
    for i, w in enumerate(_windows(z["dff_windows"])):
        with torch.no_grad():
            got = df.compute_offsets(torch.from_numpy(z["dff_pos"]), torch.from_numpy(z["dff_code"]), w).numpy()
        assert np.abs(got - z[f"dff_off_{i}"]).max() <= 5e-6, i


def test_half_mode_stays_close_to_the_pinned_mode(golden_dir):
    """The fp16-autocast mode is the pinned code path plus roundings: on the training-size fixture it stays within
    the fp16 noise of a 6-layer chain (what the kernels are then held to, tests/test_deform_gpu.py)."""
    z = np.load(f"{golden_dir}/deformation_full.npz")
    flat = od.flat_from_state_dict(make_deform_state_dict(int(z["dff_seed"][0])))
    pos, code = torch.from_numpy(z["dff_pos"]), torch.from_numpy(z["dff_code"])
    exact = od.compute_offsets(pos, code, flat, AABB, 2.75, half=False).numpy()
    half = od.compute_offsets(pos, code, flat, AABB, 2.75, half=True).numpy()
    assert np.abs(half - exact).max() <= 3e-3 * max(np.abs(exact).max(), 1.0)


def test_rounded_backward_mode_is_the_pinned_backward_plus_fp16_roundings(golden_dir):
    """``round_grads``: the gradient of every Linear output is rounded to fp16 on its way back (autocast's fp16 GEMM
    outputs; where csrc/deform.hip rounds).  Same code path as the mode that is pinned against the reference's autograd
    above, so it must stay within fp16 noise of it -- and it must actually differ (the flag does something), with
    fp16-representable per-sample code gradients when ``round_code_grad`` is on."""
    z = np.load(f"{golden_dir}/deformation_full.npz")
    base = od.flat_from_state_dict(make_deform_state_dict(int(z["dff_seed"][0])))
    pos, gw = torch.from_numpy(z["dff_pos"]), torch.from_numpy(z["dff_gw"]).double()

    def grads(**kw):
        flat = base.clone().requires_grad_(True)
        code = torch.from_numpy(z["dff_code"]).double().requires_grad_(True)
        off = od.compute_offsets(pos, code, flat, AABB, 2.75, half=True, **kw)
        (off * gw).sum().backward()
        return code.grad.numpy(), flat.grad.numpy()

    gc0, gp0 = grads()
    gc1, gp1 = grads(round_grads=True)
    gc2, _ = grads(round_grads=True, round_code_grad=False)
    assert 0 < np.abs(gp1 - gp0).max() <= 5e-3 * np.abs(gp0).max()
    assert 0 < np.abs(gc1 - gc0).max() <= 5e-3 * np.abs(gc0).max()
    assert np.array_equal(gc1, gc1.astype(np.float16).astype(np.float64))
    assert not np.array_equal(gc2, gc2.astype(np.float16).astype(np.float64))
    # and the rounded chain is still the reference's chain: held to the golden autograd within the fp16 noise
    assert np.abs(gp1 - z["dff_gparams"]).max() <= 3e-2 * np.abs(z["dff_gparams"]).max()


def test_mlp_oracle_rounded_backward():
    """oracle/mlp.py ``round_dz``: dZ rounded to fp16 per layer (tcnn's __half gradient matrices, csrc/mlp.hip's MFMA
    operands) -- within fp16 noise of the exact chain, not equal to it."""
    from oracle import mlp as omlp
    rng = np.random.default_rng(0)
    nh, B = 1, 200
    p = (rng.standard_normal(omlp.param_count(nh)) * 0.2).astype(np.float32)
    x = rng.standard_normal((B, 32)).astype(np.float16)
    dout = rng.standard_normal((B, 16))
    dW0, dx0 = omlp.mlp_bwd(x, p, nh, 16, 0, dout)
    dW1, dx1 = omlp.mlp_bwd(x, p, nh, 16, 0, dout, round_dz=True)
    assert 0 < np.abs(dW1 - dW0).max() <= 2e-3 * np.abs(dW0).max()
    assert 0 < np.abs(dx1 - dx0).max() <= 2e-3 * np.abs(dx0).max()


@pytest.mark.parametrize("window", [None, 2.75])
def test_slot_terms_restatement_equals_the_direct_forward(window):
    """The forward with the code columns k >= 48 of the two input layers summed per code row first (what
    nsx_deform_fwd_rows computes) is the direct forward: in float64 without roundings to 1e-12; with the fp16 roundings of
    both routes at the same places, equal up to the rare hidden unit whose pre-activation sits on a rounding boundary."""
    torch.manual_seed(5)
    S, T = 700, 9
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    n = od.flat_layout(128, 128)[1]
    flat = torch.randn(n, dtype=torch.float64) * 0.08
    pos = torch.rand(S, 3, dtype=torch.float64) * (aabb[1] - aabb[0]).double() + aabb[0].double()
    table = torch.randn(T, 128, dtype=torch.float64) * 0.3
    slot = torch.randint(0, T, (S,))
    for half in (False, True):
        a = od.compute_offsets(pos, table[slot], flat, aabb, window, half=half)
        b = od.compute_offsets_slot_terms(pos, table, slot, flat, aabb, window, half=half)
        d = (a - b).abs()
        if not half:
            assert d.max().item() <= 1e-11, d.max().item()
        else:
            assert d.max().item() <= 3e-3 * a.abs().max().item() + 2e-5
            assert (d <= 1e-9).float().mean().item() >= 0.9
