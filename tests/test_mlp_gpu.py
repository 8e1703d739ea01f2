"""GPU parity: fused MFMA MLP kernels (mlp_base / mlp_head shapes) vs the numpy oracle."""
import numpy as np
import pytest
import torch

from oracle import mlp as omlp

pytestmark = pytest.mark.gpu


def _params(nh, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    n = omlp.param_count(nh)
    # asymmetric, transpose-detecting weights
    p = (rng.standard_normal(n) * 0.25 * scale).astype(np.float32)
    return p.astype(np.float16).astype(np.float32)


@pytest.mark.parametrize("nh,in_dim,n_out,act", [(0, 32, 16, 0), (1, 32, 3, 1), (1, 18, 3, 1), (0, 7, 5, 0), (1, 32, 16, 0)])
@pytest.mark.parametrize("B", [1, 31, 32, 33, 1000])
def test_mlp_forward(nh, in_dim, n_out, act, B, cuda):
    from nersemble_amd import functional as F
    p = _params(nh, 10 * nh + in_dim)
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)
    want = omlp.mlp_fwd(x, p, nh, n_out, act).astype(np.float32)
    out = F.fused_mlp(torch.from_numpy(p).to(cuda), nh, n_out, act, b=torch.from_numpy(x).to(cuda))
    got = out.float().cpu().numpy()
    assert got.shape == (B, n_out)
    # fp32 MFMA accumulation + fp16 rounding of hidden activations: a flipped rounding in a hidden unit moves
    # an output by ~2^-11 * |w| -> a few fp16 ulps of the output scale
    tol = 4 * 2.0 ** -10 * max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() <= tol, float(np.abs(got - want).max())


def test_mlp_head_two_segment_input(cuda):
    """mlp_head reads (dir+1)/2 (fp32) and 15 geo features out of mlp_base's [B,16] output (cols 1..15)."""
    from nersemble_amd import functional as F
    B = 517
    p = _params(1, 5)
    rng = np.random.default_rng(1)
    dirs = rng.standard_normal((B, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    base = rng.standard_normal((B, 16)).astype(np.float16)
    x = np.concatenate([((dirs + 1) / 2).astype(np.float16), base[:, 1:]], axis=1)
    want = omlp.mlp_fwd(x, p, 1, 3, 1).astype(np.float32)
    out = F.fused_mlp(torch.from_numpy(p).to(cuda), 1, 3, 1, a=torch.from_numpy(dirs).to(cuda), a_mul=0.5, a_add=0.5,
                      b=torch.from_numpy(base).to(cuda), b_off=1, b_dim=15)
    assert np.abs(out.float().cpu().numpy() - want).max() <= 4 * 2.0 ** -10


@pytest.mark.parametrize("nh,in_dim,n_out,act", [(0, 32, 16, 0), (1, 18, 3, 1), (1, 32, 16, 0)])
@pytest.mark.parametrize("B", [5, 64, 777])
def test_mlp_backward(nh, in_dim, n_out, act, B, cuda):
    from nersemble_amd import functional as F
    p = _params(nh, 3 + nh)
    rng = np.random.default_rng(B + 1)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)
    dout = rng.standard_normal((B, n_out)).astype(np.float16)
    dW_o, dx_o = omlp.mlp_bwd(x, p, nh, n_out, act, dout.astype(np.float64), round_dz=True)
    pt = torch.from_numpy(p).to(cuda).requires_grad_(True)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    out = F.fused_mlp(pt, nh, n_out, act, b=xt)
    out.backward(torch.from_numpy(dout).to(cuda))
    dW = pt.grad.cpu().numpy()
    dx = xt.grad.float().cpu().numpy()
    # HIP rounds dZ to fp16 between layers (like tcnn) and so does the oracle (round_dz): what is left is the order of
    # the fp32 sums and single dZ elements landing on the other side of an fp16 tie; the input gradient leaves as fp16
    assert np.abs(dW - dW_o).max() <= 5e-4 * np.abs(dW_o).max() + 1e-6, float(np.abs(dW - dW_o).max() / np.abs(dW_o).max())
    d = np.abs(dx - dx_o[:, :in_dim])
    assert (d <= 2.0 ** -10 * np.abs(dx_o[:, :in_dim]) + 5e-4 * np.abs(dx_o).max() + 1e-6).all(), float(d.max() / np.abs(dx_o).max())


def test_mlp_head_backward_segments(cuda):
    from nersemble_amd import functional as F
    B = 300
    p = _params(1, 9)
    rng = np.random.default_rng(2)
    dirs = rng.standard_normal((B, 3)).astype(np.float32)
    base = rng.standard_normal((B, 16)).astype(np.float16)
    dout = rng.standard_normal((B, 3)).astype(np.float16)
    x = np.concatenate([((dirs + 1) / 2).astype(np.float16), base[:, 1:]], axis=1)
    dW_o, dx_o = omlp.mlp_bwd(x, p, 1, 3, 1, dout.astype(np.float64), round_dz=True)
    pt = torch.from_numpy(p).to(cuda).requires_grad_(True)
    bt = torch.from_numpy(base).to(cuda).requires_grad_(True)
    dt = torch.from_numpy(dirs).to(cuda).requires_grad_(True)
    out = F.fused_mlp(pt, 1, 3, 1, a=dt, a_mul=0.5, a_add=0.5, b=bt, b_off=1, b_dim=15)
    out.backward(torch.from_numpy(dout).to(cuda))
    sc = np.abs(dx_o).max()
    db = bt.grad.float().cpu().numpy()
    assert np.all(db[:, 0] == 0)
    assert (np.abs(db[:, 1:] - dx_o[:, 3:18]) <= 2.0 ** -10 * np.abs(dx_o[:, 3:18]) + 5e-4 * sc).all()      # fp16 output
    assert np.abs(dt.grad.cpu().numpy() - 0.5 * dx_o[:, :3]).max() <= 5e-4 * sc
    assert np.abs(pt.grad.cpu().numpy() - dW_o).max() <= 5e-4 * np.abs(dW_o).max()


# ---- through the operator-level drop-in: tcnn.Network / tcnn.NetworkWithInputEncoding (flat [out][in] ``params``) ----
# Reference call sites: mlp_base = tcnn.NetworkWithInputEncoding(32 -> 16, Identity encoding, 1 hidden layer, no output
# activation; nersemble_nerfacto_field.py:142-153), mlp_head = tcnn.Network(3 + 15 -> 3, 2 hidden layers, Sigmoid;
# :162-172).  The module is driven the way the reference drives it: ``module(x)`` on fp16 / fp32 inputs, gradients read from
# ``module.params.grad`` and ``x.grad``.
def _module(kind, cuda):
    from nersemble_amd import tcnn
    if kind == "base":
        net = tcnn.NetworkWithInputEncoding(
            n_input_dims=32, n_output_dims=16, encoding_config={"otype": "Identity", "n_dims_to_encode": 32},
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64,
                            "n_hidden_layers": 1})
        return net.to(cuda), 0, 32, 16, 0
    net = tcnn.Network(n_input_dims=18, n_output_dims=3,
                       network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                                       "n_neurons": 64, "n_hidden_layers": 2})
    return net.to(cuda), 1, 18, 3, 1


@pytest.mark.parametrize("kind", ["base", "head"])
@pytest.mark.parametrize("in_dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("B", [1, 96, 1001])
def test_tcnn_network_module_forward_and_backward(kind, in_dtype, B, cuda):
    net, nh, in_dim, n_out, act = _module(kind, cuda)
    assert net.params.dtype == torch.float32 and net.params.dim() == 1 and net.params.numel() == omlp.param_count(nh)
    assert net.n_output_dims == n_out
    p = _params(nh, 77 + nh)
    with torch.no_grad():
        net.params.copy_(torch.from_numpy(p))
    rng = np.random.default_rng(1000 * nh + B)
    x = rng.standard_normal((B, in_dim)).astype(np.float16)           # (fp16-representable either way)
    dout = rng.standard_normal((B, n_out)).astype(np.float16)
    xt = torch.from_numpy(x).to(cuda).to(in_dtype).requires_grad_(True)
    out = net(xt)
    assert out.dtype == torch.float16 and out.shape == (B, n_out)
    want = omlp.mlp_fwd(x, p, nh, n_out, act).astype(np.float32)
    tol = 4 * 2.0 ** -10 * max(1.0, np.abs(want).max())
    assert np.abs(out.float().detach().cpu().numpy() - want).max() <= tol
    out.backward(torch.from_numpy(dout).to(cuda))
    dW_o, dx_o = omlp.mlp_bwd(x, p, nh, n_out, act, dout.astype(np.float64), round_dz=True)
    dW = net.params.grad.cpu().numpy()
    assert dW.shape == dW_o.shape
    assert np.abs(dW - dW_o).max() <= 5e-4 * np.abs(dW_o).max() + 1e-6, float(np.abs(dW - dW_o).max() / np.abs(dW_o).max())
    assert xt.grad.dtype == in_dtype
    dx = xt.grad.float().cpu().numpy()
    d = np.abs(dx - dx_o[:, :in_dim])
    assert (d <= 2.0 ** -10 * np.abs(dx_o[:, :in_dim]) + 5e-4 * np.abs(dx_o).max() + 1e-6).all(), \
        float(d.max() / np.abs(dx_o).max())


def test_tcnn_network_params_are_row_major_out_in(cuda):
    """A single non-zero weight W0[n][i] (flat index n * 32 + i) connects input i to hidden unit n, and Wo[o][n] (flat
    index 64 * 32 + o * 64 + n) connects hidden unit n to output o -- tcnn's flat [out][in] layout of the state dict's
    ``params``; a transposed reading would route the signal elsewhere."""
    net, nh, in_dim, n_out, act = _module("base", cuda)
    n, i, o = 37, 5, 11
    p = np.zeros(omlp.param_count(0), dtype=np.float32)
    p[n * 32 + i] = 0.5
    p[64 * 32 + o * 64 + n] = 2.0
    with torch.no_grad():
        net.params.copy_(torch.from_numpy(p))
    x = np.zeros((4, 32), dtype=np.float16)
    x[:, i] = [1.0, 2.0, -1.0, 0.25]
    out = net(torch.from_numpy(x).to(cuda)).detach().float().cpu().numpy()
    want = np.zeros((4, 16), dtype=np.float32)
    want[:, o] = [1.0, 2.0, 0.0, 0.25]                               # relu(0.5 x) * 2
    assert np.array_equal(out, want)
