"""CPU restatement of the SE(3) deformation field (reference deformation_field.py:77-166,
windowed_nerf_encoding.py:33-74, util/pytorch3d.py:107-191).  TEST INFRASTRUCTURE ONLY.

Two numeric modes:
  * ``half=False``: every operation in ``dtype`` (fp64 by default), no rounding.  This mode is PINNED: it is compared
    with the outputs of the reference's own ``SE3DeformationField.compute_offsets`` (fp32, CPU) in
    tests/test_oracle_deform_cpu.py -- the tiny W = 32 / code 8 module of tests/golden/deformation.npz and the
    full-size W = 128 / code 128 module of tests/golden/deformation_full.npz (forward and autograd gradients).
  * ``half=True``: the fp16-autocast numerics the reference trains with (train_nersemble.py:160
    mixed_precision=True): Linear inputs / weights / biases rounded to fp16, wide accumulation, outputs rounded to
    fp16; PE, exponential map and warp in fp32/64.  Same code path, plus the roundings: this is the mode the HIP
    kernels (csrc/deform.hip) are held to.

Width ``W`` and warp-code dimension are parameters (6 layers, skip into layer 4 are the reference's fixed
architecture, deformation_field.py:15-21, :50-69); the flat parameter vector is in include/nsx.h order.
"""
import numpy as np
import torch

PE = 45                      # 3 x 7 frequencies x (sin, cos) + the 2*pi-scaled input (windowed_nerf_encoding.py:48,72)
W, CODE = 128, 128           # the training configuration (train_nersemble.py:84-91)
IN = PE + CODE


def flat_layout(width: int = W, code_dim: int = CODE):
    n_in = PE + code_dim
    sizes = [("W0", (width, n_in)), ("b0", (width,)), ("W1", (width, width)), ("b1", (width,)),
             ("W2", (width, width)), ("b2", (width,)), ("W3", (width, width)), ("b3", (width,)),
             ("W4", (width, n_in + width)), ("b4", (width,)), ("W5", (width, width)), ("b5", (width,)),
             ("Wr", (3, width)), ("br", (3,)), ("Wv", (3, width)), ("bv", (3,))]
    off, out = 0, {}
    for name, shp in sizes:
        n = int(np.prod(shp))
        out[name] = (off, shp)
        off += n
    return out, off


STATE_DICT_ORDER = [f"se3_field.mlp_stem.layers.{i}.{k}" for i in range(6) for k in ("weight", "bias")] + \
    ["se3_field.mlp_r.layers.0.weight", "se3_field.mlp_r.layers.0.bias",
     "se3_field.mlp_v.layers.0.weight", "se3_field.mlp_v.layers.0.bias"]


def flat_from_state_dict(sd) -> torch.Tensor:
    """The reference module's state dict (deformation_field.py:50-69 names) -> flat vector in include/nsx.h order."""
    return torch.cat([torch.as_tensor(sd[k]).reshape(-1).to(torch.float64) for k in STATE_DICT_ORDER])


def unflatten(flat: torch.Tensor, width: int = W, code_dim: int = CODE):
    lay, total = flat_layout(width, code_dim)
    assert flat.numel() == total, (flat.numel(), total)
    return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in lay.items()}


def _r16(t: torch.Tensor) -> torch.Tensor:
    """fp16 rounding with a straight-through gradient (autocast: the backward of a cast is a cast)."""
    r = t.detach().to(torch.float32).to(torch.float16).to(t.dtype)
    return t + (r - t.detach())


class _Round16Both(torch.autograd.Function):
    """fp16 rounding of the value in the forward AND of the gradient in the backward: what an fp16 tensor inside an
    autocast region is -- the gradient of a Linear's fp16 output / input is itself an fp16 tensor (torch computes
    grad_input = grad_output @ W as an fp16 GEMM), which is where csrc/deform.hip rounds its dZ tiles."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.float32).to(torch.float16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.float32).to(torch.float16).to(g.dtype)


def window_weights(windows_param, n=7):
    if windows_param is None:
        return np.ones(n, np.float32)
    bands = np.linspace(0.0, n - 1, n, dtype=np.float32)
    x = np.clip(np.float32(windows_param) - bands, 0, 1)
    return (0.5 * (1 - np.cos(np.float32(np.pi) * x))).astype(np.float32)


def encode(pn: torch.Tensor, windows_param) -> torch.Tensor:
    x = 2 * torch.pi * pn
    freqs = 2.0 ** torch.arange(7, dtype=pn.dtype)
    scaled = (x[..., None] * freqs).reshape(pn.shape[0], -1)
    enc = torch.sin(torch.cat([scaled, scaled + torch.pi / 2.0], dim=-1))
    w = torch.from_numpy(window_weights(windows_param)).to(pn.dtype)
    enc = enc * w[None, :].repeat(3, 1).reshape(-1).repeat(2)
    return torch.cat([enc, x], dim=-1)


def se3_warp(r, v, p, eps=1e-4):
    """exp([v, r]) applied to p in closed form: R p + V v with the Rodrigues coefficients of util/pytorch3d.py:10-39
    (_so3_exp_map: theta = sqrt(clamp(|r|^2, eps))) and :78-105 (_se3_V_matrix); equal to the reference's 4x4
    homogeneous product (deformation_field.py:93-100; its w component is exactly 1)."""
    nr = (r * r).sum(-1)
    theta = torch.clamp(nr, min=eps).sqrt()
    a = torch.sin(theta) / theta
    b = (1 - torch.cos(theta)) / theta ** 2
    c = (theta - torch.sin(theta)) / theta ** 3
    cr = torch.linalg.cross
    u1 = cr(r, p); u2 = cr(r, u1); w1 = cr(r, v); w2 = cr(r, w1)
    return p + a[:, None] * u1 + b[:, None] * u2 + v + b[:, None] * w1 + c[:, None] * w2


def compute_offsets(pos_world, codes, flat_params, aabb, windows_param, half=True, dtype=torch.float64,
                    width: int = None, round_grads: bool = False, round_code_grad: bool = None):
    """pos_world [S,3], codes [S,code_dim], flat_params (include/nsx.h order) -> offsets [S,3] (normalised space).
    ``width`` defaults to 128 unless the parameter count says otherwise (solved from the layout).

    ``round_grads`` (with ``half``): the BACKWARD follows the autocast numerics too -- the gradient of every Linear
    output (dZ, heads included) is rounded to fp16 before it meets the weights / the layer input, as torch's fp16
    GEMMs and csrc/deform.hip do; weight / bias gradients stay wide (the kernel accumulates them in fp32, the
    reference rounds them once more).  ``round_code_grad`` (default: as ``round_grads``): also the gradient that
    reaches the warp codes through the first cast -- the kernel's code-TABLE gradient sums fp16 dC tiles, its
    per-sample code gradient is written from the fp32 accumulators (pass False for that output)."""
    code_dim = codes.shape[1]
    if width is None:
        width = next(w for w in (128, 32, 64, 16, 256) if flat_layout(w, code_dim)[1] == flat_params.numel())
    P = unflatten(flat_params.to(dtype), width, code_dim)
    rnd = _r16 if half else (lambda t: t)                   # parameters: straight-through
    both = half and round_grads
    rnd_act = _Round16Both.apply if both else rnd           # Linear outputs
    if round_code_grad is None:
        round_code_grad = round_grads
    rnd_in = _Round16Both.apply if (half and round_code_grad) else rnd
    aabb = aabb.to(dtype)
    pn = (pos_world.to(dtype) - aabb[0]) / (aabb[1] - aabb[0])
    if half:   # the kernel normalises in fp32
        pn = ((pos_world.float() - aabb[0].float()) / (aabb[1] - aabb[0]).float()).to(dtype)
    x0 = rnd_in(torch.cat([encode(pn, windows_param), codes.to(dtype)], dim=-1))

    def lin(x, Wn, bn, relu=True):
        y = rnd_act(x @ rnd(P[Wn]).T + rnd(P[bn]))
        return torch.relu(y) if relu else y

    h = lin(x0, "W0", "b0")
    h = lin(h, "W1", "b1"); h = lin(h, "W2", "b2"); h = lin(h, "W3", "b3")
    h = lin(torch.cat([x0, h], dim=-1), "W4", "b4")
    h = lin(h, "W5", "b5")
    r = lin(h, "Wr", "br", relu=False)
    v = lin(h, "Wv", "bv", relu=False)
    warped = se3_warp(r, v, pn)
    warped = torch.where(torch.isnan(warped), pn, warped)
    return warped - pn


TERM_K0 = 48            # first input column the slot terms cover (csrc/deform.hip: TERM_K0)


def compute_offsets_slot_terms(pos_world, table, slot, flat_params, aabb, windows_param, half=True, dtype=torch.float64):
    """Restatement of ``nsx_deform_fwd_rows`` (csrc/deform.hip: deform_code_terms_kernel + deform_fwd_terms_kernel): every
    sample's code is row ``slot[s]`` of ``table`` [T, code_dim], and the code columns k >= 48 of the two layers that read
    the 173-wide input (W0, and W4 over ``cat[input, x]``, deformation_field.py:77-131) are summed per table row first,

        T_l[row] = W_l[:, 48:173] @ code16[row][3:]        (l = 0, 4; fp16 operands)

    then added to the bias while the GEMM keeps the columns k < 48 (45 positional-encoding columns + the first 3 code
    columns).  The same products as ``compute_offsets`` in another order: equal to it up to the summation order of the
    pre-activations (in float64: to ~1e-12 before the fp16 roundings that both apply at the same places)."""
    code_dim = table.shape[1]
    width = next(w for w in (128, 32, 64, 16, 256) if flat_layout(w, code_dim)[1] == flat_params.numel())
    P = unflatten(flat_params.to(dtype), width, code_dim)
    rnd = _r16 if half else (lambda t: t)
    aabb = aabb.to(dtype)
    pn = (pos_world.to(dtype) - aabb[0]) / (aabb[1] - aabb[0])
    if half:
        pn = ((pos_world.float() - aabb[0].float()) / (aabb[1] - aabb[0]).float()).to(dtype)
    slot = slot.long()
    code16 = rnd(table.to(dtype))
    x_head = rnd(torch.cat([encode(pn, windows_param), table.to(dtype)[slot][:, :TERM_K0 - 45]], dim=-1))      # [S, 48]
    n_in = 45 + code_dim
    T0 = code16[:, TERM_K0 - 45:] @ rnd(P["W0"])[:, TERM_K0:n_in].T                                            # [T, width]
    T4 = code16[:, TERM_K0 - 45:] @ rnd(P["W4"])[:, TERM_K0:n_in].T

    def act(y, relu=True):
        y = rnd(y)
        return torch.relu(y) if relu else y

    h = act(x_head @ rnd(P["W0"])[:, :TERM_K0].T + rnd(P["b0"]) + T0[slot])
    for W, b in (("W1", "b1"), ("W2", "b2"), ("W3", "b3")):
        h = act(h @ rnd(P[W]).T + rnd(P[b]))
    W4 = rnd(P["W4"])
    h = act(x_head @ W4[:, :TERM_K0].T + h @ W4[:, n_in:].T + rnd(P["b4"]) + T4[slot])
    h = act(h @ rnd(P["W5"]).T + rnd(P["b5"]))
    r = act(h @ rnd(P["Wr"]).T + rnd(P["br"]), relu=False)
    v = act(h @ rnd(P["Wv"]).T + rnd(P["bv"]), relu=False)
    warped = se3_warp(r, v, pn)
    warped = torch.where(torch.isnan(warped), pn, warped)
    return warped - pn
