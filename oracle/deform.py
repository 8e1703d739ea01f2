"""CPU restatement of the SE(3) deformation field (reference deformation_field.py:77-166,
windowed_nerf_encoding.py:33-74, util/pytorch3d.py:107-191) with the fp16-autocast numerics the reference
trains with (train_nersemble.py:160 mixed_precision=True): Linear inputs / weights / biases rounded to fp16,
wide accumulation, outputs rounded to fp16; PE, exponential map and warp in fp32/64.  TEST INFRASTRUCTURE ONLY.

The fp32 (no rounding) mode of this file is checked against nersemble_amd's torch mirror, which is itself pinned
by goldens from the reference's own module (tests/test_glue_cpu.py)."""
import numpy as np
import torch

W, PE, CODE = 128, 45, 128
IN = PE + CODE


def flat_layout():
    sizes = [("W0", (W, IN)), ("b0", (W,)), ("W1", (W, W)), ("b1", (W,)), ("W2", (W, W)), ("b2", (W,)),
             ("W3", (W, W)), ("b3", (W,)), ("W4", (W, IN + W)), ("b4", (W,)), ("W5", (W, W)), ("b5", (W,)),
             ("Wr", (3, W)), ("br", (3,)), ("Wv", (3, W)), ("bv", (3,))]
    off, out = 0, {}
    for name, shp in sizes:
        n = int(np.prod(shp))
        out[name] = (off, shp)
        off += n
    return out, off


def unflatten(flat: torch.Tensor):
    lay, total = flat_layout()
    assert flat.numel() == total
    return {k: flat[o:o + int(np.prod(s))].reshape(s) for k, (o, s) in lay.items()}


def _r16(t: torch.Tensor) -> torch.Tensor:
    """fp16 rounding with a straight-through gradient (autocast: the backward of a cast is a cast)."""
    r = t.detach().to(torch.float32).to(torch.float16).to(t.dtype)
    return t + (r - t.detach())


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
    nr = (r * r).sum(-1)
    theta = torch.clamp(nr, min=eps).sqrt()
    a = torch.sin(theta) / theta
    b = (1 - torch.cos(theta)) / theta ** 2
    c = (theta - torch.sin(theta)) / theta ** 3
    cr = torch.linalg.cross
    u1 = cr(r, p); u2 = cr(r, u1); w1 = cr(r, v); w2 = cr(r, w1)
    return p + a[:, None] * u1 + b[:, None] * u2 + v + b[:, None] * w1 + c[:, None] * w2


def compute_offsets(pos_world, codes, flat_params, aabb, windows_param, half=True, dtype=torch.float64):
    """pos_world [S,3], codes [S,128], flat_params [127750] -> offsets [S,3] (normalised space)."""
    P = unflatten(flat_params.to(dtype))
    rnd = _r16 if half else (lambda t: t)
    aabb = aabb.to(dtype)
    pn = (pos_world.to(dtype) - aabb[0]) / (aabb[1] - aabb[0])
    if half:   # the kernel normalises in fp32
        pn = ((pos_world.float() - aabb[0].float()) / (aabb[1] - aabb[0]).float()).to(dtype)
    x0 = rnd(torch.cat([encode(pn, windows_param), codes.to(dtype)], dim=-1))

    def lin(x, Wn, bn, relu=True):
        y = rnd(x @ rnd(P[Wn]).T + rnd(P[bn]))
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
