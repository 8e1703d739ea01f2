"""numpy restatement of tcnn FullyFusedMLP as used by the reference field
(nersemble_nerfacto_field.py:142-172, calls :285 / :377).  TEST INFRASTRUCTURE ONLY.

tcnn is un-vendored ("parity unpinned"): restated from its published design: fp16 weights,
no biases, row-major [out][in] matrices W0 [64][32] | Wh [64][64]* | Wo [16][64] in one flat vector, input
zero-padded to 32, output padded to 16, ReLU hidden activations stored in fp16, output None/Sigmoid.
Accumulation is carried in float64 here and rounded once per layer to fp16 (the spec for the HIP kernels).
"""
import numpy as np

W, IN, OUT = 64, 32, 16


def param_count(n_hidden_mats: int) -> int:
    return W * IN + n_hidden_mats * W * W + OUT * W


def split_params(params: np.ndarray, n_hidden_mats: int):
    p = np.asarray(params, dtype=np.float32).astype(np.float16).astype(np.float64)
    w0 = p[:W * IN].reshape(W, IN)
    off = W * IN
    wh = []
    for _ in range(n_hidden_mats):
        wh.append(p[off:off + W * W].reshape(W, W))
        off += W * W
    wo = p[off:off + OUT * W].reshape(OUT, W)
    return w0, wh, wo


def _r16(a):
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float16).astype(np.float64)


def pad_input(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x)
    out = np.zeros((x.shape[0], IN), dtype=np.float64)
    out[:, :x.shape[1]] = _r16(x)
    return out


def mlp_fwd(x, params, n_hidden_mats, n_out, out_act, return_acts=False):
    """x [B, in_dim<=32] (rounded to fp16) -> [B, n_out] np.float16."""
    w0, wh, wo = split_params(params, n_hidden_mats)
    h = pad_input(x)
    acts = [h]
    h = _r16(np.maximum(h @ w0.T, 0))
    acts.append(h)
    for m in wh:
        h = _r16(np.maximum(h @ m.T, 0))
        acts.append(h)
    z = h @ wo.T
    y = 1.0 / (1.0 + np.exp(-z)) if out_act == 1 else z
    out = y[:, :n_out].astype(np.float32).astype(np.float16)
    if return_acts:
        return out, acts, y
    return out


def mlp_bwd(x, params, n_hidden_mats, n_out, out_act, dout, round_dz=False):
    """Returns (dparams fp64 flat, dx fp64 [B, 32]) for upstream dout [B, n_out] (float).

    ``round_dz``: the gradient of every layer's pre-activation (dZ) is rounded to fp16 before it is used -- in the
    weight-gradient product AND in the product that carries it to the layer below -- which is what tcnn's
    FullyFusedMLP backward does (its dL/doutput and all intermediate gradients are __half matrices) and where
    csrc/mlp.hip rounds (the MFMA operands are fp16; accumulation is wide).  Without it the chain is exact float64."""
    w0, wh, wo = split_params(params, n_hidden_mats)
    _, acts, y = mlp_fwd(x, params, n_hidden_mats, n_out, out_act, return_acts=True)
    rz = _r16 if round_dz else (lambda a: a)
    B = acts[0].shape[0]
    dz = np.zeros((B, OUT))
    dz[:, :n_out] = np.asarray(dout, dtype=np.float64)
    if out_act == 1:
        dz = dz * y * (1 - y)
    dz = rz(dz)
    grads = []
    h = acts[-1]
    grads.append(dz.T @ h)                 # dWo
    dh = dz @ wo
    for li in range(len(wh) - 1, -1, -1):
        dzh = rz(dh * (acts[2 + li] > 0))
        grads.append(dzh.T @ acts[1 + li])
        dh = dzh @ wh[li]
    dz0 = rz(dh * (acts[1] > 0))
    grads.append(dz0.T @ acts[0])
    dx = dz0 @ w0
    grads = grads[::-1]                    # W0, Wh..., Wo
    return np.concatenate([g.reshape(-1) for g in grads]), dx
