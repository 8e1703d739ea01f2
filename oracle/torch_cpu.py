"""PyTorch-CPU restatement of the encoder path (HashEnsemble forward + mlp_base) -- the CPU baseline SURVEY.md 8(d)
defines: "the build's own pure-PyTorch CPU restatement of HashEnsemble (+MLPs ...) timed on the GPU box's host with
torch.set_num_threads(all cores)".  The reference itself has no CPU encoder (tinycudann is CUDA-only), so this is what
its ``HashEnsemble.forward`` (hash_ensemble.py:93-158: C tcnn HashGrid encodings -> stack -> rearrange -> window ->
blend einsum) would cost with a torch-native hash-grid encoder: batched index arithmetic, ``index_select`` gathers,
trilinear weights and one einsum -- no Python loop over samples.

TEST / BENCH INFRASTRUCTURE ONLY (bench.py ``cpu_baseline``; tests/test_oracle_hash.py holds it to the C oracle)."""
import numpy as np
import torch

PRIMES = (1, 2654435761, 805459861)


def grid_levels(geom):
    return [(float(geom.scale[l]), int(geom.res[l]), int(geom.size[l]), int(geom.offset[l])) for l in range(geom.n_levels)]


def _dense(size: int, res: int) -> bool:
    stride = 1
    for _ in range(3):
        if stride > size:
            break
        stride *= res
    return size >= stride


@torch.no_grad()
def hash_ensemble_forward(x: torch.Tensor, tables: torch.Tensor, code: torch.Tensor, geom, H: int) -> torch.Tensor:
    """x [S,3] fp32 in [0,1); tables [C, total, F_enc] fp16 (tcnn layout); code [S,H] fp32 (already windowed).
    Returns [S, 2L] fp16 like the reference: per-encoding features accumulate in fp32 and round to fp16 (tcnn), the blend
    runs in fp16 inputs / fp32 accumulate."""
    S = x.shape[0]
    C, total, F = tables.shape
    P = F // 2
    code16 = code.to(torch.float16).to(torch.float32)
    outs = []
    m32 = 0xFFFFFFFF
    for scale, res, size, off in grid_levels(geom):
        pos = x * scale + 0.5                                   # tcnn: fma(scale, x, 0.5)
        g = torch.floor(pos)
        w = pos - g
        g = g.to(torch.int64)
        acc = torch.zeros((S, C, F), dtype=torch.float32)
        dense = _dense(size, res)
        for corner in range(8):
            cw = torch.ones((S,), dtype=torch.float32)
            idx = torch.zeros((S,), dtype=torch.int64)
            stride = 1
            for d in range(3):
                bit = (corner >> d) & 1
                gd = (g[:, d] + bit) & m32
                cw = cw * (w[:, d] if bit else 1.0 - w[:, d])
                if dense:
                    idx = (idx + gd * stride) & m32
                    stride *= res
                else:
                    idx = idx ^ ((gd * PRIMES[d]) & m32)
            idx = idx % size + off
            feats = tables.index_select(1, idx)                 # [C, S, F] fp16 gather: the 512*H bytes per sample
            acc += cw[:, None, None] * feats.permute(1, 0, 2).to(torch.float32)
        emb = acc.to(torch.float16).to(torch.float32)           # what each tcnn encoding hands back
        # 'b c (p f) -> b f (c p)'  (hash_ensemble.py:110-112), then the blend einsum (:155-156)
        emb = emb.reshape(S, C, P, 2).permute(0, 3, 1, 2).reshape(S, 2, C * P)[:, :, :H]
        outs.append(torch.einsum("sfh,sh->sf", emb, code16))
    return torch.cat(outs, dim=1).to(torch.float16)


@torch.no_grad()
def mlp_base_forward(feats: torch.Tensor, params: torch.Tensor) -> torch.Tensor:
    """tcnn FullyFusedMLP 32 -> 64 -> 16 (no biases, ReLU, fp16 activations) as two matmuls."""
    p = params.to(torch.float16).to(torch.float32)
    w0, wo = p[:64 * 32].reshape(64, 32), p[64 * 32:].reshape(16, 64)
    h = torch.relu(feats.to(torch.float32) @ w0.T).to(torch.float16).to(torch.float32)
    return (h @ wo.T).to(torch.float16)


def _time_once(S, tables, params, geom, H, g):
    import time
    x = torch.rand((S, 3), generator=g)
    code = torch.randn((S, H), generator=g)
    t0 = time.time()
    mlp_base_forward(hash_ensemble_forward(x, tables, code, geom, H), params)
    return time.time() - t0


def time_encoder_sweep(H: int, geom, sizes=(1 << 16, 1 << 18, 1 << 20), budget_s: float = 25.0, max_threads=None, seed=0):
    """Times hash_ensemble_forward + mlp_base_forward on uniformly random samples.  The intra-op thread count is chosen
    by a short calibration on 4096 samples (torch's CPU gathers do not scale to hundreds of threads); every size is
    only started if its projected time fits the remaining budget.  Returns (list of dict(samples, seconds,
    samples_per_s), threads used)."""
    import time
    t_begin = time.time()
    g = torch.Generator().manual_seed(seed)
    F = 8 if 2 * H >= 8 else 2 * H
    C = (2 * H + 7) // 8
    tables = ((torch.rand((C, geom.total_entries, F), generator=g) - 0.5)).to(torch.float16)
    params = (torch.rand((64 * 32 + 16 * 64,), generator=g) - 0.5) * 0.5
    max_threads = int(max_threads or torch.get_num_threads())
    best_t, best_rate = None, 0.0
    # ascending: few threads are quick to try; hundreds of threads on these small ops can be 20x slower
    for t in sorted({min(max_threads, 8), min(max_threads, 16), min(max_threads, 32), min(max_threads, 64), max_threads}):
        torch.set_num_threads(t)
        _time_once(1024, tables, params, geom, H, g)                       # warm up this thread-pool size
        dt = _time_once(4096, tables, params, geom, H, g)
        if 4096 / dt > best_rate:
            best_t, best_rate = t, 4096 / dt
        elif 4096 / dt < 0.5 * best_rate:                                  # clearly past the optimum
            break
        if time.time() - t_begin > 0.3 * budget_s:
            break
    # (round 6: the figure SURVEY.md 8(d) asks for beside the calibrated one -- the same encoder on ALL cores, ~3 s of samples)
    all_cores = None
    if max_threads != best_t and time.time() - t_begin < 0.5 * budget_s:
        torch.set_num_threads(max_threads)
        _time_once(256, tables, params, geom, H, g)
        # batches of 256 samples until ~3 s have passed (at most 16): on a host shared with other jobs hundreds of torch
        # threads ran anywhere between 27 and 500 samples/s from one call to the next -- a sample count projected from a probe
        # took 56 s and 428 s on two boxes; a batch count bounded by the clock cannot
        t_all, n_all = 0.0, 0
        while t_all < 3.0 and n_all < 16 * 256:
            t_all += _time_once(256, tables, params, geom, H, g)
            n_all += 256
        all_cores = {"threads": max_threads, "samples": n_all, "seconds": round(t_all, 3), "samples_per_s": n_all / max(t_all, 1e-6)}
    time_encoder_sweep.all_cores = all_cores
    torch.set_num_threads(best_t)
    out = []
    rate = best_rate
    for S in sizes:
        left = budget_s - (time.time() - t_begin)
        if out and S / rate > left:                                      # would not fit: stop the sweep here
            break
        if not out and S / rate > left:                                  # even the first size is too slow: shrink it
            S = max(4096, int(rate * left * 0.8) // 4096 * 4096)
        dt = _time_once(S, tables, params, geom, H, g)
        rate = S / dt
        out.append({"samples": S, "seconds": round(dt, 3), "samples_per_s": rate})
    return out, best_t
