"""GPU parity: fused HashEnsemble HIP kernels (through the C ABI) vs the CPU oracle and the reference
golden vectors.  Integer outputs bit-exact; fp outputs within the stated tolerances."""
import numpy as np
import pytest
import torch

import oracle
from oracle import hashgrid as ohg
from tests.helpers import SMALL_GEOM_KW, REF_GEOM_KW, make_tcnn_tables

pytestmark = pytest.mark.gpu
FP16_EPS = 2.0 ** -10


def _native_geom(kw):
    from nersemble_amd import _lib
    return _lib.grid_geometry(**kw)


def _setup(H, kw, seed, B, cuda, amplitude=0.5):
    from nersemble_amd import functional as F
    go = oracle.grid_geometry(**kw)
    gn = _native_geom(kw)
    tabs = make_tcnn_tables(H, go, seed, amplitude)
    f16, master = F.tables_from_tcnn(torch.from_numpy(tabs).to(cuda), H, gn)
    rng = np.random.default_rng(seed + 1)
    x = rng.random((B, 3), dtype=np.float32)
    if B >= 3:
        x[0] = 0.0
        x[1] = 0.999999
        x[2] = [0.5, 0.25, 0.75]
    code = (rng.standard_normal((B, H)) * 0.7).astype(np.float32)
    return go, gn, tabs, f16, master, x, code


@pytest.mark.parametrize("kw", [SMALL_GEOM_KW, REF_GEOM_KW])
def test_indices_bit_exact(kw, cuda):
    from nersemble_amd import functional as F
    go, gn = oracle.grid_geometry(**kw), _native_geom(kw)
    rng = np.random.default_rng(5)
    x = rng.random((4099, 3), dtype=np.float32)
    x[:4] = [[0, 0, 0], [0.999999, 0.999999, 0.999999], [1.0, 1.0, 1.0], [0.5, 0.5, 0.5]]
    # out-of-contract inputs must still agree (uint32 wrap + modulo), the field zeroes them in practice
    x[4] = [-0.25, 1.5, 3.0]
    want, _ = ohg.indices(x, go)
    got = F.hash_indices(torch.from_numpy(x).to(cuda), gn).cpu().numpy().astype(np.uint32)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
def test_layout_roundtrip_and_permutation(H, cuda):
    from nersemble_amd import functional as F
    go, gn, tabs, f16, master, _, _ = _setup(H, SMALL_GEOM_KW, 40 + H, 8, cuda)
    back = F.tables_to_tcnn(master, H, gn).cpu().numpy()
    assert np.array_equal(back, tabs)
    P = 4 if 2 * H >= 8 else H
    m = master.cpu().numpy()
    for h in sorted({0, H - 1, H // 2}):
        c, p = divmod(h, P)
        for f in (0, 1):
            assert np.array_equal(m[:, f, h], tabs[c, :, p * 2 + f])
    assert np.array_equal(f16.cpu().numpy().astype(np.float32), m)


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("B", [1, 7, 96, 1031])
def test_forward_matches_oracle(H, B, cuda):
    from nersemble_amd import functional as F
    go, gn, tabs, f16, master, x, code = _setup(H, SMALL_GEOM_KW, 100 + H, B, cuda)
    out = F.hash_ensemble(torch.from_numpy(x).to(cuda), master, f16, torch.from_numpy(code).to(cuda), H, gn)
    assert out.dtype == torch.float16 and out.shape == (B, 32)
    want = ohg.ensemble_fwd(x, tabs.astype(np.float16).view(np.uint16), H, go, code).astype(np.float32)
    got = out.float().cpu().numpy()
    # fp32 accumulation in a different order + one fp16 rounding: <= 1 fp16 ulp of the value
    tol = FP16_EPS * np.abs(want) + 1e-6
    assert (np.abs(got - want) <= tol).all(), float(np.abs(got - want).max())
    assert (got == want).mean() > 0.98


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
def test_forward_matches_reference_golden(H, cuda, golden_dir):
    """HIP kernel vs outputs of the reference's own HashEnsemble.forward (module-level API, windows incl.)."""
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    z = np.load(f"{golden_dir}/hash_ensemble.npz")
    kw = SMALL_GEOM_KW
    cfg = HashEnsembleConfig(H, TCNNHashEncodingConfig(n_levels=kw["n_levels"], log2_hashmap_size=kw["log2_hashmap_size"],
                                                       base_resolution=kw["base_resolution"],
                                                       per_level_scale=kw["per_level_scale"]),
                             disable_initial_hash_ensemble=True, use_soft_transition=True)
    he = HashEnsemble(cfg)
    go = oracle.grid_geometry(**kw)
    tabs = make_tcnn_tables(H, go, 100 + H)
    he.load_state_dict({f"hash_encodings.{c}.params": torch.from_numpy(tabs[c].reshape(-1)) for c in range(tabs.shape[0])})
    he = he.to(cuda)
    x = torch.from_numpy(z["he_x"]).to(cuda)
    code = torch.from_numpy(z[f"he_code_H{H}"]).to(cuda)
    for wi, w in enumerate(z[f"he_windows_H{H}"]):
        w = None if np.isnan(w) else float(w)
        with torch.no_grad():
            out = he(x, code.clone(), window_hash_encodings=w).float().cpu().numpy()
        ref = z[f"he_out_H{H}_w{wi}"].astype(np.float32)
        tol = 2.0 * FP16_EPS * np.abs(ref).max() + 1e-6
        assert np.abs(out - ref).max() <= tol, (H, w, float(np.abs(out - ref).max()))


def test_code_index_path_equals_gathered_codes(cuda):
    from nersemble_amd import functional as F
    H, B, T = 16, 517, 9
    go, gn, tabs, f16, master, x, _ = _setup(H, SMALL_GEOM_KW, 9, B, cuda)
    rng = np.random.default_rng(2)
    emb = torch.from_numpy((rng.standard_normal((T, H)) * 0.5).astype(np.float32)).to(cuda)
    ts = torch.from_numpy(rng.integers(0, T, B).astype(np.int32)).to(cuda)
    win = torch.from_numpy(ohg.posenc_window(5.5, 0, H - 1, H)).to(cuda)
    xt = torch.from_numpy(x).to(cuda)
    a = F.hash_ensemble(xt, master, f16, emb, H, gn, code_index=ts, window=win)
    b = F.hash_ensemble(xt, master, f16, emb[ts.long()].contiguous(), H, gn, window=win)
    assert torch.equal(a, b)


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
def test_backward_matches_oracle(H, cuda):
    from nersemble_amd import functional as F
    B = 203
    go, gn, tabs, f16, master, x, code = _setup(H, SMALL_GEOM_KW, 300 + H, B, cuda)
    rng = np.random.default_rng(8)
    # the upstream gradient of an fp16 activation arrives in fp16 (AMP semantics, as in the reference)
    dout = rng.standard_normal((B, 32)).astype(np.float16).astype(np.float32)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    ct = torch.from_numpy(code).to(cuda).requires_grad_(True)
    mt = master.clone().requires_grad_(True)
    out = F.hash_ensemble(xt, mt, f16, ct, H, gn)
    out.backward(torch.from_numpy(dout).to(cuda).half())
    dtab_o, dcode_o, dx_o = ohg.ensemble_bwd(x, tabs.astype(np.float16).view(np.uint16), H, go, code, dout)
    # table gradient: compare in the reference's tcnn layout (tests the permutation of the grad as well)
    dtab = F.tables_to_tcnn(mt.grad, H, gn).cpu().numpy()
    scale = np.abs(dtab_o).max()
    assert np.abs(dtab - dtab_o).max() <= 2e-5 * scale + 1e-7, float(np.abs(dtab - dtab_o).max())
    dcode = ct.grad.cpu().numpy()
    assert np.abs(dcode - dcode_o).max() <= 2e-5 * np.abs(dcode_o).max() + 1e-7
    dx = xt.grad.cpu().numpy()
    assert np.abs(dx - dx_o).max() <= 5e-5 * np.abs(dx_o).max() + 1e-6


@pytest.mark.parametrize("H,T", [(32, 5), (16, 24), (4, 64), (1, 3), (8, 200)])
@pytest.mark.parametrize("opened", ["most", "one", "two and a bit"])
def test_backward_with_code_index_and_window(H, T, opened, cuda):
    """Indexed codes: T <= 64 rows takes the factored table-gradient path (G scatter + expand), T = 200 the
    generic atomics path; both must equal the oracle (table, code-table, position gradients).  ``opened``: how far the
    coarse-to-fine window has come (train_nersemble.py:77-78: one grid for the first 40 000 steps)."""
    from nersemble_amd import functional as F
    B = 311
    window_value = {"most": 0.63 * H + 0.2, "one": 1.0, "two and a bit": 2.37}[opened]
    go, gn, tabs, f16, master, x, _ = _setup(H, SMALL_GEOM_KW, 77 + H, B, cuda)
    rng = np.random.default_rng(12)
    emb = (rng.standard_normal((T, H)) * 0.5).astype(np.float32)
    ts = rng.integers(0, T, B).astype(np.int32)
    win = ohg.posenc_window(window_value, 0, H - 1, H)
    dout = rng.standard_normal((B, 32)).astype(np.float16).astype(np.float32)
    et = torch.from_numpy(emb).to(cuda).requires_grad_(True)
    mt = master.clone().requires_grad_(True)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    out = F.hash_ensemble(xt, mt, f16, et, H, gn,
                          code_index=torch.from_numpy(ts).to(cuda), window=torch.from_numpy(win).to(cuda))
    out.backward(torch.from_numpy(dout).to(cuda).half())
    codew = emb[ts] * win[None]
    dtab_o, dcw, dx_o = ohg.ensemble_bwd(x, tabs.astype(np.float16).view(np.uint16), H, go, codew, dout)
    want = np.zeros_like(emb)
    np.add.at(want, ts, dcw * win[None])
    assert np.abs(et.grad.cpu().numpy() - want).max() <= 1e-4 * np.abs(want).max() + 1e-6
    dtab = F.tables_to_tcnn(mt.grad, H, gn).cpu().numpy()
    assert np.abs(dtab - dtab_o).max() <= 2e-5 * np.abs(dtab_o).max() + 1e-7
    assert np.abs(xt.grad.cpu().numpy() - dx_o).max() <= 5e-5 * np.abs(dx_o).max() + 1e-6


def test_empty_batch(cuda):
    from nersemble_amd import functional as F
    H = 4
    go, gn, tabs, f16, master, _, _ = _setup(H, SMALL_GEOM_KW, 1, 4, cuda)
    out = F.hash_ensemble(torch.zeros((0, 3), device=cuda), master, f16, torch.zeros((0, H), device=cuda), H, gn)
    assert out.shape == (0, 32)


def test_full_size_properties_h32(cuda):
    """BASELINE-size run (reference geometry, H=32, 2^20 samples): properties that need no oracle pass:
    linearity in the code, agreement with the oracle on a random subsample, determinism."""
    from nersemble_amd import functional as F
    H, B = 32, 1 << 20
    gn = _native_geom(REF_GEOM_KW)
    go = oracle.grid_geometry(**REF_GEOM_KW)
    gen = torch.Generator(device=cuda).manual_seed(3)
    total = gn.total_entries
    master = (torch.rand((total, 2, H), device=cuda, generator=gen) - 0.5)
    f16 = master.half()
    master = f16.float()
    x = torch.rand((B, 3), device=cuda, generator=gen)
    c1 = torch.randn((B, H), device=cuda, generator=gen) * 0.5
    c2 = torch.randn((B, H), device=cuda, generator=gen) * 0.5
    c1, c2 = c1.half().float(), c2.half().float()
    o1 = F.hash_ensemble(x, master, f16, c1, H, gn).float()
    o2 = F.hash_ensemble(x, master, f16, c2, H, gn).float()
    o12 = F.hash_ensemble(x, master, f16, (c1 + c2).half().float(), H, gn).float()
    lin = (o1 + o2 - o12).abs().max().item()
    assert lin <= 8 * FP16_EPS * max(1.0, o12.abs().max().item())
    assert torch.equal(o1, F.hash_ensemble(x, master, f16, c1, H, gn).float())
    # oracle on a subsample of 2048 samples (tables converted back to the reference layout)
    sel = torch.randperm(B, device=cuda, generator=gen)[:2048]
    tc = F.tables_to_tcnn(master, H, gn).cpu().numpy().astype(np.float16)
    want = ohg.ensemble_fwd(x[sel].cpu().numpy(), tc.view(np.uint16), H, go, c1[sel].cpu().numpy()).astype(np.float32)
    got = o1[sel].cpu().numpy()
    assert (np.abs(got - want) <= FP16_EPS * np.abs(want) + 1e-6).all()


@pytest.mark.parametrize("F_enc", [2, 4, 8])
def test_tcnn_shaped_hashgrid_encoding(F_enc, cuda):
    """tcnn.Encoding(HashGrid) compatibility operator vs the oracle's single-encoding restatement (+ gradients)."""
    from nersemble_amd import tcnn
    kw = SMALL_GEOM_KW
    enc = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": kw["n_levels"], "n_features_per_level": F_enc,
                            "log2_hashmap_size": kw["log2_hashmap_size"], "base_resolution": kw["base_resolution"],
                            "per_level_scale": kw["per_level_scale"], "interpolation": "Linear"}).to(cuda)
    go = oracle.grid_geometry(**kw)
    rng = np.random.default_rng(F_enc)
    tab = ((rng.random((go.total_entries, F_enc), dtype=np.float32) - 0.5)).astype(np.float16)
    with torch.no_grad():
        enc.params.copy_(torch.from_numpy(tab.astype(np.float32).reshape(-1)))
    B = 333
    x = rng.random((B, 3), dtype=np.float32)
    xt = torch.from_numpy(x).to(cuda).requires_grad_(True)
    out = enc(xt)
    want = ohg.hashgrid_fwd(x, tab.view(np.uint16), go).astype(np.float32)
    assert (np.abs(out.float().detach().cpu().numpy() - want) <= FP16_EPS * np.abs(want) + 1e-6).all()
    # gradients, value by value, through the ensemble oracle: an encoding with F_enc features per level IS the
    # HashEnsemble of H = F_enc / 2 grids (tcnn feature j = p * 2 + f of level l belongs to grid p, hash_ensemble.py:110-112),
    # so one oracle backward per grid p with the one-hot code e_p and that grid's two columns of dout gives the table
    # gradient per entry (only grid p's features are touched) and its share of dL/dx; their sums are what the operator
    # must return.  A wrong corner weight, a swapped feature plane or a swapped level changes single entries.
    H = F_enc // 2
    L = kw["n_levels"]
    dout = rng.standard_normal((B, L * F_enc)).astype(np.float16)
    out.backward(torch.from_numpy(dout).to(cuda))
    d3 = dout.astype(np.float32).reshape(B, L, H, 2)
    dtab_want = np.zeros((go.total_entries, F_enc), dtype=np.float64)
    dx_want = np.zeros((B, 3), dtype=np.float64)
    for p in range(H):
        code = np.zeros((B, H), dtype=np.float32)
        code[:, p] = 1.0
        dtab_p, _, dx_p = ohg.ensemble_bwd(x, tab.view(np.uint16)[None], H, go, code,
                                           np.ascontiguousarray(d3[:, :, p, :]).reshape(B, L * 2))
        others = [j for j in range(F_enc) if j // 2 != p]
        assert not dtab_p[0][:, others].any()                   # (the oracle's own layout: grid p <-> features 2p, 2p+1)
        dtab_want += dtab_p[0]
        dx_want += dx_p
    dtab = enc.params.grad.view(go.total_entries, F_enc).cpu().numpy()
    sc = np.abs(dtab_want).max()
    assert np.abs(dtab - dtab_want).max() <= 2e-5 * sc + 1e-7, float(np.abs(dtab - dtab_want).max() / sc)
    assert (np.abs(dtab_want) > 1e-3 * sc).sum() > 1000          # a gradient that exercises every level
    dx = xt.grad.cpu().numpy()
    assert np.abs(dx - dx_want).max() <= 5e-5 * np.abs(dx_want).max() + 1e-6, \
        float(np.abs(dx - dx_want).max() / np.abs(dx_want).max())


def test_reference_layout_ensemble_from_tcnn_encodings_matches_fused(cuda):
    """The operator-level drop-in (C separate tcnn.Encoding(HashGrid) + rearrange + einsum, i.e. what the reference's
    own HashEnsemble.forward does) equals the fused native HashEnsemble on the same parameters."""
    import einops
    from nersemble_amd import tcnn
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    H, kw = 16, SMALL_GEOM_KW
    cfg = HashEnsembleConfig(H, TCNNHashEncodingConfig(n_levels=kw["n_levels"], log2_hashmap_size=kw["log2_hashmap_size"],
                                                       base_resolution=kw["base_resolution"],
                                                       per_level_scale=kw["per_level_scale"]))
    he = HashEnsemble(cfg).to(cuda)
    with torch.no_grad():
        he.tables.mul_(4000)
    sd = he.state_dict()
    encs = []
    for c in range(4):
        e = tcnn.Encoding(3, {"otype": "HashGrid", "n_levels": kw["n_levels"], "n_features_per_level": 8,
                              "log2_hashmap_size": kw["log2_hashmap_size"], "base_resolution": kw["base_resolution"],
                              "per_level_scale": kw["per_level_scale"]}).to(cuda)
        with torch.no_grad():
            e.params.copy_(sd[f"hash_encodings.{c}.params"])
        encs.append(e)
    g = torch.Generator(device=cuda).manual_seed(0)
    x = torch.rand((257, 3), device=cuda, generator=g)
    code = torch.randn((257, H), device=cuda, generator=g) * 0.5
    emb = torch.stack([e(x) for e in encs], dim=1)                                   # hash_ensemble.py:102-106
    emb = einops.rearrange(emb, 'b c (l p f) -> b (l f) (c p) ', l=kw["n_levels"], p=4, f=2)   # :112
    ref = torch.einsum('bdh,bh->bd', emb, code.to(emb))                               # :155-156
    fused = he(x, code)
    tol = 2.0 * FP16_EPS * ref.float().abs().max().item() + 1e-6
    assert (fused.float() - ref.float()).abs().max().item() <= tol


@pytest.mark.parametrize("H", [1, 4, 16, 32])
@pytest.mark.parametrize("win", [None, 1.0, 1.5, 5.25])
def test_preblended_eval_grid_matches_ensemble_forward(H, win, cuda):
    """Eval fast path: blending the tables with one time code and sampling the resulting 2-feature grid
    equals the per-sample blend up to fp16 rounding order (blend-then-interpolate vs interpolate-then-blend)."""
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    cfg = HashEnsembleConfig(H, TCNNHashEncodingConfig(n_levels=8, log2_hashmap_size=13), True, True)
    he = HashEnsemble(cfg, seed=3).to(cuda)
    with torch.no_grad():
        he.tables.mul_(5000)
    g = torch.Generator(device=cuda).manual_seed(H)
    x = torch.rand((5000, 3), device=cuda, generator=g)
    code = torch.randn((1, H), device=cuda, generator=g)
    with torch.no_grad():
        want = he(x, code.expand(x.shape[0], H).contiguous(), window_hash_encodings=win).float()
        table = he.preblend(code, window_hash_encodings=win)
        got = he.forward_preblended(x, table).float()
    assert table.shape == (he.geom.total_entries, 2) and table.dtype == torch.float16
    scale = want.abs().max().item() + 1e-12
    assert (got - want).abs().max().item() <= 4e-3 * scale, (got - want).abs().max().item() / scale


@pytest.mark.parametrize("H", [4, 16, 32])
@pytest.mark.parametrize("coherent", [False, True])
def test_split_backward_equals_fused(H, coherent, cuda):
    """The factored backward as two launches (gather half: dcode / dx; scatter half: G, no tables, no codes) against the
    fused kernel: dcode and dx BIT-equal (same code path, the scatter compiled out), G up to the order of the fp32
    atomics.  ``coherent``: consecutive samples along rays, so that the duplicate-cell merging is exercised."""
    import ctypes as C
    from nersemble_amd._lib import check, lib, ptr, stream
    B, T = 5003, 13
    go, gn, tabs, f16, master, x, code = _setup(H, SMALL_GEOM_KW, 300 + H, B, cuda)
    rng = np.random.default_rng(H)
    if coherent:
        o = rng.random((50, 1, 3), dtype=np.float32) * 0.5 + 0.1
        d = rng.standard_normal((50, 1, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        t = (np.arange(101, dtype=np.float32) * 0.0023)[None, :, None]
        x = np.clip((o + d * t).reshape(-1, 3)[:B], 0.0, 0.999).astype(np.float32)
        B = x.shape[0]
    xt = torch.from_numpy(x).to(cuda)
    table = torch.from_numpy((rng.standard_normal((T, H)) * 0.6).astype(np.float32)).to(cuda)
    slot = torch.from_numpy(np.sort(rng.integers(0, T, B)).astype(np.int32)).to(cuda)
    dout = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32)).to(cuda)
    total = gn.total_entries

    def run(G, dc, dx, nf):
        check(lib().nsx_hash_ensemble_bwd_factored(ptr(xt), B, ptr(f16), H, C.byref(gn), ptr(table), table.stride(0), T,
                                                   ptr(slot), None, ptr(dout), ptr(G), ptr(dc), ptr(dx), ptr(nf), None, stream()),
              "nsx_hash_ensemble_bwd_factored")

    G_f = torch.zeros((T, total, 2), device=cuda)
    dc_f, dx_f = torch.empty((B, H), device=cuda), torch.empty((B, 3), device=cuda)
    nf = torch.zeros((1,), device=cuda)
    run(G_f, dc_f, dx_f, nf)
    dc_g, dx_g = torch.empty((B, H), device=cuda), torch.empty((B, 3), device=cuda)
    run(None, dc_g, dx_g, None)                                                     # gather half
    assert torch.equal(dc_g, dc_f) and torch.equal(dx_g, dx_f)
    for blocks in (1, 8):
        G_s = torch.zeros((T, total, 2), device=cuda)
        nf_s = torch.zeros((1,), device=cuda)
        check(lib().nsx_hash_ensemble_bwd_scatter(ptr(xt), B, C.byref(gn), T, ptr(slot), ptr(dout), ptr(G_s), ptr(nf_s),
                                                  blocks, None, stream()), "nsx_hash_ensemble_bwd_scatter")
        sc = G_f.abs().max().item()
        assert sc > 0 and (G_s - G_f).abs().max().item() <= 2e-5 * sc
        assert torch.equal((G_s != 0), (G_f != 0)) and nf_s.item() == 0
    # a non-finite upstream gradient raises the flag (GradScaler's inf check on the table gradient)
    dout[B // 2, 7] = float("inf")
    nf_s = torch.zeros((1,), device=cuda)
    check(lib().nsx_hash_ensemble_bwd_scatter(ptr(xt), B, C.byref(gn), T, ptr(slot), ptr(dout), ptr(torch.zeros_like(G_f)),
                                              ptr(nf_s), 2, None, stream()), "nsx_hash_ensemble_bwd_scatter")
    assert nf_s.item() == 1
    # empty batch: no launch, no error
    check(lib().nsx_hash_ensemble_bwd_scatter(ptr(xt), 0, C.byref(gn), T, ptr(slot), ptr(dout), ptr(G_f), None, 2,
                                              None,
                                              stream()), "nsx_hash_ensemble_bwd_scatter")


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("order", ["sorted", "random"])
def test_code_gradient_summed_in_the_kernel(H, order, cuda):
    """nsx_hash_ensemble_bwd_codesum: the code gradient reduced per code row inside the kernel (LDS sums per block +
    one second-stage block per row, window chain rule folded in) against the per-sample gradient of
    nsx_hash_ensemble_bwd_factored summed with float64 on the host.  dx and G are the fused kernel's.  ``sorted``:
    samples of a row are adjacent (a ray's samples: the DPP run-merge path), ``random``: every lane its own row."""
    import ctypes as C
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    B, T = 4099, 24
    go, gn, tabs, f16, master, x, _ = _setup(H, SMALL_GEOM_KW, 500 + H, B, cuda)
    rng = np.random.default_rng(3 * H)
    xt = torch.from_numpy(x).to(cuda)
    table = torch.from_numpy((rng.standard_normal((T, H)) * 0.6).astype(np.float32)).to(cuda)
    sl = rng.integers(0, T, B)
    if order == "sorted":
        sl = np.sort(sl)
    slot = torch.from_numpy(sl.astype(np.int32)).to(cuda)
    win_np = ohg.posenc_window(0.6 * H + 0.3, 0, H - 1, H).astype(np.float32)
    win = torch.from_numpy(win_np).to(cuda)
    dout = torch.from_numpy(rng.standard_normal((B, 32)).astype(np.float32)).to(cuda)
    total = gn.total_entries

    G_f = torch.zeros((T, total, 2), device=cuda)
    dc_f, dx_f = torch.empty((B, H), device=cuda), torch.empty((B, 3), device=cuda)
    check(lib().nsx_hash_ensemble_bwd_factored(ptr(xt), B, ptr(f16), H, C.byref(gn), ptr(table), table.stride(0), T,
                                               ptr(slot), ptr(win), ptr(dout), ptr(G_f), ptr(dc_f), ptr(dx_f), None,
                                               None,
                                               stream()), "nsx_hash_ensemble_bwd_factored")
    want = np.zeros((T, H), dtype=np.float64)
    np.add.at(want, sl, dc_f.cpu().numpy().astype(np.float64) * win_np[None].astype(np.float64))

    def run(n, G, rows, dx, n_device=None):
        check(lib().nsx_hash_ensemble_bwd_codesum(ptr(xt), n, ptr(f16), H, C.byref(gn), ptr(table), table.stride(0), T,
                                                  ptr(slot), ptr(win), ptr(dout), ptr(G), ptr(rows),
                                                  ptr(F.codesum_scratch(T, H, cuda)), ptr(dx), None, ptr(n_device),
                                                  stream()),
              "nsx_hash_ensemble_bwd_codesum")

    for with_G in (True, False):
        G = torch.zeros((T, total, 2), device=cuda) if with_G else None
        rows = torch.full((T, H), float("nan"), device=cuda)
        dx = torch.empty((B, 3), device=cuda)
        run(B, G, rows, dx)
        got = rows.cpu().numpy().astype(np.float64)
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max() + 1e-7, float(np.abs(got - want).max())
        assert torch.equal(dx, dx_f)
        if with_G:
            assert (G - G_f).abs().max().item() <= 2e-5 * G_f.abs().max().item()
    # rows no sample uses come out as exact zeros
    unused = np.setdiff1d(np.arange(T), sl)
    assert (got[unused] == 0).all()
    # device-side sample count (the entry points' n_device argument): only the first n rows contribute; n = 0 gives zeros
    for n in (B // 3, 0):
        n_dev = torch.tensor([n], dtype=torch.int64, device=cuda)
        rows = torch.full((T, H), float("nan"), device=cuda)
        run(B, None, rows, torch.empty((B, 3), device=cuda), n_device=n_dev)
        w2 = np.zeros((T, H), dtype=np.float64)
        np.add.at(w2, sl[:n], dc_f.cpu().numpy()[:n].astype(np.float64) * win_np[None].astype(np.float64))
        assert np.abs(rows.cpu().numpy() - w2).max() <= 2e-5 * np.abs(want).max() + 1e-7
    # empty batch: zeros, no launch
    rows = torch.full((T, H), float("nan"), device=cuda)
    run(0, None, rows, None)
    assert (rows == 0).all()
