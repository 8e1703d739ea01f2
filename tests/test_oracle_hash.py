"""CPU: pins the oracle's HashEnsemble restatement against fixtures produced by the reference's own
HashEnsemble.forward (tests/golden/make_golden.py), and checks geometry known answers."""
import numpy as np
import pytest

import oracle
from oracle import hashgrid as ohg
from tests.helpers import SMALL_GEOM_KW, REF_GEOM_KW, make_tcnn_tables, ens_layout

FP16_EPS = 2.0 ** -10


def test_reference_geometry_known_answers():
    # level sizes of the reference config, 6 299 960 entries per encoding
    g = oracle.grid_geometry(**REF_GEOM_KW)
    sizes = [int(g.size[i]) for i in range(16)]
    assert sizes[:5] == [4096, 13824, 39304, 117656, 357912]
    assert sizes[5:] == [2 ** 19] * 11
    assert g.total_entries == 6299960
    assert [int(g.res[i]) for i in range(16)] == [16, 24, 34, 49, 71, 102, 148, 213, 308, 446, 646, 934, 1352, 1956,
                                                  2831, 4096]
    assert float(g.scale[0]) == 15.0
    assert abs(float(g.scale[15]) - 4094.9985) < 1e-3      # one ulp from flipping the resolution to 4097


def test_fp16_conversion_matches_numpy():
    L = oracle.lib()
    allh = np.arange(65536, dtype=np.uint16)
    f = np.array([L.nsxo_h2f(int(h)) for h in allh], dtype=np.float32)
    ref = allh.view(np.float16).astype(np.float32)
    assert ((f == ref) | (np.isnan(f) & np.isnan(ref))).all()
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(20000) * 10.0 ** rng.integers(-9, 5, 20000)).astype(np.float32)
    with np.errstate(over="ignore"):
        want = v.astype(np.float16).view(np.uint16)
    got = np.array([L.nsxo_f2h(float(a)) for a in v], dtype=np.uint16)
    assert (got == want).all()


def test_indices_in_range_and_dense_formula():
    g = oracle.grid_geometry(**SMALL_GEOM_KW)
    rng = np.random.default_rng(1)
    x = rng.random((500, 3), dtype=np.float32)
    idx, w = ohg.indices(x, g)
    for l in range(g.n_levels):
        assert idx[:, l].max() < g.size[l]
    assert (w >= 0).all() and (w < 1).all()
    # level 0 is dense 16^3: index = cx + 16*cy + 256*cz (mod 4096)
    p = np.float32(15.0) * x + np.float32(0.5)
    c = np.floor(p).astype(np.int64)
    want = (c[:, 0] + 16 * c[:, 1] + 256 * c[:, 2]) % 4096
    assert (idx[:, 0, 0] == want).all()
    want7 = ((c[:, 0] + 1) + 16 * (c[:, 1] + 1) + 256 * (c[:, 2] + 1)) % 4096
    assert (idx[:, 0, 7] == want7).all()


@pytest.mark.parametrize("H", [1, 2, 4, 8, 16, 32])
def test_fused_oracle_matches_reference_forward(H, golden_dir):
    """oracle fused ensemble == reference HashEnsemble.forward (real reference code, stub encoder) to fp16
    rounding: the reference rounds per-grid features and the window product to fp16 before the einsum, the
    fused form rounds once."""
    z = np.load(f"{golden_dir}/hash_ensemble.npz")
    g = oracle.grid_geometry(**SMALL_GEOM_KW)
    tabs = make_tcnn_tables(H, g, 100 + H).astype(np.float16).view(np.uint16)
    x, code = z["he_x"], z[f"he_code_H{H}"]
    for wi, w in enumerate(z[f"he_windows_H{H}"]):
        w = None if np.isnan(w) else float(w)
        out = ohg.ensemble_fwd(x, tabs, H, g, ohg.windowed_code(code, H, w)).astype(np.float32)
        ref = z[f"he_out_H{H}_w{wi}"].astype(np.float32)
        tol = 2.0 * FP16_EPS * np.abs(ref).max() + 1e-6
        assert np.abs(out - ref).max() <= tol, (H, w, np.abs(out - ref).max(), tol)


def test_single_encoding_equals_ensemble_with_onehot_code():
    """Rearrange map check inside the oracle itself: grid h of the ensemble == features (p*2+f) of encoding c."""
    H = 8
    g = oracle.grid_geometry(**SMALL_GEOM_KW)
    f_enc, P, C = ens_layout(H)
    tabs = make_tcnn_tables(H, g, 7).astype(np.float16)
    rng = np.random.default_rng(3)
    x = rng.random((33, 3), dtype=np.float32)
    per_enc = [ohg.hashgrid_fwd(x, tabs[c].view(np.uint16), g).astype(np.float32) for c in range(C)]
    for h in range(H):
        code = np.zeros((33, H), np.float32)
        code[:, h] = 1.0
        out = ohg.ensemble_fwd(x, tabs.view(np.uint16), H, g, code).astype(np.float32)
        c, p = divmod(h, P)
        want = per_enc[c].reshape(33, g.n_levels, f_enc)[:, :, p * 2:p * 2 + 2].reshape(33, -1)
        assert np.array_equal(out, want)


def test_posenc_window_known_answers(golden_dir):
    z = np.load(f"{golden_dir}/hash_ensemble.npz")
    i = 0
    while f"pw_{i}" in z:
        w, n = z[f"pw_{i}_args"]
        got = ohg.posenc_window(float(w), 0, int(n) - 1, int(n))
        assert np.allclose(got, z[f"pw_{i}"], atol=1e-6), i
        i += 1
    assert i >= 6
    assert np.allclose(ohg.posenc_window(3.25, 0, 31, 32)[:6], [1, 1, 1, 0.1464, 0, 0], atol=1e-4)


def test_ensemble_bwd_matches_finite_differences():
    """oracle backward (dcode, dx) vs central differences of a double-precision evaluation of the same sum."""
    H = 4
    g = oracle.grid_geometry(n_levels=4, per_level_scale=1.5, base_resolution=4, log2_hashmap_size=8)
    tabs = make_tcnn_tables(H, g, 11).astype(np.float16)
    rng = np.random.default_rng(4)
    B = 6
    x = (rng.random((B, 3)) * 0.8 + 0.1).astype(np.float32)
    # fp16-exact codes so the fp16 rounding of the code is the identity
    code = (rng.standard_normal((B, H)) * 0.5).astype(np.float16).astype(np.float32)
    dout = rng.standard_normal((B, 2 * g.n_levels)).astype(np.float32)
    dtab, dcode, dx = ohg.ensemble_bwd(x, tabs.view(np.uint16), H, g, code, dout)

    def loss(xx, cc, tt):
        # float64 re-evaluation straight from indices/weights
        idx, w = ohg.indices(xx.astype(np.float32), g)
        f_enc, P, C = ens_layout(H)
        total = 0.0
        for b in range(B):
            for l in range(g.n_levels):
                # recompute weights in float64 from positions for smooth finite differences
                pos = np.float64(g.scale[l]) * xx[b].astype(np.float64) + 0.5
                fr = pos - np.floor(pos)
                for k in range(8):
                    wk = 1.0
                    for d in range(3):
                        wk *= fr[d] if (k >> d) & 1 else 1 - fr[d]
                    e = int(g.offset[l]) + int(idx[b, l, k])
                    for h in range(H):
                        c, p = divmod(h, P)
                        for f in range(2):
                            total += wk * tt[c, e, p * 2 + f] * cc[b, h] * dout[b, l * 2 + f]
        return total

    t64 = tabs.astype(np.float64)
    eps = 1e-3
    for (b, h) in [(0, 0), (2, 3), (5, 1)]:
        cp, cm = code.astype(np.float64).copy(), code.astype(np.float64).copy()
        cp[b, h] += eps
        cm[b, h] -= eps
        fd = (loss(x, cp, t64) - loss(x, cm, t64)) / (2 * eps)
        assert abs(fd - dcode[b, h]) <= 1e-3 * max(1.0, abs(fd)), (b, h, fd, dcode[b, h])
    epsx = 1e-5
    for (b, d) in [(1, 0), (3, 1), (4, 2)]:
        xp, xm = x.astype(np.float64).copy(), x.astype(np.float64).copy()
        xp[b, d] += epsx
        xm[b, d] -= epsx
        # stay inside the same cell for a valid derivative
        if (np.floor(np.float64(g.scale[g.n_levels - 1]) * xp[b, d] + 0.5) !=
                np.floor(np.float64(g.scale[g.n_levels - 1]) * xm[b, d] + 0.5)):
            continue
        fd = (loss(xp, code.astype(np.float64), t64) - loss(xm, code.astype(np.float64), t64)) / (2 * epsx)
        assert abs(fd - dx[b, d]) <= 2e-2 * max(1.0, abs(fd)), (b, d, fd, dx[b, d])
    # table gradient: linearity check  sum(dtab * T) == sum(dout * out(T))  (out is linear in T)
    lhs = float((dtab.astype(np.float64) * t64).sum())
    rhs = loss(x, code.astype(np.float64), t64)
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))


def test_torch_cpu_encoder_restatement_matches_the_c_oracle():
    """oracle/torch_cpu.py (the PyTorch-CPU encoder bench.py times as ``cpu_baseline``) against the C oracle."""
    import torch
    from oracle import torch_cpu, mlp as omlp
    for H, kw in ((4, SMALL_GEOM_KW), (16, SMALL_GEOM_KW), (32, SMALL_GEOM_KW)):
        g = oracle.grid_geometry(**kw)
        tabs = make_tcnn_tables(H, g, 3 + H)
        rng = np.random.default_rng(H)
        x = rng.random((2000, 3), dtype=np.float32)
        code = (rng.standard_normal((2000, H)) * 0.7).astype(np.float32)
        want = ohg.ensemble_fwd(x, tabs.astype(np.float16).view(np.uint16), H, g, code).astype(np.float32)
        got = torch_cpu.hash_ensemble_forward(torch.from_numpy(x), torch.from_numpy(tabs).to(torch.float16),
                                              torch.from_numpy(code), g, H)
        err = np.abs(got.float().numpy() - want)
        # per-encoding fp16 rounding before the blend (what the reference's C tcnn encodings do) vs the fused oracle's
        # single rounding: a few fp16 ulp of the blended value's scale
        assert np.quantile(err, 0.999) <= 2.0 ** -8 * max(1.0, np.abs(want).max()), (H, float(err.max()))
        params = (rng.standard_normal(omlp.param_count(0)) * 0.25).astype(np.float32)
        base = torch_cpu.mlp_base_forward(got, torch.from_numpy(params)).float().numpy()
        want_b = omlp.mlp_fwd(got.float().numpy(), params, 0, 16, 0).astype(np.float32)
        assert np.abs(base - want_b).max() <= 4 * 2.0 ** -10 * max(1.0, np.abs(want_b).max())


@pytest.mark.parametrize("H", [4, 16, 32])
def test_first_grid_phase_algebra(H):
    """What the compact first-grid phase of the product rests on (HashEnsemble.first_grid_phase), stated on the oracle:
    at window 1 with ``disable_initial_hash_ensemble`` the H-grid ensemble IS its first grid -- forward, position
    gradient and the gradient of grid 0 equal those of a one-grid ensemble holding grid 0, bit for bit, and every other
    grid receives exactly zero."""
    g = oracle.grid_geometry(**SMALL_GEOM_KW)
    f_enc, P, C = ens_layout(H)
    tabs = make_tcnn_tables(H, g, 11).astype(np.float16)
    rng = np.random.default_rng(5)
    B = 57
    x = rng.random((B, 3), dtype=np.float32)
    dout = rng.standard_normal((B, 2 * g.n_levels)).astype(np.float16).astype(np.float32)
    codew = ohg.windowed_code(rng.standard_normal((B, H)), H, 1.0)               # ones * window(1) = e_0
    assert np.array_equal(codew, np.eye(H, dtype=np.float32)[0][None].repeat(B, 0))
    first = np.ascontiguousarray(tabs[0:1, :, 0:2])                              # grid 0 = encoding 0, features 0..1
    out_h = ohg.ensemble_fwd(x, tabs.view(np.uint16), H, g, codew)
    out_1 = ohg.ensemble_fwd(x, first.view(np.uint16), 1, g, np.ones((B, 1), np.float32))
    assert np.array_equal(out_h.view(np.uint16), out_1.view(np.uint16))
    dtab_h, _, dx_h = ohg.ensemble_bwd(x, tabs.view(np.uint16), H, g, codew, dout)
    dtab_1, _, dx_1 = ohg.ensemble_bwd(x, first.view(np.uint16), 1, g, np.ones((B, 1), np.float32), dout)
    assert np.array_equal(dx_h, dx_1)
    assert np.array_equal(dtab_h[0, :, 0:2], dtab_1[0])
    rest = dtab_h.copy()
    rest[0, :, 0:2] = 0
    assert not rest.any()                                                         # Adam never moves the other grids


@pytest.mark.parametrize("H", [1, 4, 32])
def test_cpu_baseline_port_matches_the_checker(H):
    """bench.py's cpu_baseline times ``ensemble_fwd_fast`` (fp32 accumulation, table-driven fp16 decode); it computes what
    the double-precision checker computes, to one fp16 ulp."""
    g = oracle.grid_geometry(**SMALL_GEOM_KW)
    tabs = make_tcnn_tables(H, g, 23).astype(np.float16)
    rng = np.random.default_rng(9)
    x = rng.random((301, 3), dtype=np.float32)
    codew = ohg.windowed_code(rng.standard_normal((301, H)), H, 0.6 * H + 0.3)
    a = ohg.ensemble_fwd(x, tabs.view(np.uint16), H, g, codew).astype(np.float32)
    b = ohg.ensemble_fwd_fast(x, tabs.view(np.uint16), H, g, codew).astype(np.float32)
    ulp = np.maximum(np.abs(a), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(a - b) <= 1.01 * ulp).all(), np.abs(a - b).max()


@pytest.mark.parametrize("F_enc", [2, 4, 8])
def test_single_encoding_backward_is_the_sum_of_one_hot_ensemble_backwards(F_enc):
    """What tests/test_hash_ensemble_gpu.py::test_tcnn_shaped_hashgrid_encoding holds ``nsx_hashgrid_bwd`` to: the table
    gradient of ONE tcnn HashGrid encoding with F_enc features per level, built from ``ensemble_bwd`` with H = F_enc / 2
    grids and one-hot codes, equals the direct scatter of (trilinear corner weight x dout) through the oracle's own
    indices / weights -- entry by entry."""
    kw = SMALL_GEOM_KW
    go = oracle.grid_geometry(**kw)
    rng = np.random.default_rng(F_enc)
    tab = ((rng.random((go.total_entries, F_enc), dtype=np.float32) - 0.5)).astype(np.float16)
    B, H, L = 257, F_enc // 2, kw["n_levels"]
    x = rng.random((B, 3), dtype=np.float32)
    dout = rng.standard_normal((B, L * F_enc)).astype(np.float16).astype(np.float32)
    d3 = dout.reshape(B, L, H, 2)
    got = np.zeros((go.total_entries, F_enc))
    for p in range(H):
        code = np.zeros((B, H), dtype=np.float32)
        code[:, p] = 1.0
        dtab_p, _, _ = ohg.ensemble_bwd(x, tab.view(np.uint16)[None], H, go, code,
                                        np.ascontiguousarray(d3[:, :, p, :]).reshape(B, L * 2))
        got += dtab_p[0]
    idx, w = ohg.indices(x, go)
    want = np.zeros((go.total_entries, F_enc))
    offs = np.array(go.offset[:L + 1])
    dl = dout.astype(np.float64).reshape(B, L, F_enc)
    for l in range(L):
        for c in range(8):
            wc = np.ones(B)
            for d in range(3):
                wc *= w[:, l, d] if (c >> d) & 1 else (1 - w[:, l, d])
            np.add.at(want, offs[l] + idx[:, l, c], wc[:, None] * dl[:, l, :])
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
