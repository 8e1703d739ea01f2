"""GPU parity for SURVEY rows a5 / a7: the field-level compositions and the glue kernels of csrc/field_glue.hip against
the CPU oracle (oracle/field.py), forward and backward.

  * ``nsx_sample_positions`` / ``nsx_normalise_bwd`` / ``nsx_density_fwd`` / ``nsx_density_bwd`` alone: bit-exact for the
    position / selector arithmetic (torch's op order, contraction off), <= 2 ulp for exp;
  * ``NeRSembleNeRFactoField.get_density`` (nersemble_nerfacto_field.py:250-301) and its autograd;
  * ``NeRSembleNGPModel.field_density_fn`` (nersemble_instant_ngp.py:235-266) incl. the reference's normalised-offset-
    on-world-position behaviour, the two embedding lookups and the timestep rounding.
Tolerances (stated per assert): hash features <= 1 fp16 ulp and mlp_base <= 4 fp16 ulp of the oracle, hence
density = exp(h0) within 8e-3 relative; gradients 5e-3 of their maximum (fp16 dZ inside the fused MLP, as in tcnn).
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import field as ofield, hashgrid as ohg
from tests.helpers import SMALL_GEOM_KW, make_smooth_tcnn_tables, make_tcnn_tables, randomise_model

pytestmark = pytest.mark.gpu
AABB = np.array([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]], dtype=np.float32)
FP16 = 2.0 ** -10


def _aabb6():
    import ctypes
    return (ctypes.c_float * 6)(*[float(v) for v in AABB.reshape(-1)])


def _positions(n, seed, frac_outside=0.15):
    rng = np.random.default_rng(seed)
    ext = AABB[1] - AABB[0]
    p = (rng.random((n, 3), dtype=np.float32) * ext + AABB[0]).astype(np.float32)
    k = int(n * frac_outside)
    p[:k] += (rng.standard_normal((k, 3)) * ext).astype(np.float32)          # some far outside
    if n >= 8:
        p[-1] = AABB[0]                                   # exactly on the lower corner: pn == 0 -> not selected
        p[-2] = AABB[1]                                   # exactly on the upper corner: pn == 1 -> not selected
        p[-3] = (AABB[0] + AABB[1]) / 2
        p[-4] = [AABB[0, 0], 0.0, 0.0]                    # on one face only
    return p


# ---------------------------------------------------------------------------------------------------------
# glue kernels alone
# ---------------------------------------------------------------------------------------------------------
def test_sample_positions_kernel_bit_exact(cuda):
    from nersemble_amd import functional as F
    rng = np.random.default_rng(0)
    R, S = 300, 5000
    o = (rng.standard_normal((R, 3)) * 3).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ri = np.sort(rng.integers(0, R, S)).astype(np.int64)
    t0 = (rng.random(S) * 9 + 0.2).astype(np.float32)
    t1 = (t0 + np.float32(0.011)).astype(np.float32)
    want = ofield.sample_positions(o, d, ri, t0, t1)
    tt = lambda a: torch.from_numpy(a).to(cuda)
    got = F.sample_positions(tt(o), tt(d), tt(t0), tt(t1), tt(ri)).cpu().numpy()          # gathered through ray ids
    assert np.array_equal(got, want)
    got2 = F.sample_positions(tt(o[ri]), tt(d[ri]), tt(t0)[:, None], tt(t1)[:, None]).cpu().numpy()   # per-sample rows
    assert np.array_equal(got2, want)


def test_positions_and_normalisation_in_one_launch(cuda):
    """``nsx_sample_positions`` asked for both outputs with offsets: ``pos_world`` is the sample position WITHOUT the
    offsets (what the deformation field's backward needs), ``pos_normalised`` / ``selector`` are those of position + offset
    -- bit for bit what the two separate launches give (engine/fused_pass.py takes this form when the offsets are known)."""
    import ctypes as C
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    rng = np.random.default_rng(4)
    R, S = 200, 3001
    box = np.stack([AABB[0], AABB[1]])
    o = (box[0] + (box[1] - box[0]) * rng.random((R, 3))).astype(np.float32)
    d = rng.standard_normal((R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    ri = np.sort(rng.integers(0, R, S)).astype(np.int64)
    t0 = (rng.random(S) * 2).astype(np.float32)
    t1 = (t0 + np.float32(0.011)).astype(np.float32)
    off = (rng.standard_normal((S, 3)) * 0.3).astype(np.float32)           # large enough to move samples across the faces
    tt = lambda a: torch.from_numpy(a).to(cuda)
    og, dg, t0g, t1g, offg = tt(o[ri]), tt(d[ri]), tt(t0), tt(t1), tt(off)     # (held: the call borrows raw pointers)
    pos_want = ofield.sample_positions(o, d, ri, t0, t1)
    pn_want, sel_want = ofield.normalise(pos_want + off, AABB)
    pos = torch.empty((S, 3), device=cuda)
    pn = torch.empty((S, 3), device=cuda)
    sel = torch.empty((S,), dtype=torch.uint8, device=cuda)
    check(lib().nsx_sample_positions(ptr(og), ptr(dg), None, ptr(t0g), ptr(t1g), ptr(offg), S, _aabb6(), ptr(pos),
                                     ptr(pn), ptr(sel), None, stream()), "nsx_sample_positions")
    assert np.array_equal(pos.cpu().numpy(), pos_want)                      # no offsets in the world position
    assert np.array_equal(pn.cpu().numpy(), pn_want) and np.array_equal(sel.cpu().numpy().astype(bool), sel_want)
    assert 0.1 < sel_want.mean() < 0.95 and np.abs(pos_want + off - pos_want).max() > 0.1


@pytest.mark.parametrize("with_offsets", [False, True])
def test_normalise_selector_kernel_bit_exact_and_backward(with_offsets, cuda):
    from nersemble_amd import functional as F
    S = 4099
    p = _positions(S, 1)
    rng = np.random.default_rng(2)
    off = (rng.standard_normal((S, 3)) * 0.05).astype(np.float32) if with_offsets else None
    if with_offsets:
        off[-4:] = 0
    pw = p + off if with_offsets else p
    pn_o, sel_o = ofield.normalise(pw, AABB)
    pt = torch.from_numpy(p).to(cuda).requires_grad_(True)
    ot = torch.from_numpy(off).to(cuda).requires_grad_(True) if with_offsets else None
    pn, sel = F.normalised_positions(pt, ot, _aabb6())
    assert np.array_equal(sel.cpu().numpy().astype(bool), sel_o)
    assert not sel_o[-1] and not sel_o[-2] and sel_o[-3] and not sel_o[-4]      # open interval (0, 1) on every axis
    assert 0.2 < sel_o.mean() < 0.95
    assert np.array_equal(pn.detach().cpu().numpy(), pn_o)
    g = rng.standard_normal((S, 3)).astype(np.float32)
    pn.backward(torch.from_numpy(g).to(cuda))
    want = (g * sel_o[:, None].astype(np.float32)) / (AABB[1] - AABB[0])
    assert np.array_equal(pt.grad.cpu().numpy(), want)
    if with_offsets:
        assert np.array_equal(ot.grad.cpu().numpy(), want)


def test_density_epilogue_forward_backward(cuda):
    """density = trunc_exp(h0.float()) * selector (:286-293); backward g * exp(clamp(h0, -15, 15)) in fp16."""
    from nersemble_amd import functional as F
    rng = np.random.default_rng(3)
    S = 3001
    base = (rng.standard_normal((S, 16)) * 2).astype(np.float16)
    base[:8, 0] = [-20, -15.5, -15, 0, 11, 15, 15.5, 20]
    sel = rng.random(S) < 0.8
    sel[:8] = True
    bt = torch.from_numpy(base).to(cuda).requires_grad_(True)
    st = torch.from_numpy(sel.astype(np.uint8)).to(cuda)
    dens = F.density_from_base(bt, st)
    want = np.exp(base[:, 0].astype(np.float32)) * sel.astype(np.float32)
    got = dens.detach().cpu().numpy()[:, 0]
    assert dens.dtype == torch.float32 and dens.shape == (S, 1)
    assert (np.abs(got - want) <= 5e-7 * np.abs(want)).all(), float((np.abs(got - want) / np.maximum(want, 1e-30)).max())   # a few ulp of fp32 exp
    assert (got[~sel] == 0).all()
    g = (rng.standard_normal(S) * 1e-3).astype(np.float32)
    g[0], g[7] = 1000.0, 1e-3                   # exp(-15) * 1000 and exp(15) * 1e-3 are normal fp16 numbers
    dens.backward(torch.from_numpy(g).to(cuda)[:, None])
    db = bt.grad.float().cpu().numpy()
    want_b = (g * sel.astype(np.float32) * np.exp(np.clip(base[:, 0].astype(np.float32), -15, 15)))
    assert (db[:, 1:] == 0).all()
    assert (np.abs(db[:, 0] - want_b.astype(np.float16).astype(np.float32)) <= FP16 * np.abs(want_b) + 1e-7).all()
    # the clamp: h0 = 20 gets the gradient of h0 = 15
    assert np.isclose(db[7, 0] / g[7], np.exp(15.0), rtol=2e-3) and np.isclose(db[0, 0] / g[0], np.exp(-15.0), rtol=2e-3)


# ---------------------------------------------------------------------------------------------------------
# a5: NeRSembleNeRFactoField.get_density, forward + backward
# ---------------------------------------------------------------------------------------------------------
def _field(H, cuda, seed, smooth=False):
    from nersemble_amd.field_components.hash_ensemble import HashEnsembleConfig, TCNNHashEncodingConfig
    from nersemble_amd.fields.nersemble_nerfacto_field import NeRSembleNeRFactoField
    from oracle import mlp as omlp
    go = oracle.grid_geometry(**SMALL_GEOM_KW)
    cfg = HashEnsembleConfig(n_hash_encodings=H, hash_encoding_config=TCNNHashEncodingConfig(log2_hashmap_size=15),
                             disable_initial_hash_ensemble=True, use_soft_transition=True)
    fld = NeRSembleNeRFactoField(torch.from_numpy(AABB), num_images=4, hash_ensemble_config=cfg,
                                 max_n_samples_per_batch=512)          # several chunks
    tabs = (make_smooth_tcnn_tables if smooth else make_tcnn_tables)(H, go, seed, 0.5)
    rng = np.random.default_rng(seed + 7)
    base = (rng.uniform(-1, 1, omlp.param_count(0)) * np.sqrt(6.0 / 96)).astype(np.float16).astype(np.float32)
    head = (rng.uniform(-1, 1, omlp.param_count(1)) * np.sqrt(6.0 / 128)).astype(np.float16).astype(np.float32)
    sd = {f"hash_ensemble.hash_encodings.{c}.params": torch.from_numpy(tabs[c].reshape(-1)) for c in range(tabs.shape[0])}
    sd["mlp_base.params"], sd["mlp_head.params"] = torch.from_numpy(base), torch.from_numpy(head)
    missing, unexpected = fld.load_state_dict(sd, strict=False)
    assert not unexpected
    return fld.to(cuda), go, tabs.astype(np.float16).view(np.uint16), base, head


def _ray_samples(p, d, off, codes, code_index, cuda):
    from nersemble_amd.rays import Frustums, RaySamples
    tt = lambda a: torch.from_numpy(a).to(cuda)
    S = p.shape[0]
    zeros = torch.zeros((S, 1), device=cuda)
    fr = Frustums(origins=tt(p), directions=tt(d), starts=zeros, ends=zeros, pixel_area=zeros + 1)
    if off is not None:
        fr.set_offsets(off)
    md = {"time_codes": codes}
    if code_index is not None:
        md["time_code_index"] = code_index
    return RaySamples(frustums=fr, camera_indices=torch.zeros((S, 1), dtype=torch.long, device=cuda), metadata=md)


@pytest.mark.parametrize("H,window,use_index", [(1, None, False), (4, 3.25, False), (16, None, True), (16, 1.0, True),
                                                (16, 1.5, False), (32, 8.75, True)])
def test_get_density_forward_backward_vs_oracle(H, window, use_index, cuda):
    fld, go, tabs_u16, base, head = _field(H, cuda, 20 + H)
    S, T = 1531, 11
    p = _positions(S, 30 + H)
    rng = np.random.default_rng(40 + H)
    off = (rng.standard_normal((S, 3)) * 0.02).astype(np.float32)
    d = rng.standard_normal((S, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    table = (rng.standard_normal((T, H)) * 0.6).astype(np.float32)
    slot = rng.integers(0, T, S).astype(np.int64)
    code = table[slot]
    off_t = torch.from_numpy(off).to(cuda).requires_grad_(True)
    if use_index:
        codes_t = torch.from_numpy(table).to(cuda).requires_grad_(True)
        rs = _ray_samples(p, d, off_t, codes_t, torch.from_numpy(slot).to(cuda).int(), cuda)
    else:
        codes_t = torch.from_numpy(code).to(cuda).requires_grad_(True)
        rs = _ray_samples(p, d, off_t, codes_t, None, cuda)
    density, emb = fld.get_density(rs, window_hash_encodings=window)
    assert density.shape == (S, 1) and density.dtype == torch.float32 and emb.shape == (S, 15) and emb.dtype == torch.float16

    codew = ohg.windowed_code(code, H, window)
    fwd = ofield.get_density(p + off, AABB, tabs_u16, H, go, codew, base)
    sel = fwd["selector"]
    got_d = density.detach().cpu().numpy()[:, 0]
    assert (got_d[~sel] == 0).all() and (fwd["density"][~sel] == 0).all()
    h0 = np.abs(fwd["base"][:, 0].astype(np.float32))
    assert (np.abs(got_d - fwd["density"]) <= 8e-3 * np.maximum(1.0, h0) * fwd["density"] + 1e-12).all()
    want_e = fwd["base"][:, 1:].astype(np.float32)
    assert np.abs(emb.detach().float().cpu().numpy() - want_e).max() <= 4 * FP16 * max(1.0, np.abs(want_e).max())

    # backward: upstream gradients of the size the compositing hands down (density gradient scaled so that the fp16
    # gradient entering mlp_base neither underflows nor overflows)
    gd = (rng.standard_normal(S) * 0.05 / np.maximum(fwd["density"], 1e-3)).astype(np.float32)
    ge = (rng.standard_normal((S, 15)) * 0.05).astype(np.float16)
    ((density[:, 0] * torch.from_numpy(gd).to(cuda)).sum()
     + (emb.float() * torch.from_numpy(ge.astype(np.float32)).to(cuda)).sum()).backward()
    bwd = ofield.get_density_bwd(fwd, AABB, tabs_u16, H, go, codew, base, gd, ge)

    def close(got, want, rel, what):
        sc = np.abs(want).max()
        assert sc > 0, what
        err = np.abs(got - want).max()
        assert err <= rel * sc, (what, float(err / sc))

    close(fld.mlp_base.params.grad.cpu().numpy(), bwd["d_params"], 5e-3, "mlp_base.params")
    close(off_t.grad.cpu().numpy(), bwd["d_positions"], 5e-3, "offsets (positions)")
    from nersemble_amd import functional as F
    dt = F.tables_to_tcnn(fld.hash_ensemble.tables.grad, H, fld.hash_ensemble.geom).cpu().numpy()
    close(dt, bwd["d_table"], 5e-3, "hash tables")
    # gradient w.r.t. the conditioning code (the reference's chain: window multiply, soft transition / ones)
    dcw = bwd["d_codew"]
    if window is None:
        dcode = dcw
    elif window == 1.0:
        dcode = None                                   # code replaced by ones: no gradient reaches it
    else:
        win = ohg.posenc_window(window, 0, H - 1, H)
        dcode = dcw * win[None, :] * (np.float32(window - 1) if window < 2 else np.float32(1.0))
    if dcode is None:
        assert codes_t.grad is None or float(codes_t.grad.abs().max()) == 0.0
    else:
        if use_index:
            want_c = np.zeros((T, H), np.float64)
            np.add.at(want_c, slot, dcode.astype(np.float64))
        else:
            want_c = dcode
        close(codes_t.grad.cpu().numpy(), want_c, 5e-3, "time codes")


def test_get_outputs_rgb_vs_oracle(cuda):
    fld, go, tabs_u16, base, head = _field(4, cuda, 77)
    S = 700
    p = _positions(S, 5)
    rng = np.random.default_rng(6)
    d = rng.standard_normal((S, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    code = (rng.standard_normal((S, 4)) * 0.6).astype(np.float32)
    rs = _ray_samples(p, d, None, torch.from_numpy(code).to(cuda), None, cuda)
    with torch.no_grad():
        out = fld(rs, window_hash_encodings=None)
    from nersemble_amd.fields.nersemble_nerfacto_field import FieldHeadNames
    fwd = ofield.get_density(p, AABB, tabs_u16, 4, go, code, base)
    rgb = ofield.get_rgb(d, fwd["base"], head)
    got = out[FieldHeadNames.RGB]
    assert got.dtype == torch.float32 and got.shape == (S, 3)
    # sigmoid output in (0,1): 4 fp16 ulp of 1 on top of the <= 4 ulp of the geometry features it reads
    assert np.abs(got.cpu().numpy() - rgb).max() <= 8 * FP16


# ---------------------------------------------------------------------------------------------------------
# a7: NeRSembleNGPModel.field_density_fn
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,with_offsets,S", [(16, True, 5000), (32, False, 1), (4, True, 33), (16, True, 70001)])
def test_fused_density_equals_the_four_launches(H, with_offsets, S, cuda):
    """nsx_density_fused_fwd (normalise + selector -> pre-blended lookup -> mlp_base -> trunc_exp in one launch, features in
    registers) against the route it replaces -- nsx_sample_positions + nsx_hashgrid_fwd(F = 2) + nsx_mlp_fwd + nsx_density_fwd,
    themselves held to the oracle above: the same arithmetic operation by operation, so density AND the 16-wide mlp_base row
    are equal bit for bit (incl. samples outside the box, on its faces, ragged tiles, a device-side count)."""
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    fld, go, _, _, _ = _field(H, cuda, seed=H + S)
    fld.max_n_samples_per_batch = -1
    he = fld.hash_ensemble
    rng = np.random.default_rng(S)
    code = torch.from_numpy(rng.standard_normal(H).astype(np.float32) * 0.7).to(cuda)
    p = torch.from_numpy(_positions(S, seed=S)).to(cuda)
    off = (torch.from_numpy(rng.standard_normal((S, 3)).astype(np.float32) * 0.02).to(cuda)) if with_offsets else None
    with torch.no_grad():
        table = he.preblend(code, window_hash_encodings=None)
        md = {"preblended_table": table}
        fld.fused_eval_density = False
        d_ref, _ = fld._density_from_positions(p, off, md, None)
        base_ref = fld._base_out.clone()
        fld.fused_eval_density = True
        d_got, _ = fld._density_from_positions(p, off, md, None)
        base_got = fld._base_out.clone()
    assert d_ref.shape == d_got.shape == (S, 1) and base_got.shape == (S, 16)
    assert torch.equal(base_got.view(torch.int16), base_ref.view(torch.int16))
    assert torch.equal(d_got.view(torch.int32), d_ref.view(torch.int32))
    if S >= 64:
        assert (d_got == 0).any() and (d_got > 0).any()                  # both sides of the box are in the batch
        # a device-side count: only the first rows are written
        n_dev = torch.tensor([S // 2 + 3], dtype=torch.int64, device=cuda)
        dens = torch.full((S, 1), -1.0, device=cuda)
        bo = torch.full((S, 16), -1.0, device=cuda, dtype=torch.float16)
        check(lib().nsx_density_fused_fwd(ptr(p), ptr(off), S, fld._aabb6(), ptr(table), he.geom, ptr(fld.mlp_base.half_weights()),
                                          fld.mlp_base.n_hidden_mats, ptr(bo), 16, ptr(dens), ptr(n_dev), stream()), "fused")
        k = S // 2 + 3
        assert torch.equal(dens[:k], d_ref[:k]) and bool((dens[k:] == -1).all()) and bool((bo[k:] == -1).all())
    # wrong geometry / layout is refused
    rc = lib().nsx_density_fused_fwd(ptr(p), None, S, fld._aabb6(), ptr(table), he.geom, ptr(fld.mlp_base.half_weights()), 0,
                                     None, 0, None, None, stream())
    assert rc != 0


def _model(name, cuda, seed, log2=15):
    from nersemble_amd.models.nersemble_instant_ngp import NeRSembleNGPModel
    from nersemble_amd.rays import SceneBox
    from nersemble_amd.workloads import SCENE_BOXES, WORKLOADS, build_model_config
    w = WORKLOADS[name]
    cfg = build_model_config(w, small=True)
    cfg.hash_ensemble_config.hash_encoding_config.log2_hashmap_size = log2
    box = torch.tensor(SCENE_BOXES[w["pid"]], dtype=torch.float32)
    torch.manual_seed(seed)
    model = NeRSembleNGPModel(cfg, SceneBox(box), num_train_data=12 * w["T"])
    go = oracle.grid_geometry(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16, log2_hashmap_size=log2)
    weights = randomise_model(model, seed, go)
    return model.to(cuda), go, weights, w


@pytest.mark.parametrize("step", [0, 41000, 50000, 90000])
def test_field_density_fn_vs_oracle(step, cuda):
    """Windows from the schedulers at ``step``: hash window 1 (codes := 1), 1.375 (soft transition), 4.75, 16; the
    deformation window 0 .. 7."""
    model, go, W, w = _model("p030_h16", cuda, 11)
    H, T = w["H"], w["T"]
    model.train()
    for sched in (model.sched_window_hash_encodings, model.sched_window_deform):
        sched.update(step)
    wh, wd = model.sched_window_hash_encodings.value, model.sched_window_deform.value
    N = 3001
    pos = _positions(N, 9, frac_outside=0.1)
    rng = np.random.default_rng(10)
    ts = rng.integers(0, T, N)
    times = (ts / np.float32(T - 1)).astype(np.float32)[:, None]
    model.field.keep_density_intermediates = True
    with torch.no_grad():
        got = model.field_density_fn(torch.from_numpy(pos).to(cuda), torch.from_numpy(times).to(cuda))
    gpu_offsets = model._sigma_cache["offsets"].cpu().numpy()
    model.field.keep_density_intermediates = False
    got = got.reshape(-1).cpu().numpy()
    kw = dict(deform_params=W["deform_params"], deform_embedding=W["deform_embedding"], window_hash=wh, window_deform=wd)
    want, info = ofield.field_density_fn(pos, times, T, W["aabb"], W["tables_u16"], H, go, W["mlp_base"],
                                         W["time_embedding"], **kw)
    assert np.array_equal(info["timesteps"], ts)                              # round(t / (T-1) * (T-1)) recovers t
    # (1) the deformation stage inside the composition: normalised-space offsets within the kernel's tolerance
    sc = np.abs(info["offsets"]).max()
    assert sc > 1e-2 and np.abs(gpu_offsets - info["offsets"]).max() <= 3e-3 * sc + 2e-5
    # (2) everything after it, evaluated on the GPU's own offsets: the world-position + normalised-offset sum, the
    #     selector, codes / windows, HashEnsemble, mlp_base, trunc_exp -- tight
    pos2 = pos + gpu_offsets
    codew = info["codew"]
    fwd = ofield.get_density(pos2, W["aabb"], W["tables_u16"], H, go, codew, W["mlp_base"])
    h0 = np.abs(fwd["base"][:, 0].astype(np.float32))
    # samples whose normalised coordinate sits within fp32 noise of the box faces may flip the selector
    pn_raw = (pos2 - W["aabb"][0]) / (W["aabb"][1] - W["aabb"][0])
    edge = (np.abs(pn_raw) < 1e-6).any(1) | (np.abs(pn_raw - 1) < 1e-6).any(1)
    ok = np.abs(got - fwd["density"]) <= 8e-3 * np.maximum(1.0, h0) * fwd["density"] + 1e-12
    assert (ok | edge).all(), float(np.abs(got - fwd["density"])[~edge].max())
    # (3) end to end against the oracle's own offsets (smooth "trained-like" tables keep the conditioning bounded)
    both = info["selector"] & fwd["selector"]
    rel = np.abs(got - want)[both] / np.maximum(want[both], 1e-6)
    assert np.median(rel) <= 5e-3 and rel.max() <= 0.1, (float(np.median(rel)), float(rel.max()))
    assert 0.5 < both.mean()


def test_field_density_fn_dense_march_returns_ones(cuda):
    """disable_occupancy_grid: ones [N] (not [N,1]) without touching the networks (nersemble_instant_ngp.py:239-240)."""
    model, go, W, w = _model("p097_dense", cuda, 3, log2=12)
    pos = torch.from_numpy(_positions(100, 1)).to(cuda)
    out = model.field_density_fn(pos, torch.zeros((100, 1), device=cuda))
    assert out.shape == (100,) and bool((out == 1).all())
