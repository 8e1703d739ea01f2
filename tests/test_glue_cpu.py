"""CPU: reference-owned glue mirrored in nersemble_amd (pure torch parts) vs golden vectors produced by the
reference's own Python (tests/golden/make_golden.py)."""
import numpy as np
import torch


def test_se3_exp_map_matches_reference(golden_dir):
    from nersemble_amd.util.se3 import se3_exp_map
    z = np.load(f"{golden_dir}/deformation.npz")
    got = se3_exp_map(torch.from_numpy(z["se3_in"])).numpy()
    assert np.abs(got - z["se3_out"]).max() <= 2e-6


def test_windowed_encoding_matches_reference(golden_dir):
    from nersemble_amd.field_components.windowed_nerf_encoding import WindowedNeRFEncoding
    z = np.load(f"{golden_dir}/deformation.npz")
    enc = WindowedNeRFEncoding(3, 7, 0.0, 6.0, include_input=True)
    assert enc.get_out_dim() == 45
    for i, w in enumerate(z["pe_windows"]):
        w = None if np.isnan(w) else float(w)
        got = enc(torch.from_numpy(z["pe_x"]), windows_param=w).numpy()
        assert np.abs(got - z[f"pe_out_{i}"]).max() <= 1e-6
    out = enc(torch.tensor([[0.1, 0.2, 0.3]]), windows_param=3.5)[0]
    assert torch.allclose(out[-3:], torch.tensor([0.6283, 1.2566, 1.8850]), atol=1e-4)   # SURVEY 8c known answer


def test_deformation_offsets_match_reference(golden_dir):
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    z = np.load(f"{golden_dir}/deformation.npz")
    cfg = SE3DeformationFieldConfig(warp_code_dim=8, mlp_num_layers=6, mlp_layer_width=32)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, cfg, max_n_samples_per_batch=17)
    sd = {k[len("df_sd_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("df_sd_")}
    assert set(sd) == set(df.state_dict())            # same state-dict keys as the reference module
    df.load_state_dict(sd)
    shapes = [tuple(l.weight.shape) for l in df.se3_field.mlp_stem.layers]
    assert shapes == [(32, 53), (32, 32), (32, 32), (32, 32), (32, 85), (32, 32)]
    for i, w in enumerate(z["df_windows"]):
        w = None if np.isnan(w) else float(w)
        with torch.no_grad():
            got = df.compute_offsets(torch.from_numpy(z["df_pos"]), torch.from_numpy(z["df_code"]), w).numpy()
        assert np.abs(got - z[f"df_off_{i}"]).max() <= 5e-6, i


def test_scheduler_and_chunker_match_reference(golden_dir):
    from nersemble_amd.engine.generic_scheduler import GenericScheduler
    from nersemble_amd.util.chunker import chunked
    z = np.load(f"{golden_dir}/misc.npz")
    s = GenericScheduler(init_value=1, final_value=32, begin_step=40000, end_step=80000)
    vals = []
    for st in z["sched_steps"]:
        s.update(int(st))
        vals.append(s.get_value())
    assert np.array_equal(np.array(vals, dtype=np.float64), z["sched_vals"])
    s.eval()
    assert s.get_value() == z["sched_eval"][0]
    a, b = torch.arange(10), torch.arange(20).reshape(10, 2)
    sizes = [[len(ca), len(cb), int(ca[0]), int(cb[0, 0])] for ca, cn, cb in chunked(4, a, None, b)]
    assert np.array_equal(np.array(sizes), z["chunk_sizes"])
    assert [len(c) for c in chunked(3, a)] == z["chunk_single"].tolist()


def test_full_size_deformation_shapes():
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    cfg = SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6, mlp_layer_width=128)
    df = SE3DeformationField(torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]]), cfg)
    shapes = [tuple(l.weight.shape) for l in df.se3_field.mlp_stem.layers]
    assert shapes == [(128, 173), (128, 128), (128, 128), (128, 128), (128, 301), (128, 128)]
    assert sum(p.numel() for p in df.parameters()) == 127756


def test_base_model_losses_match_reference(golden_dir):
    """Masked RGB MSE, alpha L1, empty / near (per-ray accumulated weights, Normal CDF with sigma=(eps/3)^2) and depth
    losses vs values computed by the reference's own BaseModel (models/base.py:90-222)."""
    import types
    from nersemble_amd.models.base import BaseModel, BaseModelConfig
    from nersemble_amd.engine.generic_scheduler import GenericScheduler
    z = np.load(f"{golden_dir}/misc.npz")
    m = BaseModel.__new__(BaseModel)
    torch.nn.Module.__init__(m)
    m.config = BaseModelConfig(use_masked_rgb_loss=True, alpha_mask_threshold=0, lambda_alpha_loss=1e-2,
                               lambda_empty_loss=1e-2, lambda_near_loss=1e-4, lambda_depth_loss=1e-4)
    m.sched_eps_depth = GenericScheduler(init_value=0.9, final_value=0.01, begin_step=0, end_step=10000)
    m.sched_eps_depth.update(2500)
    assert m.sched_eps_depth.value == z["ls_eps"][0]
    m.train()
    batch = {"image": torch.from_numpy(z["ls_image"]), "alpha_map": torch.from_numpy(z["ls_alpha"]),
             "depth_maps": torch.from_numpy(z["ls_depth"])}
    fr = types.SimpleNamespace(starts=torch.from_numpy(z["ls_starts"])[:, None], ends=torch.from_numpy(z["ls_ends"])[:, None])
    rs = types.SimpleNamespace(frustums=fr)
    near, empty = m.get_near_and_empty_loss(batch, rs, torch.from_numpy(z["ls_ray_idx"]),
                                            torch.from_numpy(z["ls_weights"]), torch.from_numpy(z["ls_acc"]))
    assert abs(float(near) - z["ls_near"][0]) <= 1e-6 * abs(z["ls_near"][0]) + 1e-12
    assert abs(float(empty) - z["ls_empty"][0]) <= 1e-6 * abs(z["ls_empty"][0]) + 1e-12
    rgb = m.get_masked_rgb_loss(batch, torch.from_numpy(z["ls_rgb_pred"]))
    assert abs(float(rgb) - z["ls_rgb"][0]) <= 1e-6
    al = m.get_alpha_loss(batch, torch.from_numpy(z["ls_acc"]))
    assert abs(float(al) - z["ls_alpha_loss"][0]) <= 1e-7
    dl = m.get_depth_loss(batch, torch.from_numpy(z["ls_depth_pred"]))
    assert abs(float(dl) - z["ls_depth_loss"][0]) <= 1e-8


def test_frustum_matches_reference(golden_dir):
    """TorchFrustum(cam_to_world, intrinsics, image_dimensions) vs the reference's numpy Frustum on 4 random OpenCV
    poses x 4096 points: identical containment masks, single-point ``contains`` and the >=k-views culling count."""
    from nersemble_amd.model_components.frustum import TorchFrustum, visibility_grid
    z = np.load(f"{golden_dir}/dataformat.npz")
    pts = torch.from_numpy(z["fr_points"])
    dims = tuple(int(v) for v in z["fr_dims"])
    frusta = [TorchFrustum(torch.from_numpy(p), torch.from_numpy(k), dims) for p, k in zip(z["fr_pose"], z["fr_k"])]
    for i, fr in enumerate(frusta):
        assert fr.normals.dtype == torch.float64 and torch.allclose(fr.normals.norm(dim=1), torch.ones(4).double())
        assert np.array_equal(fr.contains_points(pts).numpy(), z["fr_mask"][i])
        assert [fr.contains(p) for p in pts[:16]] == z["fr_single"][i].tolist()
        # float32 points are promoted to the planes' float64 exactly like the reference's ``points - offsets``
        d32 = fr.signed_distances(pts.float())
        assert d32.dtype == torch.float64
    # an OpenGL pose (camera looks along -z, y up) through the convenience constructor == its OpenCV twin
    gl = torch.from_numpy(z["fr_pose"][0]).clone()
    gl[:3, 1:3] = -gl[:3, 1:3]
    k = z["fr_k"][0]
    twin = TorchFrustum.from_camera(gl, k[0, 0], k[1, 1], k[0, 2], k[1, 2], *dims)
    assert np.array_equal(twin.contains_points(pts).numpy(), z["fr_mask"][0])
    # the sampler's culling lattice: linspace end points included, >= 2 views
    aabb = torch.tensor([[-3.0, -3.0, -3.0], [3.0, 3.0, 3.0]])
    grid = visibility_grid(frusta, aabb, (9, 8, 7), 2, "cpu")
    lin = [torch.linspace(-3, 3, n) for n in (9, 8, 7)]
    lattice = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1).reshape(-1, 3)
    want = sum(f.contains_points(lattice).int() for f in frusta) >= 2
    assert grid.shape == (9, 8, 7) and torch.equal(grid.reshape(-1), want)


def test_quantizers_match_reference(golden_dir):
    """Depth / normal map codecs (data format on the input side of the path) vs the reference's util/quantization.py:
    bit-identical codes, identical decoded float32 values, known answers for the reserved mask code."""
    from nersemble_amd.util.quantization import DepthQuantizer, NormalsQuantizer, Quantizer
    z = np.load(f"{golden_dir}/dataformat.npz")
    dq = DepthQuantizer()
    depth = z["dq_in"].copy()
    codes = dq.encode(depth)
    assert codes.dtype == np.uint16 and np.array_equal(codes, z["dq_codes"])
    assert (depth[z["dq_in"] > 2] == 0).all()                      # outliers masked in place, like the reference
    dec = dq.decode(codes)
    assert dec.dtype == z["dq_decoded"].dtype and np.array_equal(dec, z["dq_decoded"])
    assert np.array_equal(dq.decode(z["dq_known_codes"]), z["dq_known"])
    assert dq.decode(np.array([[0, 1, 65535]], dtype=np.uint16)).tolist() == [[0.0, 0.0, 2.0]]
    q8 = Quantizer(min_values=-1.0, max_values=3.0, bits=8, separate_mask=False)
    c8 = q8.encode(z["q8_in"].copy())
    assert c8.dtype == np.uint8 and np.array_equal(c8, z["q8_codes"])
    assert np.array_equal(q8.decode(c8), z["q8_decoded"])
    nq = NormalsQuantizer()
    nc = nq.encode(z["nq_in"].copy())
    assert np.array_equal(nc, z["nq_codes"])
    nd = nq.decode(nc)
    assert np.array_equal(nd, z["nq_decoded"])
    present = (z["nq_in"] != 0).any(-1)
    assert np.abs(nd[present] - z["nq_in"][present]).max() < 0.03  # 8-bit angles
    assert (nd[~present] == 0).all()
    try:
        dq.encode(np.array([[-0.0, 1.0], [2.0, 5.0]], dtype=np.float32))   # > max is masked, not an error
    except AssertionError:
        raise AssertionError("DepthQuantizer must mask out-of-range depth")


def test_step_lr_matches_torch_scheduler():
    """The trainer's four-line StepLR follows torch.optim.lr_scheduler.StepLR (train_nersemble.py:243-256: 20 k-step
    stairs, gamma 0.8 / 0.5) exactly, including after a state-dict round trip."""
    from nersemble_amd.engine.trainer import StepLR
    for gamma, size in ((0.8, 20), (0.5, 7)):
        pa, pb = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
        oa, ob = torch.optim.Adam([pa], lr=5e-3), torch.optim.Adam([pb], lr=5e-3)
        mine, ref = StepLR(oa, step_size=size, gamma=gamma), torch.optim.lr_scheduler.StepLR(ob, size, gamma)
        for step in range(65):
            oa.step(), ob.step()
            mine.step(), ref.step()
            assert abs(mine.get_last_lr()[0] - ref.get_last_lr()[0]) <= 1e-12 * 5e-3 + 1e-18, (step, gamma)
            if step == 30:
                state = mine.state_dict()
                oa = torch.optim.Adam([pa], lr=5e-3)
                mine = StepLR(oa, step_size=size, gamma=gamma)
                mine.load_state_dict(state)
        assert mine.last_epoch == 65


def test_native_grad_scaler_follows_torch_gradscaler():
    """NativeGradScaler (device-resident scale / growth tracker, flags summed over the parameter groups) against
    torch.amp.GradScaler's rule (nersemble_trainer.py:185-203): x0.5 on a step with inf/NaN, x2 after growth_interval
    clean steps, 65536 to start with."""
    from nersemble_amd.engine.hash_adam import NativeGradScaler
    sc = NativeGradScaler("cpu", growth_interval=5)
    assert sc.get_scale() == 65536.0
    want, clean = 65536.0, 0
    pattern = [0, 0, 0, 0, 0, 0, 1, 0, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1]
    for flag in pattern:
        groups = [torch.tensor([float(flag > 0)]), torch.tensor([float(flag > 1)]), torch.zeros(1)]
        total = sc.update(groups)
        assert float(total) == float(flag)
        if flag:
            want, clean = want * 0.5, 0
        else:
            clean += 1
            if clean == 5:
                want, clean = want * 2.0, 0
        assert sc.get_scale() == want
    loss = torch.tensor(0.25)
    assert float(sc.scale(loss)) == 0.25 * want
    assert float(sc.inv_scale()) == pytest_approx(1.0 / want)
    off = NativeGradScaler("cpu", enabled=False)
    off.update([torch.ones(1)])
    assert off.get_scale() == 1.0 and float(off.scale(loss)) == 0.25


def pytest_approx(v):
    import pytest
    return pytest.approx(v, rel=1e-7)


def test_zeros_many_carves_one_buffer():
    """functional.zeros_many: zero tensors of mixed dtypes / shapes out of one allocation, none overlapping."""
    import torch
    from nersemble_amd.functional import zeros_many
    specs = [((5, 3), torch.float32), ((7,), torch.float16), ((0, 4), torch.int64), ((2, 2, 2), torch.uint8)]
    outs = zeros_many(specs, "cpu")
    assert [tuple(t.shape) for t in outs] == [tuple(s) for s, _ in specs]
    assert [t.dtype for t in outs] == [d for _, d in specs]
    base = outs[0].untyped_storage().data_ptr()
    assert all(t.untyped_storage().data_ptr() == base for t in outs)             # one buffer
    spans = sorted((t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in outs if t.numel())
    assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))                   # disjoint
    # offsets inside the buffer (a zero-element view has no defined data pointer: torch may answer nullptr)
    assert all((t.data_ptr() - base) % 256 == 0 for t in outs if t.numel())
    for i, t in enumerate(outs):
        assert not t.any()
        t.fill_(i + 1)
    for i, t in enumerate(outs):
        assert (t == i + 1).all()


def test_prefetched_march_bookkeeping():
    """OccGridEstimator keeps at most two counting passes ahead and hands one out only for exactly its call."""
    import torch
    from nersemble_amd.nerfacc import OccGridEstimator
    grid = OccGridEstimator(roi_aabb=torch.tensor([-1., -1, -1, 1, 1, 1]), resolution=16, levels=1)
    grid._prefetched = [{"key": 1}, {"key": 2}]
    assert grid._take_prefetched(3) is None and len(grid._prefetched) == 2
    assert grid._take_prefetched(2)["key"] == 2 and [p["key"] for p in grid._prefetched] == [1]
    assert grid._take_prefetched(2) is None
    assert grid._take_prefetched(1)["key"] == 1 and not grid._prefetched
    o = torch.zeros((4, 3))
    k1 = grid._march_key(o, o, 0.2, 1e3, 0.01, True, None)
    torch.autograd.graph.increment_version(grid.binaries)                        # what the native grid update does
    assert grid._march_key(o, o, 0.2, 1e3, 0.01, True, None) != k1


def test_march_stash_capacity_and_validity():
    """The prefetched counting pass keeps the samples' starts (``nsx_march_count_stash``): how many per ray, and when the step
    may use them (every ray fits: the overflow word that travels with the total is zero)."""
    import torch
    from nersemble_amd.nerfacc import OccGridEstimator
    grid = OccGridEstimator(roi_aabb=torch.tensor([-2.5, -1.8, -2.5, 2.2, 1.8, 2.0]), resolution=16, levels=1)
    diag = (4.7 ** 2 + 3.6 ** 2 + 4.5 ** 2) ** 0.5
    cap = grid._stash_cap(0.011)
    assert cap % 64 == 0 and diag / 0.011 + 4 <= cap < diag / 0.011 + 4 + 64          # the box diagonal bounds a unit ray
    assert grid._stash_cap(1e-4) == 0                                                 # beyond march_stash_max: no stash
    grid.march_stash = False
    assert grid._stash_cap(0.011) == 0
    stash = torch.zeros((4, 64))
    ok = {"stash": stash, "cap": 64, "total_host": torch.tensor([10, 0])}
    assert OccGridEstimator._stash_of(ok) == (stash, 64)
    assert OccGridEstimator._stash_of(dict(ok, total_host=torch.tensor([10, 1]))) is None   # some ray has more than cap
    assert OccGridEstimator._stash_of(dict(ok, stash=None)) is None and OccGridEstimator._stash_of(None) is None


def test_first_grid_planes_policy(monkeypatch):
    """``HashEnsemble.first_grid_planes``: P planes when the batch has more code rows than that, else one per row (0)."""
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble
    fgp = HashEnsemble.first_grid_planes
    he = type("H", (), {"first_grid_planes_default": HashEnsemble.first_grid_planes_default})()
    monkeypatch.delenv("NSX_FIRST_GRID_PLANES", raising=False)
    assert HashEnsemble.first_grid_planes_default == 2
    assert fgp(he, 24, 100000) == 2 and fgp(he, 2, 100000) == 0 and fgp(he, 1, 5) == 0
    he.first_grid_planes_default = 0
    assert fgp(he, 24, 100000) == 0
    monkeypatch.setenv("NSX_FIRST_GRID_PLANES", "4")
    assert fgp(he, 24, 100000) == 4 and fgp(he, 3, 100000) == 0


def test_small_group_adam_takes_the_count_back_for_tensors_the_device_left_alone():
    """``SmallGroupAdam.rollback_params`` (host side of ``nsx_multi_adam_present``): only the named tensors lose the step, and
    a ``rollback_step`` of the same step (inf / NaN found as well) does not take a second count from them."""
    import torch
    from nersemble_amd.engine.small_adam import SmallGroupAdam
    ps = [torch.nn.Parameter(torch.zeros(3)) for _ in range(3)]
    opt = SmallGroupAdam(ps)
    for _ in range(2):
        opt._last_stepped = []
        for p in ps:
            opt.advance(p)
    assert [opt.steps[p] for p in ps] == [2, 2, 2]
    opt.rollback_params([ps[1]])
    assert [opt.steps[p] for p in ps] == [2, 1, 2]
    opt.rollback_step()
    assert [opt.steps[p] for p in ps] == [1, 1, 1]
    opt.rollback_params([ps[0]])                              # nothing stepped since: nothing to take back
    assert [opt.steps[p] for p in ps] == [1, 1, 1]
