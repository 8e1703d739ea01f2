"""GPU parity: ray marching (bit-exact), per-ray scans, accumulation, distortion loss vs the C oracle."""
import numpy as np
import pytest
import torch

from oracle import march as om

pytestmark = pytest.mark.gpu

P30 = np.array([-2.5, -1.8, -2.5, 2.2, 1.8, 2.0], np.float32)
P97 = np.array([-2.2, -2.8, -2.5, 2.2, 2.2, 2.0], np.float32)


def _rays(R, seed, axis_aligned=False):
    rng = np.random.default_rng(seed)
    o = rng.standard_normal((R, 3)).astype(np.float32)
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * 9
    tgt = (rng.random((R, 3)).astype(np.float32) - 0.5) * 3
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    if axis_aligned:                      # zero direction components + rays that miss the box
        d[:8] = 0
        d[:8, 0] = 1
        o[:8] = [-9, 0.1, 0.2]
        o[8:16] = [50, 50, 50]
    return o, d.astype(np.float32)


def _grid(res, kind, seed=0):
    if kind == "ones":
        return np.ones((res, res, res), bool)
    if kind == "empty":
        return np.zeros((res, res, res), bool)
    g = np.stack(np.meshgrid(*[np.linspace(0, 1, res, endpoint=False) + 0.5 / res] * 3, indexing="ij"), -1)
    if kind == "shell":
        r = np.linalg.norm((g - 0.5) / [0.25, 0.3, 0.3], axis=-1)
        return (r < 1.0) & (r > 0.8)
    return np.random.default_rng(seed).random((res, res, res)) < 0.07


def _est(aabb, binary, cuda):
    from nersemble_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(torch.from_numpy(aabb), resolution=binary.shape[0], levels=1).to(cuda)
    est.binaries = torch.from_numpy(binary)[None].to(cuda)
    return est


@pytest.mark.parametrize("kind,aabb,res", [("ones", P97, 128), ("shell", P30, 128), ("random", P30, 64),
                                           ("empty", P30, 32), ("shell", P30, 96)])
def test_march_bit_exact(kind, aabb, res, cuda):
    R = 1024
    o, d = _rays(R, 3, axis_aligned=True)
    binary = _grid(res, kind)
    rng = np.random.default_rng(9)
    near = (0.2 + rng.random(R).astype(np.float32) * np.float32(0.011)).astype(np.float32)   # stratified near planes
    ri_o, t0_o, t1_o, packed_o, cells_o = om.march(o, d, aabb, binary, near, 1e3, 0.011, want_cells=True)
    est = _est(aabb, binary, cuda)
    ri, t0, t1, packed, cells = est.traverse(torch.from_numpy(o).to(cuda), torch.from_numpy(d).to(cuda),
                                             torch.from_numpy(near).to(cuda), 1e3, 0.011, want_cells=True)
    assert np.array_equal(packed.cpu().numpy(), packed_o)            # per-ray sample counts + offsets
    assert np.array_equal(ri.cpu().numpy(), ri_o)                    # ray indices
    assert np.array_equal(cells.cpu().numpy(), cells_o)              # occupancy-grid cell ids
    assert np.array_equal(t0.cpu().numpy().view(np.uint32), t0_o.view(np.uint32))   # t values, bitwise
    assert np.array_equal(t1.cpu().numpy().view(np.uint32), t1_o.view(np.uint32))
    if kind == "empty":
        assert ri.numel() == 0
    if kind == "ones":
        assert ri.numel() > 200 * (R - 16)


@pytest.mark.parametrize("kind,aabb,res", [("ones", P97, 128), ("shell", P30, 128), ("random", P30, 64), ("empty", P30, 32)])
def test_prefetched_pass_keeps_the_samples_and_the_step_copies_them(kind, aabb, res, cuda):
    """``nsx_march_count_stash`` + ``nsx_march_fill_from_stash`` (the counting pass a step ahead keeps the samples' starts; the
    step copies them into place) against the oracle's march, bit for bit -- through ``prefetch_march`` -> ``traverse(counted=)``
    and ``counted_march``, the way the per-kernel path and the native step take it."""
    R = 1024
    o, d = _rays(R, 3, axis_aligned=True)
    binary = _grid(res, kind)
    near = np.full(R, 0.2, np.float32)
    ri_o, t0_o, t1_o, packed_o = om.march(o, d, aabb, binary, near, 1e3, 0.011)
    est = _est(aabb, binary, cuda)
    ot, dt = torch.from_numpy(o).to(cuda), torch.from_numpy(d).to(cuda)
    args = dict(near_plane=0.2, far_plane=1e3, render_step_size=0.011, stratified=False)
    assert est.prefetch_march(ot, dt, **args)
    pre = est._take_prefetched(est._march_key(ot, dt, 0.2, 1e3, 0.011, False, None))
    assert pre is not None and pre["stash"] is not None and pre["cap"] % 64 == 0
    ri, t0, t1, packed, _ = est.traverse(ot, dt, pre["near_planes"], 1e3, 0.011, counted=pre)
    assert est._stash_of(pre) is not None and int(pre["total_host"][0]) == ri_o.shape[0]
    assert np.array_equal(packed.cpu().numpy(), packed_o)
    assert np.array_equal(ri.cpu().numpy(), ri_o)
    assert np.array_equal(t0.cpu().numpy().view(np.uint32), t0_o.view(np.uint32))
    assert np.array_equal(t1.cpu().numpy().view(np.uint32), t1_o.view(np.uint32))
    # the native step's entry: counted_march hands the stash on
    assert est.prefetch_march(ot, dt, **args)
    near_t, packed2, S = est.counted_march(ot, dt, 0.2, 1e3, 0.011, False)
    assert est.last_march_prefetched and S == ri_o.shape[0] and torch.equal(packed2, packed)
    assert (est.last_march_stash is not None) and est.last_march_stash[1] == pre["cap"]
    if S:
        stash = est.last_march_stash[0]
        n0 = int(packed_o[np.argmax(packed_o[:, 1]), 1])
        r0 = int(np.argmax(packed_o[:, 1]))
        assert np.array_equal(stash[r0, :n0].cpu().numpy().view(np.uint32), t0_o[packed_o[r0, 0]:packed_o[r0, 0] + n0].view(np.uint32))
    # a stash too short for some ray: the pass says so, and the step marches in place (same result)
    est.march_stash_max = 10 ** 6
    est._stash_cap = lambda step: 64
    assert est.prefetch_march(ot, dt, **args)
    pre = est._take_prefetched(est._march_key(ot, dt, 0.2, 1e3, 0.011, False, None))
    ri3, t03, t13, packed3, _ = est.traverse(ot, dt, pre["near_planes"], 1e3, 0.011, counted=pre)
    longest = int(packed_o[:, 1].max()) if R else 0
    assert (est._stash_of(pre) is None) == (longest > 64)
    assert np.array_equal(ri3.cpu().numpy(), ri_o) and np.array_equal(t03.cpu().numpy().view(np.uint32), t0_o.view(np.uint32))
    assert np.array_equal(t13.cpu().numpy().view(np.uint32), t1_o.view(np.uint32))


def test_sampling_contract_and_visibility(cuda):
    """OccGridEstimator.sampling: sigma_fn filtering == oracle transmittance/alpha thresholds."""
    R = 512
    o, d = _rays(R, 5)
    binary = _grid(128, "shell")
    est = _est(P30, binary, cuda)
    est.occs.fill_(0.5)                     # occs.mean() > alpha_thre -> alpha_thre stays 1e-2
    ot, dt = torch.from_numpy(o).to(cuda), torch.from_numpy(d).to(cuda)

    def sigma_fn(t0, t1, ri):
        p = ot[ri] + dt[ri] * ((t0 + t1) / 2)[:, None]
        return 40.0 * torch.exp(-(p ** 2).sum(-1))

    ri, t0, t1 = est.sampling(ot, dt, sigma_fn=sigma_fn, near_plane=0.2, far_plane=1e3, render_step_size=0.011,
                              early_stop_eps=0.0, alpha_thre=1e-2, stratified=False)
    near = np.full(R, 0.2, np.float32)
    ri_o, t0_o, t1_o, packed_o = om.march(o, d, P30, binary, near, 1e3, 0.011)
    p = o[ri_o] + d[ri_o] * ((t0_o + t1_o) / 2)[:, None]
    sig = (40.0 * np.exp(-(p.astype(np.float64) ** 2).sum(-1))).astype(np.float32)
    w, T, a = om.render_weights(t0_o, t1_o, sig, packed_o)
    keep = (T >= 0.0) & (a >= 1e-2)
    # samples whose alpha sits within fp32 noise of the threshold may legitimately differ
    border = np.abs(a - 1e-2) < 1e-6
    got = set(zip(ri.cpu().numpy().tolist(), t0.cpu().numpy().view(np.uint32).tolist()))
    want = set(zip(ri_o[keep].tolist(), t0_o[keep].view(np.uint32).tolist()))
    amb = set(zip(ri_o[border].tolist(), t0_o[border].view(np.uint32).tolist()))
    assert (got ^ want) <= amb
    assert len(got) > 1000


def _packed_case(seed, R=300, maxn=300, cuda=None):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, maxn, R)
    counts[::7] = 0
    counts[5] = 1
    counts[6] = 64
    counts[8] = 65
    ray_idx = np.repeat(np.arange(R), counts).astype(np.int64)
    S = len(ray_idx)
    t0 = np.concatenate([np.sort(rng.random(c)) * 3 + 0.2 for c in counts]).astype(np.float32) if S else np.zeros(0, np.float32)
    t1 = (t0 + 0.011).astype(np.float32)
    sigma = (rng.random(S) ** 3 * 30).astype(np.float32)
    packed = om.pack_info(ray_idx, R)
    return ray_idx, t0, t1, sigma, packed


@pytest.mark.parametrize("R", [1, 63, 4096, 4097, 2 * 4096 + 37, 40000])
def test_pack_info_over_tile_boundaries(R, cuda):
    """The scan walks the rays in tiles of 256 threads x 16 rays: sizes around / beyond one tile, empty rays included."""
    from nersemble_amd import nerfacc as nf
    rng = np.random.default_rng(R)
    counts = rng.integers(0, 5, R)
    counts[rng.random(R) < 0.3] = 0
    ray_idx = np.repeat(np.arange(R), counts).astype(np.int64)
    packed = nf.pack_info(torch.from_numpy(ray_idx).to(cuda), R).cpu().numpy()
    want = np.stack([np.cumsum(counts) - counts, counts], axis=1)
    assert np.array_equal(packed, want)
    assert np.array_equal(packed, om.pack_info(ray_idx, R))


def test_pack_info_and_render_weights(cuda):
    from nersemble_amd import nerfacc as nf
    ray_idx, t0, t1, sigma, packed_o = _packed_case(1)
    R = packed_o.shape[0]
    ri = torch.from_numpy(ray_idx).to(cuda)
    packed = nf.pack_info(ri, R)
    assert np.array_equal(packed.cpu().numpy(), packed_o)
    sg = torch.from_numpy(sigma).to(cuda).requires_grad_(True)
    w, T, a = nf.render_weight_from_density(torch.from_numpy(t0).to(cuda), torch.from_numpy(t1).to(cuda), sg,
                                            packed_info=packed)
    w_o, T_o, a_o = om.render_weights(t0, t1, sigma, packed_o)
    assert np.abs(w.detach().cpu().numpy() - w_o).max() <= 2e-6
    assert np.abs(T.cpu().numpy() - T_o).max() <= 2e-6
    assert np.abs(a.cpu().numpy() - a_o).max() <= 2e-6
    gw = np.random.default_rng(3).standard_normal(len(sigma)).astype(np.float32)
    w.backward(torch.from_numpy(gw).to(cuda))
    ds_o = om.render_weights_bwd(t0, t1, sigma, packed_o, gw)
    assert np.abs(sg.grad.cpu().numpy() - ds_o).max() <= 1e-5 * max(1.0, np.abs(ds_o).max())


@pytest.mark.parametrize("Cc", [None, 1, 3])
def test_accumulate_along_rays(Cc, cuda):
    from nersemble_amd import nerfacc as nf
    ray_idx, t0, t1, sigma, packed_o = _packed_case(2)
    R = packed_o.shape[0]
    rng = np.random.default_rng(4)
    w = rng.random(len(ray_idx)).astype(np.float32)
    v = rng.standard_normal((len(ray_idx), Cc)).astype(np.float32) if Cc else None
    wt = torch.from_numpy(w).to(cuda).requires_grad_(True)
    vt = torch.from_numpy(v).to(cuda).requires_grad_(True) if Cc else None
    out = nf.accumulate_along_rays(wt, vt, ray_indices=torch.from_numpy(ray_idx).to(cuda), n_rays=R)
    want = om.accumulate(w, v, packed_o)
    assert np.abs(out.detach().cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    g = rng.standard_normal(want.shape).astype(np.float32)
    out.backward(torch.from_numpy(g).to(cuda))
    vv = v if Cc else np.ones((len(w), 1), np.float32)
    dw = (vv * g[ray_idx]).sum(1)
    assert np.abs(wt.grad.cpu().numpy() - dw).max() <= 1e-5 * max(1.0, np.abs(dw).max())
    if Cc:
        assert np.abs(vt.grad.cpu().numpy() - w[:, None] * g[ray_idx]).max() <= 1e-6


def test_flatten_eff_distloss(cuda):
    from nersemble_amd.distloss import flatten_eff_distloss
    ray_idx, t0, t1, sigma, packed_o = _packed_case(6)
    # torch_efficient_distloss semantics: n_rays = ray_id.max()+1
    n_rays = int(ray_idx.max()) + 1
    w_o, _, _ = om.render_weights(t0, t1, sigma, packed_o)
    m, iv = ((t0 + t1) * 0.5).astype(np.float32), (t1 - t0).astype(np.float32)
    loss_o, g_o = om.distloss(w_o, m, iv, om.pack_info(ray_idx, n_rays), n_rays)
    wt = torch.from_numpy(w_o).to(cuda).requires_grad_(True)
    loss = flatten_eff_distloss(wt, torch.from_numpy(m).to(cuda), torch.from_numpy(iv).to(cuda),
                                torch.from_numpy(ray_idx).to(cuda))
    assert abs(loss.item() - loss_o) <= 1e-5 * max(1e-3, abs(loss_o))
    (loss * 3.0).backward()
    assert np.abs(wt.grad.cpu().numpy() - 3.0 * g_o).max() <= 2e-5 * max(1e-3, np.abs(g_o).max() * 3)


def test_distloss_selection_matches_reference_golden(cuda, golden_dir):
    """BaseModel.get_dist_loss sample selection (ray_indices < max_rays), midpoints, intervals: golden from the
    reference's own models/base.py:224-249 with a recording stub."""
    from nersemble_amd.models.base import select_dist_loss_samples
    z = np.load(f"{golden_dir}/misc.npz")
    ri = torch.from_numpy(z["dl_ray_idx"]).to(cuda)
    w, m, iv, rid = select_dist_loss_samples(ri, torch.from_numpy(z["dl_weights"]).to(cuda),
                                             torch.from_numpy(z["dl_starts"]).to(cuda)[:, None],
                                             torch.from_numpy(z["dl_ends"]).to(cuda)[:, None], max_rays=5)
    assert np.array_equal(rid.cpu().numpy(), z["dl_sel_ray_id"])
    assert np.array_equal(w.cpu().numpy(), z["dl_sel_w"])
    assert np.array_equal(m.cpu().numpy(), z["dl_sel_m"])
    assert np.array_equal(iv.cpu().numpy(), z["dl_sel_interval"])


def test_fused_composite_equals_separate_operators(cuda):
    """nsx_composite_fwd/bwd == render_weight_from_density + RGB(white bg) / accumulation / depth('expected', clipped)
    renderers built from the separate nerfacc-shaped operators (values and gradients)."""
    from nersemble_amd import nerfacc as nf
    from nersemble_amd.model_components.renderers import AccumulationRenderer, DepthRenderer, RGBRenderer
    from nersemble_amd.rays import Frustums, RaySamples
    ray_idx, t0, t1, sigma, packed_o = _packed_case(9)
    R, S = packed_o.shape[0], len(ray_idx)
    rng = np.random.default_rng(10)
    rgbv = rng.random((S, 3)).astype(np.float32)
    aux = rng.standard_normal((S, 3)).astype(np.float32)
    ri = torch.from_numpy(ray_idx).to(cuda)
    packed = nf.pack_info(ri, R)
    t0t, t1t = torch.from_numpy(t0).to(cuda), torch.from_numpy(t1).to(cuda)
    g_rgb = torch.from_numpy(rng.standard_normal((R, 3)).astype(np.float32)).to(cuda)
    g_acc = torch.from_numpy(rng.standard_normal((R, 1)).astype(np.float32)).to(cuda)
    g_dep = torch.from_numpy(rng.standard_normal((R, 1)).astype(np.float32)).to(cuda)
    g_w = torch.from_numpy(rng.standard_normal(S).astype(np.float32)).to(cuda)

    def run(fused):
        sg = torch.from_numpy(sigma).to(cuda).requires_grad_(True)
        c = torch.from_numpy(rgbv).to(cuda).requires_grad_(True)
        if fused:
            w, rgb, acc, dep, ax = nf.composite(t0t, t1t, sg, c, packed, background=1.0, aux=torch.from_numpy(aux).to(cuda))
        else:
            w = nf.render_weight_from_density(t0t, t1t, sg, packed_info=packed)[0]
            rs = RaySamples(Frustums(None, None, t0t[:, None], t1t[:, None], None))
            rgb = RGBRenderer("white").train()(c, w[:, None], ri, R, packed)
            acc = AccumulationRenderer()(w[:, None], ri, R, packed)
            dep = DepthRenderer()(w[:, None], rs, ri, R, packed)
            ax = nf.accumulate_along_rays(w.detach(), torch.from_numpy(aux).to(cuda), ri, R, packed)
        ((rgb * g_rgb).sum() + (acc * g_acc).sum() + (dep * g_dep).sum() + (w * g_w).sum()).backward()
        return [t.detach() for t in (w, rgb, acc, dep, ax, sg.grad, c.grad)]

    a, b = run(True), run(False)
    names = ["weights", "rgb", "acc", "depth", "aux", "dsigma", "drgb"]
    for n, x, y in zip(names, a, b):
        tol = 2e-5 * max(1.0, y.abs().max().item())
        assert (x - y).abs().max().item() <= tol, (n, (x - y).abs().max().item(), tol)
    # oracle cross-check of the forward
    w_o, _, _ = om.render_weights(t0, t1, sigma, packed_o)
    assert np.abs(a[0].cpu().numpy() - w_o).max() <= 2e-6
    acc_o = om.accumulate(w_o, None, packed_o)
    assert np.abs(a[2].cpu().numpy() - acc_o).max() <= 1e-5


def test_fused_sample_losses_match_reference_golden_and_separate_ops(cuda, golden_dir):
    """Fused dist + empty + near kernel vs (1) the values the reference's BaseModel produced for the golden batch and
    (2) the separate mirrored implementations incl. gradients w.r.t. the weights."""
    from nersemble_amd import nerfacc as nf
    from nersemble_amd.distloss import fused_sample_losses, flatten_eff_distloss
    z = np.load(f"{golden_dir}/misc.npz")
    ri = torch.from_numpy(z["ls_ray_idx"]).to(cuda)
    R = int(z["ls_depth"].shape[0])
    packed = nf.pack_info(ri, R)
    t0, t1 = torch.from_numpy(z["ls_starts"]).to(cuda), torch.from_numpy(z["ls_ends"]).to(cuda)
    w = torch.from_numpy(z["ls_weights"][:, 0]).to(cuda).requires_grad_(True)
    depth = torch.from_numpy(z["ls_depth"]).to(cuda)
    eps = float(z["ls_eps"][0])
    trio = fused_sample_losses(w, t0, t1, packed, depth, eps, 5000, R)
    assert abs(1e-2 * trio[1].item() - z["ls_empty"][0]) <= 1e-5 * abs(z["ls_empty"][0])
    assert abs(1e-4 * trio[2].item() - z["ls_near"][0]) <= 1e-4 * abs(z["ls_near"][0])
    gsel = torch.tensor([0.7, -1.3, 2.1], device=cuda)
    (trio * gsel).sum().backward()
    g_fused = w.grad.clone()
    # separate implementations
    import types
    from nersemble_amd.models.base import BaseModel, BaseModelConfig
    from nersemble_amd.engine.generic_scheduler import GenericScheduler
    m = BaseModel.__new__(BaseModel)
    torch.nn.Module.__init__(m)
    m.config = BaseModelConfig(lambda_empty_loss=1.0, lambda_near_loss=1.0, lambda_dist_loss=1.0)
    m.sched_eps_depth = GenericScheduler(0.9, 0.01, 0, 10000)
    m.sched_eps_depth.update(2500)
    m.train()
    w2 = torch.from_numpy(z["ls_weights"][:, 0]).to(cuda).requires_grad_(True)
    rs = types.SimpleNamespace(frustums=types.SimpleNamespace(starts=t0[:, None], ends=t1[:, None]))
    near, empty = m.get_near_and_empty_loss({"depth_maps": depth}, rs, ri, w2[:, None], None)
    dist = flatten_eff_distloss(w2, (t0 + t1) * 0.5, t1 - t0, ri)
    assert abs(trio[0].item() - dist.item()) <= 1e-5 * abs(dist.item())
    assert abs(trio[1].item() - empty.item()) <= 1e-5 * abs(empty.item())
    assert abs(trio[2].item() - near.item()) <= 1e-4 * abs(near.item())
    (0.7 * dist - 1.3 * empty + 2.1 * near).backward()
    assert (g_fused - w2.grad).abs().max().item() <= 1e-4 * w2.grad.abs().max().item()


def test_gather_rows_matches_indexing(cuda):
    """nsx_gather_rows: several arrays of different row widths compacted by one index in one launch (bit-exact)."""
    from nersemble_amd.functional import gather_rows
    g = torch.Generator(device=cuda).manual_seed(0)
    n_src = 10_001
    a = torch.randn((n_src, 32), device=cuda, generator=g).half()          # 64-B rows -> 16-B pieces
    b = torch.randn((n_src, 3), device=cuda, generator=g)                  # 12-B rows -> 4-B pieces
    c = torch.randint(0, 1 << 40, (n_src,), device=cuda, generator=g)      # int64
    d = torch.randn((n_src,), device=cuda, generator=g)
    e = torch.randn((n_src, 16), device=cuda, generator=g).half()[1:]      # misaligned base (32-B rows, 4-B pieces)
    for n in (0, 1, 7777):
        idx = torch.randint(0, n_src - 1, (n,), device=cuda, generator=g)
        got = gather_rows(idx, a, b, c, d, e)
        for t, o in zip((a, b, c, d, e), got):
            assert o.dtype == t.dtype and torch.equal(o, t[idx])


@pytest.mark.parametrize("R,mean_n", [(1, 300), (517, 90), (4096, 200)])
def test_ray_wise_compaction_equals_the_scan_compaction(R, mean_n, cuda):
    """The sampler tail of the native step (csrc/step.hip): visibility with per-ray counts (nsx_render_visibility) ->
    nsx_pack_info -> one wave per ray writes the kept indices (nsx_compact_rays) -> ONE gather for the per-sample rows and
    the rays' rows (nsx_gather_rows_via) -- against the round-4 sequence it replaces: nsx_render_weights_fwd ->
    nsx_compact_mask -> nsx_gather_rows x 3 -> nsx_ray_histogram -> nsx_pack_info.  Bit for bit, zero tails included; rays
    without samples, rays longer than a wave, all / none visible."""
    import ctypes as C
    from nersemble_amd._lib import lib, ptr, stream, check
    g = torch.Generator(device=cuda).manual_seed(R)
    counts = torch.randint(0, 2 * mean_n, (R,), device=cuda, generator=g)
    counts[torch.rand((R,), device=cuda, generator=g) < 0.15] = 0
    if R > 3:
        counts[1], counts[2] = 64, 65
    S = int(counts.sum().item())
    if S == 0:
        counts[0] = 5
        S = 5
    packed = torch.stack([torch.cumsum(counts, 0) - counts, counts], 1).contiguous()
    m_ri = torch.repeat_interleave(torch.arange(R, device=cuda), counts)
    t0 = torch.rand((S,), device=cuda, generator=g)
    t1 = t0 + 0.01 + 0.02 * torch.rand((S,), device=cuda, generator=g)
    org, dirs = torch.randn((R, 3), device=cuda, generator=g), torch.randn((R, 3), device=cuda, generator=g)
    off = torch.randn((S, 3), device=cuda, generator=g)
    feat = torch.randn((S, 32), device=cuda, generator=g).half()
    base = torch.randn((S, 16), device=cuda, generator=g).half()
    slot = torch.randint(0, 24, (S,), device=cuda, generator=g, dtype=torch.int32)
    thre = torch.tensor([0.01], device=cuda)

    def arrays(n):
        return [torch.full((S,) + sh, -7, dtype=dt, device=cuda) for sh, dt in
                (((), torch.int64), ((), torch.float32), ((), torch.float32), ((3,), torch.float32), ((3,), torch.float32),
                 ((3,), torch.float32), ((32,), torch.float16), ((16,), torch.float16), ((), torch.int32))][:n]

    def pointers(ts):
        return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    for sig_scale, thre_v in ((3.0, 0.01), (0.0, 0.01), (1e-4, 0.0), (1e4, 0.01)):   # a mix / nothing / everything / the rays' heads
        thre.fill_(thre_v)
        dens = torch.rand((S,), device=cuda, generator=g) * sig_scale * 20
        # -- the sequence of round 4
        vis_a = torch.zeros((S,), dtype=torch.uint8, device=cuda)
        check(lib().nsx_render_weights_fwd(ptr(t0), ptr(t1), ptr(dens), ptr(packed), R, None, None, None, ptr(vis_a), 1e-4, 0.0,
                                           ptr(thre), stream()), "vis")
        keep_a = torch.full((S,), -1, dtype=torch.int64, device=cuda)
        n_a = torch.zeros((1,), dtype=torch.int64, device=cuda)
        scratch = torch.empty((int(lib().nsx_occ_scratch_bytes(S)),), dtype=torch.uint8, device=cuda)
        check(lib().nsx_compact_mask(ptr(vis_a), S, ptr(keep_a), ptr(n_a), ptr(scratch), stream()), "compact")
        out_a = arrays(9)
        rb = (C.c_int64 * 3)(8, 4, 4)
        check(lib().nsx_gather_rows(3, pointers([m_ri, t0, t1]), rb, pointers(out_a[:3]), ptr(keep_a), S, ptr(n_a), stream()), "g1")
        rb = (C.c_int64 * 2)(12, 12)
        check(lib().nsx_gather_rows(2, pointers([org, dirs]), rb, pointers(out_a[3:5]), ptr(out_a[0]), S, ptr(n_a), stream()), "g2")
        rb = (C.c_int64 * 4)(12, 64, 32, 4)
        check(lib().nsx_gather_rows(4, pointers([off, feat, base, slot]), rb, pointers(out_a[5:]), ptr(keep_a), S, ptr(n_a),
                                    stream()), "g3")
        cnt_a = torch.zeros((R,), dtype=torch.int64, device=cuda)
        check(lib().nsx_ray_histogram(ptr(out_a[0]), S, R, ptr(cnt_a), ptr(n_a), stream()), "hist")
        pk_a, tot_a = torch.zeros((R, 2), dtype=torch.int64, device=cuda), torch.zeros((1,), dtype=torch.int64, device=cuda)
        check(lib().nsx_pack_info(ptr(cnt_a), R, ptr(pk_a), ptr(tot_a), stream()), "pack")
        # -- the sequence of round 5
        vis_b = torch.zeros((S,), dtype=torch.uint8, device=cuda)
        cnt_b = torch.full((R,), -3, dtype=torch.int64, device=cuda)
        check(lib().nsx_render_visibility(ptr(t0), ptr(t1), ptr(dens), ptr(packed), R, ptr(vis_b), ptr(cnt_b), 1e-4, 0.0,
                                          ptr(thre), stream()), "vis2")
        pk_b, tot_b = torch.zeros((R, 2), dtype=torch.int64, device=cuda), torch.zeros((1,), dtype=torch.int64, device=cuda)
        check(lib().nsx_pack_info(ptr(cnt_b), R, ptr(pk_b), ptr(tot_b), stream()), "pack2")
        keep_b = torch.full((S,), -1, dtype=torch.int64, device=cuda)
        n_b = torch.zeros((1,), dtype=torch.int64, device=cuda)
        check(lib().nsx_compact_rays(ptr(vis_b), ptr(packed), ptr(pk_b), R, ptr(keep_b), ptr(tot_b), ptr(n_b), stream()), "rays")
        out_b = arrays(9)
        rb = (C.c_int64 * 9)(8, 4, 4, 12, 12, 12, 64, 32, 4)
        hop = (C.c_uint8 * 9)(0, 0, 0, 1, 1, 0, 0, 0, 0)
        check(lib().nsx_gather_rows_via(9, pointers([m_ri, t0, t1, org, dirs, off, feat, base, slot]), rb, pointers(out_b),
                                        ptr(keep_b), ptr(m_ri), hop, S, ptr(n_b), stream()), "via")
        torch.cuda.synchronize()
        n = int(n_a.item())
        assert int(n_b.item()) == n == int(tot_b.item()) == int(tot_a.item()) == int(vis_a.sum().item())
        assert n == (0 if sig_scale == 0.0 else S if thre_v == 0.0 else n) and (0 < n < S or sig_scale in (0.0, 1e-4))
        assert torch.equal(vis_a, vis_b) and torch.equal(keep_a[:n], keep_b[:n])
        assert torch.equal(cnt_a, cnt_b) and torch.equal(pk_a, pk_b)
        for x, y in zip(out_a, out_b):
            assert torch.equal(x.view(torch.uint8), y.view(torch.uint8))           # (bytes: the zero tails included)
