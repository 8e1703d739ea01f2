"""BASELINE.json configs[0] -- "single timestep (static), n_hash_encodings=1, 64x64 render, 256 rays/batch, ... on CPU
(plumbing, no GPU)": the CPU restatement of the whole static path (oracle/render.py) renders a 64 x 64 image in
256-ray batches.  Checks the plumbing between the per-stage oracles (each of which is pinned separately): packed
layout, batch-size independence, compositing identities, background handling."""
import numpy as np
import torch

from oracle import hashgrid, mlp, render
from oracle.capi import grid_geometry

AABB = np.array([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]], dtype=np.float32)
RES = 32


def _scene(seed=0):
    rng = np.random.default_rng(seed)
    g = grid_geometry(n_levels=8, per_level_scale=1.5, base_resolution=8, log2_hashmap_size=12)
    f_enc, _, c = hashgrid.ens_layout(1)
    tables = rng.uniform(-0.5, 0.5, size=(c, g.total_entries, f_enc)).astype(np.float16).view(np.uint16)
    base = rng.uniform(-1, 1, size=mlp.param_count(0)).astype(np.float32) * np.sqrt(6.0 / (32 + 64))
    head = rng.uniform(-1, 1, size=mlp.param_count(1)).astype(np.float32) * np.sqrt(6.0 / (64 + 64))
    # occupancy: voxels whose centre lies in an ellipsoid around the box centre
    ax = [(np.arange(RES) + 0.5) / RES * (AABB[1, a] - AABB[0, a]) + AABB[0, a] for a in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    ctr = AABB.mean(0)
    binary = ((X - ctr[0]) / 1.0) ** 2 + ((Y - ctr[1]) / 1.2) ** 2 + ((Z - ctr[2]) / 1.3) ** 2 <= 1.0
    return g, tables, base, head, binary


def _camera_rays(n=64):
    from nersemble_amd.cameras import Cameras
    ctr = torch.tensor(AABB.mean(0))
    eye = ctr + torch.tensor([0.0, -9.0, 0.5])
    fwd = (ctr - eye) / (ctr - eye).norm()
    right = torch.linalg.cross(fwd, torch.tensor([0.0, 0.0, 1.0]))
    right = right / right.norm()
    up = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4)[:3]
    c2w[:, 0], c2w[:, 1], c2w[:, 2], c2w[:, 3] = right, up, -fwd, eye
    cams = Cameras(c2w[None], fx=2.6 * n, fy=2.6 * n, cx=n / 2, cy=n / 2, width=n, height=n)
    b = cams.generate_rays(0).flatten()
    return b.origins.numpy(), b.directions.numpy()


def test_static_config_renders_64x64_in_256_ray_batches():
    g, tables, base, head, binary = _scene()
    o, d = _camera_rays(64)
    assert o.shape == (4096, 3)
    args = (AABB, binary, tables, g, base, head)
    batches = [render.render_static(o[i:i + 256], d[i:i + 256], *args) for i in range(0, 4096, 256)]
    rgb = np.concatenate([b["rgb"] for b in batches])
    acc = np.concatenate([b["accumulation"] for b in batches])
    depth = np.concatenate([b["depth"] for b in batches])
    counts = np.concatenate([b["num_samples_per_ray"] for b in batches])
    assert rgb.shape == (4096, 3) and acc.shape == (4096, 1) and depth.shape == (4096, 1)
    # one 4096-ray batch gives the same image bit for bit (rays are independent; the packed layout is per batch)
    whole = render.render_static(o, d, *args)
    # (a batch in which no ray hits the grid carries the reference's one fake zero-length sample on its ray 0,
    # nersemble_volumetric_sampler.py:109-115: count 1, weight 0 -- it renders as background)
    fake = np.zeros(4096, dtype=bool)
    for k, b in enumerate(batches):
        if b["num_samples_per_ray"].sum() == 1 and b["t_starts"][0] == b["t_ends"][0] == 1.0:
            assert b["weights"][0] == 0.0 and np.array_equal(b["rgb"][0], np.ones(3, np.float32))
            fake[k * 256] = True
    assert fake.any()
    counts = np.where(fake, 0, counts)
    assert np.array_equal(whole["num_samples_per_ray"], counts)
    assert np.array_equal(whole["rgb"], rgb) and np.array_equal(whole["accumulation"], acc)
    # packed layout: sorted ray indices, counts = histogram, samples of a ray are consecutive lattice steps
    ri, t0, t1 = whole["ray_indices"], whole["t_starts"], whole["t_ends"]
    assert (np.diff(ri) >= 0).all() and np.array_equal(np.bincount(ri, minlength=4096), counts)
    assert (t1 > t0).all() and np.allclose(t1 - t0, 0.011, atol=2e-6)
    same_ray = ri[1:] == ri[:-1]
    assert (t0[1:][same_ray] >= t1[:-1][same_ray] - 1e-6).all()
    # compositing identities
    assert (acc >= 0).all() and (acc <= 1 + 1e-5).all() and (rgb >= -1e-6).all() and (rgb <= 1 + 1e-5).all()
    w = whole["weights"]
    assert np.allclose(np.bincount(ri, weights=w.astype(np.float64), minlength=4096), acc[:, 0], atol=1e-5)
    hit = counts > 0
    assert 0.2 < hit.mean() < 0.9                                      # the ellipsoid fills part of the frame
    assert np.array_equal(rgb[~hit], np.ones_like(rgb[~hit])) and (acc[~hit] == 0).all()      # white background
    t_mid = (t0 + t1) / 2
    assert (depth[hit] >= t_mid.min() - 1e-6).all() and (depth[hit] <= t_mid.max() + 1e-6).all()
    # densities are finite and positive inside the box; sigmoid colours in (0, 1)
    assert np.isfinite(whole["density"]).all() and (whole["density"] >= 0).all()
    assert (whole["rgb_samples"] > 0).all() and (whole["rgb_samples"] < 1).all()
    mse = float(((rgb - 1.0) ** 2).mean())
    assert np.isfinite(10 * np.log10(1.0 / mse))                       # the PSNR the metric is quoted with


def test_static_render_stages_agree_with_their_own_oracles():
    """The composition calls the per-stage oracles with the layouts their own tests use (spot checks by hand)."""
    g, tables, base, head, binary = _scene(1)
    o, d = _camera_rays(16)
    out = render.render_static(o, d, AABB, binary, tables, g, base, head)
    ri, t0, t1 = out["ray_indices"], out["t_starts"], out["t_ends"]
    pos = render.sample_positions(o, d, ri, t0, t1)
    pn, sel = render.normalise(pos, AABB)
    assert sel.all() and (pn > 0).all() and (pn < 1).all()            # marched samples lie inside the box
    k = min(50, ri.shape[0])
    feats = hashgrid.ensemble_fwd(pn[:k], tables, 1, g, np.ones((k, 1), np.float32))
    single = hashgrid.hashgrid_fwd(pn[:k], tables[0], g)               # H = 1, code 1: the ensemble IS the one grid
    assert np.array_equal(feats.view(np.uint16), single.view(np.uint16))
    h = mlp.mlp_fwd(feats.astype(np.float32), base, 0, 16, 0)
    assert np.allclose(out["density"][:k], np.exp(h[:, 0].astype(np.float32)), rtol=1e-6)
    # alpha compositing by hand for the first ray that has samples
    r = int(ri[0])
    m = ri == r
    sig, dt = out["density"][m].astype(np.float64), (t1[m] - t0[m]).astype(np.float64)
    alpha = 1 - np.exp(-sig * dt)
    T = np.concatenate([[1.0], np.cumprod(1 - alpha)[:-1]])
    assert np.allclose(out["weights"][m], T * alpha, atol=1e-6)
    assert np.allclose(out["rgb"][r], (T * alpha) @ out["rgb_samples"][m].astype(np.float64) + (1 - (T * alpha).sum()),
                       atol=1e-5)


def test_dynamic_render_reduces_to_static_and_follows_time():
    """render_dynamic with H = 1, a constant code of 1 and no deformation field IS render_static; with a deformation
    field whose last layers are zero the offsets vanish; different timesteps give different images."""
    from oracle import deform
    g, tables, base, head, binary = _scene(2)
    o, d = _camera_rays(12)
    R = o.shape[0]
    static = render.render_static(o, d, AABB, binary, tables, g, base, head)
    one = np.ones((3, 1), np.float32)
    dyn = render.render_dynamic(o, d, np.full(R, 0.5, np.float32), AABB, binary, tables, 1, g, base, head, one, 3)
    for k in ("rgb", "depth", "accumulation", "num_samples_per_ray"):
        assert np.array_equal(static[k], dyn[k]), k
    assert (dyn["timesteps"] == 1).all() and not dyn["offsets"].any() and not dyn["deformation"].any()

    # H = 4 grids blended with per-timestep codes; deformation field with zero rotation / translation heads
    rng = np.random.default_rng(3)
    H, T = 4, 5
    f_enc, _, c = hashgrid.ens_layout(H)
    tables4 = rng.uniform(-0.5, 0.5, size=(c, g.total_entries, f_enc)).astype(np.float16).view(np.uint16)
    codes = rng.normal(0, 0.5, size=(T, H)).astype(np.float32)
    dcodes = rng.normal(0, 0.1, size=(T, 128)).astype(np.float32)
    lay, total = deform.flat_layout()
    flat = (rng.uniform(-1, 1, size=total) * 0.05).astype(np.float32)
    for name in ("Wr", "br", "Wv", "bv"):
        off, shp = lay[name]
        flat[off:off + int(np.prod(shp))] = 0.0
    times = np.where(np.arange(R) % 2 == 0, 0.0, 1.0).astype(np.float32)
    kw = dict(time_embedding=codes, n_timesteps=T, deform_embedding=dcodes, window_hash=float(H), window_deform=7.0)
    a = render.render_dynamic(o, d, times, AABB, binary, tables4, H, g, base, head, deform_params=flat, **kw)
    assert set(np.unique(a["timesteps"]).tolist()) <= {0, T - 1}
    assert np.abs(a["offsets"]).max() <= 1e-6                      # zero screw axis -> identity warp
    b = render.render_dynamic(o, d, times, AABB, binary, tables4, H, g, base, head, deform_params=None, **kw)
    assert np.allclose(a["rgb"], b["rgb"], atol=1e-5)
    # the same rays at the other timestep see another blend of the grids
    c_ = render.render_dynamic(o, d, 1.0 - times, AABB, binary, tables4, H, g, base, head, deform_params=None, **kw)
    hit = a["num_samples_per_ray"] > 0
    assert np.array_equal(a["num_samples_per_ray"], c_["num_samples_per_ray"])      # marching ignores time
    assert np.abs(a["rgb"][hit] - c_["rgb"][hit]).max() > 1e-3
    # a live deformation field moves the samples and is reported by the deformation renderer
    flat2 = flat.copy()
    off, shp = lay["bv"]
    flat2[off:off + 3] = np.array([0.01, -0.02, 0.005], np.float32)
    e = render.render_dynamic(o, d, times, AABB, binary, tables4, H, g, base, head, deform_params=flat2, **kw)
    assert np.allclose(e["offsets"], np.array([0.01, -0.02, 0.005]), atol=2e-5)     # pure translation (fp16 bias)
    assert np.allclose(e["deformation"][hit], e["accumulation"][hit] * np.array([0.01, -0.02, 0.005]), atol=1e-4)
