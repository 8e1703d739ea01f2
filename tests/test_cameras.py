"""Pinhole ray generation (the step in front of the hot path): csrc/raygen.hip and the host mirror
in nersemble_amd/cameras.py against oracle/cameras.py (numpy restatement of nerfstudio's RayGenerator ->
Cameras.generate_rays as the reference's datamanager calls it, datamanager/nersemble_datamanager.py:76-81)."""
import numpy as np
import pytest
import torch

from oracle import cameras as ocam


def _rig(n=7, seed=0):
    rng = np.random.default_rng(seed)
    q = rng.standard_normal((n, 3, 3))
    rot = np.stack([np.linalg.qr(m)[0] for m in q]).astype(np.float32)          # orthonormal, either handedness
    c2w = np.concatenate([rot, rng.standard_normal((n, 3, 1)).astype(np.float32) * 9], axis=2)
    fx = (rng.random(n) * 800 + 1800).astype(np.float32)
    fy = (fx * (1 + rng.standard_normal(n) * 0.01)).astype(np.float32)
    cx = (550 + rng.standard_normal(n) * 20).astype(np.float32)
    cy = (802 + rng.standard_normal(n) * 20).astype(np.float32)
    return c2w, fx, fy, cx, cy


def _indices(n_cams, R, seed=1, W=1100, H=1604):
    rng = np.random.default_rng(seed)
    idx = np.stack([rng.integers(0, n_cams, R), rng.integers(0, H, R), rng.integers(0, W, R)], axis=1)
    idx[:4] = [[0, 0, 0], [n_cams - 1, H - 1, W - 1], [1, 0, W - 1], [2, H - 1, 0]]       # image corners
    return idx


def _cameras(rig, device="cpu"):
    from nersemble_amd.cameras import Cameras
    c2w, fx, fy, cx, cy = rig
    return Cameras(torch.from_numpy(c2w), torch.from_numpy(fx), torch.from_numpy(fy), torch.from_numpy(cx),
                   torch.from_numpy(cy), 1100, 1604).to(device)


def test_oracle_known_answers():
    """Identity pose: the principal point looks down -z; one pixel to the right tilts the ray by atan(1 / fx); the pixel
    footprint is ~ 1 / (fx fy) at the centre."""
    c2w = np.concatenate([np.eye(3, dtype=np.float32), np.array([[1.0], [2.0], [3.0]], np.float32)], axis=1)[None]
    fx, fy, cx, cy = (np.array([v], np.float32) for v in (2000.0, 1000.0, 10.0, 20.0))
    o, d, a = ocam.generate_rays(c2w, fx, fy, cx, cy, [0, 0, 0], [20.0, 20.0, 21.0], [10.0, 11.0, 10.0])
    assert np.array_equal(o, np.tile([[1.0, 2.0, 3.0]], (3, 1)))
    assert np.array_equal(d[0], [0.0, 0.0, -1.0])
    assert np.isclose(d[1, 0] / -d[1, 2], 1 / 2000.0, rtol=1e-6) and d[1, 1] == 0
    assert np.isclose(d[2, 1] / -d[2, 2], -1 / 1000.0, rtol=1e-6)                  # image y grows downwards
    assert np.isclose(a[0, 0], 1 / (2000.0 * 1000.0), rtol=1e-3)
    # ray_generator adds the pixel centre
    o2, d2, _ = ocam.ray_generator(c2w, fx, fy, cx, cy, np.array([[0, 19, 9]]))
    d_ref = np.array([(9.5 - 10) / 2000, -(19.5 - 20) / 1000, -1.0])
    assert np.allclose(d2[0], d_ref / np.linalg.norm(d_ref), atol=1e-7)


def test_host_mirror_matches_oracle():
    from nersemble_amd.cameras import RayGenerator
    rig = _rig()
    idx = _indices(len(rig[0]), 3000)
    b = RayGenerator(_cameras(rig))(torch.from_numpy(idx))
    o, d, a = ocam.ray_generator(*rig, idx)
    assert np.array_equal(b.origins.numpy(), o)
    assert np.abs(b.directions.numpy() - d).max() <= 2.4e-7                       # einsum sums in another order
    assert np.abs(b.pixel_area.numpy() - a).max() <= 2e-3 * a.max()
    assert np.array_equal(b.camera_indices.numpy()[:, 0], idx[:, 0])
    assert np.abs(np.linalg.norm(b.directions.numpy(), axis=1) - 1).max() <= 2e-7


@pytest.mark.gpu
def test_native_ray_generation_matches_oracle(cuda):
    """csrc/raygen.hip (one launch) against the oracle: same fp32 operation order, correctly rounded div / sqrt on both
    sides -- origins bit-exact, directions within 1 ulp, pixel areas within 1e-5 relative."""
    from nersemble_amd.cameras import RayGenerator
    rig = _rig(seed=3)
    for R in (1, 4096, 100_003):
        idx = _indices(len(rig[0]), max(R, 4), seed=R)[:R] if R >= 4 else _indices(len(rig[0]), 4, seed=R)[:R]
        b = RayGenerator(_cameras(rig, cuda))(torch.from_numpy(idx).to(cuda))
        o, d, a = ocam.ray_generator(*rig, idx)
        assert b.origins.is_cuda and np.array_equal(b.origins.cpu().numpy(), o)
        assert np.abs(b.directions.cpu().numpy() - d).max() <= 1.2e-7
        assert (np.abs(b.pixel_area.cpu().numpy() - a) <= 1e-5 * a).all()
    # a whole evaluation image, row-major
    cams = _cameras(rig, cuda)
    cams.rescale_output_resolution(0.05)
    img = cams.generate_rays(2)
    h, w = int(cams.height[2, 0]), int(cams.width[2, 0])
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32) + 0.5, np.arange(w, dtype=np.float32) + 0.5, indexing="ij")
    o, d, a = ocam.generate_rays(cams.camera_to_worlds.cpu().numpy(), *(t.cpu().numpy() for t in (cams.fx, cams.fy, cams.cx, cams.cy)),
                                 np.full(h * w, 2), ys.reshape(-1), xs.reshape(-1))
    assert img.directions.shape == (h, w, 3) and np.abs(img.directions.cpu().numpy().reshape(-1, 3) - d).max() <= 1.2e-7
    assert (np.abs(img.pixel_area.cpu().numpy().reshape(-1, 1) - a) <= 1e-5 * a).all()
