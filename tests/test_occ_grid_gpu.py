"""GPU parity for SURVEY row a10: the native occupancy-grid update (csrc/occ_grid.hip behind
``OccGridEstimator.update_every_n_steps``; reference nersemble_instant_ngp.py:184-196 -> nerfacc 0.5.2 ``_update``)
against oracle/occgrid.c.  Cell ids, query positions and random timesteps BIT-EXACT; ``occs`` bit-exact and ``binaries``
equal for equal query values; the density queries themselves within the field tolerance (tests/test_field_gpu.py)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import field as ofield, occgrid as og
from tests.helpers import randomise_model

pytestmark = pytest.mark.gpu
P30 = np.array([-2.5, -1.8, -2.5, 2.2, 1.8, 2.0], dtype=np.float32)


def _estimator(res, cuda, binary=None, seed=11, T=100):
    from nersemble_amd.nerfacc import OccGridEstimator
    est = OccGridEstimator(torch.from_numpy(P30), resolution=res, levels=1).to(cuda)
    est.rng_seed, est.n_timesteps = seed, T
    if binary is not None:
        est.binaries.copy_(torch.from_numpy(binary)[None].to(cuda))
    return est


@pytest.mark.parametrize("res,fill", [(32, 0.07), (32, 0.6), (128, 0.07), (128, 0.4), (20, 0.0), (20, 1.0)])
def test_cell_selection_bit_exact(res, fill, cuda):
    rng = np.random.default_rng(res)
    binary = rng.random((res,) * 3) < fill
    est = _estimator(res, cuda, binary, seed=0x1234567890ABCDEF, T=100)
    for step, warm in ((0, True), (240, True), (256, False), (4096, False)):
        cells, pos, ts, times = est.sample_cells(step, warmup=warm)
        c_o, p_o, ts_o, t_o = og.sample_cells(binary, P30, warm, est.rng_seed, step, 100)
        assert np.array_equal(cells.cpu().numpy(), c_o), (res, fill, step)
        assert np.array_equal(pos.cpu().numpy(), p_o)
        assert np.array_equal(ts.cpu().numpy(), ts_o)
        assert np.array_equal(times.cpu().numpy()[:, 0], t_o)
        n = res ** 3 // 4
        assert cells.shape[0] == (res ** 3 if warm else n + min(n, int(binary.sum())))


@pytest.mark.parametrize("res", [32, 128])
@pytest.mark.parametrize("scale", [0.002, 0.05])          # mean below / above occ_thre
def test_ema_max_threshold_bit_exact(res, scale, cuda):
    rng = np.random.default_rng(res + int(scale * 1000))
    N = res ** 3
    est = _estimator(res, cuda)
    occs0 = (rng.random(N) ** 4 * scale * 3).astype(np.float32)
    est.occs.copy_(torch.from_numpy(occs0).to(cuda))
    M = N // 2
    cells = rng.integers(0, N, M).astype(np.int32)
    cells[:1000] = cells[1000:2000]                         # duplicates
    vals = (rng.random(M) ** 4 * scale * 2).astype(np.float32)
    vals[5], vals[6] = np.nan, -1.0
    for _ in range(2):                                      # twice: the scratch is left clean
        est.apply_update(torch.from_numpy(cells).to(cuda), torch.from_numpy(vals).to(cuda), 0.01, 0.95)
    o1, b1, _ = og.update(occs0, np.zeros(N, bool), cells, vals, 0.95, 0.01)
    o2, b2, thre = og.update(o1, b1, cells, vals, 0.95, 0.01)
    assert np.array_equal(est.occs.cpu().numpy(), o2)
    assert np.array_equal(est.binaries.cpu().numpy().reshape(-1), b2)
    assert (thre < np.float32(0.01)) == (scale < 0.01) and 0.0 < b2.mean() < 1.0


def _model(cuda, res=32):
    from nersemble_amd.models.nersemble_instant_ngp import NeRSembleNGPModel
    from nersemble_amd.rays import SceneBox
    from nersemble_amd.workloads import SCENE_BOXES, WORKLOADS, build_model_config
    w = WORKLOADS["p030_h16"]
    cfg = build_model_config(w, small=True)
    cfg.hash_ensemble_config.hash_encoding_config.log2_hashmap_size = 15
    cfg.grid_resolution = res
    cfg.use_view_frustum_culling = False
    box = torch.tensor(SCENE_BOXES[30], dtype=torch.float32)
    torch.manual_seed(0)
    model = NeRSembleNGPModel(cfg, SceneBox(box), num_train_data=12, occ_seed=99)
    go = oracle.grid_geometry(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16, log2_hashmap_size=15)
    W = randomise_model(model, 5, go)
    # densities around the grid's threshold: occ = exp(h0) * 0.011 against occ_thre = 0.01
    return model.to(cuda).train(), go, W, w


def test_model_update_occupancy_grid_vs_oracle(cuda):
    model, go, W, w = _model(cuda)
    cfg, grid = model.config, model.occupancy_grid
    model.sched_window_deform.update(8000)
    model.sched_window_hash_encodings.update(60000)
    wh, wd = model.sched_window_hash_encodings.value, model.sched_window_deform.value
    occs_o = np.zeros(grid.occs.shape[0], np.float32)
    bin_o = np.zeros_like(occs_o, dtype=bool)
    for step in (0, 16, 256, 272):                           # two warm-up updates, two sampled ones
        warm = step < cfg.occupancy_grid_warmup_steps
        # -- the oracle's update from the oracle's own state
        c_o, p_o, ts_o, t_o = og.sample_cells(bin_o.reshape((32,) * 3), P30, warm, 99, step, w["T"])
        d_o, _ = ofield.field_density_fn(p_o, t_o, w["T"], W["aabb"], W["tables_u16"], w["H"], go, W["mlp_base"],
                                         W["time_embedding"], deform_params=W["deform_params"],
                                         deform_embedding=W["deform_embedding"], window_hash=wh, window_deform=wd)
        # -- the model's queries, piece by piece (same state as the oracle: asserted below)
        grid.rng_seed, grid.n_timesteps = 99, w["T"]
        cells, pos, ts, times = grid.sample_cells(step, warmup=warm)
        assert np.array_equal(cells.cpu().numpy(), c_o) and np.array_equal(pos.cpu().numpy(), p_o)
        assert np.array_equal(ts.cpu().numpy(), ts_o)
        with torch.no_grad():
            d_g = model.field_density_fn(pos, times).reshape(-1).cpu().numpy()
        rel = np.abs(d_g - d_o) / np.maximum(d_o, 1e-6)
        inside = d_o > 0
        assert np.median(rel[inside]) <= 5e-3 and rel[inside].max() <= 0.1, (step, float(rel[inside].max()))
        # -- the whole callback on the model (occ_eval_fn wiring, time hand-over, EMA, threshold) ...
        model.update_occupancy_grid(step)
        # ... equals the oracle's rule applied to the GPU's query values, bit for bit
        occs_o, bin_o, thre = og.update(occs_o, bin_o, c_o, d_g * np.float32(cfg.render_step_size),
                                        cfg.occupancy_grid_ema_decay, cfg.occ_thre)
        assert np.array_equal(grid.occs.cpu().numpy(), occs_o), step
        assert np.array_equal(grid.binaries.cpu().numpy().reshape(-1), bin_o), step
        assert 0.02 < bin_o.mean() < 0.98, (step, bin_o.mean())
    # steps that are not multiples of 16 leave the grid alone; evaluation mode refuses
    before = grid.occs.clone()
    model.update_occupancy_grid(273)
    assert torch.equal(before, grid.occs)
    model.eval()
    with pytest.raises(RuntimeError):
        model.update_occupancy_grid(288)


def test_dense_march_config_keeps_the_grid_full(cuda):
    """--disable_occupancy_grid: density_fn answers ones, so every update leaves all cells occupied
    (nersemble_instant_ngp.py:239-240 with :185-196)."""
    from nersemble_amd.workloads import build_workload
    trainer, data, _ = build_workload("p097_dense", device="cuda:0", small=True, n_rays=64)
    model = trainer.model
    for step in (0, 256):
        model.update_occupancy_grid(step)
        assert bool(model.occupancy_grid.binaries.all())
