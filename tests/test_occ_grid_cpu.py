"""CPU: the occupancy-grid update oracle (oracle/occgrid.c): Philox4x32-10 known answers (Random123's
published vectors), the structure of nerfacc's cell selection, and the EMA-max / threshold rule on hand-made cases."""
import numpy as np

from oracle import occgrid as og

AABB = np.array([-2.5, -1.8, -2.5, 2.2, 1.8, 2.0], dtype=np.float32)


def test_philox_known_answers():
    assert og.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert og.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert og.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_warmup_visits_every_cell_once_with_jitter_inside_the_cell():
    res = 16
    b = np.zeros((res,) * 3, bool)
    cells, pos, ts, times = og.sample_cells(b, AABB, True, seed=7, step=0, n_timesteps=10)
    assert np.array_equal(cells, np.arange(res ** 3, dtype=np.int32))
    ijk = np.stack(np.unravel_index(cells, (res,) * 3), -1)
    x = (pos - AABB[:3]) / (AABB[3:] - AABB[:3]) * res
    assert ((x >= ijk - 1e-4) & (x < ijk + 1 + 1e-4)).all()
    frac = x - ijk
    assert 0.45 < frac.mean() < 0.55 and frac.min() < 0.01 and frac.max() > 0.99     # U[0,1) jitter
    assert ts.min() == 0 and ts.max() == 9 and np.allclose(times, ts / 9.0)
    # another step draws other numbers; the same (seed, step) repeats exactly
    again = og.sample_cells(b, AABB, True, seed=7, step=0, n_timesteps=10)
    other = og.sample_cells(b, AABB, True, seed=7, step=16, n_timesteps=10)
    assert np.array_equal(again[1], pos) and not np.array_equal(other[1], pos)


def test_selection_after_warmup_uniform_then_occupied():
    res = 16
    N, n = res ** 3, res ** 3 // 4
    rng = np.random.default_rng(0)
    few = rng.random((res,) * 3) < 0.1                      # fewer occupied cells than n: all of them, in order
    cells, pos, ts, times = og.sample_cells(few, AABB, False, seed=3, step=256, n_timesteps=1)
    occ_ids = np.flatnonzero(few.reshape(-1))
    assert cells.shape[0] == n + len(occ_ids) and np.array_equal(cells[n:], occ_ids)
    assert (cells[:n] >= 0).all() and (cells[:n] < N).all() and len(np.unique(cells[:n])) > 0.8 * n * (1 - np.exp(-1))
    assert (times == 0).all() and (ts == 0).all()           # one timestep: time 0
    many = rng.random((res,) * 3) < 0.7                     # more than n: n draws from the occupied list
    cells2, *_ = og.sample_cells(many, AABB, False, seed=3, step=256, n_timesteps=1)
    assert cells2.shape[0] == 2 * n and many.reshape(-1)[cells2[n:]].all()
    assert np.array_equal(cells2[:n], cells[:n])            # the uniform half does not depend on the grid


def test_ema_max_and_threshold_rules():
    occs = np.array([0.5, 0.0, 0.02, 0.0, 0.3, 0.0, 0.4, 0.2], np.float32)
    cells = np.array([0, 0, 2, 4, 4, 1, 5, 5, 6, 7], np.int32)
    vals = np.array([0.1, 0.6, 0.001, np.nan, 0.2, 0.004, -1.0, 0.0, np.nan, -3.0], np.float32)
    o, b, thre = og.update(occs, np.zeros(8, bool), cells, vals, ema_decay=0.95, occ_thre=0.01)
    d = np.float32(0.95)
    want = np.array([0.6,                         # duplicates: the larger of them beats 0.5 * 0.95
                     0.004, np.float32(0.02) * d, 0.0,                      # cell 3 untouched
                     np.float32(0.3) * d,                                   # NaN counts as 0, 0.2 < decayed
                     0.0,                                                   # negative counts as 0
                     np.float32(0.4) * d,                                   # ONLY a NaN: the cell still decays (nerfacc decays
                     np.float32(0.2) * d], np.float32)                      # every queried cell); only a negative: same
    assert np.array_equal(o, want)
    # torch.maximum(occs * decay, occ) -- nerfacc's rule -- agrees wherever no NaN is involved
    import torch
    ref = torch.from_numpy(occs.copy())
    for c in np.unique(cells):
        v = torch.from_numpy(vals[cells == c])
        if not torch.isnan(v).any():
            ref[c] = torch.maximum(ref[c] * 0.95, v.max())
    ok = np.array([c not in (4, 6) for c in range(8)])
    assert np.array_equal(ref.numpy()[ok], o[ok])
    assert thre == np.float32(0.01) and np.array_equal(b, want > 0.01)     # the mean clamps to occ_thre
    o2, b2, thre2 = og.update(np.zeros(4, np.float32), np.zeros(4, bool), np.array([1], np.int32),
                              np.array([0.004], np.float32))
    assert np.isclose(thre2, 0.001) and b2.tolist() == [False, True, False, False]   # mean below occ_thre
