"""Floater filter (nersemble_amd/util/connected_components.py) against golden outputs of the reference's own
``nersemble/util/connected_components.py`` (tests/golden/make_golden.py, numpy + scipy.ndimage run for real; the
cc3d connected-component call is stubbed there with scipy.ndimage.label -> that step is "parity unpinned").
Bit-exact: the outputs are boolean masks."""
import os

import numpy as np
import pytest
import torch

from nersemble_amd.util.connected_components import (extract_top_k_connected_component, filter_occupancy_grid,
                                                     gaussian_filter_integer, largest_connected_component)

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "occupancy_filter.npz"))
R = 40


def _unpack(key):
    return torch.from_numpy(np.unpackbits(G[key])[:R ** 3].astype(bool).reshape(R, R, R))


class _Grid:
    def __init__(self, device):
        self.resolution = torch.tensor([R, R, R])
        self.device = device
        self.occs = torch.from_numpy(G["occ_in"].reshape(-1).copy()).to(device)
        self.binaries = _unpack("occ_binaries_in").reshape(1, R, R, R).to(device)


def _check(device):
    occs = torch.from_numpy(G["occ_in"]).to(device)
    for i in range(3):
        thr, s_thin, s_ero = G[f"occ_args_{i}"]
        got = extract_top_k_connected_component(occs, threshold=float(thr), sigma_thinning=float(s_thin),
                                                sigma_erosion=float(s_ero))[0]
        want = _unpack(f"occ_mask_{i}")
        assert got.dtype == torch.uint8 and torch.equal(got.bool().cpu(), want), i
        assert 0 < want.sum() < R ** 3                       # the floaters were removed, the head kept
    g = _Grid(device)
    filter_occupancy_grid(g, threshold=0.6, sigma_erosion=5)
    assert torch.equal(g.binaries.cpu().reshape(R, R, R), _unpack("occ_binaries_out"))


def test_filter_matches_reference_golden_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_filter_matches_reference_golden_gpu(cuda):
    _check(cuda)


def test_gaussian_filter_integer_is_scipys():
    import scipy.ndimage as ndi
    rng = np.random.default_rng(3)
    a = (rng.random((9, 30, 17)) * 255).astype(np.uint8)            # radius 20 > axis length 9: multiple reflections
    for sigma in (1, 2.5, 5):
        want = ndi.gaussian_filter(a, sigma=sigma)
        got = gaussian_filter_integer(torch.from_numpy(a), sigma)
        assert np.array_equal(got.numpy().astype(np.uint8), want), sigma
    b = (rng.random((20, 20, 20)) > 0.8).astype(np.int64) * 100
    assert np.array_equal(gaussian_filter_integer(torch.from_numpy(b), 5).numpy().astype(np.int64),
                          ndi.gaussian_filter(b, sigma=5))


def test_largest_component_edge_cases():
    import scipy.ndimage as ndi
    assert not largest_connected_component(torch.zeros((5, 6, 7), dtype=torch.bool)).any()
    rng = np.random.default_rng(1)
    m = rng.random((18, 16, 20)) < 0.32                               # many small components, snake-like ones included
    lab, n = ndi.label(m)
    counts = np.bincount(lab.ravel())[1:]
    best = counts.max()
    got = largest_connected_component(torch.from_numpy(m)).numpy()
    assert got.sum() == best and (lab[got] == lab[got][0]).all()      # one whole component of maximal size
    # diagonal neighbours are NOT connected (connectivity 6)
    d = torch.zeros((4, 4, 4), dtype=torch.bool)
    d[0, 0, 0] = d[1, 1, 1] = d[1, 1, 2] = True
    got = largest_connected_component(d)
    assert got.sum() == 2 and got[1, 1, 1] and got[1, 1, 2]


def test_connectivity_is_the_reference_six_not_twenty_six():
    """The reference asks cc3d for ``connectivity = 6`` explicitly (util/connected_components.py:75-80; cc3d's own default
    is 26).  Known-answer volume on which the two differ: a 3x3x3 block and a 2x2x2 block that touch only at a corner, a
    bar that touches the big block only along an edge, and a far-away 2x2x2 block.  Face connectivity keeps them apart
    (largest = the 27 voxels of the big block); with 18- or 26-connectivity they would merge.  The mirror, scipy's label()
    with the face structure (the stub used when the golden was generated) and the known answer agree; the 26-structure
    gives something else -- so the golden cannot hide a connectivity mix-up."""
    import scipy.ndimage as ndi
    v = np.zeros((10, 10, 10), dtype=bool)
    v[1:4, 1:4, 1:4] = True                      # 27 voxels
    v[4:6, 4:6, 4:6] = True                      # corner contact with the big block at (3,3,3)-(4,4,4)
    v[4, 4, 1:3] = False
    v[4:6, 1:3, 4:6] = True                      # edge contact: shares the edge x=3|4, z=3|4 (no common face)
    v[7:9, 7:9, 7:9] = True                      # isolated
    got = largest_connected_component(torch.from_numpy(v)).numpy()
    assert got.sum() == 27 and got[1:4, 1:4, 1:4].all()
    lab6, _ = ndi.label(v)                                                       # default structure: faces only
    big6 = lab6 == np.argmax(np.bincount(lab6.ravel())[1:]) + 1
    assert np.array_equal(got, big6)
    lab26, _ = ndi.label(v, structure=np.ones((3, 3, 3)))
    big26 = lab26 == np.argmax(np.bincount(lab26.ravel())[1:]) + 1
    assert big26.sum() > 27 and not np.array_equal(got, big26)
    lab18, _ = ndi.label(v, structure=ndi.generate_binary_structure(3, 2))
    assert (lab18 == np.argmax(np.bincount(lab18.ravel())[1:]) + 1).sum() > 27
