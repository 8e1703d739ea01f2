"""Device-side pixel sampler (nersemble_amd/data/pixel_sampler.py) against golden outputs of the reference's own
``NeRSemblePixelSampler.collate_image_dataset_batch`` under the same torch seed (tests/golden/make_golden.py;
nerfstudio's PixelSampler base class is a restated stub there).  Integer / gathered outputs: bit-exact."""
import os

import numpy as np
import pytest
import torch

from nersemble_amd.data.pixel_sampler import ADDITIONAL_METADATA, NeRSemblePixelSampler, add_metadata_to_ray_bundle
from nersemble_amd.rays import RayBundle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pixel_sampler.npz"))
KEYS = ("image", "alpha_map", "depth_map", "timesteps", "cam_ids", "image_idx")


def _batch(device="cpu"):
    return {k: torch.from_numpy(G[f"px_in_{k}"]).to(device) for k in KEYS}


def test_collate_matches_reference_golden():
    sampler = NeRSemblePixelSampler(64, additional_metadata=ADDITIONAL_METADATA)
    torch.manual_seed(2024)
    col = sampler.sample(_batch())
    want = {k[len("px_out_"):]: G[k] for k in G.files if k.startswith("px_out_")}
    assert set(col) == set(want)
    for k, v in want.items():
        assert np.array_equal(col[k].numpy(), v), k
    # per-pixel entries are gathered at (c, y, x); per-image entries by c; indices carry the dataset image index
    assert col["image"].shape == (64, 3) and col["depth_map"].shape == (64, 12, 9) and col["timesteps"].shape == (64,)
    assert set(col["indices"][:, 0].tolist()) <= set(G["px_in_image_idx"].tolist())


def test_masked_sampling_matches_reference_golden():
    sampler = NeRSemblePixelSampler(64, additional_metadata=ADDITIONAL_METADATA)
    batch = _batch()
    batch["mask"] = torch.from_numpy(G["px_mask"])
    torch.manual_seed(2025)
    col = sampler.collate_image_dataset_batch(batch, 32)
    for k in (k for k in G.files if k.startswith("px_outm_")):
        assert np.array_equal(col[k[len("px_outm_"):]].numpy(), G[k]), k


def test_metadata_reaches_the_ray_bundle():
    sampler = NeRSemblePixelSampler(16, additional_metadata=ADDITIONAL_METADATA)
    col = sampler.sample(_batch())
    o = torch.zeros((16, 3))
    bundle = RayBundle(origins=o, directions=o + 1, pixel_area=o[:, :1], camera_indices=col["indices"][:, :1])
    add_metadata_to_ray_bundle(bundle, col)
    assert bundle.metadata["timesteps"].shape == (16, 1) and bundle.metadata["cam_ids"].shape == (16, 1)
    assert torch.equal(bundle.metadata["timesteps"][:, 0], col["timesteps"])


@pytest.mark.gpu
def test_sampler_stays_on_the_device(cuda):
    sampler = NeRSemblePixelSampler(4096, additional_metadata=ADDITIONAL_METADATA)
    col = sampler.sample(_batch(cuda))
    assert all(v.is_cuda for v in col.values())
    c, y, x = col["indices"][:, 0], col["indices"][:, 1], col["indices"][:, 2]
    assert int(y.max()) < 12 and int(x.max()) < 9 and int(y.min()) >= 0


@pytest.mark.gpu
def test_device_gather_matches_reference_golden(cuda):
    """The gather half on the DEVICE at the reference's own sampled pixels: the golden's (image, y, x) triples (torch's CPU
    generator under seed 2024 -- a device generator draws other numbers) fed to ``collate_at`` with the image batch
    resident in HBM; every collated entry equal to the reference's output bit for bit."""
    sampler = NeRSemblePixelSampler(64, additional_metadata=ADDITIONAL_METADATA)
    absolute = G["px_out_indices"]
    image_idx = G["px_in_image_idx"]
    local = absolute.copy()
    local[:, 0] = np.searchsorted(np.sort(image_idx), absolute[:, 0])           # dataset image index -> batch-local number
    order = np.argsort(image_idx)
    local[:, 0] = order[local[:, 0]]
    assert np.array_equal(image_idx[local[:, 0]], absolute[:, 0])
    col = sampler.collate_at(_batch(cuda), torch.from_numpy(local).to(cuda))
    want = {k[len("px_out_"):]: G[k] for k in G.files if k.startswith("px_out_")}
    assert set(col) == set(want) and all(v.is_cuda for v in col.values())
    for k, v in want.items():
        assert np.array_equal(col[k].cpu().numpy(), v), k
    # and the masked variant's pixels
    batch = _batch(cuda)
    batch["mask"] = torch.from_numpy(G["px_mask"]).to(cuda)
    absm = G["px_outm_indices"]
    locm = absm.copy()
    locm[:, 0] = order[np.searchsorted(np.sort(image_idx), absm[:, 0])]
    colm = sampler.collate_at(batch, torch.from_numpy(locm).to(cuda))
    for k in (k for k in G.files if k.startswith("px_outm_")):
        assert np.array_equal(colm[k[len("px_outm_"):]].cpu().numpy(), G[k]), k
