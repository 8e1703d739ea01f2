"""NeRSembleVanillaDataManager mirror (datamanager/nersemble_datamanager.py) over the synthetic rig: batch / ray-bundle
contract of SURVEY.md 8b, image cache behaviour, eval iteration.  CPU."""
import pytest
import torch


def _rig(T=6):
    from nersemble_amd.data.synthetic import SyntheticNeRSembleData
    box = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    return SyntheticNeRSembleData(box, n_timesteps=T, n_rays=64, device="cpu")


def _manager(rig, n_train=10, **cfg):
    from nersemble_amd.data.datamanager import NeRSembleVanillaDataManager, NeRSembleVanillaDataManagerConfig
    train_images = [(int(rig.train_cams[i % 12]), i % rig.n_timesteps) for i in range(n_train)]
    eval_images = [(int(c), t) for t in (0, 3) for c in rig.eval_cams]
    config = NeRSembleVanillaDataManagerConfig(**{"train_num_rays_per_batch": 128, "eval_num_rays_per_batch": 32,
                                                  "train_num_images_to_sample_from": 4,
                                                  "train_num_times_to_repeat_images": 3, **cfg})
    g = torch.Generator().manual_seed(5)
    dm = NeRSembleVanillaDataManager(config, rig.image_dataset(train_images, downscale=64),
                                     rig.image_dataset(eval_images, downscale=64), device="cpu", generator=g)
    return dm, train_images, eval_images


def test_next_train_contract():
    rig = _rig()
    dm, train_images, _ = _manager(rig)
    torch.manual_seed(0)
    bundle, batch = dm.next_train(0)
    R = 128
    assert len(bundle) == R and bundle.origins.shape == (R, 3) and bundle.times.shape == (R, 1)
    assert batch["image"].shape == (R, 3) and batch["alpha_map"].shape == (R, 1) and batch["alpha_map"].dtype == torch.uint8
    assert batch["depth_maps"].shape == (R,) and batch["indices"].shape == (R, 3)
    assert batch["timesteps"].shape == (R,) and batch["cam_ids"].shape == (R,)
    for key in ("timesteps", "cam_ids"):                              # per-image attributes, per ray, [R, 1]
        assert bundle.metadata[key].shape == (R, 1) and torch.equal(bundle.metadata[key][:, 0], batch[key])
    img = batch["indices"][:, 0]
    want_cam = torch.tensor([train_images[i][0] for i in img.tolist()])
    want_t = torch.tensor([train_images[i][1] for i in img.tolist()])
    assert torch.equal(batch["cam_ids"], want_cam) and torch.equal(batch["timesteps"], want_t)
    assert torch.equal(bundle.camera_indices[:, 0], img)
    assert torch.allclose(bundle.times[:, 0], want_t.float() / (rig.n_timesteps - 1))
    # the pixels are the ones of the analytic scene seen through the same ray
    rgb, alpha, depth = rig.render_ground_truth(bundle.origins, bundle.directions, bundle.times[:, 0])
    assert torch.allclose(batch["image"], rgb, atol=1e-6) and torch.equal(batch["alpha_map"], alpha)
    assert torch.allclose(batch["depth_maps"], depth, atol=1e-5)
    # ray = pixel centre of (y, x) through the image's camera
    cams = dm.train_dataset.cameras
    i = 7
    one = cams.generate_rays(int(img[i]))
    y, x = int(batch["indices"][i, 1]), int(batch["indices"][i, 2])
    assert torch.allclose(one.directions[y, x], bundle.directions[i], atol=1e-6)
    # code slots for the model: position of each ray's image in the cached batch + the batch's timesteps
    slots, per_image_t = bundle.metadata["image_index"][:, 0].long(), bundle.metadata["_image_timesteps"]
    assert per_image_t.shape == (4,) and torch.equal(per_image_t[slots].long(), batch["timesteps"])
    assert dm.get_train_rays_per_batch() == 128 and dm.train_count == 1


def test_image_cache_is_reused_then_redrawn():
    rig = _rig()
    dm, _, _ = _manager(rig)
    seen = []
    for step in range(7):
        _, batch = dm.next_train(step)
        seen.append(frozenset(batch["indices"][:, 0].tolist()))
    images_per_draw = [frozenset().union(*seen[i:i + 3]) for i in (0, 3)]
    assert all(len(s) <= 4 for s in images_per_draw)                  # 4 images per draw, held for 3 steps
    assert images_per_draw[0] != images_per_draw[1] or len(dm.train_dataset) <= 4
    assert len(dm.train_dataset._cached_items) <= len(dm.train_dataset)
    # all-images mode collates once
    dm_all, _, _ = _manager(rig, train_num_images_to_sample_from=-1)
    a = next(dm_all.iter_train_image_dataloader)
    b = next(dm_all.iter_train_image_dataloader)
    assert a is b and a["image"].shape[0] == len(dm_all.train_dataset)


def test_cache_limit_and_compression():
    rig = _rig()
    ds = rig.image_dataset([(0, 0), (1, 1), (2, 2)], downscale=64, max_cached_items=2, use_cache_compression=True)
    first = ds[0]["image"].clone()
    again = ds[0]["image"]
    assert torch.allclose(again, (first * 255).round() / 255) and (again - first).abs().max() <= 0.5 / 255 + 1e-7
    ds[1], ds[2]
    assert set(ds._cached_items) == {0, 1} and ds._cached_items[0]["image"].dtype == torch.uint8
    assert ds.metadata["camera_frustums"] is rig.camera_frustums and len(ds) == 3


def test_eval_iteration_feeds_the_evaluation_loop():
    rig = _rig()
    dm, _, eval_images = _manager(rig)
    bundle, batch = dm.next_eval(0)
    assert len(bundle) == 32 and bundle.metadata["cam_ids"].shape == (32, 1)
    idx, cam_bundle, cam_batch = dm.next_eval_image(0)
    h, w = cam_bundle.shape
    assert idx == 0 and cam_batch["image"].shape == (h, w, 3) and cam_bundle.metadata["timesteps"].shape == (h, w, 1)
    assert int(cam_bundle.metadata["cam_ids"][0, 0, 0]) == eval_images[0][0]
    assert dm.next_eval_image(1)[0] == 1
    views = list(dm.fixed_indices_eval_dataloader)
    assert len(views) == len(eval_images) == 8
    for (b, item), (cam, t) in zip(views, eval_images):
        assert item["cam_ids"] == cam and item["timesteps"] == t
        assert float(b.times.flatten()[0]) == pytest.approx(t / (rig.n_timesteps - 1))
    tidx, tb, tbatch = dm.next_train_image(0)
    assert tidx == 0 and tb.metadata["cam_ids"].shape == tb.shape + (1,)
    from nersemble_amd.data.datamanager import NeRSembleVanillaDataManagerConfig
    with pytest.raises(NotImplementedError):
        dm.config = NeRSembleVanillaDataManagerConfig(patch_size=2)
        dm._get_pixel_sampler(dm.train_dataset, 16)


@pytest.mark.gpu
def test_training_and_evaluation_through_the_datamanager(cuda):
    """datamanager.next_train -> trainer.train_iteration -> evaluate_novel_views over the datamanager's eval loader:
    the reference's pipeline order (VanillaPipeline.get_train_loss_dict / evaluate_nersemble.py) on the native path."""
    import math
    from nersemble_amd import evaluation as ev
    from nersemble_amd.data.datamanager import NeRSembleVanillaDataManager, NeRSembleVanillaDataManagerConfig
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(11)
    trainer, rig, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    train_images = [(int(rig.train_cams[i % 12]), (5 * i) % rig.n_timesteps) for i in range(12)]
    eval_images = [(int(c), t) for t in (0, 4) for c in rig.eval_cams]
    cfg = NeRSembleVanillaDataManagerConfig(train_num_rays_per_batch=512, train_num_images_to_sample_from=6,
                                            train_num_times_to_repeat_images=4)
    dm = NeRSembleVanillaDataManager(cfg, rig.image_dataset(train_images, downscale=32),
                                     rig.image_dataset(eval_images, downscale=64), device="cuda:0",
                                     generator=torch.Generator().manual_seed(1))
    first = last = None
    for step in range(10):
        bundle, batch = dm.next_train(step)
        assert bundle.origins.is_cuda and batch["image"].is_cuda and bundle.metadata["_image_timesteps"].numel() == 6
        loss = float(trainer.train_iteration(step, bundle, batch)[0].detach())
        assert math.isfinite(loss)
        first = loss if first is None else first
        last = loss
    assert last < first
    model = trainer.model
    model.eval()
    model.config.eval_num_rays_per_chunk = 256
    res = ev.evaluate_novel_views(model, dm.fixed_indices_eval_dataloader,
                                  time_to_timestep=lambda t: round(t * (rig.n_timesteps - 1)))
    assert len(res.per_cam) == 4 and math.isfinite(res.mean.regular.psnr) and math.isfinite(res.mean.masked.psnr)
    assert 0.0 <= res.mean.regular.ssim <= 1.0
    model.train()
