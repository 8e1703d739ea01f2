"""GPU: end-to-end training iterations on the native path (small tables): loss decreases, the factored-gradient
optimizer path (single GPU) and the dense-gradient path (what data-parallel ranks run before the all-reduce) give the
same trajectory, checkpoints use the reference's state-dict keys and round-trip."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(factored, steps=6, seed=0, workload="p030_h16"):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(seed)
    trainer, data, info = build_workload(workload, device="cuda:0", small=True, n_rays=512, factored_table_grad=factored)
    losses = []
    for step in range(steps):
        bundle, batch = data.next_train(step)
        loss, loss_dict, metrics = trainer.train_iteration(step, bundle, batch)
        losses.append(loss.item())
    trainer.consolidate()          # (compact first-grid phase, the trainer's default: the callers read `tables` directly)
    return trainer, losses, metrics


def test_training_loss_decreases_and_paths_agree(cuda):
    t_f, l_f, m_f = _run(True, steps=12)
    t_d, l_d, m_d = _run(False, steps=12)
    assert all(np.isfinite(l_f)) and all(np.isfinite(l_d))
    assert l_f[-1] < l_f[0] * 0.8
    # identical data, identical init: factored-on-the-fly Adam == expand + dense Adam up to atomics order.  The two
    # trajectories separate slowly (fp32 summation order -> Adam with eps 1e-15 -> which samples survive the
    # visibility test): tight over the first steps, loose over the whole run
    assert np.allclose(l_f[:5], l_d[:5], rtol=2e-3, atol=1e-5), (l_f, l_d)
    assert np.allclose(l_f, l_d, rtol=3e-2, atol=1e-4), (l_f, l_d)
    # after ONE step the tables agree entry by entry; later steps diverge chaotically on a few entries (Adam with
    # eps = 1e-15 turns summation-order noise of cancelling gradients into +-lr steps, and density changes near the
    # pruning threshold change the sample set), which is why the long run is compared through the loss only
    t1, _, _ = _run(True, steps=1)
    t2, _, _ = _run(False, steps=1)
    a = t1.model.field.hash_ensemble.tables.detach()
    b = t2.model.field.hash_ensemble.tables.detach()
    d = (a - b).abs()
    assert (d <= 1e-5).float().mean().item() >= 0.9999
    assert d.mean().item() <= 1e-7


def test_static_h1_config_runs(cuda):
    """BASELINE configs[0]-like: single timestep, one hash grid (the reference divides by n_timesteps-1 = 0 in the
    occupancy callback, nersemble_instant_ngp.py:189-190; here time == 0)."""
    trainer, losses, metrics = _run(None, steps=4, workload="static_h1")
    assert all(np.isfinite(losses))


def test_checkpoint_roundtrip_reference_keys(cuda):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(3)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256)
    for step in range(2):
        trainer.train_iteration(step, *data.next_train(step))
    sd = trainer.model.state_dict()
    for k in ("field.hash_ensemble.hash_encodings.0.params", "field.hash_ensemble.hash_encodings.3.params",
              "field.mlp_base.params", "field.mlp_head.params", "time_embedding.weight",
              "time_embedding_deformation.weight", "deformation_field.se3_field.mlp_stem.layers.4.weight",
              "deformation_field.aabb", "scene_aabb", "occupancy_grid.occs", "occupancy_grid.binaries"):
        assert k in sd, k
    assert "field.hash_ensemble.tables" not in sd
    torch.manual_seed(4)
    trainer2, data2, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256)
    trainer2.model.load_state_dict(sd)
    # window schedulers are run-time state, not checkpoint state (as in the reference): bring them to the same step
    for cb in trainer2.callbacks[1:]:
        cb.run(1)
    trainer.model.eval(); trainer2.model.eval()
    bundle, batch, hw = data.eval_image_rays(cam=3, timestep=5, downscale=64)
    with torch.no_grad():
        o1 = trainer.model(bundle)["rgb"]
        o2 = trainer2.model(bundle)["rgb"]
    assert torch.equal(o1, o2)
    assert o1.shape[0] == hw[0] * hw[1] and float(o1.min()) >= 0 and float(o1.max()) <= 1     # eval mode clamps


def test_sigma_pass_reuse_is_exact(cuda):
    """The main pass reuses offsets / hash features / mlp_base outputs computed by the sampler's no-grad density pass
    for the samples that survive pruning: outputs must be BIT-identical to recomputing them, gradients equal up to
    atomics order."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(5)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512, factored_table_grad=False)
    model = trainer.model
    for step in range(3):
        trainer.train_iteration(step, *data.next_train(step))
    bundle, batch = data.next_train(7)
    res = {}
    for reuse in (True, False):
        model.reuse_sigma_pass = reuse
        model.zero_grad(set_to_none=True)
        torch.manual_seed(123)                       # same stratified jitter
        with torch.autocast("cuda", dtype=torch.float16, cache_enabled=False):
            out = model(bundle)
            loss = sum(model.get_loss_dict(out, batch).values())
        (loss * 1024.0).backward()
        res[reuse] = (out["rgb"].detach().clone(), out["weights"][0].detach().clone(), loss.item(),
                      model.field.mlp_head.params.grad.clone(), model.field.hash_ensemble.tables.grad.clone(),
                      model.time_embedding_deformation.weight.grad.clone())
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and a[2] == b[2]
    for i in (3, 4, 5):
        d = (a[i] - b[i]).abs().max().item()
        assert d <= 1e-4 * max(b[i].abs().max().item(), 1e-12), (i, d)


@pytest.mark.parametrize("n_rays", [512, 257, 2305])
def test_fused_step_losses_match_composed_losses(n_rays, cuda):
    """csrc/losses.hip (all loss terms, their sum, metrics, and the gradients w.r.t. rgb / accumulation / depth /
    weights in 2+2 launches) against the operator-by-operator losses of models/base.py on the same outputs.
    fp32 reductions in a different order: 1e-5 relative.  (Ray counts on either side of the reduction's rounds of
    256 threads x 8 rays.)"""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(5)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=n_rays)
    for step in range(3):
        trainer.train_iteration(step, *data.next_train(step))
    model = trainer.model
    model.train()
    bundle, batch = data.next_train(3)

    def losses(fused):
        model.fuse_step_losses = fused
        for p in model.parameters():
            p.grad = None
        sink = model.field.hash_ensemble.grad_sink
        if sink is not None:
            sink.clear()
        torch.manual_seed(11)                      # the sampler's jitter
        out = model(bundle)
        leaves = {}
        for k in ("rgb", "accumulation", "depth"):
            leaves[k] = out[k].detach().clone().requires_grad_(True)
            out[k] = leaves[k]
        w = out["weights"][0].detach().clone().requires_grad_(True)
        out["weights"] = (w,)
        leaves["weights"] = w
        metrics = model.get_metrics_dict(out, batch)
        ld = model.get_loss_dict(out, batch, metrics)
        total = getattr(ld, "total", None)
        if fused:
            assert total is not None
        else:
            assert total is None
            import functools
            total = functools.reduce(torch.add, ld.values())
        (total * 1024.0).backward()
        return ({k: float(v) for k, v in ld.items()}, float(total), {k: float(v) for k, v in metrics.items()},
                {k: v.grad.detach().clone() for k, v in leaves.items()})

    ld_f, tot_f, m_f, g_f = losses(True)
    ld_c, tot_c, m_c, g_c = losses(False)
    model.fuse_step_losses = True
    assert set(ld_f) == set(ld_c) and list(ld_f) == list(ld_c), (ld_f, ld_c)
    for k in ld_c:
        assert ld_f[k] == pytest.approx(ld_c[k], rel=1e-5, abs=1e-9), k
    assert tot_f == pytest.approx(tot_c, rel=1e-5)
    assert set(m_f) == set(m_c)
    for k in m_c:
        assert m_f[k] == pytest.approx(m_c[k], rel=1e-5), k
    for k in g_c:
        scale = g_c[k].abs().max().item() + 1e-20
        assert (g_f[k] - g_c[k]).abs().max().item() <= 2e-5 * scale, k
        assert g_c[k].abs().max().item() > 0, k


def test_deferred_scheduler_step_matches_immediate(cuda):
    """The LR-scheduler step is applied lazily (no end-of-step host sync): after flush the scheduler state equals
    one step per clean iteration."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(1)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256)
    for step in range(4):
        trainer.train_iteration(step, *data.next_train(step))
    trainer.flush_scheduler_step()
    for sch in trainer.schedulers.values():
        assert sch.last_epoch == 4


def test_eval_preblend_fast_path_matches_per_sample_blend(cuda):
    """Rendering an evaluation image (one timestep for all rays) through the pre-blended grid gives the image of the
    regular path: same samples, colours within fp16 blend-order noise."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(2)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    for step in range(6):
        trainer.train_iteration(step, *data.next_train(step))
    model = trainer.model
    model.eval()
    bundle, _, _ = data.eval_image_rays(cam=1, timestep=3, downscale=64)
    outs = {}
    with torch.no_grad():
        for fast in (False, True):
            model.eval_preblend = fast
            torch.manual_seed(7)
            outs[fast] = model(bundle)
            assert (model._eval_blend_cache[0] is not None) == fast or fast is False
    a, b = outs[False], outs[True]
    n_a, n_b = a["num_samples_per_ray"].sum().item(), b["num_samples_per_ray"].sum().item()
    assert abs(n_a - n_b) <= 0.01 * max(n_a, 1)           # a density at the alpha threshold may flip a sample
    assert (a["rgb"] - b["rgb"]).abs().max().item() <= 2e-2
    assert (a["rgb"] - b["rgb"]).abs().mean().item() <= 2e-3
    assert (a["accumulation"] - b["accumulation"]).abs().mean().item() <= 2e-3
    # a bundle with several timesteps must not take the fast path
    model.eval_preblend = True
    tb, _ = data.next_train(99)
    with torch.no_grad():
        model(tb)
    assert model._eval_blend is None
    model.train()


def test_eval_image_through_the_fused_density_pass_is_the_same_image(cuda):
    """An evaluation bundle (one timestep) with the fused density pass (one launch: pre-blended lookup -> mlp_base -> trunc_exp;
    the sampler's sigma_fn reads ONE deformation code row) against the same bundle with the four-launch route: the same
    samples per ray and the same image, bit for bit."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(3)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    for step in range(6):
        trainer.train_iteration(step, *data.next_train(step))
    model = trainer.model
    model.eval()
    bundle, _, _ = data.eval_image_rays(cam=2, timestep=5, downscale=64)
    outs = {}
    with torch.no_grad():
        for fused in (False, True):
            model.field.fused_eval_density = fused
            torch.manual_seed(7)
            outs[fused] = model(bundle)
    a, b = outs[False], outs[True]
    assert int(a["num_samples_per_ray"].sum()) > 0
    assert torch.equal(a["num_samples_per_ray"], b["num_samples_per_ray"])
    for k in ("rgb", "accumulation", "depth"):
        assert torch.equal(a[k], b[k]), k
    model.train()


def test_dense_march_config_runs(cuda):
    """BASELINE configs[3]-like: --disable_occupancy_grid --lambda_dist_loss 0 (every cell occupied, density_fn == 1 for
    the visibility pass, no distortion loss -> the operator-by-operator loss path)."""
    trainer, losses, metrics = _run(None, steps=4, workload="p097_dense")
    assert all(np.isfinite(losses))
    assert trainer.model.config.disable_occupancy_grid and trainer.model.occupancy_grid.binaries.all()
    assert float(metrics["num_samples_per_batch"]) > 0


def test_cached_parameter_packs_follow_fused_optimizers(cuda):
    """torch's fused optimizers update parameters without bumping Tensor._version; the cached fp16 working tables and
    the cached deformation weight fragments must still follow them."""
    from nersemble_amd.field_components.deformation_field import SE3DeformationField, SE3DeformationFieldConfig
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    torch.manual_seed(0)
    df = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(cuda)
    pos = (torch.rand(200, 3) * (aabb[1] - aabb[0]) + aabb[0]).to(cuda)
    codes = torch.randn(200, 128, device=cuda) * 0.3
    opt = torch.optim.Adam(df.parameters(), lr=1e-2, fused=True)
    off0 = df.compute_offsets(pos, codes, None)
    off0.square().sum().backward()
    opt.step()
    with torch.no_grad():
        off1 = df.compute_offsets(pos, codes, None)
        fresh = SE3DeformationField(aabb, SE3DeformationFieldConfig(warp_code_dim=128)).to(cuda)
        fresh.load_state_dict(df.state_dict())
        off_fresh = fresh.compute_offsets(pos, codes, None)
    assert not torch.equal(off0.detach(), off1) and torch.equal(off1, off_fresh)

    he = HashEnsemble(HashEnsembleConfig(4, TCNNHashEncodingConfig(n_levels=4, log2_hashmap_size=10), True, True)).to(cuda)
    with torch.no_grad():
        he.tables.mul_(1000)
    x = torch.rand((500, 3), device=cuda)
    code = torch.randn((500, 4), device=cuda)
    opt = torch.optim.Adam([he.tables], lr=1e-2, fused=True)
    y0 = he(x, code)
    y0.float().square().sum().backward()
    opt.step()
    with torch.no_grad():
        y1 = he(x, code)
    assert torch.equal(he.half_tables(), he.tables.detach().half()) and not torch.equal(y0.detach(), y1)


def test_eval_preblend_cache_follows_training(cuda):
    """eval(t0) -> train k steps -> eval(t0): the pre-blended grid must be rebuilt (the table optimizers write through
    raw pointers and torch's fused Adam leaves Tensor._version alone, so the cache keys on the optimizer-step count)."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(5)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    model = trainer.model
    model.sched_window_hash_encodings.update(90000)          # window saturated: only the weights change below
    for step in range(3):
        trainer.train_iteration(step, *data.next_train(step))
    bundle, _, _ = data.eval_image_rays(cam=1, timestep=3, downscale=64)

    def render(fast):
        model.eval()
        model.eval_preblend = fast
        with torch.no_grad():
            return model(bundle)["rgb"].clone()

    first = render(True)
    blend_before = model._eval_blend_cache[1].clone()
    model.train()
    for step in range(3, 40):
        trainer.train_iteration(step, *data.next_train(step))
    fast = render(True)
    assert not torch.equal(model._eval_blend_cache[1], blend_before)          # rebuilt from the trained tables
    slow = render(False)
    assert (fast - slow).abs().mean().item() <= 2e-3 and (fast - slow).abs().max().item() <= 2e-2
    assert (fast - first).abs().mean().item() > 5 * (fast - slow).abs().mean().item()   # training did move the image
    model.train()


def test_resume_from_checkpoint_continues_the_run(cuda):
    """Model + training state (Adam moments of all groups -- the table moments in the reference's tcnn layout --, step
    counts, StepLR counters, loss scale) through the nerfstudio checkpoint dict; the resumed run takes the same steps."""
    from nersemble_amd.util.checkpoint import nerfstudio_checkpoint_from_model, resume_trainer_from_checkpoint
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(8)
    a, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    batches = [data.next_train(s) for s in range(7)]
    for step in range(4):
        a.train_iteration(step, *batches[step])
    ckpt = nerfstudio_checkpoint_from_model(a.model, 4, trainer=a)
    assert set(ckpt) >= {"step", "pipeline", "optimizers", "scalers"}
    # the reference's shape: one torch.optim.Adam state dict per parameter group of get_param_groups; `fields` holds the C
    # tcnn encodings (flat moments in tcnn layout) in front of mlp_base / mlp_head, each with its own `step`
    assert set(ckpt["optimizers"]) == {"fields", "embeddings", "deformation_field"}
    he = a.model.field.hash_ensemble
    C = he.n_tcnn_encodings
    fields = ckpt["optimizers"]["fields"]
    # the reference's numbering (tests/golden/state_manifest.json): direction_encoding.params (empty, no state), the C hash
    # encodings, position_encoding.params (empty, no state), mlp_base, mlp_head
    tab_ids, mlp_ids = list(range(1, C + 1)), [C + 2, C + 3]
    assert fields["param_groups"][0]["params"] == list(range(C + 4)) and set(fields["state"]) == set(tab_ids + mlp_ids)
    assert all(int(fields["state"][i]["step"]) == 4 for i in tab_ids + mlp_ids)
    for c in range(C):
        assert fields["state"][1 + c]["exp_avg"].shape == ckpt["pipeline"][f"_model.field.hash_ensemble.hash_encodings.{c}.params"].shape
    assert fields["state"][C + 2]["exp_avg"].shape == ckpt["pipeline"]["_model.field.mlp_base.params"].shape
    assert float(sum(fields["state"][i]["exp_avg_sq"].abs().sum() for i in tab_ids)) > 0
    # torch.optim.Adam over the reference's parameter list accepts it (what nerfstudio's Optimizers.load_optimizers does)
    sd = ckpt["pipeline"]
    ref_params = [torch.nn.Parameter(sd["_model.field.direction_encoding.params"].clone())] + \
        [torch.nn.Parameter(sd[f"_model.field.hash_ensemble.hash_encodings.{c}.params"].clone()) for c in range(C)] + \
        [torch.nn.Parameter(sd["_model.field.position_encoding.params"].clone()),
         torch.nn.Parameter(sd["_model.field.mlp_base.params"].clone()),
         torch.nn.Parameter(sd["_model.field.mlp_head.params"].clone())]
    ref_adam = torch.optim.Adam(ref_params, lr=5e-3, eps=1e-15)
    ref_adam.load_state_dict(fields)
    assert int(ref_adam.state[ref_params[1]]["step"]) == 4 and ref_params[0] not in ref_adam.state
    # the time codes are not trained while the window is closed: no state for time_embedding.weight, as in torch
    emb = ckpt["optimizers"]["embeddings"]
    assert set(emb["state"]) == {1} and emb["param_groups"][0]["params"] == [0, 1]
    occ_a = (a.model.occupancy_grid.occs.clone(), a.model.occupancy_grid.binaries.clone())

    def snapshot(tr):
        """Every tensor the next step depends on."""
        snap = {"p:" + n: p.detach().clone() for n, p in tr.model.named_parameters()}
        snap["f16"] = tr.model.field.hash_ensemble.half_tables().clone()
        for key, opt in tr.optimizers.items():
            for i, (p, st) in enumerate(opt.state.items()):
                for k, v in st.items():
                    snap[f"o:{key}:{i}:{k}"] = v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v))
            snap[f"lr:{key}"] = torch.tensor([g["lr"] for g in opt.param_groups])
        snap["scale"] = tr.grad_scaler._scale.clone()
        snap["growth"] = tr.grad_scaler._growth_tracker.clone()
        return snap

    snap_a = snapshot(a)
    def three_more_steps(trainer):
        out = []
        for step in range(4, 7):
            torch.manual_seed(100 + step)                     # the marcher's near-plane jitter draws from the global generator
            out.append(trainer.train_iteration(step, *batches[step])[0].item())
        return out

    losses_a = three_more_steps(a)

    torch.manual_seed(1234)                                   # different initial weights: everything comes from the file
    b, _, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    assert resume_trainer_from_checkpoint(ckpt, b) == 4
    assert torch.equal(b.model.occupancy_grid.occs, occ_a[0]) and torch.equal(b.model.occupancy_grid.binaries, occ_a[1])
    assert b.grad_scaler.get_scale() == ckpt["scalers"]["scale"]
    snap_b = snapshot(b)
    assert set(snap_a) == set(snap_b), set(snap_a) ^ set(snap_b)
    for k in snap_a:
        assert torch.equal(snap_a[k].cpu().float(), snap_b[k].cpu().float()), k
    for sched in (b.model.sched_window_deform, b.model.sched_window_hash_encodings):
        sched.update(3)                                       # (window schedules are functions of the step)
    losses_b = three_more_steps(b)
    # same data, same jitter, same weights, same moments: the runs agree up to the order of the fp32 atomics
    assert np.allclose(losses_a, losses_b, rtol=1e-2), (losses_a, losses_b)
    sa = a.optimizers["fields/tables"].state[a.model.field.hash_ensemble.tables]["step"]
    sb = b.optimizers["fields/tables"].state[b.model.field.hash_ensemble.tables]["step"]
    assert sa == sb == 7


def test_split_scatter_backward_trains_like_the_fused_one(cuda):
    """NSX_SPLIT_SCATTER path (scatter half of the factored backward on its own stream, consumers ordered by
    ``FactoredGradSink.wait_scatter``): same trajectory as the fused kernel up to the order of the atomics."""
    from nersemble_amd.workloads import build_workload
    runs = {}
    for split in (False, True):
        torch.manual_seed(4)
        trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
        trainer.model.fuse_main_pass = False                 # the split lives in the modular path's HashEnsemble backward
        sink = trainer.model.field.hash_ensemble.grad_sink
        sink.split_scatter, sink.scatter_blocks_per_cu = split, 2
        losses = []
        for step in range(6):
            torch.manual_seed(50 + step)
            losses.append(trainer.train_iteration(step, *data.next_train(step))[0].item())
        trainer.flush_scheduler_step()
        assert (sink.scatter_stream is not None) == split
        runs[split] = (losses, trainer.model.field.hash_ensemble.tables.detach().clone())
    assert np.allclose(runs[True][0][:4], runs[False][0][:4], rtol=2e-3), runs
    assert np.allclose(runs[True][0], runs[False][0], rtol=3e-2), runs


@pytest.mark.parametrize("window_open", [False, True])
def test_fused_main_pass_equals_modular_path(window_open, cuda):
    """engine/fused_pass.py (the kept samples' main pass as ONE autograd node) against the modular path (nine autograd
    Functions): the same kernels in the same order, so the loss vector is equal bit for bit and the gradients agree up
    to the order of the fp32 atomics; with and without reuse of the sigma_fn pass's forward values.  ``window_open``:
    every hash grid on (the schedule after step 80 000, train_nersemble.py:77-78) -- the time codes are trained and
    their gradient comes out of the HashEnsemble backward summed per code row."""
    from nersemble_amd.workloads import build_workload
    for reuse in (True, False):
        res = {}
        for fused in (False, True):
            torch.manual_seed(6)
            trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512,
                                              window_hash=(0, 1) if window_open else None)
            model = trainer.model
            model.fuse_main_pass, model.reuse_sigma_pass = fused, reuse
            if window_open:                                       # open from step 0 on (the ramp is over before it)
                model.sched_window_hash_encodings.begin_step, model.sched_window_hash_encodings.end_step = -2, -1
            calls = []
            orig = model.fused_train_forward

            def counted(*a, _orig=orig, _calls=calls, **k):
                r = _orig(*a, **k)
                _calls.append(r is not None)
                return r

            model.fused_train_forward = counted
            losses, terms = [], None
            for step in range(5):
                torch.manual_seed(70 + step)                      # same near-plane jitter on both sides
                loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step))
                losses.append(loss.item())
                if step == 0:
                    terms = {k: v.item() for k, v in loss_dict.items()}
                    terms.update({"m:" + k: float(v) for k, v in metrics.items()})
                    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
                    model.field.hash_ensemble.wait_tables()
                    tables = model.field.hash_ensemble.tables.detach().clone()       # after ONE optimizer step
            trainer.flush_scheduler_step()
            assert all(calls) == fused and len(calls) == 5
            res[fused] = (losses, terms, grads, tables)
        (l_m, t_m, g_m, tab_m), (l_f, t_f, g_f, tab_f) = res[False], res[True]
        assert l_m[0] == l_f[0] and t_m == t_f, (reuse, l_m[0], l_f[0], t_m, t_f)     # forward: bit for bit
        assert set(g_m) == set(g_f)
        assert ("time_embedding.weight" in g_m) == window_open            # the codes get a gradient once the window is open
        if window_open:
            assert g_m["time_embedding.weight"].abs().max().item() > 0
        for name in g_m:
            sc = g_m[name].abs().max().item()
            assert (g_m[name] - g_f[name]).abs().max().item() <= 1e-4 * sc + 1e-12, (reuse, name)
        assert np.allclose(l_m, l_f, rtol=2e-3), (reuse, l_m, l_f)
        # one Adam step (+-lr per touched entry): equal unless a cancelling gradient changed sign with the atomics' order;
        # later steps diverge chaotically from there, which is why the long run is compared through the loss
        d = (tab_m - tab_f).abs()
        assert (d <= 1e-5).float().mean().item() >= 0.9995


def test_early_table_step_keeps_gradscaler_skip_semantics(cuda):
    """The table optimizer launched from inside the backward (HashTableAdam.arm_early_step) still skips its group as a
    whole when the scaled gradients overflow: tables, fused-MLP parameters and all moments untouched, the scale halves,
    the step count does not advance; the next (finite) step goes through."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(9)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512, compact_first_grid=False)
    trainer.early_table_step = True                                      # opt-in (measured slower, see the trainer)
    model = trainer.model
    he = model.field.hash_ensemble
    trainer.train_iteration(0, *data.next_train(0))                     # a normal step first (moments exist)
    trainer.flush_scheduler_step()
    he.wait_tables()
    opt = trainer.optimizers["fields/tables"]
    before = {"tables": he.tables.detach().clone(), "f16": he.half_tables().clone(),
              "m": opt.state[he.tables]["exp_avg"].clone(), "base": model.field.mlp_base.params.detach().clone(),
              "deform": model.deformation_field.se3_field.mlp_stem.layers[0].weight.detach().clone()}
    trainer.grad_scaler._scale.fill_(2.0 ** 60)                          # every fp16 gradient overflows
    armed = []
    orig = opt._early_step
    opt._early_step = lambda: (orig(), armed.append(opt.stepped_early))[0]
    trainer.train_iteration(1, *data.next_train(1))
    trainer.flush_scheduler_step()
    he.wait_tables()
    assert armed == [True]                                               # the early path was taken ...
    assert torch.equal(he.tables, before["tables"]) and torch.equal(he.half_tables(), before["f16"])   # ... and skipped
    assert torch.equal(opt.state[he.tables]["exp_avg"], before["m"])
    assert torch.equal(model.field.mlp_base.params, before["base"])
    assert opt.state[he.tables]["step"] == 1 and trainer.optimizers["fields"].step_count == 1
    assert trainer.grad_scaler.get_scale() == 2.0 ** 59
    trainer.grad_scaler._scale.fill_(65536.0)
    trainer.train_iteration(2, *data.next_train(2))
    trainer.flush_scheduler_step()
    he.wait_tables()
    assert not torch.equal(he.tables, before["tables"]) and opt.state[he.tables]["step"] == 2
    assert not torch.equal(model.field.mlp_base.params, before["base"])


def test_device_side_sample_counts_equal_host_side_counts(cuda):
    """The training fast path with the kept-sample count left on the device (one host read-back per step: the marcher's
    own) against the same path with the count read back (``nonzero``): the kernels process the same rows, so the loss
    vector is equal bit for bit, the gradients up to the order of atomics; the padded tails never leak into a result."""
    import ctypes as C
    from nersemble_amd import _lib
    from nersemble_amd.workloads import build_workload
    res = {}
    for on_device in (False, True):
        torch.manual_seed(12)
        trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
        model = trainer.model
        model.device_sample_counts = on_device
        model.occupancy_grid.occs.fill_(0.5)                      # visibility threshold at its cap: pruning does bite
        losses, kept, marched = [], [], []
        for step in range(24):
            torch.manual_seed(90 + step)
            loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step))
            losses.append(loss.item())
            kept.append(int(metrics["num_samples_per_batch"]))
            marched.append(model.occupancy_grid.last_n_marched)
            if step == 0:
                terms = {k: v.item() for k, v in loss_dict.items()}
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        trainer.flush_scheduler_step()
        assert (model.occupancy_grid.last_n_kept is not None) == on_device
        res[on_device] = (losses, kept, terms, grads, marched)
    (l_h, k_h, t_h, g_h, m_h), (l_d, k_d, t_d, g_d, m_d) = res[False], res[True]
    assert m_h[0] == m_d[0] and k_h[0] == k_d[0] and 0 < k_h[0] <= m_h[0]      # same marched set, same kept count
    assert any(k < m for k, m in zip(k_d, m_d)), (k_d, m_d)                  # the visibility test does prune on the way
    assert l_h[0] == l_d[0] and t_h == t_d                              # forward: bit for bit
    for name in g_h:
        sc = g_h[name].abs().max().item()
        assert (g_h[name] - g_d[name]).abs().max().item() <= 1e-4 * sc + 1e-12, name
    assert np.allclose(l_h[:6], l_d[:6], rtol=2e-3) and k_h[:3] == k_d[:3], (l_h, l_d, k_h, k_d)
    assert np.allclose(l_h, l_d, rtol=5e-2)


def test_device_count_argument_of_the_c_abi(cuda):
    """``n_device`` (include/nsx.h, "Device-side element counts"): a per-sample entry point handed a device-side count
    touches only the first ``min(*n_device, rows)`` rows -- straight through the C ABI, no state between calls: the same
    call with NULL right after it processes every row.  Then the Python convenience over it (``_lib.device_count`` +
    ``ndev``): the pointer goes to calls of exactly the block's capacity, blocks do not nest."""
    import ctypes as C
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, device_count, lib, ndev, ptr, stream
    src = torch.arange(40, device=cuda, dtype=torch.float32).reshape(10, 4)
    idx = torch.tensor([9, 8, 7, 6, 5, 4, 3, 2, 1, 0], device=cuda)
    n = torch.tensor([3], device=cuda, dtype=torch.int64)

    def gather_c(count_ptr):
        dst = torch.full((10, 4), -1.0, device=cuda)
        srcs, dsts, rb = (C.c_void_p * 1)(src.data_ptr()), (C.c_void_p * 1)(dst.data_ptr()), (C.c_int64 * 1)(16)
        check(lib().nsx_gather_rows(1, srcs, rb, dsts, ptr(idx), 10, count_ptr, stream()), "nsx_gather_rows")
        return dst

    part, full = gather_c(ptr(n)), gather_c(None)
    assert torch.equal(part[:3], src[idx[:3]]) and bool((part[3:] == 0).all())          # (gather zero-fills its tail)
    assert torch.equal(full, src[idx])
    # a kernel that leaves rows beyond the count alone: the density epilogue
    base = torch.randn(10, 16, device=cuda).half()
    sel = torch.ones(10, dtype=torch.uint8, device=cuda)
    dens = torch.full((10, 1), -7.0, device=cuda)
    check(lib().nsx_density_fwd(ptr(base), 16, ptr(sel), 10, ptr(dens), ptr(n), stream()), "nsx_density_fwd")
    assert torch.equal(dens[:3, 0], torch.exp(base[:3, 0].float())) and bool((dens[3:] == -7.0).all())
    # the Python scope
    with device_count(n, 10):
        assert ndev(10).value == n.data_ptr() and not ndev(5).value
        (part2,) = F.gather_rows(idx, src, zero_fill=True)
        (other,) = F.gather_rows(idx[:5], src, zero_fill=True)              # size != capacity: every row
        with pytest.raises(RuntimeError):
            with device_count(n, 10):
                pass
    assert not ndev(10).value
    (full2,) = F.gather_rows(idx, src)
    assert torch.equal(part2, part) and torch.equal(other, src[idx[:5]]) and torch.equal(full2, src[idx])
    n.fill_(0)
    assert bool((gather_c(ptr(n)) == 0).all())
    n.fill_(25)                                                               # more than the capacity: capped
    assert torch.equal(gather_c(ptr(n)), src[idx])


def test_march_counted_one_step_ahead_is_the_same_march(cuda):
    """``OccGridEstimator.prefetch_march`` (the counting pass of the traversal on a side stream, one step ahead) followed
    by ``sampling`` == ``sampling`` alone, bit for bit, when the jitter draws the same random numbers; a grid update in
    between discards the prefetched pass."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(21)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    for step in range(3):
        trainer.train_iteration(step, *data.next_train(step))
    model = trainer.model
    grid, cfg = model.occupancy_grid, model.config
    bundle, _ = data.next_train(3)
    o, d = bundle.origins.contiguous(), bundle.directions.contiguous()
    kw = dict(near_plane=cfg.near_plane, far_plane=cfg.far_plane, render_step_size=cfg.render_step_size, stratified=True)

    def sample():
        out = grid.sampling(rays_o=o, rays_d=d, sigma_fn=None, alpha_thre=0.0, early_stop_eps=0.0, **kw)
        return [t.clone() for t in out], grid.last_march_prefetched

    torch.manual_seed(5)
    ref, used = sample()
    assert not used and ref[0].shape[0] > 0
    torch.manual_seed(5)
    assert grid.prefetch_march(o, d, **kw)
    got, used = sample()
    assert used
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # a later call without a new prefetch marches on its own
    torch.manual_seed(5)
    got, used = sample()
    assert not used and all(torch.equal(a, b) for a, b in zip(ref, got))
    # the grid changes between prefetch and use: the prefetched pass is dropped
    torch.manual_seed(5)
    assert grid.prefetch_march(o, d, **kw)
    cells = torch.arange(0, grid.cells_per_lvl, 7, dtype=torch.int32, device="cuda:0")
    grid.apply_update(cells, torch.zeros(cells.shape[0], device="cuda:0"), occ_thre=0.5, ema_decay=0.0)
    torch.manual_seed(5)
    fresh_after, used = sample()
    assert not used
    assert fresh_after[0].shape[0] < ref[0].shape[0]                     # cells were emptied: fewer samples


def test_trainer_prefetches_the_next_batch(cuda):
    """``train_iteration(..., next_ray_bundle=)``: every step that is not preceded by a grid update finds its sample count
    on the host; the run trains like one without the prefetch (same data, different jitter draws)."""
    from nersemble_amd.workloads import build_workload

    def run(prefetch, steps=36):
        torch.manual_seed(8)
        trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
        trainer.prefetch_march = prefetch
        batches = [data.next_train(s) for s in range(steps + 1)]
        used, losses = [], []
        for s in range(steps):
            loss, _, _ = trainer.train_iteration(s, *batches[s], next_ray_bundle=batches[s + 1][0])
            used.append(bool(trainer.model.occupancy_grid.last_march_prefetched))
            losses.append(loss.detach())
        return used, [float(l) for l in losses]

    used, losses = run(True)
    assert used == [s > 0 and s % 16 != 0 for s in range(36)], used
    used0, losses0 = run(False)
    assert not any(used0)
    assert np.isfinite(losses).all() and losses[-1] < 0.8 * losses[0]
    assert abs(np.mean(losses[-8:]) - np.mean(losses0[-8:])) < 0.25 * np.mean(losses0[-8:])


def _compact_run(compact, steps, window_hash=None, seed=31, ramp=True, window_at_step0=None, planes=None):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(seed)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512, compact_first_grid=compact,
                                      window_hash=window_hash)
    he = trainer.model.field.hash_ensemble
    he.compact_window_ramp = ramp
    if planes is not None:
        he.first_grid_planes_default = planes
    if window_at_step0 is not None:
        # a schedule whose value at step 0 is `window_at_step0` and that barely moves over a few steps
        sch = trainer.model.sched_window_hash_encodings
        span = 100000.0
        sch.begin_step = -(window_at_step0 - 1.0) / (he.n_hash_encodings - 1.0) * span
        sch.end_step = sch.begin_step + span
    init = he.tables.detach().clone()
    losses, in_phase, first = [], [], None
    for step in range(steps):
        torch.manual_seed(500 + step)                                   # same near-plane jitter on both sides
        loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step))
        losses.append(loss.item())
        in_phase.append(he._compact["width"] if he._compact is not None else 0)
        if step == 0:
            first = {k: v.item() for k, v in loss_dict.items()}
            trainer.consolidate()
            he.wait_tables()
            first["tables"] = he.tables.detach().clone()
    trainer.flush_scheduler_step()
    return trainer, init, losses, in_phase, first


def test_compact_first_grid_phase_is_the_same_training(cuda):
    """``NeRSembleTrainer(compact_first_grid=True)``: while the window keeps one hash grid on, a contiguous copy of that
    grid is trained with the H = 1 kernels.  Same loss bit for bit in the first forward, the same table after one
    optimizer step (up to the order of the atomics), the other grids untouched, the same run afterwards."""
    t_c, init, l_c, phase_c, f_c = _compact_run(True, 12)
    t_f, _, l_f, phase_f, f_f = _compact_run(False, 12)
    assert all(phase_c) and not any(phase_f)
    tab_c, tab_f = f_c.pop("tables"), f_f.pop("tables")
    assert f_c == f_f, (f_c, f_f)                                            # every loss term of step 0: bit for bit
    d = (tab_c - tab_f).abs()
    assert (d <= 1e-5).float().mean().item() >= 0.9995
    assert torch.equal(tab_c[:, :, 1:], init[:, :, 1:])                      # Adam does not move a grid that is off
    assert torch.equal(tab_f[:, :, 1:], init[:, :, 1:])
    assert (tab_c[:, :, 0] != init[:, :, 0]).any()
    assert np.allclose(l_c[:5], l_f[:5], rtol=2e-3), (l_c, l_f)
    assert np.allclose(l_c, l_f, rtol=5e-2), (l_c, l_f)
    # checkpoints see the trained grid: model.state_dict() and the optimizer's moments in the reference layout
    he = t_c.model.field.hash_ensemble
    sd_c = t_c.model.state_dict()
    assert he._compact is not None                                           # saving does not end the phase
    assert torch.equal(he.tables.detach()[:, :, 0:1], he._compact["master"])
    assert torch.equal(he.half_tables()[:, :, 0:1], he._compact["f16"])
    from nersemble_amd import functional as Fn
    tc = Fn.tables_to_tcnn(he.tables.detach(), he.n_hash_encodings, he.geom)
    assert torch.equal(sd_c["field.hash_ensemble.hash_encodings.0.params"], tc[0].reshape(-1))
    st = t_c.state_dict()["optimizers"]["fields"]["state"][1]        # (position 0 is the empty direction_encoding.params)
    st_f = t_f.state_dict()["optimizers"]["fields"]["state"][1]
    assert int(st["step"]) == int(st_f["step"]) == 12
    a, b = st["exp_avg_sq"], st_f["exp_avg_sq"]
    assert a.abs().sum().item() > 0
    assert abs(a.abs().sum().item() - b.abs().sum().item()) <= 0.05 * b.abs().sum().item()
    # evaluation leaves the phase (pre-blended grids and everything else read the full layout) and agrees with the full run
    t_c.model.eval()
    assert he._compact is None
    t_c.model.train()


@pytest.mark.parametrize("planes", [1, 4])
def test_first_grid_phase_with_fewer_gradient_planes_is_the_same_training(planes, cuda):
    """``HashEnsemble.first_grid_planes``: in the compact first-grid phase every code row is the same one, so the factored
    gradient may keep P planes (a sample adds to plane ``slot % P``) instead of one per code row -- the sum over the planes,
    which is all the optimizer reads, is the same up to the order of the additions."""
    t_p, init, l_p, phase_p, f_p = _compact_run(True, 8, planes=planes)
    t_r, _, l_r, phase_r, f_r = _compact_run(True, 8, planes=0)
    assert all(phase_p) and all(phase_r)
    cache_p, cache_r = t_p.model.field.hash_ensemble.grad_sink._cache, t_r.model.field.hash_ensemble.grad_sink._cache
    assert list(cache_p) == [planes] and list(cache_r)[0] > 4            # G: [planes][entries][2] against one plane per row
    tab_p, tab_r = f_p.pop("tables"), f_r.pop("tables")
    assert f_p == f_r, (f_p, f_r)                                            # the forward does not know about planes
    d = (tab_p - tab_r).abs()
    assert (d <= 1e-5).float().mean().item() >= 0.9995
    assert torch.equal(tab_p[:, :, 1:], init[:, :, 1:])
    assert (tab_p[:, :, 0] != init[:, :, 0]).any()
    assert np.allclose(l_p[:5], l_r[:5], rtol=2e-3), (l_p, l_r)
    assert np.allclose(l_p, l_r, rtol=5e-2), (l_p, l_r)


def test_compact_phase_ends_when_the_window_opens(cuda):
    """Window schedule (4, 12): one grid until step 4, then the window grows -- the compact copy is written back into
    column 0 of the full layout and training continues there."""
    t_c, init, l_c, phase_c, _ = _compact_run(True, 10, window_hash=(4, 12), ramp=False)
    t_f, _, l_f, phase_f, _ = _compact_run(False, 10, window_hash=(4, 12))
    assert phase_c == [int(s <= 4) for s in range(10)], phase_c
    assert not any(phase_f)
    assert np.allclose(l_c[:5], l_f[:5], rtol=2e-3), (l_c, l_f)
    assert np.allclose(l_c, l_f, rtol=5e-2), (l_c, l_f)
    a = t_c.model.field.hash_ensemble
    b = t_f.model.field.hash_ensemble
    a.wait_tables(), b.wait_tables()
    assert (a.tables.detach()[:, :, 1] != init[:, :, 1]).any()               # the second grid trains once it is on


@pytest.mark.parametrize("window,width", [(1.5, 2), (3.5, 4), (5.5, 8)])
def test_compact_window_ramp_widths_are_the_same_training(window, width, cuda):
    """Window ramp (train_nersemble.py:77-78; hash_ensemble.py:133-138): while ceil(window) <= width < H the grids
    width ... H - 1 have zero window weight, zero gradient and zero Adam moments; the module trains a contiguous copy of the
    first `width` grids with the H = width kernels.  Against the full layout at the same window: every loss term of the
    first step bit for bit, the tables after one optimizer step up to the order of the atomics -- the grids beyond the
    width untouched --, the time codes' gradient in place (their first `width` columns), the same run afterwards."""
    t_c, init, l_c, phase_c, f_c = _compact_run(True, 6, window_at_step0=window)
    t_f, _, l_f, phase_f, f_f = _compact_run(False, 6, window_at_step0=window)
    assert phase_c == [width] * 6 and not any(phase_f), (phase_c, phase_f)
    tab_c, tab_f = f_c.pop("tables"), f_f.pop("tables")
    assert f_c == f_f, (f_c, f_f)                                            # every loss term of step 0: bit for bit
    d = (tab_c - tab_f).abs()
    assert (d <= 1e-5).float().mean().item() >= 0.9995
    n_on = int(np.ceil(window))                                              # grids with a non-zero window weight
    assert torch.equal(tab_c[:, :, n_on:], init[:, :, n_on:])                # Adam does not move a grid that is off
    assert torch.equal(tab_f[:, :, n_on:], init[:, :, n_on:])
    assert (tab_c[:, :, n_on - 1] != init[:, :, n_on - 1]).any()             # the last grid that is on trains
    assert np.allclose(l_c[:4], l_f[:4], rtol=2e-3), (l_c, l_f)
    ga = t_c.model.time_embedding.weight.grad
    gb = t_f.model.time_embedding.weight.grad
    assert ga is not None and gb is not None and ga.shape == gb.shape
    assert bool((ga[:, n_on:] == 0).all()) and ga[:, :n_on].abs().max().item() > 0
    # checkpoints see the trained grids and their moments
    he = t_c.model.field.hash_ensemble
    st = t_c.state_dict()["optimizers"]["fields"]["state"][1]
    st_f = t_f.state_dict()["optimizers"]["fields"]["state"][1]
    assert int(st["step"]) == int(st_f["step"]) == 6
    a, b = st["exp_avg_sq"], st_f["exp_avg_sq"]
    assert abs(a.abs().sum().item() - b.abs().sum().item()) <= 0.05 * b.abs().sum().item()
    assert torch.equal(he.tables.detach()[:, :, :width], he._compact["master"])


def test_compact_layout_is_handed_over_at_every_doubling(cuda):
    """A schedule that walks the window from 1 to 8.5 in 17 steps: the compact copy goes 1 -> 2 -> 4 -> 8 grids and ends
    when the window passes H / 2 (H = 16); every hand-over writes the trained grids and their moments back and cuts the
    next width from the full layout.  The run follows the full-layout run."""
    t_c, init, l_c, phase_c, _ = _compact_run(True, 19, window_hash=(2, 32))
    t_f, _, l_f, phase_f, _ = _compact_run(False, 19, window_hash=(2, 32))
    want = [1, 1, 1] + [2, 2] + [4] * 4 + [8] * 8 + [0, 0]        # window(step) = 1 + (step - 2) / 2
    assert phase_c == want, phase_c
    assert not any(phase_f)
    assert np.allclose(l_c[:4], l_f[:4], rtol=2e-3), (l_c, l_f)
    assert np.allclose(l_c, l_f, rtol=5e-2), (l_c, l_f)
    a, b = t_c.model.field.hash_ensemble, t_f.model.field.hash_ensemble
    a.wait_tables(), b.wait_tables()
    for h in (0, 1, 3, 7, 8):                                       # every grid that was on has moved, in both runs
        assert (a.tables.detach()[:, :, h] != init[:, :, h]).any() and (b.tables.detach()[:, :, h] != init[:, :, h]).any()
    assert torch.equal(a.tables.detach()[:, :, 10:], init[:, :, 10:])


def test_fused_pass_takes_the_dense_configuration(cuda):
    """BASELINE.json configs[3] (`--disable_occupancy_grid --lambda_dist_loss 0`, train_nersemble.py) on the fused main pass:
    the sampler's sigma_fn answers ones (every marched sample is kept, nothing to reuse), the distortion term is absent
    from the loss dict, and the pass runs UN-CHUNKED although the step has several `max_n_samples_per_batch` chunks --
    against the modular path, which walks the chunks with the reference's operator structure: same loss terms, same
    gradients (different summation order only)."""
    from nersemble_amd.workloads import build_workload
    res = {}
    for fused in (False, True):
        torch.manual_seed(11)
        trainer, data, _ = build_workload("p097_dense", device="cuda:0", small=True, n_rays=384)
        model = trainer.model
        model.fuse_main_pass = fused
        model.config.max_n_samples_per_batch = 1 << 14                   # several chunks per step on the modular path
        model.field.max_n_samples_per_batch = 1 << 14
        if model.deformation_field is not None:
            model.deformation_field.max_n_samples_per_batch = 1 << 14
        calls = []
        orig = model.fused_train_forward

        def counted(*a, _orig=orig, _calls=calls, **k):
            r = _orig(*a, **k)
            _calls.append(r is not None)
            return r

        model.fused_train_forward = counted
        torch.manual_seed(70)
        loss, loss_dict, metrics = trainer.train_iteration(0, *data.next_train(0))
        assert calls == [fused]
        assert int(metrics["num_samples_per_batch"]) > 3 * (1 << 14)     # really more than one chunk
        assert "dist_loss" not in loss_dict                              # lambda_dist_loss = 0: the term is absent
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        res[fused] = (loss.item(), {k: v.item() for k, v in loss_dict.items()}, grads,
                      int(metrics["num_samples_per_batch"]))
        trainer.flush_scheduler_step()
    (l_m, t_m, g_m, n_m), (l_f, t_f, g_f, n_f) = res[False], res[True]
    assert n_m == n_f and set(t_m) == set(t_f)
    assert np.isclose(l_m, l_f, rtol=1e-5), (l_m, l_f)
    for k in t_m:
        assert np.isclose(t_m[k], t_f[k], rtol=1e-4, atol=1e-9), (k, t_m[k], t_f[k])
    assert set(g_m) == set(g_f)
    for name in g_m:
        sc = g_m[name].abs().max().item()
        assert (g_m[name] - g_f[name]).abs().max().item() <= 2e-3 * sc + 1e-12, name
