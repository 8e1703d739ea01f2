"""nerfstudio checkpoint wire format (util/checkpoint.py): round trip through the reference's key layout on CPU."""
import torch

from nersemble_amd.util.checkpoint import (load_nerfstudio_checkpoint, model_state_from_pipeline,
                                           nerfstudio_checkpoint_from_model)


def _model(seed):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(seed)
    trainer, _, _ = build_workload("p030_h16", device="cpu", small=True, n_rays=64)
    return trainer.model


def test_roundtrip_through_nerfstudio_checkpoint(tmp_path):
    a, b = _model(1), _model(2)
    with torch.no_grad():                       # the tables are seeded like tcnn (1337), not by torch's generator
        a.field.hash_ensemble.tables.add_(torch.randn_like(a.field.hash_ensemble.tables) * 1e-3)
    ckpt = nerfstudio_checkpoint_from_model(a, step=1234)
    keys = set(ckpt["pipeline"])
    # the reference's names: one tcnn parameter vector per 8-feature encoding, flat fused-MLP parameters, nn.Linear keys
    for k in ("_model.field.hash_ensemble.hash_encodings.0.params", "_model.field.hash_ensemble.hash_encodings.3.params",
              "_model.field.mlp_base.params", "_model.field.mlp_head.params", "_model.time_embedding.weight",
              "_model.deformation_field.se3_field.mlp_stem.layers.4.weight", "_model.occupancy_grid.occs"):
        assert k in keys, k
    assert "_model.field.hash_ensemble.tables" not in keys
    # what a real checkpoint additionally carries
    ckpt["pipeline"]["datamanager.train_camera_optimizer.pose_adjustment"] = torch.zeros(3, 6)
    ckpt["pipeline"] = {("module." + k if i % 2 else k): v for i, (k, v) in enumerate(ckpt["pipeline"].items())}
    path = str(tmp_path / "step-000001234.ckpt")
    torch.save(ckpt, path)
    assert not torch.equal(a.field.hash_ensemble.tables, b.field.hash_ensemble.tables)
    step, missing, unexpected = load_nerfstudio_checkpoint(path, b)
    assert step == 1234 and missing == [] and unexpected == []
    sa, sb = a.state_dict(), b.state_dict()
    assert set(sa) == set(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert torch.equal(a.field.hash_ensemble.tables, b.field.hash_ensemble.tables)     # native layout restored


def test_rejects_foreign_files():
    import pytest
    with pytest.raises(KeyError):
        load_nerfstudio_checkpoint({"model": {}}, torch.nn.Linear(1, 1))
    with pytest.raises(KeyError):
        load_nerfstudio_checkpoint({"pipeline": {"datamanager.x": torch.zeros(1)}}, torch.nn.Linear(1, 1))
    assert model_state_from_pipeline({"_model.a": 1, "module._model.b": 2, "other": 3}) == {"a": 1, "b": 2}


def test_run_folder_config_yml_and_checkpoint_lookup(tmp_path, golden_dir):
    """SURVEY 8 f3 -- a run as the reference stores it: ``config.yml`` (yaml.dump of the trainer config: python-object tags
    of nerfstudio / nersemble classes that are not installed here, a pickled torch scene box, PosixPaths) next to
    ``checkpoints/step-*.ckpt``.  ``nersemble_eval_setup`` (util/setup.py:14-71 in the reference) reads the config without
    those packages, builds the model from ``pipeline.model``, picks the newest checkpoint and loads it.
    The fixture is produced by tests/golden/make_golden.py::gen_config_yml from the reference's own field names."""
    import torch
    from nersemble_amd.models.nersemble_instant_ngp import NeRSembleNGPModel
    from nersemble_amd.rays import SceneBox
    from nersemble_amd.util.checkpoint import nerfstudio_checkpoint_from_model
    from nersemble_amd.util.setup import (find_checkpoint, model_config_from_nerfstudio, nersemble_eval_setup,
                                          try_load_config)
    cfg_path = f"{golden_dir}/config.yml"
    config = try_load_config(cfg_path)
    assert type(config).__name__ == "NeRSembleTrainerConfig" and config.run_name == "NERS-9999"
    assert config.pipeline.datamanager.dataparser.participant_id == 30
    assert torch.equal(config.pipeline.datamanager.dataparser.scene_box,
                       torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]]))
    assert str(config.relative_model_dir) == "checkpoints" and config.optimizers["deformation_field"]["optimizer"].lr == 1e-3
    model_cfg, unknown = model_config_from_nerfstudio(config.pipeline.model)
    # every field of the reference's model config has a place here; what is reported belongs to nerfstudio's collider /
    # loss-coefficient plumbing, which the path does not use
    assert sorted(unknown) == ["model.collider_params", "model.enable_collider", "model.loss_coefficients"]
    assert model_cfg.n_timesteps == 100 and model_cfg.latent_dim_time == 16 and model_cfg.max_n_samples_per_batch == 2 ** 20
    assert model_cfg.hash_ensemble_config.n_hash_encodings == 16
    assert model_cfg.hash_ensemble_config.hash_encoding_config.log2_hashmap_size == 19
    assert model_cfg.deformation_field_config.warp_code_dim == 128 and model_cfg.deformation_field_config.skip_connections == (4,)
    assert model_cfg.window_hash_encodings_end == 80000 and model_cfg.lambda_dist_loss == 1e-4

    # a small twin of that run (2^12-entry tables keep the CPU test quick): write two checkpoints, load the newest.
    # The loaded config dumps back to the tags it was read from (what the reference's save_config does with the real classes)
    config.pipeline.model.hash_ensemble_config.hash_encoding_config.log2_hashmap_size = 12
    import yaml
    run = tmp_path / "NERS-9999"
    (run / "checkpoints").mkdir(parents=True)
    text = yaml.dump(config)
    assert "!!python/object:nersemble.nerfstudio.models.nersemble_instant_ngp.NeRSembleNGPModelConfig" in text
    (run / "config.yml").write_text(text)
    small_cfg, _ = model_config_from_nerfstudio(config.pipeline.model)
    torch.manual_seed(0)
    src = NeRSembleNGPModel(small_cfg, SceneBox(config.pipeline.datamanager.dataparser.scene_box), num_train_data=1)
    torch.save(nerfstudio_checkpoint_from_model(src, 50000), run / "checkpoints" / "step-000050000.ckpt")
    with torch.no_grad():
        src.time_embedding.weight.add_(1.0)
        src.field.hash_ensemble.tables.mul_(3.0)
    torch.save(nerfstudio_checkpoint_from_model(src, 300000), run / "checkpoints" / "step-000300000.ckpt")
    assert find_checkpoint(run / "checkpoints")[1] == 300000 and find_checkpoint(run / "checkpoints", 50000)[1] == 50000
    loaded_cfg, model, path, step = nersemble_eval_setup(run / "config.yml", run / "checkpoints", eval_num_rays_per_chunk=2048,
                                                          device="cpu")
    assert step == 300000 and path.name == "step-000300000.ckpt" and not model.training
    assert model.config.eval_num_rays_per_chunk == 2048
    want = src.state_dict()
    got = model.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    older = nersemble_eval_setup(run / "config.yml", run / "checkpoints", checkpoint=50000, device="cpu")[1]
    assert not torch.equal(older.time_embedding.weight, model.time_embedding.weight)
    import pytest
    with pytest.raises(FileNotFoundError):
        find_checkpoint(run / "checkpoints", 123)


def test_fields_optimizer_state_has_the_reference_shape():
    """``optimizers['fields']`` of a checkpoint is ONE torch.optim.Adam state dict over ``field.parameters()`` in the
    reference's registration order (nersemble_nerfacto_field.py:99-172: the C tcnn hash encodings, mlp_base, mlp_head;
    nerfstudio's ``Optimizers.load_optimizers`` hands it to ``Adam.load_state_dict``).  The natively stepped tables are
    merged in / split off losslessly; entries of parameter-free tcnn encodings (empty ``params``) that a reference
    checkpoint may list are ignored; a mismatch raises a clear error."""
    import pytest
    import torch
    from nersemble_amd.engine.trainer import _merge_table_state, _split_table_state
    C, n_small = 3, 2
    g = torch.Generator().manual_seed(0)
    table = {"step": 7, "lr": 4e-3, "exp_avg": [torch.randn(40, generator=g) for _ in range(C)],
             "exp_avg_sq": [torch.rand(40, generator=g) for _ in range(C)]}
    small_params = [torch.nn.Parameter(torch.randn(5, generator=g)), torch.nn.Parameter(torch.randn(6, generator=g))]
    opt = torch.optim.Adam(small_params, lr=5e-3, eps=1e-15)
    for p in small_params:
        p.grad = torch.randn(p.shape, generator=g)
    opt.step()
    merged = _merge_table_state(table, opt.state_dict())
    assert merged["param_groups"][0]["params"] == list(range(C + n_small)) and merged["param_groups"][0]["lr"] == 4e-3
    # the reference side accepts it
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(40)) for _ in range(C)] +
                           [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(6))], lr=5e-3, eps=1e-15)
    ref.load_state_dict(merged)
    assert [int(st["step"]) for st in ref.state.values()] == [7, 7, 7, 1, 1]
    # and what the reference writes comes back apart
    tab, small = _split_table_state(ref.state_dict(), C, n_small, "fields")
    assert tab["step"] == 7 and all(torch.equal(a, b) for a, b in zip(tab["exp_avg"], table["exp_avg"]))
    assert small["param_groups"][0]["params"] == [0, 1] and torch.equal(small["state"][1]["exp_avg"],
                                                                         opt.state_dict()["state"][1]["exp_avg"])
    # a reference file with the empty `params` of tcnn's Identity / Frequency encodings in the list
    sd = ref.state_dict()
    e = {"step": torch.tensor(7.0), "exp_avg": torch.zeros(0), "exp_avg_sq": torch.zeros(0)}
    shifted = {0: e}
    for i, st in sd["state"].items():
        shifted[i + 1 + (i >= C)] = st
    shifted[C + 1] = dict(e)
    sd2 = {"state": shifted, "param_groups": [dict(sd["param_groups"][0], params=list(range(C + n_small + 2)))]}
    tab2, small2 = _split_table_state(sd2, C, n_small, "fields")
    assert all(torch.equal(a, b) for a, b in zip(tab2["exp_avg_sq"], table["exp_avg_sq"])) and set(small2["state"]) == {0, 1}
    # before the first step there are no moments
    tab0, _ = _split_table_state(_merge_table_state(dict(table, step=0), opt.state_dict()), C, n_small, "fields")
    assert tab0["step"] == 0 and tab0["exp_avg"] is None
    with pytest.raises(KeyError, match="hash encodings"):
        _split_table_state(merged, C + 1, n_small, "fields")


def test_config_loader_constructs_nothing_a_file_names(tmp_path):
    """A ``config.yml`` of unknown origin: names outside the allow-list (paths, OrderedDict, the pieces of a dumped tensor)
    become inert placeholders -- nothing is imported, nothing is called."""
    from nersemble_amd.util.setup import ConfigNode, try_load_config
    marker = tmp_path / "ran"
    text = (f"a: !!python/object/apply:os.system ['touch {marker}']\n"
            f"b: !!python/object/apply:subprocess.check_output [['touch', '{marker}']]\n"
            "c: !!python/name:os.system\n"
            "d: !!python/module:shutil\n"
            "e: !!python/object/new:os.system ['true']\n"
            "p: !!python/object/apply:pathlib.PosixPath [x, y]\n"
            "t: !!python/tuple [1, 2]\n")
    path = tmp_path / "config.yml"
    path.write_text(text)
    cfg = try_load_config(path)
    assert not marker.exists()
    assert isinstance(cfg["a"], ConfigNode) and cfg["a"].args == [f"touch {marker}"]
    assert isinstance(cfg["b"], ConfigNode) and isinstance(cfg["e"], ConfigNode)
    assert isinstance(cfg["c"], type) and issubclass(cfg["c"], ConfigNode) and cfg["c"]._class == "os.system"
    assert type(cfg["d"]).__name__ == "module" and not hasattr(cfg["d"], "rmtree")
    assert str(cfg["p"]) == "x/y" and cfg["t"] == (1, 2)
