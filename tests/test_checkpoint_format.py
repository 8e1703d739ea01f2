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


# ---- the reference's own module tree (tests/golden/state_manifest.json, generated by make_golden.py::gen_state_manifest
# from NeRSembleNGPModel.populate_modules / get_param_groups of the reference under module-tree stubs) --------------------
def _manifest(golden_dir, H):
    import json
    with open(f"{golden_dir}/state_manifest.json") as f:
        return json.load(f)["configs"][f"H{H}"]


def _full_size_model(entry):
    from nersemble_amd.models.nersemble_instant_ngp import NeRSembleNGPModel
    from nersemble_amd.rays import SceneBox
    from nersemble_amd.workloads import SCENE_BOXES, build_model_config
    c = entry["config"]
    cfg = build_model_config(dict(H=c["n_hash_encodings"], T=c["n_timesteps"], disable_occ=False, lambda_dist=1e-4,
                                  win=(40000, 80000)))
    cfg.use_view_frustum_culling = False
    assert cfg.hash_ensemble_config.hash_encoding_config.log2_hashmap_size == c["log2_hashmap_size"]
    return NeRSembleNGPModel(cfg, SceneBox(torch.tensor(SCENE_BOXES[30], dtype=torch.float32)),
                             num_train_data=c["num_train_data"])


def _expanded_group_names(model, group):
    """Names of a parameter group's members in order, the native ``tables`` standing for its C tcnn encodings."""
    name_of = {id(p): n for n, p in model.named_parameters()}
    out = []
    for p in group:
        n = name_of[id(p)]
        if n.endswith("hash_ensemble.tables"):
            out += [n[:-len("tables")] + f"hash_encodings.{c}.params" for c in range(model.field.hash_ensemble.n_tcnn_encodings)]
        else:
            out.append(n)
    return out


import pytest


@pytest.mark.parametrize("H", [1, 16])          # (H = 32 -- 403 M parameters -- runs with the GPU tests)
def test_state_dict_and_param_groups_equal_the_reference_manifest(H, golden_dir):
    """f3 pinned to the reference itself: every key / shape / dtype of ``NeRSembleNGPModel.state_dict()`` at the
    reference's table size, and the membership AND ORDER of every optimizer group, equal what the reference's own
    ``populate_modules`` / ``get_param_groups`` produce (nersemble_instant_ngp.py:81-179, :502-514;
    nersemble_nerfacto_field.py:99-172; hash_ensemble.py:84-91; deformation_field.py:50-69,129-131)."""
    entry = _manifest(golden_dir, H)
    model = _full_size_model(entry)
    got = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in model.state_dict().items()}
    want = {e["key"]: (e["shape"], e["dtype"]) for e in entry["state_dict"]}
    assert set(got) == set(want), (sorted(set(got) - set(want)), sorted(set(want) - set(got)))
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])
    groups = model.get_param_groups()
    assert list(groups) == list(entry["param_groups"])
    for g, names in entry["param_groups"].items():
        assert _expanded_group_names(model, groups[g]) == names, g
    for n, p in model.named_parameters():
        if n in entry["requires_grad"]:
            assert p.requires_grad == entry["requires_grad"][n], n


def _small_trainer(seed=0):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(seed)
    trainer, _, _ = build_workload("p030_h16", device="cpu", small=True, n_rays=64)
    return trainer


def _fake_training_state(trainer, seed=0):
    """Moments and step counts as after some steps (the native table step needs the GPU; the wire format does not)."""
    g = torch.Generator().manual_seed(seed)
    topt = trainer.optimizers[trainer.group_of_tables()]
    st = topt._state()
    st["step"] = 7
    st["exp_avg"].copy_(torch.randn(st["exp_avg"].shape, generator=g) * 1e-3)
    st["exp_avg_sq"].copy_(torch.rand(st["exp_avg_sq"].shape, generator=g) * 1e-6)
    Hn = trainer.model.field.hash_ensemble.n_hash_encodings
    st["exp_avg"][:, :, Hn:] = 0                # (padding grids hold no state)
    st["exp_avg_sq"][:, :, Hn:] = 0
    for key, opt in trainer.optimizers.items():
        if key.endswith("/tables"):
            continue
        for p in (p for pg in opt.param_groups for p in pg["params"]):
            if key == "embeddings" and p is trainer.model.time_embedding.weight:
                continue                         # (no gradient before the window opens: no Adam state, as in torch)
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
        opt.zero_grad(set_to_none=True)


def _reference_side_adam(trainer, group):
    """What nerfstudio builds on the reference side: ``torch.optim.Adam(list(group parameters))`` with one flat tensor per
    tcnn encoding (shapes of this model's twin in the reference's layout)."""
    he = trainer.model.field.hash_ensemble
    per_enc = he.geom.total_entries * (8 if 2 * he.n_hash_encodings >= 8 else 2 * he.n_hash_encodings)
    params = []
    for kind, x in trainer.group_layout[group]:
        params.append(torch.nn.Parameter(torch.zeros(per_enc) if kind == "table" else torch.zeros(tuple(x.shape))))
    return torch.optim.Adam(params, lr=1.0, eps=1e-15)


def test_optimizer_state_travels_both_ways_in_the_reference_numbering(golden_dir):
    """``optimizers[group]`` of a checkpoint is ONE ``torch.optim.Adam.state_dict()`` per group of ``get_param_groups`` in the
    reference's numbering (nerfstudio's ``Optimizers.load_optimizers`` hands it to ``Adam.load_state_dict``): ``fields`` has
    C + 4 members -- ``direction_encoding.params`` (empty), the C hash encodings, ``position_encoding.params`` (empty),
    ``mlp_base``, ``mlp_head`` -- ``deformation_field`` starts with the frozen ``aabb``; members that never get a
    gradient are listed in ``param_groups`` and have NO state entry.  (a) what this trainer writes loads into the
    reference-side Adam objects; (b) what those write -- a genuine reference layout -- loads into a second trainer, and
    moments / steps / learning rates arrive where they belong."""
    a = _small_trainer(1)
    _fake_training_state(a, 3)
    entry = _manifest(golden_dir, 16)
    sd = a.state_dict()["optimizers"]
    assert sorted(sd) == sorted(entry["param_groups"])
    C = a.model.field.hash_ensemble.n_tcnn_encodings
    ref_files = {}
    for group, names in entry["param_groups"].items():
        assert sd[group]["param_groups"][0]["params"] == list(range(len(names))), group
        no_grad = [i for i, (kind, _) in enumerate(a.group_layout[group]) if kind == "none"]
        assert no_grad == [i for i, n in enumerate(names)
                           if n.endswith(("direction_encoding.params", "position_encoding.params", "deformation_field.aabb"))]
        assert not (set(no_grad) & set(sd[group]["state"]))
        ref = _reference_side_adam(a, group)
        ref.load_state_dict(sd[group])                                  # (a) the reference side accepts it
        ref_files[group] = ref.state_dict()                            # what a reference run would write back
        assert set(ref_files[group]["state"]) == set(sd[group]["state"])
    f = ref_files["fields"]
    assert len(f["param_groups"][0]["params"]) == C + 4 and 0 not in f["state"] and C + 1 not in f["state"]
    assert [int(f["state"][i]["step"]) for i in range(1, C + 1)] == [7] * C
    assert 0 not in ref_files["deformation_field"]["state"] and len(ref_files["deformation_field"]["param_groups"][0]["params"]) == 17
    assert set(ref_files["embeddings"]["state"]) == {1}                 # time_embedding.weight has not started
    b = _small_trainer(2)
    b.load_state_dict({"optimizers": ref_files, "schedulers": {}, "scalers": {}})         # (b)
    ta, tb = a.optimizers[a.group_of_tables()], b.optimizers[b.group_of_tables()]
    assert tb._state()["step"] == 7
    assert torch.equal(ta._state()["exp_avg"], tb._state()["exp_avg"])
    assert torch.equal(ta._state()["exp_avg_sq"], tb._state()["exp_avg_sq"])
    for key in ("fields", "embeddings", "deformation_field"):
        pa = [p for pg in a.optimizers[key].param_groups for p in pg["params"]]
        pb = [p for pg in b.optimizers[key].param_groups for p in pg["params"]]
        assert len(pa) == len(pb)
        for x, y in zip(pa, pb):
            sa, sb_ = a.optimizers[key].state.get(x, {}), b.optimizers[key].state.get(y, {})
            assert set(sa) == set(sb_)
            for k in sa:
                assert torch.equal(torch.as_tensor(sa[k]), torch.as_tensor(sb_[k])), (key, k)
    # a file that holds moments where this model has a member without a gradient is another layout: refused
    bad = {k: {"state": dict(v["state"]), "param_groups": v["param_groups"]} for k, v in ref_files.items()}
    bad["fields"]["state"][0] = {"step": torch.tensor(1.0), "exp_avg": torch.zeros(5), "exp_avg_sq": torch.zeros(5)}
    with pytest.raises(KeyError, match="never has a gradient"):
        _small_trainer(3).load_state_dict({"optimizers": bad})


def test_checkpoints_of_earlier_rounds_still_load():
    """Round 3 numbered the ``fields`` group as the C encodings + the two MLPs (no empty encodings) and the deformation
    group without its ``aabb``; round 2 kept the tables under ``fields/tables: {native_table_adam: ...}``."""
    a = _small_trainer(1)
    _fake_training_state(a, 5)
    sd = a.state_dict()["optimizers"]
    C = a.model.field.hash_ensemble.n_tcnn_encodings

    def drop(group_sd, positions):
        keep = [i for i in group_sd["param_groups"][0]["params"] if i not in positions]
        renum = {old: new for new, old in enumerate(keep)}
        return {"state": {renum[i]: st for i, st in group_sd["state"].items() if i in renum},
                "param_groups": [dict(group_sd["param_groups"][0], params=list(range(len(keep))))]}

    r3 = {"fields": drop(sd["fields"], {0, C + 1}), "embeddings": sd["embeddings"],
          "deformation_field": drop(sd["deformation_field"], {0})}
    b = _small_trainer(2)
    b.load_state_dict({"optimizers": r3})
    ta, tb = a.optimizers[a.group_of_tables()], b.optimizers[b.group_of_tables()]
    assert tb._state()["step"] == 7 and torch.equal(ta._state()["exp_avg"], tb._state()["exp_avg"])
    x = [p for pg in a.optimizers["deformation_field"].param_groups for p in pg["params"]][3]
    y = [p for pg in b.optimizers["deformation_field"].param_groups for p in pg["params"]][3]
    assert torch.equal(a.optimizers["deformation_field"].state[x]["exp_avg"], b.optimizers["deformation_field"].state[y]["exp_avg"])
    # round 2
    r2 = dict(r3)
    small_only = drop(sd["fields"], set(range(C + 2)))
    r2["fields"] = small_only
    r2["fields/tables"] = {"native_table_adam": ta.table_state()}
    c = _small_trainer(3)
    c.load_state_dict({"optimizers": r2})
    tc = c.optimizers[c.group_of_tables()]
    assert tc._state()["step"] == 7 and torch.equal(ta._state()["exp_avg_sq"], tc._state()["exp_avg_sq"])
    with pytest.raises(KeyError, match="reference's numbering"):
        _small_trainer(4).load_state_dict({"optimizers": dict(r3, fields=drop(sd["fields"], {0, 1, 2}))})


def test_checkpoint_keys_are_the_manifests(golden_dir):
    """``nerfstudio_checkpoint_from_model`` writes ``_model.`` + exactly the reference's key set (small tables here: the
    names do not depend on the table size) and a resumed trainer continues from it."""
    from nersemble_amd.util.checkpoint import resume_trainer_from_checkpoint
    a = _small_trainer(1)
    _fake_training_state(a, 9)
    ckpt = nerfstudio_checkpoint_from_model(a.model, 77, trainer=a)
    want = {"_model." + e["key"] for e in _manifest(golden_dir, 16)["state_dict"]}
    assert set(ckpt["pipeline"]) == want
    b = _small_trainer(2)
    assert resume_trainer_from_checkpoint(ckpt, b) == 77
    for (k, v), (_, w) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(v, w), k
    assert b.optimizers[b.group_of_tables()]._state()["step"] == 7


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
