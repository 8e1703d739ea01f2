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
