"""nerfstudio checkpoint wire format -- SURVEY.md 8(f) rank 3.

The reference stores / restores models through nerfstudio's trainer (``step-{step:09d}.ckpt`` = ``torch.save({"step",
"pipeline": pipeline.state_dict(), "optimizers", "scalers"})``; restored by ``nersemble_eval_setup`` ->
``eval_load_checkpoint``, util/setup.py:14-66).  ``pipeline.state_dict()`` prefixes every model tensor with ``_model.``
and also carries datamanager entries.  The model mirrors in this package already speak the reference's tensor layout
(``hash_encodings.{c}.params`` in tcnn order, ``mlp_base.params`` / ``mlp_head.params`` flat, the deformation
``nn.Linear`` keys), so loading a checkpoint is key plumbing only.
"""
from typing import Dict, Tuple, Union

import torch

MODEL_PREFIX = "_model."


def model_state_from_pipeline(pipeline_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The ``_model.*`` entries of a nerfstudio pipeline state dict, prefix removed (everything else -- datamanager
    buffers, a DDP ``module.`` wrapper prefix -- is dropped / unwrapped)."""
    out = {}
    for key, value in pipeline_state.items():
        if key.startswith("module."):
            key = key[len("module."):]
        if key.startswith(MODEL_PREFIX):
            out[key[len(MODEL_PREFIX):]] = value
    return out


def load_nerfstudio_checkpoint(checkpoint: Union[str, Dict], model: torch.nn.Module, strict: bool = True
                               ) -> Tuple[int, list, list]:
    """Load ``checkpoint`` (a path or the already ``torch.load``-ed dict) into ``model``.
    Returns (step, missing_keys, unexpected_keys).  ``strict``: raise if a model parameter is missing from the file."""
    if isinstance(checkpoint, str):
        checkpoint = torch.load(checkpoint, map_location="cpu")
    if "pipeline" not in checkpoint:
        raise KeyError("not a nerfstudio checkpoint: no 'pipeline' entry (keys: %s)" % sorted(checkpoint)[:8])
    state = model_state_from_pipeline(checkpoint["pipeline"])
    if not state:
        raise KeyError("the checkpoint's pipeline state holds no '_model.*' tensors")
    result = model.load_state_dict(state, strict=False)
    # run-time buffers that are rebuilt on construction are allowed to be absent from either side, and so are the
    # EMPTY parameters of the reference's module tree (tests/golden/state_manifest.json) that files written by this
    # package before round 4 did not list
    optional = ("grid_coords", "grid_indices", "tables_f16", "device_indicator_param", "direction_encoding.params",
                "position_encoding.params")
    missing = [k for k in result.missing_keys if not k.endswith(optional)]
    if strict and missing:
        raise KeyError(f"checkpoint lacks model tensors: {missing[:10]}{' ...' if len(missing) > 10 else ''}")
    return int(checkpoint.get("step", -1)), missing, list(result.unexpected_keys)


def nerfstudio_checkpoint_from_model(model: torch.nn.Module, step: int, trainer=None) -> Dict:
    """The inverse: a dict in nerfstudio's checkpoint format.  With ``trainer`` (``engine.trainer.NeRSembleTrainer``) the
    ``optimizers`` / ``scalers`` entries carry the training state in the reference's shape -- ``optimizers[group]`` is a
    ``torch.optim.Adam.state_dict()`` per parameter group of ``get_param_groups`` (``fields``: the C tcnn hash encodings'
    moments in tcnn layout, then ``mlp_base`` / ``mlp_head``; ``embeddings``; ``deformation_field``), with torch's
    per-parameter ``step`` -- so that a run can resume here or there; data-parallel runs gather the sharded table state
    first.  ``schedulers`` (StepLR counters per group) is an extra entry nerfstudio 0.3.1's loader does not read."""
    training = {"optimizers": {}, "scalers": {}}
    if trainer is not None:
        trainer.consolidate()
        full = trainer.state_dict()
        training = {"optimizers": full["optimizers"], "scalers": full["scalers"], "schedulers": full["schedulers"]}
    # a SNAPSHOT: ``state_dict()`` hands out views of the live parameters / optimizer moments, which the next training
    # step would rewrite under the checkpoint
    return _snapshot({"step": int(step), "pipeline": {MODEL_PREFIX + k: v for k, v in model.state_dict().items()},
                      **training})


def _snapshot(obj):
    if torch.is_tensor(obj):
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: _snapshot(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_snapshot(v) for v in obj)
    return obj


def resume_trainer_from_checkpoint(checkpoint: Dict, trainer) -> int:
    """Model weights + training state from a checkpoint written by ``nerfstudio_checkpoint_from_model(..., trainer)``."""
    step, _, _ = load_nerfstudio_checkpoint(checkpoint, trainer.model, strict=True)
    if checkpoint.get("optimizers"):
        trainer.load_state_dict({"optimizers": checkpoint["optimizers"], "scalers": checkpoint.get("scalers", {}),
                                 "schedulers": checkpoint.get("schedulers", {})})
    return step
