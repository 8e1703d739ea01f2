"""Loading a trained run from disk -- mirror of the reference's ``util/setup.py:14-88`` (``nersemble_eval_setup``,
``try_load_config``) and of the checkpoint lookup it delegates to nerfstudio (``eval_load_checkpoint``: the newest
``step-*.ckpt`` of the run's checkpoint folder unless a step is named; ``model_manager/base.py:24-46`` names the folder
layout: ``checkpoints/step-$.ckpt`` next to ``config.yml``).

A run's ``config.yml`` is ``yaml.dump`` of nerfstudio / nersemble config dataclasses: python-object tags for classes of
packages that are not installed here (and need not be: only their FIELDS are wanted).  ``try_load_config`` therefore loads
the file with a loader that builds a plain attribute container for every class it cannot import and lets everything else
(paths, tuples, torch tensors such as the scene box) construct as usual.  From the ``pipeline.model`` node the model config
of this package is filled field by field; fields this package does not know are reported, not dropped silently.

No datamanager is set up (the dataset is gated and absent): this is the reference's ``test_mode="inference"``.
"""
import dataclasses
import pathlib
import re
import types
from pathlib import Path
from typing import List, Optional, Tuple

import torch
import yaml

from ..field_components.deformation_field import SE3DeformationFieldConfig
from ..field_components.hash_ensemble import HashEnsembleConfig, TCNNHashEncodingConfig
from ..models.nersemble_instant_ngp import NeRSembleNGPModel, NeRSembleNGPModelConfig
from ..rays import SceneBox
from .checkpoint import load_nerfstudio_checkpoint


class ConfigNode:
    """Stand-in for an instance of a config class whose package is not installed (a plain attribute container, so that it
    pickles / yaml-dumps like the dataclass instance it stands for); ``_class`` names the original."""

    def __init__(self, **fields):
        self.__dict__.update(fields)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={v!r}' for k, v in vars(self).items())})"


def _load_storage_from_bytes(b):
    """What ``torch.storage._load_from_bytes`` does for a tensor dumped into the yaml (the scene box), restricted to
    tensor data: the original unpickles arbitrary objects."""
    import io
    return torch.load(io.BytesIO(b), weights_only=True)


# The only python names a run's ``config.yml`` needs CONSTRUCTED: paths, ordered dicts, and the pieces of a dumped
# torch tensor.  Everything else -- config classes of packages that are not installed, ``_target`` class references,
# optimizer classes -- becomes an inert ``ConfigNode`` placeholder: no import, no call.
_ALLOWED_NAMES = {
    "pathlib.PosixPath": pathlib.PosixPath, "pathlib.WindowsPath": pathlib.WindowsPath, "pathlib.Path": pathlib.Path,
    "pathlib.PurePosixPath": pathlib.PurePosixPath, "pathlib.PureWindowsPath": pathlib.PureWindowsPath,
    "collections.OrderedDict": __import__("collections").OrderedDict,
    "torch._utils._rebuild_tensor_v2": torch._utils._rebuild_tensor_v2,
    "torch.storage._load_from_bytes": _load_storage_from_bytes,
    "torch.float32": torch.float32, "torch.float64": torch.float64, "torch.float16": torch.float16,
    "torch.int64": torch.int64, "torch.int32": torch.int32, "torch.uint8": torch.uint8, "torch.bool": torch.bool,
}


class _LenientLoader(yaml.Loader):
    """A loader for ``yaml.dump``-ed nerfstudio / nersemble configs (the reference reads them with ``yaml.Loader``,
    util/setup.py:76) that neither needs the dumped classes' packages nor executes anything a file names: python names
    on ``_ALLOWED_NAMES`` resolve to what they are, every other name / class becomes a placeholder class whose
    instances are ``ConfigNode``s carrying the dumped fields.  Opening a run folder of unknown origin therefore cannot
    import modules or call functions of the file's choosing."""

    def find_python_name(self, name, mark, unsafe=False):
        hit = _ALLOWED_NAMES.get(name)
        if hit is not None:
            if hit is pathlib.PosixPath or hit is pathlib.WindowsPath:
                return getattr(pathlib, hit.__name__)      # (try_load_config's PosixPath <-> WindowsPath retry swaps these)
            return hit
        # named and "located" like the original, so that yaml.dump of a loaded config writes the original tags again
        # (the reference's save_config, model_manager/base.py:44-46)
        module, _, cls_name = name.rpartition(".")
        return type(cls_name or name, (ConfigNode,), {"_class": name, "__module__": module})

    def find_python_module(self, name, mark, unsafe=False):
        return types.ModuleType(name)

    def make_python_instance(self, suffix, node, args=None, kwds=None, newobj=False, unsafe=False):
        cls = self.find_python_name(suffix, node.start_mark)
        if isinstance(cls, type) and issubclass(cls, ConfigNode):
            # `object:` / `object/apply:` / `object/new:` of a class that is not constructed here: keep what the node says
            fields = {str(k): v for k, v in (kwds or {}).items()}
            if args:
                fields["args"] = list(args)
            return cls(**fields)
        return super().make_python_instance(suffix, node, args, kwds, newobj)


def try_load_config(config_path) -> ConfigNode:
    """util/setup.py:73-88 (incl. its PosixPath <-> WindowsPath retry)."""
    text = pathlib.Path(config_path).read_text()
    posix_path = pathlib.PosixPath
    try:
        return yaml.load(text, Loader=_LenientLoader)
    except NotImplementedError:
        pathlib.PosixPath = pathlib.WindowsPath
        return yaml.load(text, Loader=_LenientLoader)
    finally:
        pathlib.PosixPath = posix_path


def _fill(dc_type, node, unknown: List[str], where: str):
    """A dataclass of this package from a config node: fields the dataclass has are copied, others reported."""
    if node is None:
        return None
    values = dict(vars(node)) if not isinstance(node, dict) else dict(node)
    names = {f.name for f in dataclasses.fields(dc_type)}
    kwargs = {}
    for key, value in values.items():
        if key.startswith("_"):
            continue                                      # _target etc.
        if key in names:
            kwargs[key] = value
        else:
            unknown.append(f"{where}.{key}")
    return dc_type(**kwargs)


def model_config_from_nerfstudio(model_node) -> Tuple[NeRSembleNGPModelConfig, List[str]]:
    """``config.pipeline.model`` (a dumped ``NeRSembleNGPModelConfig``, nersemble_instant_ngp.py:40-77 on nerfstudio's
    ``InstantNGPModelConfig``) -> this package's model config + the list of fields it had no place for."""
    unknown: List[str] = []
    values = dict(vars(model_node))
    he_node, df_node = values.pop("hash_ensemble_config", None), values.pop("deformation_field_config", None)
    cfg = _fill(NeRSembleNGPModelConfig, types.SimpleNamespace(**values), unknown, "model")
    if he_node is not None:
        he_values = dict(vars(he_node))
        enc_node = he_values.pop("hash_encoding_config", None)
        he = _fill(HashEnsembleConfig, types.SimpleNamespace(**he_values), unknown, "model.hash_ensemble_config")
        if enc_node is not None:
            he.hash_encoding_config = _fill(TCNNHashEncodingConfig, enc_node, unknown,
                                            "model.hash_ensemble_config.hash_encoding_config")
        cfg.hash_ensemble_config = he
    if df_node is not None:
        df_values = {k: v for k, v in vars(df_node).items()}
        skip = df_values.get("skip_connections")
        if skip is not None:
            df_values["skip_connections"] = tuple(skip)
        cfg.deformation_field_config = _fill(SE3DeformationFieldConfig, types.SimpleNamespace(**df_values), unknown,
                                             "model.deformation_field_config")
    return cfg, unknown


def find_checkpoint(checkpoint_folder, step: Optional[int] = None) -> Tuple[Path, int]:
    """nerfstudio ``eval_load_checkpoint``: ``step-{step:09d}.ckpt`` of the folder, the newest one when no step is named."""
    folder = Path(checkpoint_folder)
    if not folder.is_dir():
        raise FileNotFoundError(f"checkpoint folder {folder} does not exist")
    if step is None:
        steps = sorted(int(m.group(1)) for m in (re.fullmatch(r"step-(\d+)\.ckpt", p.name) for p in folder.iterdir()) if m)
        if not steps:
            raise FileNotFoundError(f"no step-*.ckpt in {folder}")
        step = steps[-1]
    path = folder / f"step-{int(step):09d}.ckpt"
    if not path.exists():
        raise FileNotFoundError(f"checkpoint {path} does not exist")
    return path, int(step)


def nersemble_eval_setup(config_path, checkpoint_folder, eval_num_rays_per_chunk: Optional[int] = None,
                         checkpoint: Optional[int] = None, scene_box: Optional[torch.Tensor] = None,
                         device=None) -> Tuple[ConfigNode, NeRSembleNGPModel, Path, int]:
    """util/setup.py:14-71 for ``test_mode="inference"``: config -> model in evaluation mode with the checkpoint's weights.
    Returns (config, model, checkpoint path, step).  The scene box comes from the argument or from the dataparser node."""
    config = try_load_config(config_path)
    model_node = config.pipeline.model
    if eval_num_rays_per_chunk:
        model_node.eval_num_rays_per_chunk = eval_num_rays_per_chunk
    config.load_dir = Path(checkpoint_folder)
    if scene_box is None:
        scene_box = getattr(config.pipeline.datamanager.dataparser, "scene_box", None)
    if scene_box is None:
        raise ValueError("no scene box: neither passed in nor stored in config.pipeline.datamanager.dataparser.scene_box")
    scene_box = torch.as_tensor(scene_box, dtype=torch.float32).reshape(2, 3)
    model_cfg, unknown = model_config_from_nerfstudio(model_node)
    config.unmapped_model_fields = unknown
    device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    checkpoint_path, step = find_checkpoint(checkpoint_folder, checkpoint)
    # (tensors and primitives only: a checkpoint of unknown origin cannot run code through the unpickler)
    state = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    # the image count only sizes nerfstudio's (unused) appearance embedding: take it from the checkpoint if it is there
    num_train_data = int(getattr(config.pipeline.datamanager, "train_num_images", 0) or 1)
    model = NeRSembleNGPModel(model_cfg, SceneBox(scene_box), num_train_data=num_train_data)
    load_nerfstudio_checkpoint(state, model, strict=True)
    model = model.to(device).eval()
    config.load_step = step
    return config, model, checkpoint_path, step
