"""Generates tests/golden/*.npz by importing the REFERENCE's own Python (read-only, /root/reference)
in this container.  Only the resulting small fixtures travel; the reference source never does.

Run:  python tests/golden/make_golden.py          (needs /root/reference; CPU only)

What is pinned (reference-owned code, executed for real):
  * HashEnsemble.forward  (hash_ensemble.py:93-158): rearrange map (c,p)->h, disable-initial (w==1),
    soft transition (1<w<2), cosine grid window, blend einsum -- with a stub ``tinycudann.Encoding``
    whose output is the oracle's tcnn HashGrid restatement on seeded tables.
  * posenc_window (hash_ensemble.py:12-28)
  * WindowedNeRFEncoding.forward (windowed_nerf_encoding.py:33-74)
  * se3_exp_map (util/pytorch3d.py:107-191)
  * SE3DeformationField.compute_offsets (deformation_field.py:148-166) with a tiny seeded config, and at the
    training size (6 x 128, code 128) with forward outputs + autograd gradients -> deformation_full.npz
  * GenericScheduler (engine/generic_scheduler.py), chunked (util/chunker.py)
  * BaseModel.get_dist_loss sample selection / midpoint construction (models/base.py:224-249) with a
    stub flatten_eff_distloss that records its arguments.
  * BaseModel masked-RGB / alpha / empty / near / depth losses (models/base.py:90-222) on a synthetic packed batch.
  * NeRSemblePixelSampler.collate_image_dataset_batch (data/nersemble_pixel_sampler.py:23-69) under a fixed seed
    (nerfstudio's PixelSampler base restated as a stub).
  * extract_top_k_connected_component / filter_occupancy_grid (util/connected_components.py:28-139) on a synthetic
    40^3 grid (numpy + scipy.ndimage for real; cc3d.largest_k stubbed with scipy.ndimage.label).
  * Frustum.contains / contains_points (model_components/frustum.py) and Quantizer / DepthQuantizer / NormalsQuantizer
    (util/quantization.py): imported and run as they are (numpy only) -> dataformat.npz.
Stubs (third-party packages that are not installed): tinycudann, nerfstudio.*, jaxtyping,
torch_efficient_distloss, cc3d, nerfacc.  The nerfstudio MLP / NeRFEncoding / SceneBox stubs restate nerfstudio 0.3.1
 -> that sub-part stays "parity unpinned".
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import oracle  # noqa: E402
from oracle import hashgrid as ohg  # noqa: E402
from tests.helpers import make_tcnn_tables, SMALL_GEOM_KW  # noqa: E402


# ------------------------------------------------------------------------------------------------
# stubs
# ------------------------------------------------------------------------------------------------
def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


_ENC_COUNTER = {"n": 0}
_ENC_CTX = {}


class StubEncoding(nn.Module):
    """tinycudann.Encoding stand-in: HashGrid output = oracle restatement on seeded tables."""

    def __init__(self, n_input_dims, encoding_config, **kw):
        super().__init__()
        self.cfg = encoding_config
        self.n_input_dims = n_input_dims
        if encoding_config["otype"] == "HashGrid":
            self.geom = oracle.grid_geometry(encoding_config["n_levels"], encoding_config["per_level_scale"],
                                             encoding_config["base_resolution"], encoding_config["log2_hashmap_size"])
            self.f_enc = encoding_config["n_features_per_level"]
            self.n_output_dims = encoding_config["n_levels"] * self.f_enc
            c = _ENC_COUNTER["n"]
            _ENC_COUNTER["n"] += 1
            tables = _ENC_CTX["tables"]          # [C, total, F_enc] float32, prepared by the caller
            self.params = nn.Parameter(torch.from_numpy(tables[c].reshape(-1).copy()))
        else:
            self.n_output_dims = n_input_dims

    def forward(self, x):
        if self.cfg["otype"] != "HashGrid":
            return x.half()
        tab = self.params.detach().numpy().astype(np.float16).reshape(self.geom.total_entries, self.f_enc)
        out = ohg.hashgrid_fwd(x.detach().numpy(), tab.view(np.uint16), self.geom)
        return torch.from_numpy(out.copy())


tcnn = _mod("tinycudann")
tcnn.Encoding = StubEncoding

jax = _mod("jaxtyping")


class _Sub:
    def __getitem__(self, item):
        return torch.Tensor


jax.Shaped = _Sub()
jax.Float = _Sub()

ns = _mod("nerfstudio")
for sub in ["cameras", "cameras.rays", "data", "data.scene_box", "field_components", "field_components.encodings",
            "utils", "utils.math"]:
    _mod("nerfstudio." + sub)


class RaySamples:  # only the name is needed by deformation_field.py's annotations
    pass


sys.modules["nerfstudio.cameras.rays"].RaySamples = RaySamples


class SceneBox:
    @staticmethod
    def get_normalized_positions(positions, aabb):
        aabb_lengths = aabb[1] - aabb[0]
        return (positions - aabb[0]) / aabb_lengths


sys.modules["nerfstudio.data.scene_box"].SceneBox = SceneBox


class MLP(nn.Module):
    """Restatement of nerfstudio 0.3.1 field_components.MLP."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim=None, skip_connections=None,
                 activation=nn.ReLU(), out_activation=None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.num_layers = num_layers
        self.layer_width = layer_width
        self.skip_connections = skip_connections
        self._skip = set(skip_connections) if skip_connections else set()
        self.activation = activation
        self.out_activation = out_activation
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    assert i not in self._skip
                    layers.append(nn.Linear(in_dim, layer_width))
                elif i in self._skip:
                    layers.append(nn.Linear(layer_width + in_dim, layer_width))
                else:
                    layers.append(nn.Linear(layer_width, layer_width))
            layers.append(nn.Linear(layer_width, self.out_dim))
        self.layers = nn.ModuleList(layers)

    def forward(self, in_tensor):
        x = in_tensor
        for i, layer in enumerate(self.layers):
            if i in self._skip:
                x = torch.cat([in_tensor, x], -1)
            x = layer(x)
            if self.activation is not None and i < len(self.layers) - 1:
                x = self.activation(x)
        if self.out_activation is not None:
            x = self.out_activation(x)
        return x


sys.modules["nerfstudio.field_components"].MLP = MLP


class NeRFEncoding(nn.Module):
    def __init__(self, in_dim, num_frequencies, min_freq_exp, max_freq_exp, include_input=False):
        super().__init__()
        self.in_dim = in_dim
        self.num_frequencies = num_frequencies
        self.include_input = include_input

    def get_out_dim(self):
        out = self.in_dim * self.num_frequencies * 2
        if self.include_input:
            out += self.in_dim
        return out


sys.modules["nerfstudio.field_components.encodings"].NeRFEncoding = NeRFEncoding
sys.modules["nerfstudio.utils.math"].expected_sin = lambda x, v: torch.exp(-0.5 * v) * torch.sin(x)


# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# nerfstudio 0.3.1 / nerfacc 0.5.2 / torchmetrics MODULE-TREE stubs (what registers which sub-module, parameter and
# buffer under which name -- restated from the published packages; "upstream-stub" in the manifest)
# ------------------------------------------------------------------------------------------------
def _install_model_stubs():
    """Idempotent.  Everything the reference's models/base.py, models/nersemble_instant_ngp.py, fields/
    nersemble_nerfacto_field.py and model_components/*.py import from packages that are not installed."""
    if "nerfstudio.models.base_model" in sys.modules and hasattr(sys.modules["nerfstudio.models.base_model"], "Model"):
        return
    import dataclasses
    from dataclasses import dataclass, field
    for sub in ["engine", "engine.callbacks", "models", "models.base_model", "models.instant_ngp", "fields",
                "fields.base_field", "fields.nerfacto_field", "field_components.activations",
                "field_components.embedding", "field_components.field_heads", "field_components.spatial_distortions",
                "model_components", "model_components.losses", "model_components.renderers",
                "model_components.ray_samplers", "utils.colormaps"]:
        if "nerfstudio." + sub not in sys.modules:
            _mod("nerfstudio." + sub)
    cb = sys.modules["nerfstudio.engine.callbacks"]

    class TrainingCallback:
        def __init__(self, where_to_run=None, func=None, update_every_num_iters=None, iters=None, args=None, kwargs=None):
            self.where_to_run, self.func, self.update_every_num_iters = where_to_run, func, update_every_num_iters
            self.args, self.kwargs = args or [], kwargs or {}

    cb.TrainingCallback = TrainingCallback
    cb.TrainingCallbackAttributes = object
    cb.TrainingCallbackLocation = types.SimpleNamespace(BEFORE_TRAIN_ITERATION="before", AFTER_TRAIN_ITERATION="after")
    bm = sys.modules["nerfstudio.models.base_model"]

    @dataclass
    class ModelConfig:
        _target: type = None
        enable_collider: bool = True
        collider_params: dict = None
        loss_coefficients: dict = None
        eval_num_rays_per_chunk: int = 4096

    class Model(nn.Module):
        """nerfstudio Model.__init__: stores the arguments, calls ``populate_modules()``, then registers the empty
        ``device_indicator_param`` (which therefore sits LAST in ``parameters()`` / the state dict)."""

        def __init__(self, config, scene_box, num_train_data, **kwargs):
            super().__init__()
            self.config, self.scene_box, self.num_train_data, self.kwargs = config, scene_box, num_train_data, kwargs
            self.render_aabb, self.collider = None, None
            self.populate_modules()
            self.callbacks = None
            self.device_indicator_param = nn.Parameter(torch.empty(0))

        def populate_modules(self):
            pass

        def get_training_callbacks(self, training_callback_attributes):
            return []

    bm.Model, bm.ModelConfig = Model, ModelConfig
    ngp = sys.modules["nerfstudio.models.instant_ngp"]

    @dataclass
    class InstantNGPModelConfig(ModelConfig):
        enable_collider: bool = False
        grid_resolution: int = 128
        grid_levels: int = 4
        max_res: int = 2048
        log2_hashmap_size: int = 19
        alpha_thre: float = 0.01
        cone_angle: float = 0.004
        render_step_size: float = None
        near_plane: float = 0.05
        far_plane: float = 1e3
        use_appearance_embedding: bool = False
        background_color: str = "random"
        disable_scene_contraction: bool = False

    class NGPModel(Model):
        def get_param_groups(self):
            if self.field is None:
                raise ValueError("populate_fields() must be called before get_param_groups")
            return {"fields": list(self.field.parameters())}

    ngp.InstantNGPModelConfig, ngp.NGPModel = InstantNGPModelConfig, NGPModel
    # fields
    class Field(nn.Module):
        def __init__(self):
            super().__init__()

    class TCNNNerfactoField(Field):
        pass

    sys.modules["nerfstudio.fields.base_field"].Field = Field
    sys.modules["nerfstudio.fields.base_field"].shift_directions_for_tcnn = lambda d: (d + 1.0) / 2.0
    sys.modules["nerfstudio.fields.nerfacto_field"].TCNNNerfactoField = TCNNNerfactoField
    sys.modules["nerfstudio.field_components.activations"].trunc_exp = torch.exp
    sys.modules["nerfstudio.field_components.embedding"].Embedding = nn.Embedding
    fh = sys.modules["nerfstudio.field_components.field_heads"]
    for name in ("PredNormalsFieldHead", "SemanticFieldHead", "TransientDensityFieldHead", "TransientRGBFieldHead",
                 "UncertaintyFieldHead"):
        setattr(fh, name, type(name, (nn.Module,), {}))
    import enum
    fh.FieldHeadNames = enum.Enum("FieldHeadNames", {"RGB": "rgb", "DENSITY": "density"})
    sd = sys.modules["nerfstudio.field_components.spatial_distortions"]
    sd.SpatialDistortion = type("SpatialDistortion", (nn.Module,), {})
    sd.SceneContraction = type("SceneContraction", (nn.Module,), {"__init__": lambda self, order=None: nn.Module.__init__(self)})
    rays = sys.modules["nerfstudio.cameras.rays"]
    for name in ("RayBundle", "Frustums"):
        if not hasattr(rays, name):
            setattr(rays, name, type(name, (), {}))
    sys.modules["nerfstudio.model_components.losses"].MSELoss = nn.MSELoss
    rn = sys.modules["nerfstudio.model_components.renderers"]
    for name in ("AccumulationRenderer", "DepthRenderer", "RGBRenderer"):
        setattr(rn, name, type(name, (nn.Module,), {"__init__": lambda self, *a, **k: nn.Module.__init__(self)}))
    rs = sys.modules["nerfstudio.model_components.ray_samplers"]

    class Sampler(nn.Module):
        def __init__(self, num_samples=None):
            super().__init__()
            self.num_samples = num_samples

    class VolumetricSampler(Sampler):
        """nerfstudio ray_samplers.VolumetricSampler: keeps the estimator as a SUB-MODULE (``sampler.occupancy_grid.*``
        appears in the state dict a second time)."""

        def __init__(self, occupancy_grid, density_fn=None):
            super().__init__()
            assert occupancy_grid is not None
            self.density_fn = density_fn
            self.occupancy_grid = occupancy_grid

    rs.Sampler, rs.VolumetricSampler, rs.DensityFn = Sampler, VolumetricSampler, object
    ut = sys.modules["nerfstudio.utils"]
    ut.writer = types.SimpleNamespace(put_scalar=lambda **k: None)
    ut.colormaps = sys.modules["nerfstudio.utils.colormaps"]
    ut.colormaps.ColormapOptions = object
    # nerfacc 0.5.2
    na = sys.modules.get("nerfacc") or _mod("nerfacc")

    class OccGridEstimator(nn.Module):
        """nerfacc 0.5.2 OccGridEstimator's registered state: persistent buffers ``resolution`` int32 [3], ``aabbs``
        [levels, 6], ``occs`` [levels * cells], ``binaries`` bool [levels, r, r, r]; non-persistent ``grid_coords``,
        ``grid_indices`` and AbstractEstimator's ``_dummy``."""

        def __init__(self, roi_aabb, resolution=128, levels=1, **kw):
            super().__init__()
            self.register_buffer("_dummy", torch.empty(0), persistent=False)
            res = torch.tensor([resolution] * 3, dtype=torch.int32)
            cells = int(resolution) ** 3
            self.register_buffer("resolution", res)
            self.register_buffer("aabbs", torch.empty((levels, 6), device="meta"))
            self.register_buffer("occs", torch.empty((levels * cells,), device="meta"))
            self.register_buffer("binaries", torch.empty([levels] + [resolution] * 3, dtype=torch.bool, device="meta"))
            self.register_buffer("grid_coords", torch.empty((cells, 3), dtype=torch.int64, device="meta"), persistent=False)
            self.register_buffer("grid_indices", torch.empty((cells,), dtype=torch.int64, device="meta"), persistent=False)

    na.OccGridEstimator = OccGridEstimator
    # torchmetrics / dreifus / distloss: modules without state of their own in this manifest (the real LPIPS carries its
    # network's weights under ``lpips.net.*`` -- third-party, listed under "upstream_omitted")
    tm = _mod("torchmetrics")
    tm.PeakSignalNoiseRatio = type("PeakSignalNoiseRatio", (nn.Module,), {"__init__": lambda self, **k: nn.Module.__init__(self)})
    _mod("torchmetrics.functional").structural_similarity_index_measure = lambda *a, **k: None
    _mod("torchmetrics.image")
    _mod("torchmetrics.image.lpip").LearnedPerceptualImagePatchSimilarity = type(
        "LearnedPerceptualImagePatchSimilarity", (nn.Module,), {"__init__": lambda self, **k: nn.Module.__init__(self)})
    _mod("dreifus"), _mod("dreifus.util")
    _mod("dreifus.util.colormap").apply_scene_flow_colormap = lambda *a, **k: None
    if "torch_efficient_distloss" not in sys.modules:
        _mod("torch_efficient_distloss").flatten_eff_distloss = lambda *a, **k: torch.tensor(0.0)


class ManifestTCNNModule(nn.Module):
    """tinycudann ``Module``: ONE flat fp32 ``params`` nn.Parameter -- also when the object has no parameters at all
    (Identity / Frequency / SphericalHarmonics encodings register an EMPTY one).  Sizes: HashGrid = sum of the level
    sizes x n_features_per_level; FullyFusedMLP = pad16(in) x W + (hidden - 1) x W x W + W x pad16(out), no biases.
    Allocated on the ``meta`` device: names, shapes and dtypes are what the manifest records."""

    def _register(self, n_params: int):
        self.params = nn.Parameter(torch.empty(int(n_params), dtype=torch.float32, device="meta"))


def _pad16(n):
    return (int(n) + 15) // 16 * 16


class ManifestEncoding(ManifestTCNNModule):
    def __init__(self, n_input_dims, encoding_config, **kw):
        super().__init__()
        ot = encoding_config["otype"]
        if ot == "HashGrid":
            g = oracle.grid_geometry(encoding_config["n_levels"], encoding_config["per_level_scale"],
                                     encoding_config["base_resolution"], encoding_config["log2_hashmap_size"])
            self.n_output_dims = encoding_config["n_levels"] * encoding_config["n_features_per_level"]
            self._register(g.total_entries * encoding_config["n_features_per_level"])
        elif ot == "Identity":
            self.n_output_dims = n_input_dims
            self._register(0)
        elif ot == "Frequency":
            self.n_output_dims = n_input_dims * 2 * encoding_config["n_frequencies"]
            self._register(0)
        elif ot == "SphericalHarmonics":
            self.n_output_dims = encoding_config["degree"] ** 2
            self._register(0)
        else:
            raise NotImplementedError(ot)


class ManifestNetwork(ManifestTCNNModule):
    def __init__(self, n_input_dims, n_output_dims, network_config, **kw):
        super().__init__()
        assert network_config["otype"] == "FullyFusedMLP"
        w, nh = network_config["n_neurons"], network_config["n_hidden_layers"]
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self._register(_pad16(n_input_dims) * w + (nh - 1) * w * w + w * _pad16(n_output_dims))


class ManifestNetworkWithInputEncoding(ManifestNetwork):
    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, **kw):
        enc = ManifestEncoding(n_input_dims, encoding_config)
        assert enc.params.numel() == 0, "encoding parameters would be prepended to the network's"
        super().__init__(enc.n_output_dims, n_output_dims, network_config)


def gen_hash_ensemble(out):
    from nersemble.nerfstudio.field_components.hash_ensemble import (HashEnsemble, HashEnsembleConfig,
                                                                     TCNNHashEncodingConfig, posenc_window)
    rng = np.random.default_rng(20240501)
    B = 96
    x = rng.random((B, 3), dtype=np.float32)
    x[0] = [0.0, 0.0, 0.0]
    x[1] = [0.999999, 0.999999, 0.999999]
    x[2] = [0.5, 0.25, 0.75]
    out["he_x"] = x
    geom = oracle.grid_geometry(**SMALL_GEOM_KW)
    for H in (1, 2, 4, 8, 16, 32):
        tables = make_tcnn_tables(H, geom, seed=100 + H, amplitude=0.5)      # [C,total,F_enc] fp32
        _ENC_COUNTER["n"] = 0
        _ENC_CTX["tables"] = tables
        cfg = HashEnsembleConfig(n_hash_encodings=H,
                                 hash_encoding_config=TCNNHashEncodingConfig(
                                     n_levels=SMALL_GEOM_KW["n_levels"],
                                     log2_hashmap_size=SMALL_GEOM_KW["log2_hashmap_size"],
                                     base_resolution=SMALL_GEOM_KW["base_resolution"],
                                     per_level_scale=SMALL_GEOM_KW["per_level_scale"]),
                                 disable_initial_hash_ensemble=True, use_soft_transition=True)
        he = HashEnsemble(cfg)
        assert len(he.hash_encodings) == tables.shape[0]
        code = (rng.standard_normal((B, H)) * 0.7).astype(np.float32)
        out[f"he_code_H{H}"] = code
        windows = [None, 1, 1.5, 3.25, float(H)]
        for wi, w in enumerate(windows):
            with torch.no_grad():
                y = he(torch.from_numpy(x), torch.from_numpy(code.copy()), window_hash_encodings=w)
            assert y.dtype == torch.float16 and y.shape == (B, 2 * SMALL_GEOM_KW["n_levels"])
            out[f"he_out_H{H}_w{wi}"] = y.numpy()
        out[f"he_windows_H{H}"] = np.array([np.nan if w is None else w for w in windows], dtype=np.float64)
    # posenc_window known answers
    for i, (w, n) in enumerate([(3.25, 32), (0.0, 32), (1.0, 32), (32.0, 32), (7.5, 16), (0.4, 1)]):
        out[f"pw_{i}"] = posenc_window(w, 0, n - 1, n).numpy()
        out[f"pw_{i}_args"] = np.array([w, n], dtype=np.float64)


def gen_deformation(out):
    from nersemble.nerfstudio.field_components.windowed_nerf_encoding import WindowedNeRFEncoding
    from nersemble.nerfstudio.field_components.deformation_field import (SE3DeformationField,
                                                                          SE3DeformationFieldConfig)
    from nersemble.util.pytorch3d import se3_exp_map
    rng = np.random.default_rng(77)
    # windowed PE
    enc = WindowedNeRFEncoding(in_dim=3, num_frequencies=7, min_freq_exp=0.0, max_freq_exp=6.0, include_input=True)
    xp = rng.random((40, 3), dtype=np.float32)
    xp[0] = [0.1, 0.2, 0.3]
    out["pe_x"] = xp
    for i, w in enumerate([None, 0.0, 3.5, 7.0, 0.25]):
        with torch.no_grad():
            out[f"pe_out_{i}"] = enc(torch.from_numpy(xp), windows_param=w).numpy()
    out["pe_windows"] = np.array([np.nan, 0.0, 3.5, 7.0, 0.25])
    # se3 exp map: random + edge cases (zero rotation -> eps clamp, tiny, large angle)
    screw = (rng.standard_normal((64, 6)) * 0.5).astype(np.float32)
    screw[0] = 0
    screw[1, 3:] = 0
    screw[2, 3:] = [1e-4, 0, 0]
    screw[3, 3:] = [3.0, 0.1, -0.2]
    screw[4, 3:] = [0, 6.0, 0]
    screw[5] = [1, 2, 3, 1e-3, 1e-3, 1e-3]
    out["se3_in"] = screw
    out["se3_out"] = se3_exp_map(torch.from_numpy(screw)).numpy()
    # tiny deformation field, fp32
    torch.manual_seed(1234)
    cfg = SE3DeformationFieldConfig(warp_code_dim=8, mlp_num_layers=6, mlp_layer_width=32)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, cfg, max_n_samples_per_batch=17)
    # make the heads non-trivial so the SE(3) part is exercised (reference init is ~identity)
    with torch.no_grad():
        df.se3_field.mlp_r.layers[-1].weight.mul_(2e4)
        df.se3_field.mlp_v.layers[-1].weight.mul_(2e4)
    pos = (torch.rand(50, 3) * (aabb[1] - aabb[0]) + aabb[0])
    codes = torch.randn(50, 8) * 0.3
    out["df_pos"] = pos.numpy()
    out["df_code"] = codes.numpy()
    for k, v in df.state_dict().items():
        out["df_sd_" + k] = v.numpy()
    for i, w in enumerate([None, 0.0, 2.75, 7.0]):
        with torch.no_grad():
            out[f"df_off_{i}"] = df.compute_offsets(pos, codes, w).numpy()
    out["df_windows"] = np.array([np.nan, 0.0, 2.75, 7.0])


def gen_deformation_full(out):
    """The reference's own SE3DeformationField at the TRAINING size (6 x 128, warp code 128, train_nersemble.py:84-91),
    fp32 on CPU: offsets for four window values and, for one of them, the autograd gradients w.r.t. the warp codes,
    the positions' normalised encoding input (through the codes only -- positions are data) and every parameter.
    The weights come from tests/helpers.make_deform_state_dict (numpy PCG64, seed in the fixture), so the fixture
    holds inputs and outputs only."""
    from nersemble.nerfstudio.field_components.deformation_field import (SE3DeformationField,
                                                                          SE3DeformationFieldConfig)
    from tests.helpers import make_deform_state_dict, DEFORM_KEYS
    seed = 4711
    cfg = SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6, mlp_layer_width=128)
    aabb = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    df = SE3DeformationField(aabb, cfg, max_n_samples_per_batch=29)
    sd = {k: torch.from_numpy(v) for k, v in make_deform_state_dict(seed).items()}
    sd["aabb"] = aabb
    assert set(sd) == set(df.state_dict())
    df.load_state_dict(sd)
    rng = np.random.default_rng(seed + 1)
    pos = (rng.random((64, 3), dtype=np.float32) * (aabb[1] - aabb[0]).numpy() + aabb[0].numpy()).astype(np.float32)
    pos[0] = [3.0, 0.0, 0.0]                                   # outside the box: the field extrapolates, no clamp
    codes = (rng.standard_normal((64, 128)) * 0.3).astype(np.float32)
    gw = rng.standard_normal((64, 3)).astype(np.float32)
    out["dff_seed"] = np.array([seed])
    out["dff_pos"], out["dff_code"], out["dff_gw"] = pos, codes, gw
    wins = [None, 0.0, 2.75, 7.0]
    out["dff_windows"] = np.array([np.nan, 0.0, 2.75, 7.0])
    for i, w in enumerate(wins):
        with torch.no_grad():
            out[f"dff_off_{i}"] = df.compute_offsets(torch.from_numpy(pos), torch.from_numpy(codes), w).numpy()
    c = torch.from_numpy(codes).requires_grad_(True)
    off = df.compute_offsets(torch.from_numpy(pos), c, 2.75)
    (off * torch.from_numpy(gw)).sum().backward()
    out["dff_gcode"] = c.grad.numpy()
    named = dict(df.named_parameters())
    out["dff_gparams"] = torch.cat([named[k].grad.reshape(-1) for k in DEFORM_KEYS]).numpy()


def gen_misc(out):
    from nersemble.nerfstudio.engine.generic_scheduler import GenericScheduler
    from nersemble.util.chunker import chunked
    s = GenericScheduler(init_value=1, final_value=32, begin_step=40000, end_step=80000)
    steps = np.array([0, 39999, 40000, 40001, 50000, 60000, 79999, 80000, 80001, 300000])
    vals = []
    for st in steps:
        s.update(int(st))
        vals.append(s.get_value())
    out["sched_steps"] = steps
    out["sched_vals"] = np.array(vals, dtype=np.float64)
    s.eval()
    out["sched_eval"] = np.array([s.get_value()], dtype=np.float64)
    a = torch.arange(10)
    b = torch.arange(20).reshape(10, 2)
    sizes = []
    for ca, cn, cb in chunked(4, a, None, b):
        assert cn is None
        sizes.append([len(ca), len(cb), int(ca[0]), int(cb[0, 0])])
    out["chunk_sizes"] = np.array(sizes)
    single = [len(c) for c in chunked(3, a)]
    out["chunk_single"] = np.array(single)


def gen_distloss_selection(out):
    _install_model_stubs()
    sys.modules["nerfstudio.utils"].writer = types.SimpleNamespace()
    rec = {}
    ted = _mod("torch_efficient_distloss")

    def flatten_eff_distloss(w, m, interval, ray_id):
        rec["w"], rec["m"], rec["interval"], rec["ray_id"] = w, m, interval, ray_id
        return torch.tensor(2.0)

    ted.flatten_eff_distloss = flatten_eff_distloss
    from nersemble.nerfstudio.models.base import BaseModel, BaseModelConfig
    model = BaseModel.__new__(BaseModel)
    nn.Module.__init__(model)
    cfg = BaseModelConfig.__new__(BaseModelConfig)
    cfg.lambda_dist_loss = 1e-4
    cfg.dist_loss_max_rays = 5
    model.config = cfg
    rng = np.random.default_rng(5)
    counts = np.array([3, 0, 4, 2, 0, 1, 5, 2])
    ray_idx = np.repeat(np.arange(len(counts)), counts)
    S = len(ray_idx)
    starts = np.sort(rng.random(S).astype(np.float32)) * 3
    ends = starts + 0.011
    weights = rng.random((S, 1)).astype(np.float32)
    fr = types.SimpleNamespace(starts=torch.from_numpy(starts)[:, None], ends=torch.from_numpy(ends.astype(np.float32))[:, None])
    rs = types.SimpleNamespace(frustums=fr)
    loss = model.get_dist_loss(rs, torch.from_numpy(ray_idx), torch.from_numpy(weights))
    out["dl_ray_idx"] = ray_idx
    out["dl_starts"] = starts
    out["dl_ends"] = ends.astype(np.float32)
    out["dl_weights"] = weights
    out["dl_sel_w"] = rec["w"].numpy()
    out["dl_sel_m"] = rec["m"].numpy()
    out["dl_sel_interval"] = rec["interval"].numpy()
    out["dl_sel_ray_id"] = rec["ray_id"].numpy()
    out["dl_loss"] = np.array([float(loss)])      # = lambda * stub value


def gen_losses(out):
    """BaseModel losses (models/base.py:90-222) on a synthetic packed batch: masked RGB MSE, alpha L1, empty / near
    (cumsum bookkeeping, Normal CDF with sigma=(eps/3)^2) and depth losses."""
    from nersemble.nerfstudio.models.base import BaseModel, BaseModelConfig
    from nersemble.nerfstudio.engine.generic_scheduler import GenericScheduler
    model = BaseModel.__new__(BaseModel)
    nn.Module.__init__(model)
    cfg = BaseModelConfig.__new__(BaseModelConfig)
    cfg.use_masked_rgb_loss = True
    cfg.alpha_mask_threshold = 0
    cfg.lambda_alpha_loss = 1e-2
    cfg.lambda_empty_loss = 1e-2
    cfg.lambda_near_loss = 1e-4
    cfg.lambda_depth_loss = 1e-4
    cfg.lambda_dist_loss = 0
    cfg.dist_loss_max_rays = 5000
    model.config = cfg
    model.rgb_loss = nn.MSELoss()
    model._dummy = nn.Parameter(torch.zeros(1))
    type(model).device = property(lambda self: torch.device("cpu"))
    model.sched_eps_depth = GenericScheduler(init_value=0.9, final_value=0.01, begin_step=0, end_step=10000)
    model.sched_eps_depth.update(2500)
    model.train()
    rng = np.random.default_rng(11)
    R = 40
    counts = rng.integers(1, 30, R)              # the reference's bookkeeping needs every ray to own >= 1 sample
    ray_idx = np.repeat(np.arange(R), counts)
    S = len(ray_idx)
    starts = np.concatenate([8.0 + np.sort(rng.random(c)) * 2.0 for c in counts]).astype(np.float32)
    ends = (starts + 0.011).astype(np.float32)
    weights = (rng.random((S, 1)) * 0.1).astype(np.float32)
    depth = (8.5 + rng.random(R) * 1.0).astype(np.float32)
    depth[::5] = 0                                # rays without depth information
    alpha = rng.integers(0, 256, (R, 1)).astype(np.uint8)
    alpha[::3] = 255
    alpha[1::7] = 0
    image = rng.random((R, 3)).astype(np.float32)
    rgb_pred = rng.random((R, 3)).astype(np.float32)
    w_t = torch.from_numpy(weights)
    acc = torch.zeros(R).index_add_(0, torch.from_numpy(ray_idx), w_t[:, 0])[:, None]
    depth_pred = torch.from_numpy((8.4 + rng.random((R, 1))).astype(np.float32))
    batch = {"image": torch.from_numpy(image), "alpha_map": torch.from_numpy(alpha), "depth_maps": torch.from_numpy(depth)}
    fr = types.SimpleNamespace(starts=torch.from_numpy(starts)[:, None], ends=torch.from_numpy(ends)[:, None])
    rs = types.SimpleNamespace(frustums=fr)
    near, empty = model.get_near_and_empty_loss(batch, rs, torch.from_numpy(ray_idx), w_t, acc)
    out["ls_ray_idx"], out["ls_starts"], out["ls_ends"], out["ls_weights"] = ray_idx, starts, ends, weights
    out["ls_depth"], out["ls_alpha"], out["ls_image"], out["ls_rgb_pred"] = depth, alpha, image, rgb_pred
    out["ls_acc"], out["ls_depth_pred"] = acc.numpy(), depth_pred.numpy()
    out["ls_eps"] = np.array([model.sched_eps_depth.value], dtype=np.float64)
    out["ls_near"] = np.array([float(near)])
    out["ls_empty"] = np.array([float(empty)])
    out["ls_rgb"] = np.array([float(model.get_masked_rgb_loss(batch, torch.from_numpy(rgb_pred)))])
    out["ls_alpha_loss"] = np.array([float(model.get_alpha_loss(batch, acc))])
    out["ls_depth_loss"] = np.array([float(model.get_depth_loss(batch, depth_pred))])


def gen_occupancy_filter(out):
    """filter_occupancy_grid / extract_top_k_connected_component (util/connected_components.py:28-139), run for real
    with numpy + scipy.ndimage.  Stub: ``cc3d.largest_k`` (not installed) = largest 6-connected component via
    scipy.ndimage.label -> the connected-component step itself stays "parity unpinned"."""
    import scipy.ndimage as ndi

    def largest_k(arr, k=1, connectivity=6, delta=0, return_N=False):
        assert k == 1 and connectivity == 6
        lab, n = ndi.label(arr > 0)                      # default structure = 6-connectivity in 3-D
        res = np.zeros(arr.shape, dtype=np.uint32)
        if n > 0:
            counts = np.bincount(lab.ravel())[1:]
            res[lab == (int(np.argmax(counts)) + 1)] = 1
        return (res, min(n, 1)) if return_N else res

    cc3d = _mod("cc3d")
    cc3d.largest_k = largest_k
    if "nerfacc" not in sys.modules:
        _mod("nerfacc")
    if not hasattr(sys.modules["nerfacc"], "OccGridEstimator"):       # (the module-tree stub, if it is installed already)
        sys.modules["nerfacc"].OccGridEstimator = object
    from nersemble.util.connected_components import extract_top_k_connected_component, filter_occupancy_grid

    rng = np.random.default_rng(4242)
    R = 40
    zz, yy, xx = np.meshgrid(np.arange(R), np.arange(R), np.arange(R), indexing="ij")

    def blob(c, r, amp):
        d2 = (xx - c[0]) ** 2 + (yy - c[1]) ** 2 + (zz - c[2]) ** 2
        return amp * np.exp(-d2 / (2.0 * r * r))

    occs = blob((20, 19, 21), 7.0, 6.0) + blob((6, 7, 30), 2.5, 5.0) + blob((33, 31, 8), 2.0, 7.0)   # head + 2 floaters
    occs[19:21, 19:21, 6:16] += 4.0                       # a thin bridge towards a floater (cut by the thinning blur)
    occs += rng.random((R, R, R)) * 0.4                   # noise floor
    occs = occs.astype(np.float32)
    out["occ_in"] = occs
    for i, (thr, s_thin, s_ero) in enumerate([(0.6, 1, 5), (0.6, 1, 2), (0.8, 2, 3)]):
        m = extract_top_k_connected_component(occs.copy(), threshold=thr, sigma_thinning=s_thin, sigma_erosion=s_ero)[0]
        out[f"occ_mask_{i}"] = np.packbits(m.astype(bool).ravel())
        out[f"occ_args_{i}"] = np.array([thr, s_thin, s_ero], dtype=np.float64)

    class FakeGrid:
        resolution = torch.tensor([R, R, R])
        device = "cpu"

    g = FakeGrid()
    g.occs = torch.from_numpy(occs.reshape(-1).copy())
    g.binaries = torch.from_numpy(rng.random((1, R, R, R)) < 0.7)
    out["occ_binaries_in"] = np.packbits(g.binaries.numpy().ravel())
    filter_occupancy_grid(g, threshold=0.6, sigma_erosion=5)
    out["occ_binaries_out"] = np.packbits(g.binaries.numpy().ravel())


def gen_pixel_sampler(out):
    """NeRSemblePixelSampler.collate_image_dataset_batch (data/nersemble_pixel_sampler.py:23-69), run for real on a
    synthetic image batch under a fixed torch seed.  Stub: nerfstudio's PixelSampler base class (constructor +
    uniform ``sample_method``, restated from nerfstudio 0.3.1)."""
    ps = _mod("nerfstudio.data.pixel_samplers") if "nerfstudio.data.pixel_samplers" not in sys.modules \
        else sys.modules["nerfstudio.data.pixel_samplers"]
    if "nerfstudio.data" not in sys.modules:
        _mod("nerfstudio.data")

    class PixelSampler:
        def __init__(self, num_rays_per_batch, keep_full_image=False, **kwargs):
            self.num_rays_per_batch = num_rays_per_batch
            self.keep_full_image = keep_full_image

        def sample_method(self, batch_size, num_images, image_height, image_width, mask=None, device="cpu"):
            if isinstance(mask, torch.Tensor):
                nonzero_indices = torch.nonzero(mask[..., 0], as_tuple=False)
                chosen = torch.randint(0, nonzero_indices.shape[0], (batch_size,))
                return nonzero_indices[chosen]
            return torch.floor(torch.rand((batch_size, 3), device=device)
                               * torch.tensor([num_images, image_height, image_width], device=device)).long()

    ps.PixelSampler = PixelSampler
    from nersemble.nerfstudio.data.nersemble_pixel_sampler import NeRSemblePixelSampler
    g = torch.Generator().manual_seed(11)
    N, Hh, Ww = 5, 12, 9
    batch = {"image": torch.rand((N, Hh, Ww, 3), generator=g),
             "alpha_map": (torch.rand((N, Hh, Ww, 1), generator=g) * 255).to(torch.uint8),
             "depth_map": torch.rand((N, Hh, Ww), generator=g),
             "timesteps": torch.tensor([3, 3, 17, 0, 42]), "cam_ids": torch.tensor([0, 5, 2, 2, 7]),
             "image_idx": torch.tensor([10, 11, 25, 4, 31])}
    for k, v in batch.items():
        out[f"px_in_{k}"] = v.numpy()
    sampler = NeRSemblePixelSampler(64, additional_metadata=["depth_map", "timesteps", "cam_ids"])
    torch.manual_seed(2024)
    col = sampler.collate_image_dataset_batch({k: v.clone() for k, v in batch.items()}, 64)
    for k, v in col.items():
        out[f"px_out_{k}"] = v.numpy()
    mask = (torch.rand((N, Hh, Ww, 1), generator=g) > 0.6)
    torch.manual_seed(2025)
    colm = sampler.collate_image_dataset_batch({**{k: v.clone() for k, v in batch.items()}, "mask": mask}, 32)
    out["px_mask"] = mask.numpy()
    for k, v in colm.items():
        out[f"px_outm_{k}"] = v.numpy()


def gen_dataformat(out):
    """Frustum / HalfSpaceCollection (model_components/frustum.py:17-26,60-100,148-193 -- the numpy twin of the
    TorchFrustum the dataparser builds, which needs ``.cuda()``) and the depth / normal codecs
    (util/quantization.py:33-120), all imported and run for real: no stubs."""
    from nersemble.nerfstudio.model_components.frustum import Frustum
    from nersemble.util.quantization import DepthQuantizer, NormalsQuantizer, Quantizer
    rng = np.random.default_rng(77)
    poses, ks, masks, singles = [], [], [], []
    pts = rng.uniform(-3.0, 3.0, size=(4096, 3))
    dims = (1100, 1604)
    for i in range(4):
        # random OpenCV pose looking roughly at the origin from ~9 units away
        eye = rng.normal(size=3)
        eye = 9.0 * eye / np.linalg.norm(eye)
        fwd = -eye / np.linalg.norm(eye) + 0.05 * rng.normal(size=3)
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.9 else np.array([0.0, 1.0, 0.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        pose = np.eye(4)
        pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, down, fwd, eye
        k = np.array([[2200.0 + 50 * i, 0.0, 550.0 + 3 * i], [0.0, 2190.0 - 20 * i, 802.0 - 5 * i], [0.0, 0.0, 1.0]])
        fr = Frustum(pose, k, dims)
        poses.append(pose), ks.append(k)
        masks.append(fr.contains_points(pts))
        singles.append(np.array([fr.contains(p) for p in pts[:16]]))
    out["fr_pose"], out["fr_k"], out["fr_dims"] = np.stack(poses), np.stack(ks), np.array(dims)
    out["fr_points"], out["fr_mask"], out["fr_single"] = pts, np.stack(masks), np.stack(singles)

    depth = rng.uniform(0.0, 2.4, size=(48, 40)).astype(np.float32)
    depth[rng.random(depth.shape) < 0.2] = 0
    out["dq_in"] = depth.copy()
    dq = DepthQuantizer()
    codes = dq.encode(depth.copy())
    out["dq_codes"], out["dq_decoded"] = codes, dq.decode(codes)
    known = np.array([[0, 1, 2, 32768, 65535]], dtype=np.uint16)
    out["dq_known_codes"], out["dq_known"] = known, dq.decode(known)
    q8 = Quantizer(min_values=-1.0, max_values=3.0, bits=8, separate_mask=False)
    vals = rng.uniform(-1.0, 3.0, size=(17, 5))
    out["q8_in"], out["q8_codes"] = vals, q8.encode(vals.copy())
    out["q8_decoded"] = q8.decode(out["q8_codes"])
    normals = rng.normal(size=(24, 20, 3))
    normals /= np.linalg.norm(normals, axis=-1, keepdims=True)
    normals[..., 2] = -np.abs(normals[..., 2]) * 0.4          # camera-facing: theta within [pi/3, pi]
    normals /= np.linalg.norm(normals, axis=-1, keepdims=True)
    normals[rng.random(normals.shape[:2]) < 0.25] = 0
    nq = NormalsQuantizer()
    ncodes = nq.encode(normals.copy())
    out["nq_in"], out["nq_codes"], out["nq_decoded"] = normals, ncodes, nq.decode(ncodes)


def gen_config_yml(path):
    """A run's ``config.yml`` as the reference writes it (``model_manager/base.py:44-46``: ``yaml.dump(config)`` of the
    ``NeRSembleTrainerConfig`` built in ``scripts/train/train_nersemble.py:146-262``).  The config CLASSES live in packages
    that are not installed (nerfstudio) or import them (nersemble.nerfstudio.*), so the objects dumped here are instances of
    dynamically created classes that carry the reference's module paths and class names -- PyYAML's dumper only writes
    ``cls.__module__ + "." + cls.__name__`` -- and the reference's own FIELD NAMES and DEFAULTS, read from its source files
    with ``ast`` (no import); nerfstudio's inherited fields are the ones SURVEY.md A.3 lists.  Values as in
    ``train_nersemble.py`` for participant 30 with 16 hash grids."""
    import ast
    import pathlib
    import yaml
    ref = pathlib.Path("/root/reference/src/nersemble")

    def fields_of(rel, cls_name):
        tree = ast.parse((ref / rel).read_text())
        out = {}
        for node in ast.walk(tree):
            if isinstance(node, ast.ClassDef) and node.name == cls_name:
                for st in node.body:
                    if isinstance(st, ast.AnnAssign) and isinstance(st.target, ast.Name) and st.value is not None:
                        try:
                            out[st.target.id] = ast.literal_eval(st.value)
                        except Exception:
                            out[st.target.id] = None
        return out

    def make(module, name, **values):
        cls = type(name, (), {"__module__": module})
        obj = cls()
        obj.__dict__.update(values)
        return obj

    def cls_ref(module, name):
        return type(name, (), {"__module__": module})

    enc = make("nersemble.nerfstudio.field_components.hash_ensemble", "TCNNHashEncodingConfig",
               **fields_of("nerfstudio/field_components/hash_ensemble.py", "TCNNHashEncodingConfig"))
    he_f = fields_of("nerfstudio/field_components/hash_ensemble.py", "HashEnsembleConfig")
    he_f.update(n_hash_encodings=16, hash_encoding_config=enc, disable_initial_hash_ensemble=True, use_soft_transition=True)
    he = make("nersemble.nerfstudio.field_components.hash_ensemble", "HashEnsembleConfig", **he_f)
    df_f = fields_of("nerfstudio/field_components/deformation_field.py", "SE3DeformationFieldConfig")
    df_f.update(warp_code_dim=128, mlp_num_layers=6, mlp_layer_width=128)
    df = make("nersemble.nerfstudio.field_components.deformation_field", "SE3DeformationFieldConfig", **df_f)
    model_f = {  # nerfstudio 0.3.1 ModelConfig + InstantNGPModelConfig (UPSTREAM)
        "_target": cls_ref("nersemble.nerfstudio.models.nersemble_instant_ngp", "NeRSembleNGPModel"),
        "enable_collider": False, "collider_params": None, "loss_coefficients": {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0},
        "eval_num_rays_per_chunk": 4096, "grid_resolution": 128, "grid_levels": 1, "max_res": 2048,
        "log2_hashmap_size": 19, "alpha_thre": 0.01, "cone_angle": 0.0, "render_step_size": 0.011, "near_plane": 0.2,
        "far_plane": 1000.0, "use_appearance_embedding": False, "background_color": "white",
        "disable_scene_contraction": True}
    model_f.update(fields_of("nerfstudio/models/base.py", "BaseModelConfig"))
    own = fields_of("nerfstudio/models/nersemble_instant_ngp.py", "NeRSembleNGPModelConfig")
    own.pop("_target", None)
    model_f.update(own)
    model_f.update(early_stop_eps=0, occ_thre=0.01, max_n_samples_per_batch=2 ** 20, n_timesteps=100, latent_dim_time=16,
                   use_masked_rgb_loss=True, alpha_mask_threshold=0, lambda_alpha_loss=1e-2, lambda_near_loss=1e-4,
                   lambda_empty_loss=1e-2, lambda_depth_loss=1e-4, lambda_dist_loss=1e-4, use_hash_ensemble=True,
                   hash_ensemble_config=he, use_deformation_field=True, use_separate_deformation_time_embedding=True,
                   deformation_field_config=df, disable_occupancy_grid=False, window_hash_encodings_begin=40000,
                   window_hash_encodings_end=80000, window_deform_begin=0, window_deform_end=20000,
                   use_view_frustum_culling=False, view_frustum_culling=2)
    model = make("nersemble.nerfstudio.models.nersemble_instant_ngp", "NeRSembleNGPModelConfig", **model_f)
    dataparser = make("nersemble.nerfstudio.dataparser.nersemble_dataparser", "NeRSembleDataParserConfig",
                      _target=cls_ref("nersemble.nerfstudio.dataparser.nersemble_dataparser", "NeRSembleDataParser"),
                      participant_id=30, sequence_name="EXP-2-eyes", start_timestep=0, n_timesteps=100, skip_timesteps=3,
                      scale_factor=9, scene_box=torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]]))
    datamanager = make("nersemble.nerfstudio.datamanager.nersemble_datamanager", "NeRSembleVanillaDataManagerConfig",
                       _target=cls_ref("nersemble.nerfstudio.datamanager.nersemble_datamanager", "NeRSembleVanillaDataManager"),
                       dataparser=dataparser, train_num_rays_per_batch=4096, eval_num_rays_per_batch=1024,
                       train_num_images_to_sample_from=24, train_num_times_to_repeat_images=20,
                       eval_num_images_to_sample_from=36, use_cache_compression=False, max_cached_items=10000)
    pipeline = make("nerfstudio.pipelines.base_pipeline", "VanillaPipelineConfig",
                    _target=cls_ref("nerfstudio.pipelines.base_pipeline", "VanillaPipeline"), datamanager=datamanager,
                    model=model)

    def opt(lr, gamma):
        return {"optimizer": make("nerfstudio.engine.optimizers", "AdamOptimizerConfig",
                                  _target=torch.optim.Adam, lr=lr, eps=1e-15, max_norm=None, weight_decay=0),
                "scheduler": make("nersemble.nerfstudio.engine.step_lr_scheduler", "StepLRSchedulerConfig",
                                  step_size=20000, gamma=gamma)}

    config = make("nersemble.nerfstudio.config.nersemble_trainer_config", "NeRSembleTrainerConfig",
                  _target=cls_ref("nersemble.nerfstudio.engine.nersemble_trainer", "NeRSembleTrainer"),
                  output_dir=pathlib.PosixPath("/models/nersemble"), method_name="nersemble", experiment_name="NERS-9999",
                  project_name="nersemble", run_name="NERS-9999", relative_model_dir=pathlib.PosixPath("checkpoints/"),
                  steps_per_save=50000, steps_per_eval_batch=500, steps_per_eval_image=500,
                  steps_per_eval_all_images=50000, max_num_iterations=300001, mixed_precision=True,
                  save_only_latest_checkpoint=True, load_dir=None, load_step=None, log_gradients=False, vis="wandb",
                  pipeline=pipeline,
                  optimizers={"fields": opt(5e-3, 0.8), "deformation_field": opt(1e-3, 0.5), "embeddings": opt(5e-3, 0.8)})
    pathlib.Path(path).write_text(yaml.dump(config))


def gen_state_manifest(path):
    """tests/golden/state_manifest.json: the reference's OWN module tree, instantiated for real --
    ``NeRSembleNGPModel.populate_modules`` (nersemble_instant_ngp.py:81-179) with ``NeRSembleNeRFactoField``
    (nersemble_nerfacto_field.py:32-226), ``HashEnsemble`` (hash_ensemble.py:69-91), ``SE3DeformationField``
    (deformation_field.py:119-131), the two ``nn.Embedding`` tables, ``scene_aabb``, and ``get_param_groups``
    (:502-514) -- for the configurations of ``train_nersemble.py`` with H = 1, 16 and 32 hash grids.  Recorded per
    configuration: every ``state_dict()`` key with shape and dtype, in order; every parameter group of
    ``get_param_groups()`` as the ordered list of parameter NAMES (the order ``torch.optim.Adam.state_dict()`` numbers
    them in); which entries the reference's own code registers ("reference") and which come from the restated
    third-party module trees ("upstream-stub": nerfacc's estimator buffers, nerfstudio's ``device_indicator_param``).
    tcnn objects are the ``Manifest*`` stand-ins above (meta tensors: names / shapes / dtypes only)."""
    import json
    _install_model_stubs()
    tc = sys.modules["tinycudann"]
    saved = {k: getattr(tc, k, None) for k in ("Encoding", "Network", "NetworkWithInputEncoding")}
    tc.Encoding, tc.Network, tc.NetworkWithInputEncoding = ManifestEncoding, ManifestNetwork, ManifestNetworkWithInputEncoding
    try:
        from nersemble.nerfstudio.field_components.deformation_field import SE3DeformationFieldConfig
        from nersemble.nerfstudio.field_components.hash_ensemble import HashEnsembleConfig, TCNNHashEncodingConfig
        from nersemble.nerfstudio.models.nersemble_instant_ngp import NeRSembleNGPModel, NeRSembleNGPModelConfig
        doc = {"source": "tests/golden/make_golden.py::gen_state_manifest -- the reference's NeRSembleNGPModel instantiated "
                         "under module-tree stubs of tinycudann / nerfstudio 0.3.1 / nerfacc 0.5.2 / torchmetrics",
               "upstream_omitted": ["lpips.net.* (torchmetrics LPIPS keeps its pretrained network's weights in the module "
                                    "tree; a real checkpoint lists them, this manifest cannot)"],
               "configs": {}}
        for H, T in ((1, 1), (16, 100), (32, 100)):
            scene_box = types.SimpleNamespace(aabb=torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]]))
            cfg = NeRSembleNGPModelConfig(
                # train_nersemble.py:184-241
                render_step_size=0.011, near_plane=0.2, far_plane=1e3, cone_angle=0, alpha_thre=1e-2, occ_thre=1e-2,
                early_stop_eps=0, background_color="white", grid_levels=1, disable_scene_contraction=True,
                max_n_samples_per_batch=2 ** 20, n_timesteps=T, latent_dim_time=H, use_masked_rgb_loss=True,
                alpha_mask_threshold=0, lambda_alpha_loss=1e-2, lambda_near_loss=1e-4, lambda_empty_loss=1e-2,
                lambda_depth_loss=1e-4, lambda_dist_loss=1e-4, use_hash_ensemble=True,
                hash_ensemble_config=HashEnsembleConfig(n_hash_encodings=H, hash_encoding_config=TCNNHashEncodingConfig(),
                                                        disable_initial_hash_ensemble=True, use_soft_transition=True),
                use_deformation_field=True, use_separate_deformation_time_embedding=True,
                deformation_field_config=SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6,
                                                                   mlp_layer_width=128),
                window_hash_encodings_begin=40000, window_hash_encodings_end=80000, window_deform_begin=0,
                window_deform_end=20000, use_view_frustum_culling=False)
            model = NeRSembleNGPModel(cfg, scene_box, 12 * T, metadata={"camera_frustums": None})
            upstream = ("occupancy_grid.", "sampler.occupancy_grid.", "device_indicator_param")
            state = [{"key": k, "shape": list(v.shape), "dtype": str(v.dtype).replace("torch.", ""),
                      "origin": "upstream-stub" if k.startswith(upstream) else "reference"}
                     for k, v in model.state_dict().items()]
            name_of = {id(p): n for n, p in model.named_parameters()}
            groups = {g: [name_of[id(p)] for p in ps] for g, ps in model.get_param_groups().items()}
            trainable = {n: bool(p.requires_grad) for n, p in model.named_parameters()}
            doc["configs"][f"H{H}"] = {
                "config": {"n_hash_encodings": H, "latent_dim_time": H, "n_timesteps": T, "log2_hashmap_size": 19,
                           "warp_code_dim": 128, "num_train_data": 12 * T},
                "state_dict": state, "param_groups": groups,
                "requires_grad": trainable}
        with open(path, "w") as f:
            json.dump(doc, f, indent=1)
            f.write("\n")
    finally:
        for k, v in saved.items():
            if v is None:
                if hasattr(tc, k):
                    delattr(tc, k)
            else:
                setattr(tc, k, v)


def main():
    """python tests/golden/make_golden.py [hash_ensemble] [deformation] [deformation_full] [misc] [occupancy_filter] [pixel_sampler] [dataformat]   (default: all)"""
    torch.set_num_threads(4)
    which = set(sys.argv[1:]) or {"hash_ensemble", "deformation", "deformation_full", "config_yml", "misc", "occupancy_filter", "pixel_sampler", "dataformat",
                                    "state_manifest"}
    written = []
    if "hash_ensemble" in which:
        a = {}
        gen_hash_ensemble(a)
        np.savez_compressed(os.path.join(HERE, "hash_ensemble.npz"), **a)
        written.append("hash_ensemble.npz")
    if "deformation" in which:
        b = {}
        gen_deformation(b)
        np.savez_compressed(os.path.join(HERE, "deformation.npz"), **b)
        written.append("deformation.npz")
    if "deformation_full" in which:
        b2 = {}
        gen_deformation_full(b2)
        np.savez_compressed(os.path.join(HERE, "deformation_full.npz"), **b2)
        written.append("deformation_full.npz")
    if "misc" in which:
        c = {}
        gen_misc(c)
        gen_distloss_selection(c)
        gen_losses(c)
        np.savez_compressed(os.path.join(HERE, "misc.npz"), **c)
        written.append("misc.npz")
    if "occupancy_filter" in which:
        d = {}
        gen_occupancy_filter(d)
        np.savez_compressed(os.path.join(HERE, "occupancy_filter.npz"), **d)
        written.append("occupancy_filter.npz")
    if "pixel_sampler" in which:
        e = {}
        gen_pixel_sampler(e)
        np.savez_compressed(os.path.join(HERE, "pixel_sampler.npz"), **e)
        written.append("pixel_sampler.npz")
    if "dataformat" in which:
        f_ = {}
        gen_dataformat(f_)
        np.savez_compressed(os.path.join(HERE, "dataformat.npz"), **f_)
        written.append("dataformat.npz")
    if "config_yml" in which:
        gen_config_yml(os.path.join(HERE, "config.yml"))
        written.append("config.yml")
    if "state_manifest" in which:
        gen_state_manifest(os.path.join(HERE, "state_manifest.json"))
        written.append("state_manifest.json")
    for f in written:
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
