"""tcnn-shaped operator layer (``tinycudann`` Python API subset the reference imports):
``Network``, ``NetworkWithInputEncoding``, ``Encoding`` with ``.params`` (flat fp32 ``nn.Parameter``),
``.n_output_dims`` and fp16 outputs -- backed by the MFMA kernels of libnsx.so.

Reference call sites: nersemble_nerfacto_field.py:98-112 (direction encoding), :142-153 (mlp_base),
:162-172 (mlp_head).  Supported configurations are the ones the reference instantiates on the hot path:
FullyFusedMLP, 64 neurons, ReLU, 1-2 hidden layers, output None/Sigmoid, Identity encoding.
"""
import math

import torch
from torch import nn

from . import functional as F

_ACT = {"None": 0, "Sigmoid": 1}


def _xavier_flat(shapes, gen) -> torch.Tensor:
    parts = []
    for fan_out, fan_in in shapes:
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        parts.append(((torch.rand(fan_out * fan_in, generator=gen) * 2 - 1) * bound))
    return torch.cat(parts)


class Network(nn.Module):
    """``tcnn.Network(n_input_dims, n_output_dims, network_config)`` (FullyFusedMLP)."""

    def __init__(self, n_input_dims: int, n_output_dims: int, network_config: dict, seed: int = 1337):
        super().__init__()
        if network_config.get("otype", "FullyFusedMLP") != "FullyFusedMLP":
            raise NotImplementedError("only FullyFusedMLP is provided")
        if network_config.get("n_neurons", 64) != 64 or network_config.get("activation", "ReLU") != "ReLU":
            raise NotImplementedError("native FullyFusedMLP: 64 neurons, ReLU (the reference's configuration)")
        n_hidden = int(network_config.get("n_hidden_layers", 1))
        if n_hidden not in (1, 2):
            raise NotImplementedError("native FullyFusedMLP supports 1 or 2 hidden layers")
        if n_input_dims > 32 or n_output_dims > 16:
            raise NotImplementedError("native FullyFusedMLP: n_input_dims <= 32, n_output_dims <= 16")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.n_hidden_mats = n_hidden - 1
        self.out_act = _ACT[network_config.get("output_activation", "None")]
        gen = torch.Generator().manual_seed(seed)
        shapes = [(64, 32)] + [(64, 64)] * self.n_hidden_mats + [(16, 64)]
        self.params = nn.Parameter(_xavier_flat(shapes, gen))
        assert self.params.numel() == 64 * 32 + self.n_hidden_mats * 64 * 64 + 16 * 64

    def half_weights(self) -> torch.Tensor:
        """fp16 copy of ``params`` (what the kernels read), made once per optimizer step instead of once per call (tcnn
        itself casts its fp32 master parameters to fp16 on every step).  Fused optimizers update parameters without
        bumping ``Tensor._version``, hence the optimizer-step counter in the key."""
        from .field_components.deformation_field import _OPTIMIZER_STEPS
        p = self.params
        key = (p._version, p.data_ptr(), _OPTIMIZER_STEPS[0])
        if getattr(self, "_w16_key", None) != key:
            self._w16 = F.f32_to_f16(p.detach())
            self._w16_key = key
        return self._w16

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, n_input_dims] (fp16 or fp32) -> [B, n_output_dims] fp16."""
        x2 = x.reshape(-1, self.n_input_dims)
        if x2.dtype == torch.float16:
            return F.fused_mlp(self.params, self.n_hidden_mats, self.n_output_dims, self.out_act, b=x2)
        return F.fused_mlp(self.params, self.n_hidden_mats, self.n_output_dims, self.out_act, a=x2.float())


class NetworkWithInputEncoding(Network):
    """``tcnn.NetworkWithInputEncoding`` with the Identity encoding the reference uses for mlp_base when the
    HashEnsemble is enabled (nersemble_nerfacto_field.py:123-129)."""

    def __init__(self, n_input_dims: int, n_output_dims: int, encoding_config: dict, network_config: dict,
                 seed: int = 1337):
        if encoding_config.get("otype") != "Identity":
            raise NotImplementedError("native NetworkWithInputEncoding: Identity encoding (use HashEnsemble for grids)")
        super().__init__(n_input_dims, n_output_dims, network_config, seed)


class Encoding(nn.Module):
    """``tcnn.Encoding``: ``Identity`` (direction encoding with spherical_harmonics_degree = 0, the default of every
    shipped NeRSemble config, nersemble_instant_ngp.py:46), ``Frequency`` (the field's ``position_encoding``,
    nersemble_nerfacto_field.py:137-140 -- constructed by the reference, evaluated only under ``use_pred_normals``, which
    no NeRSemble config sets: constructed here too, never evaluated) and ``HashGrid`` with tcnn's parameter layout and
    U(-1e-4, 1e-4) initialisation (the operator the reference's own HashEnsemble instantiates at
    hash_ensemble.py:42-50; the fused ``nersemble_amd`` HashEnsemble is the fast path).

    Like every tcnn module, a parameter-free encoding registers an EMPTY flat ``params``: it is an entry of
    ``field.parameters()`` (hence of the ``fields`` optimizer group's numbering) and a key of the state dict
    (tests/golden/state_manifest.json, generated from the reference's module tree)."""

    def __init__(self, n_input_dims: int, encoding_config: dict, seed: int = 1337):
        super().__init__()
        otype = encoding_config.get("otype")
        self.otype = otype
        self.n_input_dims = n_input_dims
        if otype == "Identity":
            self.n_output_dims = n_input_dims
            self.params = nn.Parameter(torch.empty(0, dtype=torch.float32))
        elif otype == "Frequency":
            self.n_output_dims = n_input_dims * 2 * int(encoding_config["n_frequencies"])
            self.params = nn.Parameter(torch.empty(0, dtype=torch.float32))
        elif otype == "HashGrid":
            if n_input_dims != 3 or encoding_config.get("interpolation", "Linear") != "Linear":
                raise NotImplementedError("native HashGrid: 3-D input, Linear interpolation")
            from . import _lib
            self.f_enc = int(encoding_config.get("n_features_per_level", 2))
            if self.f_enc not in (2, 4, 8):
                raise NotImplementedError("native HashGrid: n_features_per_level in {2, 4, 8}")
            self.geom = _lib.grid_geometry(int(encoding_config.get("n_levels", 16)),
                                           float(encoding_config.get("per_level_scale", 2.0)),
                                           int(encoding_config.get("base_resolution", 16)),
                                           int(encoding_config.get("log2_hashmap_size", 19)))
            self.n_output_dims = self.geom.n_levels * self.f_enc
            gen = torch.Generator().manual_seed(seed)
            self.params = nn.Parameter((torch.rand(self.geom.total_entries * self.f_enc, generator=gen) * 2 - 1) * 1e-4)
        else:
            raise NotImplementedError(f"native Encoding: Identity, Frequency (construction only) and HashGrid (got {otype})")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.otype == "Identity":
            return x.to(torch.float16)
        if self.otype == "Frequency":
            raise NotImplementedError("tcnn Frequency encoding: only reached under use_pred_normals "
                                      "(nersemble_nerfacto_field.py:213-226), which NeRSemble never sets")
        return F.hashgrid_encoding(x, self.params.view(self.geom.total_entries, self.f_enc), self.f_enc, self.geom)
