"""SE(3) deformation field -- mirror of the reference's field_components/deformation_field.py:15-166.

``SE3WarpingField``: windowed positional encoding (45) + warp code -> 6x128 MLP with a skip into layer 4
(nerfstudio ``MLP`` semantics) -> two linear heads (rotation r, translation v) -> screw axis
[v, r] -> se3_exp_map -> homogeneous transform of the NORMALISED position, NaN fallback.
``SE3DeformationField.compute_offsets`` returns ``warped - normalised`` (deformation_field.py:148-166).
Parameter names follow the reference's state dict (``se3_field.mlp_stem.layers.{i}.weight`` ...).
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch import nn

import ctypes

from .. import functional as F
from ..rays import RaySamples, SceneBox
from ..util.chunker import chunked
from ..util.se3 import se3_exp_map
from .windowed_nerf_encoding import WindowedNeRFEncoding
from torch.optim.optimizer import register_optimizer_step_post_hook

# any torch.optim step anywhere invalidates cached parameter packs (SE3DeformationField.packed_params)
_OPTIMIZER_STEPS = [0]


def _count_optimizer_step(*_args, **_kwargs):
    _OPTIMIZER_STEPS[0] += 1


register_optimizer_step_post_hook(_count_optimizer_step)



@dataclass
class SE3DeformationFieldConfig:
    n_freq_pos = 7
    warp_code_dim: int = 8
    mlp_num_layers: int = 6
    mlp_layer_width: int = 128
    skip_connections: Tuple[int] = (4,)


class MLP(nn.Module):
    """nerfstudio 0.3.1 ``field_components.MLP``: ``num_layers`` Linear layers, layer i in ``skip_connections``
    takes cat([input, x]); activation after all but the last layer; ``out_activation`` after the last."""

    def __init__(self, in_dim, num_layers, layer_width, out_dim=None, skip_connections=None,
                 activation=nn.ReLU(), out_activation=None):
        super().__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim if out_dim is not None else layer_width
        self.num_layers, self.layer_width = num_layers, layer_width
        self._skip = set(skip_connections) if skip_connections else set()
        self.activation, self.out_activation = activation, out_activation
        layers = []
        if num_layers == 1:
            layers.append(nn.Linear(in_dim, self.out_dim))
        else:
            for i in range(num_layers - 1):
                if i == 0:
                    assert i not in self._skip, "Skip connection at layer 0 doesn't make sense."
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


class SE3WarpingField(nn.Module):
    def __init__(self, config: SE3DeformationFieldConfig) -> None:
        super().__init__()
        self.position_encoding = WindowedNeRFEncoding(in_dim=3, num_frequencies=config.n_freq_pos, min_freq_exp=0.0,
                                                      max_freq_exp=config.n_freq_pos - 1, include_input=True)
        in_dim = self.position_encoding.get_out_dim() + config.warp_code_dim
        self.mlp_stem = MLP(in_dim=in_dim, out_dim=config.mlp_layer_width, num_layers=config.mlp_num_layers,
                            layer_width=config.mlp_layer_width, skip_connections=config.skip_connections,
                            out_activation=nn.ReLU())
        self.mlp_r = MLP(in_dim=config.mlp_layer_width, out_dim=3, num_layers=1, layer_width=config.mlp_layer_width)
        self.mlp_v = MLP(in_dim=config.mlp_layer_width, out_dim=3, num_layers=1, layer_width=config.mlp_layer_width)
        # start close to the identity transformation (deformation_field.py:71-75)
        for head in (self.mlp_r, self.mlp_v):
            nn.init.uniform_(head.layers[-1].weight, a=-1e-5, b=1e-5)
            nn.init.zeros_(head.layers[-1].bias)

    def get_transform(self, positions, warp_code, windows_param=None):
        encoded_xyz = self.position_encoding(positions, windows_param=windows_param)
        feat = self.mlp_stem(torch.cat([encoded_xyz, warp_code], dim=-1))
        r = self.mlp_r(feat).reshape(-1, 3)
        v = self.mlp_v(feat).reshape(-1, 3)
        screw_axis = torch.concat([v, r], dim=-1).to(positions.dtype)
        return se3_exp_map(screw_axis).permute(0, 2, 1)

    def apply_transform(self, positions, transforms):
        p = positions.reshape(-1, 3)
        ph = torch.concat([p, torch.ones_like(p[..., :1])], dim=-1)
        wh = (transforms @ ph.unsqueeze(-1)).squeeze(-1)
        warped = (wh[..., :3] / wh[..., -1:]).to(positions.dtype)
        warped = torch.where(warped.isnan(), p, warped)          # NaN deformation -> keep the original point
        return warped.reshape(*positions.shape[: positions.ndim - 1], 3)

    def forward(self, positions, warp_code=None, windows_param=None):
        if warp_code is None:
            return None
        return self.apply_transform(positions, self.get_transform(positions, warp_code, windows_param))


class SE3DeformationField(nn.Module):
    def __init__(self, aabb: torch.Tensor, deformation_field_config: SE3DeformationFieldConfig,
                 max_n_samples_per_batch: int = -1):
        super().__init__()
        self.aabb = nn.Parameter(aabb, requires_grad=False)
        self.se3_field = SE3WarpingField(deformation_field_config)
        self.max_n_samples_per_batch = max_n_samples_per_batch

    def forward(self, ray_samples: RaySamples, warp_code: Optional[torch.Tensor] = None,
                windows_param: Optional[float] = None, code_index: Optional[torch.Tensor] = None,
                precomputed_offsets: Optional[torch.Tensor] = None) -> RaySamples:
        assert ray_samples.frustums.offsets is None, "ray samples have already been warped"
        fr = ray_samples.frustums
        positions = F.sample_positions(fr.origins, fr.directions, fr.starts, fr.ends) if fr.origins.is_cuda \
            else fr.get_positions()
        fr.base_positions = positions          # un-warped sample positions: the field's density pass starts from them
        ray_samples.frustums.set_offsets(self.compute_offsets(positions, warp_code, windows_param, code_index,
                                                              precomputed_offsets))
        return ray_samples

    # ---- native path -------------------------------------------------------------------------------
    def native_supported(self) -> bool:
        st = self.se3_field.mlp_stem
        return (len(st.layers) == 6 and st.layer_width == 128 and st._skip == {4} and st.in_dim == 173)

    def ordered_params(self):
        """The 16 nn.Linear tensors in include/nsx.h order."""
        L = self.se3_field
        parts = []
        for lyr in L.mlp_stem.layers:
            parts += [lyr.weight, lyr.bias]
        parts += [L.mlp_r.layers[0].weight, L.mlp_r.layers[0].bias, L.mlp_v.layers[0].weight, L.mlp_v.layers[0].bias]
        return parts

    def flat_params(self) -> torch.Tensor:
        """All parameters as one flat fp32 tensor in include/nsx.h order."""
        return torch.cat([p.float().reshape(-1) for p in self.ordered_params()])

    def packed_params(self) -> torch.Tensor:
        """MFMA weight fragments of the current parameter values, packed once per optimizer step (every pass of a step
        -- occupancy update, sigma_fn pass, main pass -- shares them)."""
        params = self.ordered_params()
        # fused optimizers update parameters without touching Tensor._version: count optimizer steps as well
        key = (_OPTIMIZER_STEPS[0],) + tuple((p._version, p.data_ptr()) for p in params)
        if getattr(self, "_packed_key", None) != key:
            with torch.no_grad():
                self._packed = F.deform_pack_tensors(params)
            self._packed_key = key
        return self._packed

    def _aabb6(self):
        """Host copy of ``aabb`` for the kernels' by-value box argument; follows ``load_state_dict`` / ``.to()``."""
        key = (self.aabb.data_ptr(), self.aabb._version)
        if getattr(self, "_aabb6_key", None) != key:
            self._aabb6_cache = (ctypes.c_float * 6)(*[float(v) for v in self.aabb.detach().flatten().tolist()])
            self._aabb6_key = key
        return self._aabb6_cache

    def compute_offsets(self, positions, warp_code=None, windows_param=None, code_index=None, precomputed=None):
        """``code_index`` (native extension): ``warp_code`` is a code TABLE and ``code_index[s]`` the row of sample s.
        ``precomputed``: offsets of the same samples from the step's no-grad sigma_fn pass (forward is skipped, the
        backward still runs the native kernel)."""
        if positions.is_cuda:
            if not self.native_supported():
                raise NotImplementedError("native deformation kernel: 6x128 MLP, skip at 4, warp_code_dim 128 "
                                          "(the configuration of train_nersemble.py:84-91)")
            if warp_code is None:
                return None
            max_chunk = len(positions) if self.max_n_samples_per_batch == -1 else self.max_n_samples_per_batch
            if not torch.is_grad_enabled():
                max_chunk = len(positions)            # (nothing is kept for a backward: one launch for the whole pass)
            params, packed = self.ordered_params(), self.packed_params()
            outs = []
            if code_index is None and warp_code.shape[0] == 1 and len(positions) != 1:
                # ONE code for every sample (native extension: an evaluation image's timestep) -- a one-row table
                for pos_c, pre_c in chunked(max(max_chunk, 1), positions, precomputed):
                    outs.append(F.deform_offsets(params, packed, pos_c, warp_code, self._aabb6(), windows_param, None, pre_c))
            elif code_index is None:
                for pos_c, code_c, pre_c in chunked(max(max_chunk, 1), positions, warp_code, precomputed):
                    outs.append(F.deform_offsets(params, packed, pos_c, code_c, self._aabb6(), windows_param, None, pre_c))
            else:
                for pos_c, idx_c, pre_c in chunked(max(max_chunk, 1), positions, code_index, precomputed):
                    outs.append(F.deform_offsets(params, packed, pos_c, warp_code, self._aabb6(), windows_param, idx_c,
                                                 pre_c))
            if len(outs) == 1:
                return outs[0]
            return torch.cat(outs, dim=0) if outs else positions.new_zeros((0, 3))
        # CPU tensors: plain torch restatement (used to pin the glue against the reference's goldens)
        if code_index is not None:
            warp_code = warp_code[code_index.long()]
        elif warp_code is not None and warp_code.shape[0] == 1 and len(positions) != 1:
            warp_code = warp_code.expand(len(positions), -1)
        max_chunk = len(positions) if self.max_n_samples_per_batch == -1 else self.max_n_samples_per_batch
        offsets = []
        for pos_c, code_c in chunked(max(max_chunk, 1), positions, warp_code):
            pos_n = SceneBox.get_normalized_positions(pos_c, self.aabb)
            warped = self.se3_field(pos_n, warp_code=code_c, windows_param=windows_param)
            offsets.append(warped - pos_n)
        return torch.cat(offsets, dim=0) if offsets else positions.new_zeros((0, 3))
