"""NeRSembleNeRFactoField -- host-side mirror of the reference's fields/nersemble_nerfacto_field.py:30-402 for
the configuration NeRSemble trains (hash ensemble on, Identity direction encoding, no appearance / transient /
semantic / normal heads -- those are dead code in the reference's configs).

Same public surface: ``get_density(ray_samples, window_hash_encodings) -> (density [S,1] fp32, embedding [S,15])``,
``get_outputs(ray_samples, density_embedding) -> {FieldHeadNames.RGB: [S,3] fp32}``, ``forward``, ``density_fn``;
same state-dict names (``hash_ensemble.hash_encodings.{c}.params``, ``mlp_base.params``, ``mlp_head.params``,
``aabb`` ...).  Compute is three native kernels: fused HashEnsemble, mlp_base, mlp_head (which reads the shifted
directions and the 15 geometry features in place -- no torch.cat, no Identity-encoding launch).
"""
import enum
from typing import Dict, Optional, Tuple

import torch
from torch import nn, Tensor

from .. import functional as F
from .. import tcnn
from ..field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig
from ..rays import RaySamples, SceneBox
from ..util.chunker import chunked


class FieldHeadNames(enum.Enum):
    """nerfstudio.field_components.field_heads.FieldHeadNames (the members the path produces)."""
    RGB = "rgb"
    DENSITY = "density"


class _TruncExp(torch.autograd.Function):
    """nerfstudio ``trunc_exp``: forward exp(x), backward g * exp(clamp(x, -15, 15))."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(x.clamp(-15, 15))


trunc_exp = _TruncExp.apply


def shift_directions_for_tcnn(directions: Tensor) -> Tensor:
    return (directions + 1.0) / 2.0


class NeRSembleNeRFactoField(nn.Module):

    def __init__(self, aabb: Tensor, num_images: int, num_layers: int = 2, hidden_dim: int = 64,
                 geo_feat_dim: int = 15, num_levels: int = 16, max_res: int = 2048, log2_hashmap_size: int = 19,
                 num_layers_color: int = 3, hidden_dim_color: int = 64, spatial_distortion=None,
                 use_appearance_embedding: bool = False, spherical_harmonics_degree: int = 0,
                 use_hash_ensemble: bool = True, hash_ensemble_config: Optional[HashEnsembleConfig] = None,
                 max_n_samples_per_batch: int = -1, **unused_nerfacto_kwargs) -> None:
        super().__init__()
        if spatial_distortion is not None or use_appearance_embedding or spherical_harmonics_degree > 0 \
                or not use_hash_ensemble:
            raise NotImplementedError("native field covers the NeRSemble training configuration: no scene "
                                      "contraction, no appearance embedding, SH degree 0, hash ensemble on")
        self.register_buffer("aabb", aabb)
        self.geo_feat_dim = geo_feat_dim
        self.register_buffer("max_res", torch.tensor(max_res))
        self.register_buffer("num_levels", torch.tensor(num_levels))
        self.register_buffer("log2_hashmap_size", torch.tensor(log2_hashmap_size))
        self.num_images = num_images
        self.use_hash_ensemble = use_hash_ensemble
        self.max_n_samples_per_batch = max_n_samples_per_batch
        # set by the model around the sampler's sigma_fn pass: keep hash features / base-MLP outputs for reuse
        self.keep_density_intermediates = False
        self.last_hash_features = self.last_base_out = None
        # evaluation fast path: lookup in the pre-blended grid -> mlp_base -> trunc_exp as ONE launch (csrc/density_fused.hip;
        # bit-identical to the four launches it replaces)
        self.fused_eval_density = True

        self.direction_encoding = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Identity"})
        self.hash_ensemble = HashEnsemble(hash_ensemble_config)
        # (registration order = the reference's, :98-172: it fixes the numbering of the ``fields`` optimizer group)
        self.position_encoding = tcnn.Encoding(n_input_dims=3, encoding_config={"otype": "Frequency", "n_frequencies": 2})
        self.mlp_base = tcnn.NetworkWithInputEncoding(
            n_input_dims=self.hash_ensemble.get_out_dim(), n_output_dims=1 + self.geo_feat_dim,
            encoding_config={"otype": "Identity", "n_dims_to_encode": self.hash_ensemble.get_out_dim()},
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None",
                            "n_neurons": hidden_dim, "n_hidden_layers": num_layers - 1}, seed=1338)
        self.mlp_head = tcnn.Network(
            n_input_dims=self.direction_encoding.n_output_dims + self.geo_feat_dim, n_output_dims=3,
            network_config={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "Sigmoid",
                            "n_neurons": hidden_dim_color, "n_hidden_layers": num_layers_color - 1}, seed=1339)

    # ---- density -----------------------------------------------------------------------------------
    def density_fn(self, positions: Tensor, times: Optional[Tensor] = None,
                   window_hash_encodings: Optional[float] = None, time_codes: Optional[Tensor] = None,
                   time_code_index: Optional[Tensor] = None, preblended_table: Optional[Tensor] = None) -> Tensor:
        """Occupancy / sigma_fn entry (nersemble_nerfacto_field.py:228-248)."""
        del times
        # the reference wraps the positions into dummy Frustums (starts = ends = 0) whose get_positions() returns them
        # unchanged (:236-246); the wrapper is skipped here
        density, _ = self._density_from_positions(positions, None, {"time_codes": time_codes,
                                                                    "time_code_index": time_code_index,
                                                                    "preblended_table": preblended_table},
                                                  window_hash_encodings)
        return density

    def _aabb6(self):
        """Host copy of ``aabb`` for the kernels' by-value box argument; follows ``load_state_dict`` / ``.to()``."""
        key = (self.aabb.data_ptr(), self.aabb._version)
        if getattr(self, "_aabb6_key", None) != key:
            import ctypes
            self._aabb6_cache = (ctypes.c_float * 6)(*[float(v) for v in self.aabb.detach().flatten().tolist()])
            self._aabb6_key = key
        return self._aabb6_cache

    def get_density(self, ray_samples: RaySamples, window_hash_encodings: Optional[float]) -> Tuple[Tensor, Tensor]:
        fr = ray_samples.frustums
        if fr.origins.is_cuda:
            base = getattr(fr, "base_positions", None)            # left by the deformation field for these samples
            if base is None or base.shape[0] != fr.origins.shape[0]:
                base = F.sample_positions(fr.origins, fr.directions, fr.starts, fr.ends)
            # (offsets are added in the normalisation kernel below)
            return self._density_from_positions(base, fr.offsets, ray_samples.metadata or {}, window_hash_encodings)
        return self._density_from_positions(fr.get_positions(), None, ray_samples.metadata or {}, window_hash_encodings)

    def _density_from_positions(self, positions_world: Tensor, offsets: Optional[Tensor], md: Dict,
                                window_hash_encodings: Optional[float]) -> Tuple[Tensor, Tensor]:
        """positions (+ offsets) -> scene-box normalisation, selector (:257, :268-269) -> HashEnsemble -> mlp_base ->
        trunc_exp density (:286-293)."""
        blended = md.get("preblended_table")           # eval fast path: one time code for every sample of the image
        if (blended is not None and positions_world.is_cuda and self.fused_eval_density and not torch.is_grad_enabled()
                and md.get("precomputed_base_out") is None and self.hash_ensemble.geom.n_levels == 16
                and self.mlp_base.n_output_dims == 16 and self.mlp_base.n_hidden_mats in (0, 1)):
            # (what nsx_density_fused_fwd is built for -- 16 levels, a 64-wide mlp_base [tcnn.Network admits no other width]
            # with 0 or 1 hidden matrices, 16 outputs; any other model takes the four launches below)
            return self._density_fused(positions_world, offsets, blended)
        if positions_world.is_cuda:
            positions, selector_all = F.normalised_positions(positions_world, offsets, self._aabb6())
        else:
            if offsets is not None:
                positions_world = positions_world + offsets
            positions = SceneBox.get_normalized_positions(positions_world, self.aabb)
            selector_all = None
        max_chunk = len(positions) if self.max_n_samples_per_batch == -1 else self.max_n_samples_per_batch
        if self.hash_ensemble.level_parallel is not None and self.hash_ensemble.training:
            max_chunk = len(positions)         # (every rank issues ONE exchange per pass, whatever its sample count)
        time_codes = md.get("time_codes")
        code_index = md.get("time_code_index")        # native extension: time_codes is a [T,H] table
        pre_feats = md.get("precomputed_hash_features")       # from the step's sigma_fn pass (same samples, same params)
        pre_base = md.get("precomputed_base_out")
        densities, base_outs, feats_all = [], [], []
        if code_index is None:
            chunks = ((p, c, None, f, b, sl) for p, c, f, b, sl in chunked(max(max_chunk, 1), positions, time_codes,
                                                                          pre_feats, pre_base, selector_all))
        else:
            chunks = ((p, time_codes, ci, f, b, sl) for p, ci, f, b, sl in chunked(max(max_chunk, 1), positions, code_index,
                                                                                  pre_feats, pre_base, selector_all))
        for pos_c, codes_c, idx_c, pre_f, pre_b, sel_c in chunks:
            if sel_c is None:
                # tcnn needs inputs in [0,1): zero the samples outside the scene box (:268-269)
                selector = ((pos_c > 0.0) & (pos_c < 1.0)).all(dim=-1)
                pos_c = pos_c * selector[..., None]
            if blended is not None and pre_b is not None and not torch.is_grad_enabled():
                feats = None                 # (the sigma_fn pass's mlp_base rows are reused: nobody reads the features)
            elif blended is not None:
                feats = self.hash_ensemble.forward_preblended(pos_c.view(-1, 3), blended)
            else:
                feats = self.hash_ensemble(pos_c.view(-1, 3), conditioning_code=codes_c,
                                           window_hash_encodings=window_hash_encodings, code_index=idx_c,
                                           precomputed=pre_f)
            if feats is None:
                h = pre_b.detach().view(*pos_c.shape[:-1], -1)
            else:
                h = F.fused_mlp(self.mlp_base.params, self.mlp_base.n_hidden_mats, self.mlp_base.n_output_dims,
                                self.mlp_base.out_act, b=feats, precomputed=pre_b,
                                w16=self.mlp_base.half_weights()).view(*pos_c.shape[:-1], -1)   # [S,16] fp16
            if self.keep_density_intermediates and feats is not None:
                feats_all.append(feats)
            if sel_c is not None:
                density = F.density_from_base(h, sel_c)
            else:
                density_before_activation = h[..., :1]
                density = trunc_exp(density_before_activation.to(pos_c)) * selector[..., None]
            densities.append(density)
            base_outs.append(h)
        density = densities[0] if len(densities) == 1 else torch.cat(densities, dim=0)
        base_out = base_outs[0] if len(base_outs) == 1 else torch.cat(base_outs, dim=0)
        self._base_out = base_out                      # full [S,16] tensor for the fused head read
        if self.keep_density_intermediates:
            self.last_hash_features = None if not feats_all else (feats_all[0] if len(feats_all) == 1
                                                                  else torch.cat(feats_all, dim=0))
            self.last_base_out = base_out
        return density, base_out[..., 1:]

    def _density_fused(self, positions_world: Tensor, offsets: Optional[Tensor], blended: Tensor) -> Tuple[Tensor, Tensor]:
        """``_density_from_positions`` on the pre-blended grid without gradients: one launch per chunk
        (``nsx_density_fused_fwd``); the hash features are never materialised."""
        # (round 6: ONE launch for the whole pass.  The reference's ``max_n_samples_per_batch`` -- evaluate_nersemble.py:133-145
        # walks the field in 2^20-sample pieces -- bounds the memory of activations kept for autograd; this route runs
        # without gradients and keeps 34 bytes per sample, so the 27 pieces of an evaluation image were 27 launches and two
        # ``cat`` copies of their outputs for nothing; the result does not depend on the chunking, tests/test_full_size_gpu.py)
        max_chunk = len(positions_world)
        w16, nh, geom, aabb6 = self.mlp_base.half_weights(), self.mlp_base.n_hidden_mats, self.hash_ensemble.geom, self._aabb6()
        densities, base_outs = [], []
        for pos_c, off_c in chunked(max(max_chunk, 1), positions_world, offsets):
            d, b = F.density_fused(pos_c, off_c, aabb6, blended, geom, w16, nh)
            densities.append(d)
            base_outs.append(b)
        density = densities[0] if len(densities) == 1 else torch.cat(densities, dim=0)
        base_out = base_outs[0] if len(base_outs) == 1 else torch.cat(base_outs, dim=0)
        self._base_out = base_out
        if self.keep_density_intermediates:
            self.last_hash_features, self.last_base_out = None, base_out
        return density, base_out[..., 1:]

    # ---- colour ------------------------------------------------------------------------------------
    def get_outputs(self, ray_samples: RaySamples, density_embedding: Optional[Tensor] = None,
                    base_out: Optional[Tensor] = None) -> Dict[FieldHeadNames, Tensor]:
        assert density_embedding is not None or base_out is not None
        if ray_samples.camera_indices is None:
            raise AttributeError("Camera indices are not provided.")
        directions = ray_samples.frustums.directions.reshape(-1, 3)
        max_chunk = len(ray_samples) if self.max_n_samples_per_batch == -1 else self.max_n_samples_per_batch
        if not torch.is_grad_enabled() and directions.is_cuda:
            max_chunk = len(ray_samples)                  # (no activations are kept: one launch, see _density_fused)
        if base_out is None:
            # generic path (reference signature): rebuild a [S,16] tensor with an empty density column
            base_out = torch.cat([torch.zeros_like(density_embedding[..., :1]), density_embedding], dim=-1)
        rgbs = []
        for dir_c, base_c in chunked(max(max_chunk, 1), directions, base_out):
            # mlp_head input = [(d+1)/2 (3), geo features (15)], read in place by the kernel (:313, :371-377)
            rgb = F.fused_mlp(self.mlp_head.params, self.mlp_head.n_hidden_mats, 3, self.mlp_head.out_act,
                              a=dir_c, a_mul=0.5, a_add=0.5, b=base_c, b_off=1, b_dim=self.geo_feat_dim,
                              w16=self.mlp_head.half_weights())
            rgbs.append(rgb.to(directions))
        return {FieldHeadNames.RGB: torch.cat(rgbs, dim=0)}

    def forward(self, ray_samples: RaySamples, compute_normals: bool = False,
                window_hash_encodings: Optional[float] = None) -> Dict[FieldHeadNames, Tensor]:
        if compute_normals:
            raise NotImplementedError("normals are not part of the NeRSemble training path")
        density, density_embedding = self.get_density(ray_samples, window_hash_encodings=window_hash_encodings)
        field_outputs = self.get_outputs(ray_samples, density_embedding=density_embedding, base_out=self._base_out)
        self._base_out = None
        field_outputs[FieldHeadNames.DENSITY] = density
        return field_outputs
