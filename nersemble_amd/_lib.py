"""ctypes binding of csrc/libnsx.so (C ABI: include/nsx.h).  Fails loudly when the library is absent."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "csrc", "libnsx.so")
NSX_MAX_LEVELS = 32
NSX_MAX_SLOTS = 64
NSX_MAX_GATHER = 8
NSX_MAX_ADAM_SLOTS = 192
NSX_OPT_ADAM_BLOCKS_PER_CU, NSX_OPT_MLP_BWD_HALF_BLOCKS_PER_CU, NSX_OPT_MLP_BWD0_HALF_BLOCKS_PER_CU = 0, 1, 2
NSX_OPT_LP_ONE_LAUNCH = 3
NSX_COMM_ID_BYTES = 128

c_void_p, c_int, c_int64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GridGeom(C.Structure):
    """Mirror of ``nsx_grid_geom`` (include/nsx.h)."""
    _fields_ = [
        ("n_levels", C.c_int32),
        ("log2_hashmap_size", C.c_int32),
        ("base_resolution", C.c_int32),
        ("per_level_scale", C.c_float),
        ("scale", C.c_float * NSX_MAX_LEVELS),
        ("res", C.c_uint32 * NSX_MAX_LEVELS),
        ("size", C.c_uint32 * NSX_MAX_LEVELS),
        ("offset", C.c_uint32 * (NSX_MAX_LEVELS + 1)),
        ("hashed", C.c_uint32 * NSX_MAX_LEVELS),
    ]

    @property
    def total_entries(self) -> int:
        return int(self.offset[self.n_levels])


_GEOM_P = C.POINTER(GridGeom)
NSX_MAX_TENSORS = 64
NSX_MAX_GROUPS = 8
NSX_MAX_SCALE_MIRRORS = 8


class TensorRef(C.Structure):
    """Mirror of ``nsx_tensor_ref``."""
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("group", C.c_int32), ("step", C.c_int32)]


class AdamGroup(C.Structure):
    """Mirror of ``nsx_adam_group``."""
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step", C.c_int64)]


# name -> (restype, argtypes); must list every symbol include/nsx.h declares (tests check this)
SIGNATURES = {
    "nsx_version": (c_int, []),
    "nsx_last_error": (C.c_char_p, []),
    "nsx_grid_geometry": (c_int, [c_int, c_float, c_int, c_int, _GEOM_P]),
    "nsx_padded_grids": (c_int, [c_int]),
    "nsx_set_option": (c_int, [c_int, c_int]),
    "nsx_get_option": (c_int, [c_int]),
    "nsx_check_code_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "nsx_tables_from_tcnn": (c_int, [c_void_p, c_int, _GEOM_P, c_void_p, c_void_p, c_void_p]),
    "nsx_tables_to_tcnn": (c_int, [c_void_p, c_int, _GEOM_P, c_void_p, c_void_p]),
    "nsx_hash_ensemble_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_hash_ensemble_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_int64, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_hash_ensemble_bwd_factored": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_int64, c_int,
                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p]),
    "nsx_hash_codesum_scratch_floats": (c_int64, [c_int, c_int]),
    "nsx_hash_ensemble_bwd_codesum": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_int64, c_int,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_void_p]),
    "nsx_hash_ensemble_bwd_scatter": (c_int, [c_void_p, c_int64, _GEOM_P, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_int, c_void_p, c_void_p]),
    "nsx_tables_preblend": (c_int, [c_void_p, c_int, _GEOM_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_hash_grad_expand": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_int,
                                     c_void_p]),
    "nsx_hashgrid_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_void_p]),
    "nsx_hashgrid_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_density_fused_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, _GEOM_P, c_void_p, c_int, c_void_p,
                                      c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_mlp_param_count": (c_int, [c_int]),
    "nsx_mlp_fwd": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_int, c_float, c_float, c_void_p, c_int64,
                            c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "nsx_mlp_bwd": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_int, c_float, c_float, c_void_p, c_int64,
                            c_int, c_int, c_int, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p]),
    "nsx_f32_to_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nsx_sample_positions": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_generate_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_gather_rows": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "nsx_gather_rows_via": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                    c_void_p]),
    "nsx_compact_rays": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_render_visibility": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_float, c_float,
                                      c_void_p, c_void_p]),
    "nsx_normalise_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_density_fwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_density_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_deform_param_count": (c_int, []),
    "nsx_deform_pack_bytes": (c_int64, []),
    "nsx_deform_scratch_bytes": (c_int64, [c_int64]),
    "nsx_deform_pack": (c_int, [c_void_p, c_void_p, c_void_p]),
    "nsx_deform_pack_tensors": (c_int, [C.POINTER(c_void_p), c_void_p, c_void_p]),
    "nsx_deform_fwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "nsx_deform_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_march_count": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                c_void_p, c_void_p]),
    "nsx_march_count_stash": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                      c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "nsx_march_fill_from_stash": (c_int, [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    "nsx_pack_info": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_copy_to_host_async": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nsx_march_fill": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_ray_histogram": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_render_weights_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_float, c_float, c_void_p, c_void_p]),
    "nsx_render_weights_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_accumulate_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "nsx_accumulate_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p]),
    "nsx_composite_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_composite_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_composite_fwd_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_composite_bwd_h": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_sample_losses_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_int64, c_void_p,
                                      c_void_p]),
    "nsx_sample_losses_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_int64, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_ray_losses_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
    "nsx_ray_losses_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float,
                                   c_float, c_float, c_float, c_float, c_float, c_int64, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_distloss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_float, c_void_p,
                             c_void_p, c_void_p]),
    "nsx_check_finite": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nsx_check_finite_f16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "nsx_adam_dense_f16grad": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                                       c_float, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_hash_grad_expand_f16": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_float,
                                         c_int, c_void_p]),
    "nsx_hash_grad_expand_f16_bucket": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p,
                                                c_float, c_int, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "nsx_hash_grad_expand_f16_bucket_width": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p,
                                                      c_float, c_int, c_int64, c_int64, c_int64, c_int, c_int, c_void_p,
                                                      c_void_p]),
    "nsx_hash_grad_expand_f16_bucket_width_consume": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P,
                                                              c_void_p, c_float, c_int, c_int64, c_int64, c_int64, c_int,
                                                              c_int, c_void_p, c_void_p]),
    "nsx_adam_dense_f16grad_width": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                             c_float, c_float, c_float, c_float, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_tables_unpack_width": (c_int, [c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "nsx_deform_terms_floats": (c_int64, [c_int]),
    "nsx_deform_fwd_rows": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_grad_scaler_update": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                       c_float, c_float, c_int, c_int, c_void_p]),
    "nsx_adam_hash_factored": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64, c_void_p,
                                       c_void_p, c_void_p]),
    "nsx_adam_hash_factored_consume": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int, _GEOM_P, c_void_p, c_void_p,
                                       c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_int64, c_void_p,
                                       c_void_p, c_void_p]),
    "nsx_adam_dense": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_float,
                               c_float, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_hash_indices": (c_int, [c_void_p, c_int64, _GEOM_P, c_void_p, c_void_p]),
    "nsx_multi_unscale_check": (c_int, [C.POINTER(TensorRef), c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "nsx_multi_adam": (c_int, [C.POINTER(TensorRef), c_int, C.POINTER(AdamGroup), c_int, c_void_p, c_void_p]),
    "nsx_bucket_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nsx_bucket_unpack": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "nsx_multi_adam_present": (c_int, [C.POINTER(TensorRef), c_int, C.POINTER(AdamGroup), c_int, c_void_p, c_void_p,
                                       c_void_p, c_int, c_void_p]),
    "nsx_occ_scratch_bytes": (c_int64, [c_int64]),
    "nsx_occ_compact": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_compact_mask": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_occ_sample_cells": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int64, C.c_uint64, c_int64, c_int, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_occ_update": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p,
                               c_void_p, c_void_p]),
    # training-step drivers (structs: see step_struct below)
    "nsx_step_plan_make": (c_int, [c_int64, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "nsx_step_sample_run": (c_int, [c_void_p, c_void_p]),
    "nsx_step_main_fwd": (c_int, [c_void_p, c_void_p]),
    "nsx_step_main_bwd": (c_int, [c_void_p, c_int, c_void_p]),
    "nsx_step_profile": (c_int, [c_int, c_int]),
    "nsx_step_profile_count": (c_int, []),
    "nsx_step_profile_get": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "nsx_step_profile_reset": (c_int, []),
    "nsx_step_sizeof": (c_int64, [c_int]),
    "nsx_step_echo": (c_int, [c_int, c_void_p, c_void_p, c_int]),
    # level-parallel exchange (csrc/level_parallel.hip; struct nsx_lp_layout through step_struct)
    "nsx_lp_layout_make": (c_int, [c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "nsx_lp_sizeof": (c_int64, []),
    "nsx_lp_fwd_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p,
                                c_void_p]),
    "nsx_lp_fwd_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _GEOM_P, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "nsx_lp_fwd_unpack": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_lp_bwd_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "nsx_lp_bwd_run": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _GEOM_P, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nsx_lp_bwd_unpack": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    # collectives issued by the library (csrc/comm.hip)
    "nsx_comm_library": (c_int, [C.c_char_p]),
    "nsx_comm_unique_id": (c_int, [c_void_p]),
    "nsx_comm_create": (c_int, [c_void_p, c_int, c_int, C.POINTER(c_void_p)]),
    "nsx_comm_destroy": (c_int, [c_void_p]),
    "nsx_comm_world_size": (c_int, [c_void_p]),
    "nsx_comm_rank": (c_int, [c_void_p]),
    "nsx_comm_all_reduce_sum": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "nsx_lp_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, _GEOM_P, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p]),
    "nsx_lp_backward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, _GEOM_P, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "nsx.h")
_STEP_STRUCTS = {}


def step_struct(name: str):
    """ctypes mirror of one of the training-step driver structs (``nsx_step_plan`` / ``nsx_step_sample`` /
    ``nsx_step_main``), built from its declaration in include/nsx.h -- one field per line there -- so that the two cannot
    drift apart: pointers -> c_void_p, int64_t -> c_int64, int32_t -> c_int32, float -> c_float, ``float x[n]`` -> array.
    tests/test_boundary.py holds ``ctypes.sizeof`` and a field-by-field echo against the compiled library."""
    cls = _STEP_STRUCTS.get(name)
    if cls is not None:
        return cls
    import re
    with open(HEADER_PATH) as f:
        text = f.read()
    m = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", text, re.S)
    if m is None:
        raise RuntimeError(f"{name} not found in {HEADER_PATH}")
    fields = []
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)      # comments may span lines
    for line in body.splitlines():
        line = line.strip()
        if not line:
            continue
        d = re.match(r"^(const\s+)?([A-Za-z_0-9]+)\s*(\*?)\s*([A-Za-z_0-9]+)(\[(\d+)\])?;$", line)
        if d is None:
            raise RuntimeError(f"{name}: cannot parse field line {line!r}")
        ctype, star, fname, n = d.group(2), d.group(3), d.group(4), d.group(6)
        if star:
            t = c_void_p
        else:
            t = {"int64_t": c_int64, "int32_t": C.c_int32, "int": C.c_int32, "float": c_float}[ctype]
        fields.append((fname, t * int(n) if n else t))
    cls = type(name, (C.Structure,), {"_fields_": fields, "__doc__": f"Mirror of ``{name}`` (include/nsx.h)."})
    _STEP_STRUCTS[name] = cls
    return cls


_lib = None


class KernelProfiler:
    """Optional HIP-event timing of every native call (used by bench.py for the live roofline numbers).
    Events are recorded on torch's current stream -- the stream the kernels are enqueued on."""

    def __init__(self):
        self.enabled = False
        self.watch = None          # optional set of entry-point names to time (None = all)
        self.records = []          # (name, start_event, end_event, int_args)
        self.alias = {}            # entry point -> the name it is booked under (variants of one kernel)
        self._pool = []
        self.tag = None            # set by the caller (bench.py: index of the timed step); stored with every record
        self.counted_capacity = None   # capacity of the active ``device_count`` scope, if any
        self.tags = []             # per record: (tag, counted) -- counted: the call's row count is the scope's capacity,
        #                            the kernel processed only the device-side count of that step

    def reset(self):
        self.records = []
        self.tags = []

    def prewarm(self, n: int):
        """Create (and record once) n events up front: the first record of an event allocates it in the runtime,
        which must not happen inside a timed region."""
        for _ in range(n):
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._pool.append(e)

    def event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def collect_native(self):
        """Append the records the native step drivers took of their own kernel calls (csrc/step.hip,
        nsx_step_profile*): same shape as the per-call records -- (entry-point name, span, integer arguments in the
        positions bench.py's kernel_model reads) + (tag, counted).  Call after torch.cuda.synchronize()."""
        L = lib()
        n = int(L._h.nsx_step_profile_count())
        name = C.create_string_buffer(64)
        ms, rows, info = C.c_float(), C.c_int64(), (C.c_int32 * 4)()
        for i in range(n):
            check(L._h.nsx_step_profile_get(i, name, 64, C.byref(ms), C.byref(rows), info), "nsx_step_profile_get")
            nm = name.value.decode()
            H, n_slots, counted, tag = int(info[0]), int(info[1]), bool(info[2]), int(info[3])
            ints = [H, int(rows.value)] if nm.startswith("nsx_mlp_") else [int(rows.value), H, 0, n_slots]
            self.records.append((self.alias.get(nm, nm), _Span(0.0), _Span(float(ms.value)), ints))
            self.tags.append((tag if tag >= 0 else None, counted))
        L._h.nsx_step_profile_reset()
        return n

    def summary(self):
        """name -> dict(calls, total_ms, avg_ms, samples); call after torch.cuda.synchronize()."""
        out = {}
        for name, s, e, ints in self.records:
            d = out.setdefault(name, {"calls": 0, "total_ms": 0.0})
            d["calls"] += 1
            d["total_ms"] += s.elapsed_time(e)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / max(d["calls"], 1)
        return out


class _Span:
    """A duration measured elsewhere, shaped like a pair of events (``start.elapsed_time(end)``)."""

    def __init__(self, ms: float):
        self.ms = ms

    def elapsed_time(self, other) -> float:
        return other.ms - self.ms


profiler = KernelProfiler()


class _LibProxy:
    def __init__(self, handle):
        self._h = handle

    def __getattr__(self, name):
        fn = getattr(self._h, name)
        if not profiler.enabled or not name.startswith("nsx_") or (profiler.watch is not None
                                                                   and name not in profiler.watch):
            return fn

        def timed(*args):
            s, e = profiler.event(), profiler.event()
            s.record()
            rc = fn(*args)
            e.record()
            ints = [a for a in args if isinstance(a, int)]
            profiler.records.append((profiler.alias.get(name, name), s, e, ints))
            profiler.tags.append((profiler.tag, profiler.counted_capacity is not None and bool(ints)
                                  and ints[0] == profiler.counted_capacity))
            return rc
        return timed


def lib():
    """Loads libnsx.so.  No fallback: a missing library is an error (build with __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"nersemble_amd: native library {SO_PATH} not found. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
        handle = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = _LibProxy(handle)
    return _lib


_COUNT_SCOPE = None          # (n_dev tensor, capacity) of the active ``device_count`` block, host-side Python state only


class device_count:
    """``with device_count(n_dev, capacity): ...`` -- Python-side convenience over the C ABI's explicit ``n_device``
    argument (include/nsx.h, "Device-side element counts"): the wrappers of this package ask ``ndev(n)`` for the pointer to
    hand to a per-sample entry point, and get ``n_dev`` when the call's row count equals ``capacity`` of the enclosing
    block (the arrays of ONE sample set: marched capacity, valid rows counted on the device).  The native library itself
    holds no such state any more; callers that prefer to be explicit pass the pointer themselves (engine/fused_pass.py).
    ``n_dev``: int64 device tensor with one element, or None (then the block runs unchanged)."""

    def __init__(self, n_dev, capacity: int):
        self.n_dev, self.capacity = n_dev, int(capacity)
        self._outer = None

    def __enter__(self):
        global _COUNT_SCOPE
        if self.n_dev is not None:
            if self.n_dev.dtype != torch.int64 or self.n_dev.numel() != 1:
                raise RuntimeError("device_count: n_dev must be one int64 on the device")
            if _COUNT_SCOPE is not None:
                raise RuntimeError("device_count: a device count is already attached (blocks do not nest)")
            _COUNT_SCOPE = (self.n_dev, self.capacity)
            profiler.counted_capacity = self.capacity
        return self

    def __exit__(self, *exc):
        global _COUNT_SCOPE
        if self.n_dev is not None:
            _COUNT_SCOPE = None
            profiler.counted_capacity = None
        return False


def ndev(n: int) -> c_void_p:
    """The ``n_device`` argument for a per-sample native call on ``n`` rows: the active ``device_count`` block's pointer
    if ``n`` is its capacity, else NULL."""
    sc = _COUNT_SCOPE
    if sc is not None and int(n) == sc[1]:
        return c_void_p(sc[0].data_ptr())
    return c_void_p(0)


def ndev_tensor(n: int):
    """The int64 device tensor of the active ``device_count`` block if ``n`` is its capacity, else None."""
    sc = _COUNT_SCOPE
    return sc[0] if (sc is not None and int(n) == sc[1]) else None


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().nsx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libnsx {what} failed (rc={rc}): {msg}")


def ptr(t, dtype=None) -> c_void_p:
    """Raw device pointer of a contiguous CUDA(HIP) tensor; None -> NULL."""
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("nersemble_amd native ops need device tensors (cuda/HIP); got a CPU tensor. "
                           "There is no CPU fallback.")
    if not t.is_contiguous():
        raise RuntimeError("nersemble_amd native ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    if t.device.index != torch._C._cuda_getDevice():
        # stream() hands the kernels the CURRENT device's stream: a tensor of another device would be dereferenced there
        raise RuntimeError(f"nersemble_amd native ops launch on the current device (cuda:{torch._C._cuda_getDevice()}); "
                           f"got a tensor on {t.device}. Use torch.cuda.set_device / torch.cuda.device.")
    return c_void_p(t.data_ptr())


def stream() -> c_void_p:
    """The raw HIP stream torch is currently enqueueing on (what every native call launches into)."""
    return c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def grid_geometry(n_levels=16, per_level_scale=1.4472692012786865, base_resolution=16,
                  log2_hashmap_size=19) -> GridGeom:
    g = GridGeom()
    check(lib().nsx_grid_geometry(n_levels, per_level_scale, base_resolution, log2_hashmap_size, C.byref(g)),
          "nsx_grid_geometry")
    return g


def padded_grids(H: int) -> int:
    return int(lib().nsx_padded_grids(H))
