"""BASELINE.json configs -> (trainer, synthetic data) builders.  Hyper-parameters come from the reference's
scripts/train/train_nersemble.py:59-111,184-256 (the source of truth for the path's constants)."""
from typing import Optional, Tuple

import torch

from .data.synthetic import SyntheticNeRSembleData
from .engine.trainer import NeRSembleTrainer, OptimizerConfig
from .field_components.deformation_field import SE3DeformationFieldConfig
from .field_components.hash_ensemble import HashEnsembleConfig, TCNNHashEncodingConfig
from .models.nersemble_instant_ngp import NeRSembleNGPModel, NeRSembleNGPModelConfig
from .rays import SceneBox

SCENE_BOXES = {   # train_nersemble.py:40-48
    30: [[-2.5, -1.8, -2.5], [2.2, 1.8, 2]],
    97: [[-2.2, -2.8, -2.5], [2.2, 2.2, 2]],
    124: [[-2.2, -2.5, -2.5], [2.2, 1.5, 2]],
}

WORKLOADS = {
    # name: (participant, n_hash_encodings, latent_dim_time, n_timesteps, disable_occ, lambda_dist, window_hash (begin,end))
    "static_h1": dict(pid=30, H=1, T=1, disable_occ=False, lambda_dist=1e-4, win=(40000, 80000), rays=256),
    "p030_h16": dict(pid=30, H=16, T=100, disable_occ=False, lambda_dist=1e-4, win=(40000, 80000), rays=4096),
    "p030_h32": dict(pid=30, H=32, T=100, disable_occ=False, lambda_dist=1e-4, win=(40000, 80000), rays=4096),
    "p097_dense": dict(pid=97, H=32, T=100, disable_occ=True, lambda_dist=0.0, win=(40000, 80000), rays=4096),
    "p124_dp": dict(pid=124, H=32, T=475, disable_occ=False, lambda_dist=1e-4, win=(50000, 100000), rays=4096),
}


def build_model_config(w: dict, max_n_samples_per_batch: int = 20, small: bool = False) -> NeRSembleNGPModelConfig:
    H = w["H"]
    enc = TCNNHashEncodingConfig(log2_hashmap_size=12 if small else 19)
    return NeRSembleNGPModelConfig(
        render_step_size=0.011, near_plane=0.2, far_plane=1e3, cone_angle=0, alpha_thre=1e-2, occ_thre=1e-2,
        early_stop_eps=0, background_color="white", grid_levels=1, disable_scene_contraction=True,
        max_n_samples_per_batch=-1 if max_n_samples_per_batch == -1 else 2 ** max_n_samples_per_batch,
        n_timesteps=w["T"], latent_dim_time=H,
        use_masked_rgb_loss=True, alpha_mask_threshold=0, lambda_alpha_loss=1e-2, lambda_near_loss=1e-4,
        lambda_empty_loss=1e-2, lambda_depth_loss=1e-4, lambda_dist_loss=w["lambda_dist"],
        use_hash_ensemble=True,
        hash_ensemble_config=HashEnsembleConfig(n_hash_encodings=H, hash_encoding_config=enc,
                                                disable_initial_hash_ensemble=True, use_soft_transition=True),
        use_deformation_field=True, use_separate_deformation_time_embedding=True,
        deformation_field_config=SE3DeformationFieldConfig(warp_code_dim=128, mlp_num_layers=6, mlp_layer_width=128),
        disable_occupancy_grid=w["disable_occ"],
        window_hash_encodings_begin=w["win"][0], window_hash_encodings_end=w["win"][1],
        window_deform_begin=0, window_deform_end=20000,
        use_view_frustum_culling=True, view_frustum_culling=2)


def build_workload(name: str, device="cuda:0", small: bool = False, rank: int = 0, world_size: int = 1,
                   n_rays: int = None, factored_table_grad=None, sharded_table_adam=None,
                   global_loss_normalisers: bool = False, window_hash: Optional[Tuple[int, int]] = None,
                   compact_first_grid: Optional[bool] = None, table_parallel: Optional[str] = "auto",
                   level_parallel_emulation: Optional[Tuple[int, int]] = None
                   ) -> Tuple[NeRSembleTrainer, SyntheticNeRSembleData, dict]:
    """``window_hash``: (begin, end) steps of the coarse-to-fine schedule of the hash grids instead of the workload's
    (train_nersemble.py:77-78: 40000, 80000); (0, 1) has every grid switched on from step 1 -- the state of a run after
    ``end``.  ``compact_first_grid``: None = the trainer's default (on)."""
    w = dict(WORKLOADS[name])
    if window_hash is not None:
        w["win"] = (int(window_hash[0]), int(window_hash[1]))
    box = torch.tensor(SCENE_BOXES[w["pid"]], dtype=torch.float32)
    rays = n_rays if n_rays is not None else w["rays"]
    data = SyntheticNeRSembleData(box, n_timesteps=w["T"], n_rays=rays, device=device, rank=rank)
    cfg = build_model_config(w, small=small)
    model = NeRSembleNGPModel(cfg, SceneBox(box), num_train_data=12 * w["T"],
                              metadata={"camera_frustums": data.camera_frustums}).to(device)
    if w["disable_occ"]:
        model.occupancy_grid.binaries.fill_(True)
        model.occupancy_grid.occs.fill_(1.0)
    trainer = NeRSembleTrainer(model, OptimizerConfig(), mixed_precision=True, world_size=world_size,
                               factored_table_grad=factored_table_grad, rank=rank,
                               sharded_table_adam=sharded_table_adam, global_loss_normalisers=global_loss_normalisers,
                               table_parallel=table_parallel, level_parallel_emulation=level_parallel_emulation,
                               **({} if compact_first_grid is None else {"compact_first_grid": compact_first_grid}))
    info = dict(workload=name, participant=w["pid"], n_hash_encodings=w["H"], n_timesteps=w["T"], rays=rays,
                params=sum(p.numel() for p in model.parameters()))
    return trainer, data, info
