"""Training step -- mirror of the reference's engine/nersemble_trainer.py:169-206 (``train_iteration``):
callbacks before the iteration, ``autocast(fp16, cache_enabled=False)``, GradScaler-scaled backward, Adam per
parameter group (eps 1e-15) with StepLR, LR step skipped when the scale dropped; optimizer hyper-parameters
from scripts/train/train_nersemble.py:243-256.  Adds what the reference lacks: data-parallel training, one
process per GPU, gradients averaged with RCCL all-reduce over xGMI.
"""
import functools
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from ..models.nersemble_instant_ngp import NeRSembleNGPModel
from .parallel import all_reduce_gradients
from ..rays import RayBundle


@dataclass
class OptimizerConfig:
    lr_main: float = 5e-3
    lr_deformation_field: float = 1e-3
    lr_embeddings: float = 5e-3
    eps: float = 1e-15
    step_size: int = 20000
    gamma_fields: float = 0.8
    gamma_deformation_field: float = 0.5
    gamma_embeddings: float = 0.8


class NeRSembleTrainer:
    def __init__(self, model: NeRSembleNGPModel, opt_cfg: Optional[OptimizerConfig] = None,
                 mixed_precision: bool = True, world_size: int = 1):
        self.model = model
        self.cfg = opt_cfg or OptimizerConfig()
        self.mixed_precision = mixed_precision
        self.world_size = world_size
        groups = model.get_param_groups()
        lrs = {"fields": self.cfg.lr_main, "deformation_field": self.cfg.lr_deformation_field,
               "embeddings": self.cfg.lr_embeddings}
        gammas = {"fields": self.cfg.gamma_fields, "deformation_field": self.cfg.gamma_deformation_field,
                  "embeddings": self.cfg.gamma_embeddings}
        self.optimizers, self.schedulers = {}, {}
        for name, params in groups.items():
            self.optimizers[name] = torch.optim.Adam(params, lr=lrs[name], eps=self.cfg.eps, weight_decay=0)
            self.schedulers[name] = torch.optim.lr_scheduler.StepLR(self.optimizers[name], step_size=self.cfg.step_size,
                                                                    gamma=gammas[name])
        self.grad_scaler = torch.amp.GradScaler("cuda", enabled=mixed_precision)
        self.callbacks = model.get_training_callbacks()

    # ---- data-parallel gradient averaging ------------------------------------------------------------
    def _all_reduce_grads(self) -> None:
        if self.world_size <= 1:
            return
        params = [p for opt in self.optimizers.values() for pg in opt.param_groups for p in pg["params"]]
        all_reduce_gradients(params, self.world_size)

    def train_iteration(self, step: int, ray_bundle: RayBundle, batch: Dict[str, torch.Tensor]
                        ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        self.model.train()
        for cb in self.callbacks:
            cb.run(step)
        for opt in self.optimizers.values():
            opt.zero_grad(set_to_none=True)
        dev_type = ray_bundle.origins.device.type
        with torch.autocast(device_type=dev_type, dtype=torch.float16, enabled=self.mixed_precision, cache_enabled=False):
            outputs = self.model(ray_bundle)
            metrics_dict = self.model.get_metrics_dict(outputs, batch)
            loss_dict = self.model.get_loss_dict(outputs, batch, metrics_dict)
            loss = functools.reduce(torch.add, loss_dict.values())
        self.grad_scaler.scale(loss).backward()
        self._all_reduce_grads()
        for opt in self.optimizers.values():
            self.grad_scaler.step(opt)
        scale = self.grad_scaler.get_scale()
        self.grad_scaler.update()
        if scale <= self.grad_scaler.get_scale():
            for sch in self.schedulers.values():
                sch.step()
        return loss, loss_dict, metrics_dict
