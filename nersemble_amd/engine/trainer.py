"""Training step -- mirror of the reference's engine/nersemble_trainer.py:169-206 (``train_iteration``):
callbacks before the iteration, ``autocast(fp16, cache_enabled=False)``, GradScaler-scaled backward, Adam per
parameter group (eps 1e-15) with StepLR, LR step skipped when the scale dropped; optimizer hyper-parameters
from scripts/train/train_nersemble.py:243-256.  Adds what the reference lacks: data-parallel training, one
process per GPU, gradients averaged with RCCL all-reduce over xGMI.
"""
import functools
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from ..models.nersemble_instant_ngp import NeRSembleNGPModel
from .hash_adam import HashTableAdam, NativeGradScaler
from .parallel import all_reduce_gradients, check_gradient_presence
from .level_parallel import LevelParallelTableAdam
from .sharded_adam import ShardedTableAdam
from .small_adam import SmallGroupAdam, adam_groups, unscale_and_check_groups
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


_NATIVE_SCALE_UPDATE = os.environ.get("NSX_NATIVE_SCALER", "1") != "0"      # (A/B knob: the torch route of GradScaler.update)


class StepLR:
    """lr = base_lr * gamma ** (steps // step_size) -- torch.optim.lr_scheduler.StepLR's schedule
    (train_nersemble.py:243-256) without its per-step Python machinery (0.13 ms per scheduler and step, four of them)."""

    def __init__(self, optimizer: torch.optim.Optimizer, step_size: int, gamma: float):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.last_epoch = 0
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]

    def step(self) -> None:
        self.last_epoch += 1
        if self.last_epoch % self.step_size == 0:
            for g in self.optimizer.param_groups:
                g["lr"] = g["lr"] * self.gamma

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, state) -> None:
        self.last_epoch = int(state["last_epoch"])
        self.base_lrs = list(state.get("base_lrs", self.base_lrs))
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = base * self.gamma ** (self.last_epoch // self.step_size)


class NeRSembleTrainer:
    def __init__(self, model: NeRSembleNGPModel, opt_cfg: Optional[OptimizerConfig] = None,
                 mixed_precision: bool = True, world_size: int = 1, factored_table_grad: Optional[bool] = None,
                 rank: Optional[int] = None, sharded_table_adam: Optional[bool] = None,
                 overlap_table_adam: bool = True, calibrate_table_placement: bool = True,
                 global_loss_normalisers: bool = False, early_table_step: bool = False,
                 compact_first_grid: bool = True, table_parallel: Optional[str] = "auto",
                 level_parallel_emulation: Optional[Tuple[int, int]] = None):
        """``global_loss_normalisers``: the ranks hold consecutive slices of ONE ray batch (strong scaling)
        -- loss denominators are made global so that the step equals the single-process step on the union batch."""
        self.model = model
        # ``table_parallel`` (data-parallel runs with the sharded table optimizer): "auto" = the exchange follows the
        # coarse-to-fine window -- narrow fp16 reduce-scatter / all-gather while few grids are on (engine/sharded_adam.py),
        # and from the step at which the window passes H / 2 the LEVEL-parallel exchange that moves samples instead of
        # parameters (engine/level_parallel.py); "level" = level-parallel from the first step; "shard" / None = the
        # reduce-scatter exchange throughout (rounds 1-4)
        self.table_parallel = table_parallel
        # ``level_parallel_emulation = (N, r)`` (single process, a one-rank process group): train as rank r of an N-rank
        # level-parallel job whose other ranks are replicas of this one -- the partition, the exchange and every collective
        # of such a rank (on the one-rank group), measurable on one GPU (engine/level_parallel.py, ``emulate``)
        self.level_parallel_emulation = tuple(level_parallel_emulation) if level_parallel_emulation else None
        if self.level_parallel_emulation is not None and world_size != 1:
            raise ValueError("level_parallel_emulation runs in a single process (world_size = 1)")
        # ``compact_first_grid``: while the coarse-to-fine window keeps one hash grid on (the first 40 000 steps of the
        # reference's schedule), train a contiguous copy of that grid with the H = 1 kernels instead of the 32-grid layout
        # (HashEnsemble.first_grid_phase: same values, ~1.4-1.7x per step).  Single GPU, factored table gradient.  ON by
        # default (round 3): the phase ends by itself when the window opens, ``state_dict()`` / checkpoints /
        # ``model.eval()`` see the full layout; only code that reads the ``hash_ensemble.tables`` PARAMETER or the table
        # optimizer's moments directly while the phase lasts must call ``consolidate()`` first.
        self.compact_first_grid = compact_first_grid
        # start the table optimizer from inside the backward (HashTableAdam.arm_early_step).  OFF by default: measured
        # slower (8.8 vs 8.3 ms per step early in training, 4.85 vs 4.25 ms in steady state) -- the 12 GB pass next to
        # the deformation backward triples the latter (both are HBM-bound) and crowds the step's tail off the CUs
        self.early_table_step = early_table_step
        self.prefetch_march = True                # use train_iteration's next_ray_bundle (off: for A/B measurements)
        self.cfg = opt_cfg or OptimizerConfig()
        self.mixed_precision = mixed_precision
        self.world_size = world_size
        self.rank = (dist.get_rank() if dist.is_initialized() else 0) if rank is None else rank
        model.global_loss_normalisers = dict(world_size=world_size, rank=self.rank) \
            if (global_loss_normalisers and world_size > 1) else None
        device = next(model.parameters()).device
        groups = model.get_param_groups()
        lrs = {"fields": self.cfg.lr_main, "deformation_field": self.cfg.lr_deformation_field,
               "embeddings": self.cfg.lr_embeddings}
        gammas = {"fields": self.cfg.gamma_fields, "deformation_field": self.cfg.gamma_deformation_field,
                  "embeddings": self.cfg.gamma_embeddings}
        self.optimizers, self.schedulers, self.group_of = {}, {}, {}
        # the reference's numbering of every group (``torch.optim.Adam(list(params))`` numbers its parameters by position;
        # nerfstudio builds one Adam per group of ``get_param_groups``): per position ("table", c) for the c-th tcnn hash
        # encoding the native tables stand for, ("param", p) for a tensor an optimizer here steps, ("none", p) for members
        # that never have a gradient -- tcnn's empty ``params`` of parameter-free encodings, the frozen ``aabb`` of the
        # deformation field -- which torch's Adam lists in ``param_groups`` and keeps no state for.  Checkpoints are
        # written and read in THIS numbering (tests/golden/state_manifest.json holds the reference's).
        self.group_layout = {}
        fused = device.type == "cuda"
        tables = model.field.hash_ensemble.tables
        n_enc = model.field.hash_ensemble.n_tcnn_encodings
        for name, params in groups.items():
            layout = []
            for p in params:
                if p is tables:
                    layout.extend(("table", c) for c in range(n_enc))
                elif p.requires_grad and p.numel() > 0:
                    layout.append(("param", p))
                else:
                    layout.append(("none", p))
            self.group_layout[name] = layout
            small = [x for kind, x in layout if kind == "param"]
            if fused:
                # all small groups step together in two native launches (engine/small_adam.py)
                self.optimizers[name] = SmallGroupAdam(small, lr=lrs[name], eps=self.cfg.eps)
            else:
                self.optimizers[name] = torch.optim.Adam(small, lr=lrs[name], eps=self.cfg.eps, weight_decay=0)
            self.group_of[name] = name
            if any(p is tables for p in params):
                # the 403 M-parameter hash tables: native fused step on the factored gradient.  Data-parallel runs
                # shard the optimizer state and exchange fp16 (reduce-scatter / all-gather, engine/sharded_adam.py);
                # factored_table_grad=False keeps the dense fp32 gradient + all-reduce path.
                sharded = (world_size > 1 and factored_table_grad is not False) if sharded_table_adam is None \
                    else sharded_table_adam
                if self.level_parallel_emulation is not None:
                    n_emu, r_emu = self.level_parallel_emulation
                    opt_emu = LevelParallelTableAdam(model.field.hash_ensemble, lr=lrs[name], eps=self.cfg.eps,
                                                     world_size=n_emu, rank=r_emu, emulate=True)
                    model.field.hash_ensemble._level_parallel_owner = opt_emu
                    self.optimizers[name + "/tables"] = opt_emu
                elif sharded:
                    # (the exchange follows the coarse-to-fine window: every rank's schedule holds the same value)
                    sched = getattr(model, "sched_window_hash_encodings", None)
                    self.optimizers[name + "/tables"] = ShardedTableAdam(
                        model.field.hash_ensemble, lr=lrs[name], eps=self.cfg.eps, world_size=world_size, rank=self.rank,
                        width_source=(lambda s=sched: s.value) if sched is not None else None)
                else:
                    self.optimizers[name + "/tables"] = HashTableAdam(model.field.hash_ensemble, lr=lrs[name],
                                                                      eps=self.cfg.eps,
                                                                      factored=(world_size == 1) if factored_table_grad is None
                                                                      else factored_table_grad)
                self.group_of[name + "/tables"] = name
        for key, opt in self.optimizers.items():
            self.schedulers[key] = StepLR(opt, step_size=self.cfg.step_size, gamma=gammas[self.group_of[key]])
        self.grad_scaler = NativeGradScaler(device, enabled=mixed_precision)
        self.callbacks = model.get_training_callbacks()
        self._pending, self._found_host, self._found_event = None, None, None
        self._flag_state = None       # persistent found_inf flags of the native scale update
        self._presence, self._presence_host = None, None      # data-parallel: ranks per parameter that held a gradient
        self._took_part = None                                # ... > 0 in the previous step (engine/parallel.py)
        self._presence_params = self._presence_params_late = None   # the parameters those counts are about (this / the read-back step)
        # the table optimizer's 12 GB pass runs beside the rest of the step's tail and the next step's ray marching
        self._opt_stream = torch.cuda.Stream(device) if (overlap_table_adam and device.type == "cuda") else None
        self._found_groups = []
        # where the table optimizer's streams live in HBM changes the pass by up to 20 % (engine/placement.py)
        self.placement_report = None
        table_opt = self.optimizers.get(self.group_of_tables())
        if isinstance(table_opt, ShardedTableAdam) and self._opt_stream is not None:
            # ONE side stream for the expansion + reduce-scatter (from inside the backward) and, later, the shard's Adam +
            # all-gather: HIP maps its streams onto 4 hardware queues by default, and main, ray-march prefetch, optimizer and
            # RCCL's own stream already use them.  (Measured, tools/run_{m,n,o}_r06.sh: whichever side stream is first used
            # inside the backward lands on the main stream's queue and runs in line with it; with GPU_MAX_HW_QUEUES=8 the
            # expansion does overlap the deformation backward, which then takes 3.6 x as long -- the two share the CUs -- for
            # 0.01-0.09 ms per step.  What shortened the rank's step was a smaller G: HashEnsemble.first_grid_planes.)
            table_opt.comm_stream = self._opt_stream
            sink = model.field.hash_ensemble.grad_sink
            if sink is not None:
                sink._fill_stream = self._opt_stream      # (the clear-ahead of G: that stream is idle while the forward runs)
        if calibrate_table_placement and isinstance(table_opt, HashTableAdam):
            from .placement import calibrate_table_placement as _calibrate
            self.placement_report = _calibrate(model.field.hash_ensemble, table_opt)
        # (round 6: also data-parallel ranks -- the packed buffer of the narrow exchange is the compact copy's working table,
        # engine/sharded_adam.py)
        model.field.hash_ensemble.compact_first_grid = bool(compact_first_grid and (
            (world_size == 1 and isinstance(table_opt, HashTableAdam) and table_opt.factored)
            or (isinstance(table_opt, ShardedTableAdam) and table_opt.compact_layouts_supported()
                and self.table_parallel != "level")))

    def _maybe_level_parallel(self, ray_bundle: Optional[RayBundle] = None) -> None:
        """Data-parallel runs: hand the tables from the reduce-scatter exchange (``ShardedTableAdam``) to the level-parallel
        one once the coarse-to-fine window has passed H / 2 -- a collective every rank reaches at the same step (the schedule
        is the same everywhere).  The fp32 master and both moments are gathered once; each rank keeps its levels' range."""
        key = self.group_of_tables()
        opt = self.optimizers.get(key) if key else None
        if not isinstance(opt, ShardedTableAdam) or self.table_parallel not in ("auto", "level"):
            return
        he = self.model.field.hash_ensemble
        if he.geom.n_levels % self.world_size != 0 or not he.tables.is_cuda:
            return
        if self.table_parallel == "auto":
            sched = getattr(self.model, "sched_window_hash_encodings", None)
            if sched is not None and float(sched.value) < opt.Hp / 2:
                return
        # the optimizer pass of a level owner reads one gradient plane per (rank, code row of that rank's batch): the job's
        # rows must fit NSX_MAX_ADAM_SLOTS.  Decided ONCE, together (advisor, round 5: 8 ranks with more than 24 images per
        # batch used to abort at the hand-over step): a job that does not fit stays on the reduce-scatter exchange.
        md = (ray_bundle.metadata or {}) if ray_bundle is not None else {}
        rows = int(md["_image_timesteps"].numel()) if "_image_timesteps" in md else 0
        most = torch.tensor([float(rows)], device=he.tables.device)
        dist.all_reduce(most, op=dist.ReduceOp.MAX, group=opt.group)
        from .level_parallel import MAX_ADAM_SLOTS
        if int(most.item()) * self.world_size > MAX_ADAM_SLOTS or int(most.item()) > 64:
            import warnings
            warnings.warn(f"level-parallel tables need <= {MAX_ADAM_SLOTS} code rows in the job's batch; this job has "
                          f"{self.world_size} ranks x up to {int(most.item())} rows: staying on the reduce-scatter exchange")
            self.table_parallel = "shard"
            return
        self.flush_scheduler_step()
        he.leave_first_grid_phase()                           # (a compact copy of the window ramp goes back into the full layout)
        he.wait_tables()
        opt.gather_master()
        b = opt._buffers()
        exp_avg, exp_avg_sq = opt._gather_shards(b["exp_avg"]), opt._gather_shards(b["exp_avg_sq"])
        pg = opt.param_groups[0]
        new = LevelParallelTableAdam(he, lr=pg["lr"], betas=pg["betas"], eps=pg["eps"], world_size=self.world_size,
                                     rank=self.rank, group=opt.group, step=opt._step, exp_avg=exp_avg, exp_avg_sq=exp_avg_sq)
        he._level_parallel_owner = new
        new.timing = opt.timing
        self.optimizers[key] = new
        self.schedulers[key].optimizer = new
        del exp_avg, exp_avg_sq
        opt._buf = None                                       # (the working tables stay: HashEnsemble.tables_f16 views them)

    def become_emulated_level_parallel_rank(self, n_ranks: int, rank: int, frozen: bool = True, shadow: bool = True) -> None:
        """A single-GPU run (``HashTableAdam``) goes on as rank ``rank`` of an ``n_ranks``-rank level-parallel job whose other
        ranks are replicas of this process (``LevelParallel(emulate=True)``; a one-rank process group must exist).  Made for
        measuring what such a rank computes and issues per step once the model is TRAINED: ``shadow`` keeps the feature
        columns the model sees the true ones (one extra launch per forward, priced by ``comm_report``), ``frozen`` sets
        every learning rate to zero -- the emulated backward's table gradient is a stand-in (the replicas' planes receive
        this rank's own column blocks), so the parameters must not move; every kernel still runs with its full traffic."""
        key = self.group_of_tables()
        opt = self.optimizers.get(key) if key else None
        if not isinstance(opt, HashTableAdam) or self.world_size != 1:
            raise RuntimeError("become_emulated_level_parallel_rank: a single-process run with the fused table optimizer")
        he = self.model.field.hash_ensemble
        self.flush_scheduler_step()
        he.leave_first_grid_phase()
        he.wait_tables()
        st = opt._state()
        pg = opt.param_groups[0]
        new = LevelParallelTableAdam(he, lr=pg["lr"], betas=pg["betas"], eps=pg["eps"], world_size=n_ranks, rank=rank,
                                     step=int(st["step"]), exp_avg=st["exp_avg"], exp_avg_sq=st["exp_avg_sq"], emulate=True)
        new.lp.shadow_forward = bool(shadow)
        he._level_parallel_owner = new
        he._compact_listeners = []
        he.compact_first_grid = False
        opt.state.clear()
        self.optimizers[key] = new
        self.schedulers[key].optimizer = new
        self.level_parallel_emulation = (int(n_ranks), int(rank))
        self._took_part = None
        if frozen:
            for o in self.optimizers.values():
                for g in o.param_groups:
                    g["lr"] = 0.0

    def group_of_tables(self) -> Optional[str]:
        """Key of the optimizer that owns the hash tables (``"<group>/tables"``), if there is one."""
        return next((k for k in self.optimizers if k.endswith("/tables")), None)

    # ---- data-parallel gradient averaging ------------------------------------------------------------
    def _all_reduce_grads(self) -> None:
        if self.world_size <= 1 and self.level_parallel_emulation is None:
            return
        # (the sharded table optimizer runs its own reduce-scatter; its parameter has no dense gradient -- every rank has
        # started that collective before the ones below are issued, also a rank without samples)
        for opt in self.optimizers.values():
            if isinstance(opt, (ShardedTableAdam, LevelParallelTableAdam)):
                opt.ensure_reduce_started()
        params = [p for opt in self.optimizers.values() if not isinstance(opt, (ShardedTableAdam, LevelParallelTableAdam))
                  for pg in opt.param_groups for p in pg["params"]]
        # which parameters took part in the PREVIOUS step (on any rank): its counts have reached the host by now
        self.flush_scheduler_step()
        # the native step deposited most of these gradients as views of ONE persistent buffer: reduced where they are
        # (whether there is one follows from the configuration, the same on every rank: the bucket's wire format)
        arena = None
        model = self.model
        if (getattr(model, "native_step", False) and getattr(model, "fuse_main_pass", False) and self.mixed_precision
                and model.deformation_field is not None and next(model.parameters()).is_cuda
                and model.field.hash_ensemble.geom.n_levels == 16 and model.field.mlp_base.n_output_dims == 16):
            if model._native is None:
                from .native_step import NativeStep
                model._native = NativeStep(model)
            arena = model._native.grad_arena()
        # level-parallel tables: the owners' non-finite flags ride in the same bucket (a step is skipped on every rank or on
        # none); every other flag is computed from the REDUCED gradients, identical on all ranks -- no collective of its own
        table_opt = self.optimizers.get(self.group_of_tables() or "")
        lp_flag = native_comm = None
        if isinstance(table_opt, LevelParallelTableAdam):
            lp_flag = table_opt.local_nonfinite()
            native_comm = table_opt.lp.comm           # (the exchange's own RCCL communicator spans the same ranks)
        self._presence, flags = all_reduce_gradients(params, self.world_size, takes_part=self._took_part,
                                                     force=self.level_parallel_emulation is not None, arena=arena,
                                                     extra_flags=lp_flag, native_comm=native_comm)
        # (the counts' order: the parameters that take gradients, as all_reduce_gradients lists them)
        self._presence_params = [p for p in params if p.requires_grad] if self._presence is not None else None
        if isinstance(table_opt, LevelParallelTableAdam):
            table_opt.reduced_nonfinite = flags

    def _arm_early_table_step(self):
        """Single GPU, fused main pass: let the table optimizer start from inside the backward (HashTableAdam.
        arm_early_step).  Returns (found_all, inv_scale) for ``_optimizer_step_all`` or None."""
        key = self.group_of_tables()
        opt = self.optimizers.get(key) if key else None
        if not self.early_table_step or self.world_size != 1 or not isinstance(opt, HashTableAdam) \
                or opt.he.grad_sink is None or self._opt_stream is None:
            return None
        self.flush_scheduler_step()        # the learning rates of this step depend on the previous step's outcome
        inv_scale = self.grad_scaler.inv_scale()
        groups = sorted(set(self.group_of.values()))
        found_all = torch.zeros((len(groups),), dtype=torch.float32, device=inv_scale.device)
        i = groups.index(self.group_of[key])
        opt.arm_early_step(found_all[i:i + 1], inv_scale, self._opt_stream)
        return found_all, inv_scale

    def _optimizer_step_all(self, early=None):
        """GradScaler semantics (nersemble_trainer.py:186): unscale + inf check per optimizer group, skip the step of
        a group whose gradients hold inf/NaN, then one scale update from all groups.  ``early``: (found_all, inv_scale)
        of a step whose table optimizer was armed to start inside the backward."""
        self.flush_scheduler_step()        # the learning rates of this step depend on the previous step's outcome
        scaler = self.grad_scaler
        groups = sorted(set(self.group_of.values()))
        native_update = False
        if early is not None:
            found_all, inv_scale = early
            dev = inv_scale.device
            found = {g: found_all[i:i + 1] for i, g in enumerate(groups)}
        else:
            native_update = isinstance(scaler, NativeGradScaler) and scaler._scale.is_cuda and _NATIVE_SCALE_UPDATE
            dev = scaler._scale.device
            if native_update:
                # the flags and 1 / scale live in TWO persistent slots used in turn: the scale update of step n
                # (nsx_grad_scaler_update) copies step n's flags out, clears slot n + 1 and writes ITS 1 / scale, while the
                # table optimizer of step n may still be reading slot n on its own stream (it has finished before the
                # HashEnsemble forward of step n + 1, i.e. long before slot n is written again at the end of that step)
                st = self._flag_state
                stamp = (scaler._scale._version, scaler._scale.data_ptr())
                if st is None or st["groups"] != groups or st["copy"].device != dev:
                    slots = []
                    for _ in range(2):
                        all_ = torch.zeros((len(groups),), dtype=torch.float32, device=dev)
                        slots.append({"all": all_, "views": {g: all_[i:i + 1] for i, g in enumerate(groups)},
                                      "inv": scaler.inv_scale().clone()})
                    st = self._flag_state = {"groups": groups, "slots": slots, "turn": 0, "stamp": stamp,
                                             "copy": torch.zeros((len(groups),), dtype=torch.float32, device=dev)}
                elif st["stamp"] != stamp:                       # the scale was written from torch (load_state_dict)
                    st["slots"][st["turn"]]["inv"].copy_(scaler.inv_scale())
                    st["stamp"] = stamp
                slot = st["slots"][st["turn"]]
                found_all, found, inv_scale = slot["all"], slot["views"], slot["inv"]
            else:
                inv_scale = scaler.inv_scale()
                dev = inv_scale.device
                found_all = torch.zeros((len(groups),), dtype=torch.float32, device=dev)
                found = {g: found_all[i:i + 1] for i, g in enumerate(groups)}
        table_opt = self.optimizers.get(self.group_of_tables() or "")
        stepped_early = isinstance(table_opt, HashTableAdam) and table_opt.stepped_early
        if isinstance(table_opt, HashTableAdam):
            table_opt.disarm_early_step()
            table_opt.stepped_early = False
        native_small = [(key, opt) for key, opt in self.optimizers.items() if isinstance(opt, SmallGroupAdam)]
        small_groups = [groups.index(self.group_of[key]) for key, _ in native_small]
        table = None
        if native_small:
            table = unscale_and_check_groups([o for _, o in native_small], small_groups, len(groups), found_all, inv_scale)
        for key, opt in self.optimizers.items():
            f = found[self.group_of[key]]
            if isinstance(opt, (HashTableAdam, ShardedTableAdam, LevelParallelTableAdam)):
                if not (stepped_early and opt is table_opt):
                    opt.check_finite(f)
            elif not isinstance(opt, SmallGroupAdam):
                grads = [p.grad for pg in opt.param_groups for p in pg["params"] if p.grad is not None]
                scaler.unscale_and_check(grads, f, inv_scale)
        flags_agree = isinstance(table_opt, LevelParallelTableAdam) and table_opt.reduced_nonfinite is not None
        if (self.world_size > 1 or self.level_parallel_emulation is not None) and not flags_agree:
            # a step is skipped on every rank or on none: the shard-level checks of the table gradient differ per rank
            # (level-parallel tables: the flags already agree -- see _all_reduce_grads)
            dist.all_reduce(found_all, op=dist.ReduceOp.MAX)
        if flags_agree:
            table_opt.reduced_nonfinite = None
        for key, opt in self.optimizers.items():
            f = found[self.group_of[key]]
            if isinstance(opt, HashTableAdam):
                if not (stepped_early and opt is table_opt):
                    opt.step_unhooked(found_inf=f, inv_scale=inv_scale, side_stream=self._opt_stream)
            elif isinstance(opt, (ShardedTableAdam, LevelParallelTableAdam)):
                opt.step(found_inf=f, inv_scale=inv_scale, side_stream=self._opt_stream)
            elif isinstance(opt, SmallGroupAdam):
                continue
            elif any(p.grad is not None for pg in opt.param_groups for p in pg["params"]):
                if opt.defaults.get("fused"):
                    opt.found_inf, opt.grad_scale = f.reshape(()), None
                    opt.step()
                    del opt.found_inf, opt.grad_scale
                elif f.item() == 0:
                    opt.step()
        if native_small:
            # data-parallel: a parameter NO rank had a gradient for this step is left alone on the device (its count in the
            # all-reduced bucket is 0), although every rank holds the zeros it joined the all-reduce with (advisor, round 5:
            # Adam stepped it with g = 0 -- momentum drift and a step count a single process does not take)
            present = index = None
            if self._presence is not None and self._presence_params:
                present = self._presence[0]
                index = {id(p): i for i, p in enumerate(self._presence_params)}
            adam_groups([o for _, o in native_small], small_groups, len(groups), table, found_all, present, index)
            # the fused MLPs' fp16 weight copies for the NEXT step, now: two small launches that would otherwise sit in
            # the dependent chain behind the table optimizer (tcnn.Network.half_weights is lazy) run beside it instead
            field = getattr(self.model, "field", None)
            for net in (getattr(field, "mlp_base", None), getattr(field, "mlp_head", None)):
                if net is not None and hasattr(net, "half_weights") and net.params.is_cuda:
                    net.half_weights()
        self._found_groups = groups
        if native_update:
            st = self._flag_state
            nxt = st["slots"][1 - st["turn"]]
            scaler.update_native(found_all, st["copy"], nxt["all"], nxt["inv"])
            st["turn"] = 1 - st["turn"]
            return st["copy"]
        scaler.update(list(found.values()))
        return found_all

    def train_iteration(self, step: int, ray_bundle: RayBundle, batch: Dict[str, torch.Tensor],
                        next_ray_bundle: Optional[RayBundle] = None
                        ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
        """One optimisation step.  ``next_ray_bundle`` (optional): the rays of step ``step + 1`` if the loader has them
        already -- the counting pass of their ray marching then runs beside this step (``prefetch_sampling``), and the
        next call finds the sample count on the host instead of waiting for the device.  It must be the very bundle
        object passed to that next call; anything else is simply not used."""
        if not self.model.training:               # (nn.Module.train() walks all ~100 sub-modules)
            self.model.train()
        for cb in self.callbacks:
            cb.run(step)
        if self.world_size > 1:
            self._maybe_level_parallel(ray_bundle)
        if next_ray_bundle is not None and self.prefetch_march:
            self.model.prefetch_sampling(next_ray_bundle, step + 1)
        for opt in self.optimizers.values():
            clear = getattr(opt, "clear_grads", None)          # the native optimizers' zero_grad(set_to_none=True)
            if clear is not None:
                clear()
            else:
                opt.zero_grad(set_to_none=True)
        dev_type = ray_bundle.origins.device.type
        with torch.autocast(device_type=dev_type, dtype=torch.float16, enabled=self.mixed_precision, cache_enabled=False):
            fast = self.model.fused_train_forward(ray_bundle, batch) if self.mixed_precision else None
            if fast is not None:
                loss_dict, metrics_dict, outputs = fast
            else:
                outputs = self.model(ray_bundle)
                metrics_dict = self.model.get_metrics_dict(outputs, batch)
                loss_dict = self.model.get_loss_dict(outputs, batch, metrics_dict)
            loss = getattr(loss_dict, "total", None)
            if loss is None:
                loss = functools.reduce(torch.add, loss_dict.values())
        early = self._arm_early_table_step() if fast is not None else None
        fused_vec = getattr(loss_dict, "fused", None)
        if fused_vec is not None and loss is getattr(loss_dict, "total", None):
            # the loss is one entry of the fused pass's output vector: start the backward at that node with the scaled
            # one-hot gradient `scale(loss).backward()` would have produced there
            torch.autograd.backward([fused_vec], [self.grad_scaler.loss_grad_vector(fused_vec.numel(), loss_dict.total_index)])
        else:
            self.grad_scaler.scale(loss).backward()
        self._all_reduce_grads()
        found_all = self._optimizer_step_all(early)
        # the reference skips the LR step when the scale dropped, i.e. when an inf/NaN was found (:199-203).  Reading
        # the flags here would drain the GPU queue at the end of every step; they are copied to pinned memory instead
        # and consulted right before the learning rates are next used (flush_scheduler_step).
        self._defer_scheduler_step(found_all)
        return loss, loss_dict, metrics_dict

    def _defer_scheduler_step(self, found_all: torch.Tensor) -> None:
        """found_all: one inf/NaN flag per parameter group (device)."""
        self._presence_params_late = self._presence_params if self._presence is not None else None
        if not found_all.is_cuda:
            self._pending = ("host", found_all.clone())
            self._presence_host, self._presence = (self._presence.clone() if self._presence is not None else None), None
            return
        if self._found_host is None or self._found_host.numel() != found_all.numel():
            self._found_host = torch.empty((found_all.numel(),), dtype=torch.float32).pin_memory()
            self._found_event = torch.cuda.Event()
        self._found_host.copy_(found_all, non_blocking=True)
        if self._presence is not None:
            if self._presence_host is None or self._presence_host.shape != self._presence.shape:
                self._presence_host = torch.empty(tuple(self._presence.shape), dtype=torch.float32).pin_memory()
            self._presence_host.copy_(self._presence, non_blocking=True)
            self._presence = None
        self._found_event.record()
        self._pending = ("pinned", None)

    def consolidate(self) -> None:
        """Data-parallel runs: rebuild the full fp32 master tables on every rank (before ``model.state_dict()``).
        Compact first-grid phase: write grid 0 and its moments back into the full layout."""
        self.model.field.hash_ensemble.sync_first_grid()
        for opt in self.optimizers.values():
            if isinstance(opt, (ShardedTableAdam, LevelParallelTableAdam)):
                opt.gather_master()

    # ---- checkpointing of the training state (nerfstudio's checkpoints carry "optimizers" and "scalers") -------------
    # Wire format = the reference's: ``optimizers[group] = torch.optim.Adam.state_dict()`` for the three groups of
    # ``get_param_groups`` (nersemble_instant_ngp.py:502-514; nerfstudio's Optimizers.load_optimizers does
    # ``self.optimizers[k].load_state_dict(v)`` for every key).  The ``fields`` group holds ``field.parameters()`` in
    # registration order -- ``hash_ensemble.hash_encodings.{c}.params`` (flat, tcnn layout), ``mlp_base.params``,
    # ``mlp_head.params`` -- so the natively stepped tables appear there as C per-encoding entries in front of the two
    # MLPs, each with torch's per-parameter ``step``.
    def state_dict(self) -> Dict:
        """Everything a resumed run needs besides the model: Adam moments + step counts of every group (the table
        moments in the reference's tcnn parameter layout, gathered from all ranks in data-parallel runs -- a
        collective there), StepLR counters, the loss scale and its growth tracker.  ``consolidate()`` is called first:
        in the compact first-grid phase (the default for the first 40 000 steps) the moments of grid 0 live in their
        contiguous copy, and a data-parallel run holds the fp32 master in shards -- ``model.state_dict()`` taken after
        this call is complete as well."""
        self.flush_scheduler_step()
        self.consolidate()
        tkey = self.group_of_tables()
        opts = {}
        for key, opt in self.optimizers.items():
            if isinstance(opt, (HashTableAdam, ShardedTableAdam, LevelParallelTableAdam)):
                continue
            table = self.optimizers[tkey].table_state() if (tkey is not None and self.group_of[tkey] == key) else None
            opts[key] = _to_reference_numbering(self.group_layout[key], opt, opt.state_dict(), table)
        return {"optimizers": opts,
                "schedulers": {k: s.state_dict() for k, s in self.schedulers.items() if not k.endswith("/tables")},
                "scalers": self.grad_scaler.state_dict()}

    def load_state_dict(self, state: Dict) -> None:
        self.flush_scheduler_step()
        saved_all = state.get("optimizers", {})
        tkey = self.group_of_tables()
        for key, opt in self.optimizers.items():
            if isinstance(opt, (HashTableAdam, ShardedTableAdam, LevelParallelTableAdam)):
                continue
            if key not in saved_all:
                raise KeyError(f"checkpoint has no optimizer state for the parameter group '{key}' "
                               f"(groups in the file: {sorted(saved_all)})")
            has_tables = tkey is not None and self.group_of[tkey] == key
            table_state, saved = _from_reference_numbering(self.group_layout[key], saved_all[key], key)
            if has_tables:
                legacy = saved_all.get(tkey)
                if table_state is None and isinstance(legacy, dict) and "native_table_adam" in legacy:
                    table_state = dict(legacy["native_table_adam"])     # files of round 2: tables under their own key
                if table_state is None:
                    table_state = {"step": 0, "exp_avg": None, "exp_avg_sq": None}
                table_state.setdefault("lr", saved["param_groups"][0]["lr"])
                self.optimizers[tkey].load_table_state(table_state)
            opt.load_state_dict(saved)
        for key, sch in self.schedulers.items():
            grp = self.group_of[key]
            if grp in state.get("schedulers", {}):
                sch.load_state_dict(state["schedulers"][grp])
        if state.get("scalers"):
            self.grad_scaler.load_state_dict(state["scalers"])

    def flush_scheduler_step(self) -> None:
        """Applies the LR-scheduler step of the last finished iteration (skipped if that iteration found inf/NaN).
        Called automatically before the next optimizer step; call it once after the last iteration."""
        if self._pending is None:
            return
        kind, val = self._pending
        self._pending = None
        if kind == "pinned":
            self._found_event.synchronize()
            flags = self._found_host.tolist()
        else:
            flags = val.tolist()
        if (self.world_size > 1 or self.level_parallel_emulation is not None) and self._presence_host is not None:
            counts = self._presence_host.tolist()
            check_gradient_presence(counts, self.world_size)
            self._took_part = [c > 0 for c in counts[0]]
            # tensors the device left alone in that step (nobody had a gradient): their host-side step count goes back
            listed = self._presence_params_late or []
            idle = [p for p, c in zip(listed, counts[0]) if c == 0]
            if idle:
                for opt in self.optimizers.values():
                    if isinstance(opt, SmallGroupAdam):
                        opt.rollback_params(idle)
        # the native table optimizers count their step on the host before the device decides to skip it: take the
        # count back for the groups that skipped (torch's fused Adam does the same with _foreach_sub_(steps, found_inf))
        for key, opt in self.optimizers.items():
            if isinstance(opt, (HashTableAdam, ShardedTableAdam, LevelParallelTableAdam, SmallGroupAdam)) and \
                    flags[self._found_groups.index(self.group_of[key])] != 0.0:
                opt.rollback_step()
        if not any(f != 0.0 for f in flags):
            for sch in self.schedulers.values():
                sch.step()


def _to_reference_numbering(layout, opt, small: Dict, table: Optional[Dict]) -> Dict:
    """``torch.optim.Adam.state_dict()`` of one group in the reference's numbering (``layout``, NeRSembleTrainer.
    group_layout): ``param_groups[0]["params"]`` lists EVERY position, ``state`` holds the positions that have been
    stepped -- the C tcnn encodings' flat moments with torch's per-parameter ``step``, the small tensors' -- and nothing for
    the members without a gradient (what a reference run writes: its Adam never creates state for them)."""
    small_ids = [int(i) for g in small["param_groups"] for i in g["params"]]
    small_pos = {id(p): i for i, p in zip(small_ids, (p for g in opt.param_groups for p in g["params"]))}
    state = {}
    for ref_id, (kind, x) in enumerate(layout):
        if kind == "table":
            if table is not None and int(table["step"]) > 0:
                state[ref_id] = {"step": torch.tensor(float(table["step"])), "exp_avg": table["exp_avg"][x],
                                 "exp_avg_sq": table["exp_avg_sq"][x]}
        elif kind == "param":
            st = small["state"].get(small_pos[id(x)])
            if st is not None:
                state[ref_id] = st
    groups = []
    for g in small["param_groups"]:
        g = dict(g)
        g["params"] = list(range(len(layout)))
        if table is not None:
            g["lr"] = float(table.get("lr", g["lr"]))
        groups.append(g)
    return {"state": state, "param_groups": groups[:1]}


def _from_reference_numbering(layout, saved: Dict, key: str):
    """The inverse: (table state for ``load_table_state`` or None, Adam state dict of the tensors this trainer's small
    optimizer steps).  The file's ids are matched BY POSITION in the reference's registration order -- a genuine
    reference checkpoint lists the parameter-free encodings and the frozen ``aabb`` in ``param_groups`` without a state
    entry.  Files this package wrote before round 4 numbered only the C encodings + the stepped tensors (and gave the
    empty encodings moments of zero elements in one synthetic layout): recognised by their count."""
    ids = [int(i) for g in saved["param_groups"] for i in g["params"]]
    state = {int(i): st for i, st in saved["state"].items()}
    n_table = sum(1 for kind, _ in layout if kind == "table")
    n_param = sum(1 for kind, _ in layout if kind == "param")
    if len(ids) == len(layout):
        kinds = layout
    else:
        # legacy numberings: drop ids whose moments have zero elements, then expect tables + stepped tensors only
        empty = {i for i in ids if i in state and torch.is_tensor(state[i].get("exp_avg")) and state[i]["exp_avg"].numel() == 0}
        ids = [i for i in ids if i not in empty]
        if n_table and len(ids) == n_param:
            kinds = [k for k in layout if k[0] == "param"]         # round 2: the tables under a key of their own
        elif len(ids) == n_table + n_param:
            kinds = [k for k in layout if k[0] != "none"]          # round 3: encodings + stepped tensors
        else:
            raise KeyError(f"optimizer state of group '{key}': {len(ids)} parameters in the checkpoint; this model's group has "
                           f"{len(layout)} members in the reference's numbering ({n_table} hash encodings, {n_param} "
                           f"stepped tensors, {len(layout) - n_table - n_param} without a gradient)")
    tab_ids, small_ids = [], []
    for i, (kind, x) in zip(ids, kinds):
        if kind == "table":
            tab_ids.append(i)
        elif kind == "param":
            small_ids.append(i)
        elif i in state and torch.is_tensor(state[i].get("exp_avg")) and state[i]["exp_avg"].numel() > 0:
            raise KeyError(f"optimizer state of group '{key}': position {i} holds moments, but the member there "
                           f"({tuple(x.shape)}) never has a gradient in this model -- not the same parameter layout")
    table = None
    if tab_ids:
        have = [i for i in tab_ids if i in state]
        if have and len(have) != len(tab_ids):
            raise KeyError(f"optimizer state of group '{key}': moments for {len(have)} of {len(tab_ids)} hash encodings")
        if have:
            table = {"step": int(state[tab_ids[0]]["step"]), "exp_avg": [state[i]["exp_avg"] for i in tab_ids],
                     "exp_avg_sq": [state[i]["exp_avg_sq"] for i in tab_ids]}
        else:
            table = {"step": 0, "exp_avg": None, "exp_avg_sq": None}
    pos = {old: new for new, old in enumerate(small_ids)}
    groups = []
    for g in saved["param_groups"][:1]:
        g = dict(g)
        g["params"] = list(range(len(small_ids)))
        groups.append(g)
    small_state = {pos[i]: state[i] for i in small_ids if i in state}
    return table, {"state": small_state, "param_groups": groups}
