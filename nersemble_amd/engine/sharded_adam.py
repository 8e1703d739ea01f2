"""Data-parallel optimizer step for the hash tables: reduce-scatter -> sharded Adam -> all-gather (RCCL over xGMI).

The reference is single-GPU (scripts/train/train_nersemble.py:272-274); SURVEY.md 8(e) asks for ray-sharded data
parallelism with a gradient exchange.  A dense fp32 all-reduce of the 403 M-parameter table gradient moves
2 * 7/8 * 1.6 GB per GPU and step and then every rank repeats the same 12 GB Adam pass.  Instead (ZeRO-1 layout,
sized for xGMI's point-to-point links):

  1. every rank expands its factored gradient G (functional.FactoredGradSink) into a dense **fp16** gradient, already
     divided by the world size (tcnn's own table gradients are fp16; the GradScaler's loss scale keeps the range),
  2. ``reduce_scatter_tensor``: rank r receives the summed 1/W shard r                       (7/8 * 0.8 GB per GPU),
  3. inf/NaN check of the shard, MAX-all-reduced together with the other groups' flags (GradScaler semantics are
     global: a step is skipped on every rank or on none),
  4. Adam on the shard only -- fp32 master weights and both moments exist once per node, not once per GPU
     (1/W of the 12 GB pass),
  5. ``all_gather_into_tensor`` of the fp16 working tables, in place                         (7/8 * 0.8 GB per GPU).

Per GPU and step that is 1.4 GB over the links instead of 2.8 GB, and 1.5 GB instead of 12 GB of optimizer traffic.
The fp32 master copy in ``HashEnsemble.tables`` is authoritative only inside the rank's shard;
``gather_master()`` rebuilds the full tensor (checkpointing).
"""
import ctypes as C
from typing import Optional

import torch
import torch.distributed as dist

from .. import functional as F
from .._lib import check, lib, ptr, stream
from ..field_components.hash_ensemble import HashEnsemble

SHARD_ALIGN = 1024            # elements; keeps every shard 16-byte aligned in fp16 and fp32


class NativeTableOps:
    """The three libnsx kernels of the sharded step (tests substitute a torch restatement to run the collective
    plumbing on CPU with gloo)."""

    @staticmethod
    def expand_f16(he: HashEnsemble, entry, out: torch.Tensor, scale: float, accumulate: bool) -> None:
        check(lib().nsx_hash_grad_expand_f16(ptr(entry["G"]), entry["n_rows"], ptr(entry["code"]), entry["code"].stride(0),
                                             ptr(entry["window"]), he.n_hash_encodings, C.byref(he.geom), ptr(out),
                                             float(scale), int(accumulate), stream()), "nsx_hash_grad_expand_f16")

    @staticmethod
    def check_finite_f16(x: torch.Tensor, found_inf: torch.Tensor) -> None:
        check(lib().nsx_check_finite_f16(ptr(x), x.numel(), ptr(found_inf), stream()), "nsx_check_finite_f16")

    @staticmethod
    def adam_f16grad(grad, n, master, exp_avg, exp_avg_sq, f16_out, lr, b1, b2, eps, step, inv_scale, found_inf) -> None:
        check(lib().nsx_adam_dense_f16grad(ptr(grad), int(n), ptr(master), ptr(exp_avg), ptr(exp_avg_sq), ptr(f16_out),
                                           lr, b1, b2, eps, int(step), ptr(inv_scale), ptr(found_inf), stream()),
              "nsx_adam_dense_f16grad")


class ShardedTableAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad / weight decay) for ``HashEnsemble.tables`` with the state sharded over the ranks."""
    writes_half_tables = True      # HashEnsemble keeps its fp16 working copy; this optimizer refreshes it itself


    def __init__(self, hash_ensemble: HashEnsemble, lr: float = 5e-3, betas=(0.9, 0.999), eps: float = 1e-15,
                 world_size: int = 1, rank: int = 0, group=None, ops=None, overlap_reduce: bool = True):
        self.he = hash_ensemble
        super().__init__([hash_ensemble.tables], dict(lr=lr, betas=betas, eps=eps))
        self.world_size, self.rank, self.group = int(world_size), int(rank), group
        self.ops = ops or NativeTableOps()
        hash_ensemble.grad_sink = F.FactoredGradSink()
        self.n = hash_ensemble.tables.numel()
        per = (self.n + self.world_size - 1) // self.world_size
        self.shard = (per + SHARD_ALIGN - 1) // SHARD_ALIGN * SHARD_ALIGN
        self.lo = self.rank * self.shard
        self.n_local = max(0, min(self.shard, self.n - self.lo))          # the last shards may be short or empty
        self._buf = None
        self._step = 0
        # the reduce-scatter starts inside the backward, as soon as the HashEnsemble's backward has completed G: it
        # then runs beside the deformation field's backward (which comes later in the graph) instead of after it
        self._early = None            # handle of a reduce-scatter already started for this step ("done" = finished)
        self._comm_stream = None
        if overlap_reduce:
            hash_ensemble.grad_sink.on_complete = self._start_reduce

    # ---- buffers ------------------------------------------------------------------------------------------------
    def _buffers(self):
        p = self.he.tables
        if self._buf is None or self._buf["dev"] != p.device:
            dev, padded = p.device, self.shard * self.world_size
            f16_padded = torch.zeros((padded,), dtype=torch.float16, device=dev)
            f16_padded[:self.n].copy_(p.detach().reshape(-1))
            # the forward kernels read the working tables through HashEnsemble.tables_f16: make it a view of the
            # all-gather buffer
            self.he.tables_f16 = f16_padded[:self.n].view(p.shape)
            self.he.mark_half_synced()
            self._buf = {
                "dev": dev,
                "f16": f16_padded,
                "grad_dense": torch.zeros((padded,), dtype=torch.float16, device=dev),     # tail stays zero
                "grad_shard": torch.empty((self.shard,), dtype=torch.float16, device=dev),
                "exp_avg": torch.zeros((self.shard,), dtype=torch.float32, device=dev),
                "exp_avg_sq": torch.zeros((self.shard,), dtype=torch.float32, device=dev),
            }
        return self._buf

    def _master_shard(self) -> torch.Tensor:
        return self.he.tables.data.reshape(-1)[self.lo:self.lo + self.n_local]

    # ---- step, in the two phases the trainer runs for every optimizer ----------------------------------------------
    @torch.no_grad()
    def check_finite(self, found_inf: torch.Tensor) -> None:
        """Expand + reduce-scatter the table gradient; found_inf[0] = 1 if this rank's shard (or a value this rank
        added to its own G) is inf/NaN.  The trainer MAX-reduces the flags over the ranks afterwards."""
        he, b = self.he, self._buffers()
        sink = he.grad_sink
        entries = sink.entries if sink is not None else []
        if he.tables.grad is not None:
            raise RuntimeError("ShardedTableAdam consumes the factored gradient; a dense .grad on the tables means the "
                               "HashEnsemble ran without time_code_index (not a data-parallel training configuration)")
        early, self._early = self._early, None
        if early is None:
            if sink is not None and he.tables.is_cuda:
                sink.wait_scatter()
            self._expand_and_reduce(async_op=False)              # not started from the backward: do it here
        elif early != "done":
            early.wait()                                         # the current stream waits for the collective
        self.ops.check_finite_f16(b["grad_shard"], found_inf)
        if entries and sink.nonfinite is not None:
            torch.maximum(found_inf, sink.nonfinite.to(found_inf.dtype), out=found_inf)

    @torch.no_grad()
    def ensure_reduce_started(self) -> None:
        """Every rank must issue its collectives in the same order: the table reduce-scatter (started from inside the
        backward, ``_start_reduce``), THEN the all-reduce of the small gradients (``NeRSembleTrainer._all_reduce_grads``).
        A rank whose backward never completed a G -- its rays produced no samples this step -- joins the reduce-scatter
        here, with zeros, before the trainer goes on to the small gradients; without this it would meet the other ranks'
        reduce-scatter with its all-reduce."""
        if self._early is not None:
            return
        sink = self.he.grad_sink
        if sink is not None and self.he.tables.is_cuda:
            sink.wait_scatter()
        self._expand_and_reduce(async_op=False)
        self._early = "done"

    def _expand_and_reduce(self, async_op: bool):
        """Dense fp16 gradient / world from the factored one, then the reduce-scatter (every rank joins, with zeros
        if it has no gradient)."""
        he, b = self.he, self._buffers()
        entries = he.grad_sink.entries if he.grad_sink is not None else []
        if not entries:
            b["grad_dense"][:self.n].zero_()
        for i, e in enumerate(entries):
            self.ops.expand_f16(he, e, b["grad_dense"], 1.0 / self.world_size, i > 0)
        return dist.reduce_scatter_tensor(b["grad_shard"], b["grad_dense"], op=dist.ReduceOp.SUM, group=self.group,
                                          async_op=async_op)

    @torch.no_grad()
    def _start_reduce(self) -> None:
        """``FactoredGradSink.on_complete``: called from the HashEnsemble's backward when G holds the whole step."""
        he = self.he
        self._buffers()
        if not he.tables.is_cuda:
            self._expand_and_reduce(async_op=False)
            self._early = "done"
            return
        dev = he.tables.device
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(dev)
        comm = self._comm_stream
        comm.wait_stream(torch.cuda.current_stream(dev))         # codes / flags are ready on the backward's stream
        he.grad_sink.wait_scatter(comm)                          # G is complete on the scatter's stream
        with torch.cuda.stream(comm):
            self._early = self._expand_and_reduce(async_op=True)
        for e in he.grad_sink.entries:                           # allocated on the main stream, read on this one
            for t in (e["G"], e["code"], e["window"]):
                if t is not None:
                    t.record_stream(comm)

    @torch.no_grad()
    def step(self, found_inf: Optional[torch.Tensor] = None, inv_scale: Optional[torch.Tensor] = None,
             side_stream: Optional["torch.cuda.Stream"] = None):
        """Adam on the shard + all-gather of the working tables.  ``side_stream``: run both there -- nothing else in the
        step's tail nor the next step's ray marching needs the tables; ``HashEnsemble.wait_tables`` orders the next
        reader after the all-gather (same scheme as HashTableAdam.step)."""
        he = self.he
        if side_stream is None or not he.tables.is_cuda:
            return self._step_now(found_inf, inv_scale)
        he.wait_tables()
        side_stream.wait_stream(torch.cuda.current_stream(he.tables.device))
        with torch.cuda.stream(side_stream):
            self._step_now(found_inf, inv_scale)
            done = torch.cuda.Event()
            done.record(side_stream)
        for t in (found_inf, inv_scale):
            if t is not None:
                t.record_stream(side_stream)
        he._tables_ready = done

    def _step_now(self, found_inf, inv_scale):
        he, b = self.he, self._buffers()
        group = self.param_groups[0]
        self._step += 1
        b1, b2 = group["betas"]
        if self.n_local > 0:
            self.ops.adam_f16grad(b["grad_shard"], self.n_local, self._master_shard(), b["exp_avg"], b["exp_avg_sq"],
                                  b["f16"][self.lo:self.lo + self.shard], group["lr"], b1, b2, group["eps"], self._step,
                                  inv_scale, found_inf)
        dist.all_gather_into_tensor(b["f16"], b["f16"][self.lo:self.lo + self.shard], group=self.group)
        if he.grad_sink is not None:
            he.grad_sink.clear()
        he.mark_half_synced()

    def rollback_step(self) -> None:
        """The last ``step()`` was skipped on the device (inf/NaN): it must not count (torch.optim.Adam semantics)."""
        self._step = max(0, self._step - 1)

    def clear_grads(self) -> None:
        self.zero_grad(set_to_none=True)

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        early, self._early = self._early, None
        if early is not None and early != "done":
            early.wait()                          # a reduce-scatter nobody consumed: let it finish before G is reused
        if self.he.grad_sink is not None:
            self.he.grad_sink.clear()

    # ---- checkpointing ------------------------------------------------------------------------------------------
    def _gather_shards(self, mine_local: torch.Tensor) -> torch.Tensor:
        """All ranks' shards of one fp32 per-parameter array -> the full array, shaped like ``tables``."""
        p = self.he.tables
        full = torch.zeros((self.shard * self.world_size,), dtype=torch.float32, device=p.device)
        mine = full[self.lo:self.lo + self.shard]
        mine[:self.n_local].copy_(mine_local[:self.n_local])
        dist.all_gather_into_tensor(full, mine, group=self.group)
        return full[:self.n].view(p.shape)

    @torch.no_grad()
    def table_state(self) -> dict:
        """Collective (every rank calls it): the moments of ALL shards in the reference's parameter layout, the step
        count and the learning rate -- what ``HashTableAdam.table_state`` returns on one GPU."""
        self.he.wait_tables()
        b = self._buffers()
        return {"step": int(self._step), "lr": float(self.param_groups[0]["lr"]),
                "exp_avg": self.he.to_tcnn_layout(self._gather_shards(b["exp_avg"])),
                "exp_avg_sq": self.he.to_tcnn_layout(self._gather_shards(b["exp_avg_sq"]))}

    @torch.no_grad()
    def load_table_state(self, state: dict) -> None:
        self.he.wait_tables()
        b = self._buffers()
        self._step = int(state["step"])
        self.param_groups[0]["lr"] = float(state.get("lr", self.param_groups[0]["lr"]))
        for key in ("exp_avg", "exp_avg_sq"):
            b[key].zero_()
            if state.get(key) is None:                   # a checkpoint written before the first optimizer step
                continue
            full = self.he.from_tcnn_layout(state[key]).reshape(-1)
            b[key][:self.n_local].copy_(full[self.lo:self.lo + self.n_local])
        # the working tables / master shard follow the (already loaded) model parameters
        b["f16"][:self.n].copy_(self.he.tables.detach().reshape(-1))
        self.he.mark_half_synced()

    @torch.no_grad()
    def gather_master(self) -> None:
        """Rebuild the full fp32 master tables on every rank from the shards (call before ``state_dict()``)."""
        self.he.wait_tables()                    # a step may still be running on the optimizer stream
        p = self.he.tables
        full = torch.zeros((self.shard * self.world_size,), dtype=torch.float32, device=p.device)
        mine = full[self.lo:self.lo + self.shard]
        mine[:self.n_local].copy_(self._master_shard())
        dist.all_gather_into_tensor(full, mine, group=self.group)
        p.data.reshape(-1).copy_(full[:self.n])
