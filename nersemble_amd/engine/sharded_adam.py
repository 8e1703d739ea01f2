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

Round 4: steps 1-2 run in ``n_buckets`` pieces.  Bucket k is piece k of EVERY rank's shard (``[world][shard / K]``
elements, expanded straight into that order by ``nsx_hash_grad_expand_f16_bucket``), so ``reduce_scatter(bucket k)`` hands
a rank piece k of its own shard -- the shard, the Adam pass and the all-gather are what they were -- and the expansion of
bucket k + 1 runs while bucket k is on the links; two bucket buffers alternate, the 0.8 GB dense gradient is gone
(2 / K of it).  The result is bit-identical to the one-piece exchange (same values, same reduction per element;
tests/test_parallel_cpu.py).  ``timing = True`` records HIP events around every phase (``comm_report``; bench.py's
``comm`` block).

Round 4, the exchange follows the coarse-to-fine window (``width_source``): while ``ceil(window) <= W < H`` the grids
``W ... H - 1`` have zero window weight, hence zero gradient and zero Adam moments (hash_ensemble.py:133-138), and neither
their gradient nor their (unchanged) working values need to travel.  Steps 1, 2 and 5 then move ``[entry][f][W]`` instead
of ``[entry][f][H]`` -- 1/32 of the bytes while one grid is on (steps 0 ... 40 000 of the reference's schedule, and the
configuration ``bench.py`` prices), 1/16 ... 1/2 along the ramp -- and step 4 touches W / H of the shard.  Master weights,
moments and working tables keep the full layout (nothing is handed over when W doubles); W never shrinks below the widest
exchange so far (moments, once non-zero, decay for ever).  Bit-identical to the full-width exchange, which remains the
path once the window has passed H / 2 (87 % of the schedule).
"""
import ctypes as C
import math
from typing import Callable, Optional

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
    def expand_f16_bucket(he: HashEnsemble, entry, out: torch.Tensor, scale: float, accumulate: bool, shard: int,
                          bucket: int, k: int, world: int) -> None:
        check(lib().nsx_hash_grad_expand_f16_bucket(ptr(entry["G"]), entry["n_rows"], ptr(entry["code"]),
                                                    entry["code"].stride(0), ptr(entry["window"]), he.n_hash_encodings,
                                                    C.byref(he.geom), ptr(out), float(scale), int(accumulate), int(shard),
                                                    int(bucket), int(k), int(world), stream()),
              "nsx_hash_grad_expand_f16_bucket")

    @staticmethod
    def expand_f16_bucket_width(he: HashEnsemble, entry, out: torch.Tensor, scale: float, accumulate: bool, shard: int,
                                bucket: int, k: int, world: int, width: int, beyond: Optional[torch.Tensor],
                                consume: bool = False) -> None:
        fn = lib().nsx_hash_grad_expand_f16_bucket_width_consume if consume else lib().nsx_hash_grad_expand_f16_bucket_width
        check(fn(ptr(entry["G"]), entry["n_rows"], ptr(entry["code"]),
                 entry["code"].stride(0), ptr(entry["window"]), he.n_hash_encodings, C.byref(he.geom), ptr(out), float(scale),
                 int(accumulate), int(shard), int(bucket), int(k), int(world), int(width), ptr(beyond), stream()),
              "nsx_hash_grad_expand_f16_bucket_width")

    @staticmethod
    def adam_f16grad_width(grad, n_entries, width, Hp, master, exp_avg, exp_avg_sq, f16, packed_out, lr, b1, b2, eps, step,
                           inv_scale, found_inf) -> None:
        check(lib().nsx_adam_dense_f16grad_width(ptr(grad), int(n_entries), int(width), int(Hp), ptr(master), ptr(exp_avg),
                                                 ptr(exp_avg_sq), ptr(f16), ptr(packed_out), lr, b1, b2, eps, int(step),
                                                 ptr(inv_scale), ptr(found_inf), stream()),
              "nsx_adam_dense_f16grad_width")

    @staticmethod
    def unpack_width(packed, n_entries, width, Hp, f16) -> None:
        check(lib().nsx_tables_unpack_width(ptr(packed), int(n_entries), int(width), int(Hp), ptr(f16), stream()),
              "nsx_tables_unpack_width")

    @staticmethod
    def check_finite_f16(x: torch.Tensor, found_inf: torch.Tensor) -> None:
        check(lib().nsx_check_finite_f16(ptr(x), x.numel(), ptr(found_inf), stream()), "nsx_check_finite_f16")

    @staticmethod
    def adam_f16grad(grad, n, master, exp_avg, exp_avg_sq, f16_out, lr, b1, b2, eps, step, inv_scale, found_inf) -> None:
        check(lib().nsx_adam_dense_f16grad(ptr(grad), int(n), ptr(master), ptr(exp_avg), ptr(exp_avg_sq), ptr(f16_out),
                                           lr, b1, b2, eps, int(step), ptr(inv_scale), ptr(found_inf), stream()),
              "nsx_adam_dense_f16grad")


class _Handles:
    """The work handles of a bucketed collective as one."""

    def __init__(self, handles):
        self.handles = [h for h in handles if h is not None]

    def wait(self):
        for h in self.handles:
            h.wait()


class ShardedTableAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad / weight decay) for ``HashEnsemble.tables`` with the state sharded over the ranks."""
    writes_half_tables = True      # HashEnsemble keeps its fp16 working copy; this optimizer refreshes it itself


    def __init__(self, hash_ensemble: HashEnsemble, lr: float = 5e-3, betas=(0.9, 0.999), eps: float = 1e-15,
                 world_size: int = 1, rank: int = 0, group=None, ops=None, overlap_reduce: bool = True,
                 n_buckets: int = 8, width_source: Optional[Callable[[], Optional[float]]] = None):
        self.he = hash_ensemble
        # the coarse-to-fine window of the CURRENT step as every rank's schedule gives it (None: every grid is on); the
        # exchange is restricted to the grids it has reached (module docstring).  Must be the same on all ranks.
        self.width_source = width_source
        self.Hp = int(hash_ensemble.tables.shape[-1])            # padded grid count of the [entry][f][Hp] layout
        self._max_width = 0           # widest exchange so far: grids below it may hold non-zero moments
        self._width = None            # width of the step in flight (decided with its first collective)
        self._last_width = self.Hp
        self._beyond = None           # device flag: a code was non-zero at a grid the narrow exchange left out
        self._beyond_pending = None
        self._beyond_host = None
        super().__init__([hash_ensemble.tables], dict(lr=lr, betas=betas, eps=eps))
        self.world_size, self.rank, self.group = int(world_size), int(rank), group
        self.ops = ops or NativeTableOps()
        hash_ensemble.grad_sink = F.FactoredGradSink()
        self.n = hash_ensemble.tables.numel()
        per = (self.n + self.world_size - 1) // self.world_size
        # pieces of the exchange (1: the one-piece exchange of rounds 1-3); every piece a multiple of SHARD_ALIGN elements
        self.n_buckets = max(1, int(n_buckets))
        while self.n_buckets > 1 and per < self.n_buckets * SHARD_ALIGN:
            self.n_buckets //= 2                                   # (small test tables: fewer, larger pieces)
        unit = SHARD_ALIGN * self.n_buckets
        self.shard = (per + unit - 1) // unit * unit
        self.bucket = self.shard // self.n_buckets
        self.timing = False           # HIP events around expand / reduce-scatter / shard Adam / all-gather (comm_report)
        self._events = []
        self.lo = self.rank * self.shard
        self.n_local = max(0, min(self.shard, self.n - self.lo))          # the last shards may be short or empty
        self._buf = None
        self._step = 0
        # the reduce-scatter starts inside the backward, as soon as the HashEnsemble's backward has completed G: it
        # then runs beside the deformation field's backward (which comes later in the graph) instead of after it
        self._early = None            # handle of a reduce-scatter already started for this step ("done" = finished)
        self.comm_stream = None       # the stream it runs on (the trainer hands in its optimizer stream; else one of its own)
        # narrow phases (exchange width W < H): fp32 master and both moments of the W active grids of this rank's entries
        # as CONTIGUOUS [entries][2][W] arrays -- the shard's Adam is then a dense pass over W / H of the bytes instead of
        # one 4 W-byte run per 128-byte line of the full layout (1.65 ms at W = 1 on a half-table shard: as long as the
        # full-width pass).  Written back into the full-layout shard when the width changes and before anybody reads it
        # (gather_master / table_state); the same adam_update on the same values: bit-identical.
        self._compact = None
        # round 6: while the HashEnsemble trains a compact copy of its first W grids (field_components/hash_ensemble.py:
        # the first-grid phase and the window ramp), the PACKED buffer of the narrow all-gather -- [entry][f][W] fp16 of all
        # ranks -- IS that copy's working table: the kernels read it with the H = W instances, the shard's Adam writes this
        # rank's piece, the all-gather the others', and nothing is unpacked into the 32-grid layout until the width changes
        hash_ensemble._compact_listeners = [self._on_compact]
        hash_ensemble.compact_from_f16 = True
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
                # one-piece exchange: the dense gradient (tail stays zero); bucketed: two alternating bucket buffers
                "grad_dense": torch.zeros((padded,), dtype=torch.float16, device=dev) if self.n_buckets == 1 else None,
                "buckets": [torch.zeros((self.world_size * self.bucket,), dtype=torch.float16, device=dev)
                            for _ in range(2)] if self.n_buckets > 1 else None,
                "grad_shard": torch.empty((self.shard,), dtype=torch.float16, device=dev),
                "exp_avg": torch.zeros((self.shard,), dtype=torch.float32, device=dev),
                "exp_avg_sq": torch.zeros((self.shard,), dtype=torch.float32, device=dev),
                "packed": None,       # narrow exchange: [world][shard entries][2][W] fp16 for the all-gather
            }
        return self._buf

    # ---- the width of this step's exchange --------------------------------------------------------------------------
    def _exchange_width(self) -> int:
        """Grids this step exchanges: the smallest power of two >= ceil(window), never below an earlier step's, H when
        no schedule is known or the exchange runs in one piece."""
        if self._width is not None:
            return self._width
        Hp = self.Hp
        need = Hp
        if self.width_source is not None and self.n_buckets > 1:
            w = self.width_source()
            if w is not None:
                need = 1
                while need < min(Hp, max(1, int(math.ceil(float(w))))):
                    need *= 2
        need = min(Hp, max(need, self._max_width))
        if 2 * need > Hp:
            need = Hp                 # (half of the bytes does not pay the pack / unpack passes)
        self._width = need
        return need

    def _packed(self, W: int) -> torch.Tensor:
        b = self._buffers()
        n = self.world_size * (self.shard // (2 * self.Hp)) * 2 * W
        if b["packed"] is None or b["packed"].numel() != n:
            b["packed"] = torch.zeros((n,), dtype=torch.float16, device=b["dev"])
        return b["packed"]

    def _raise_if_beyond(self) -> None:
        """The narrow exchange's premise, checked one step late (no host sync in the step): no conditioned code may be
        non-zero at a grid the exchange left out."""
        pend = self._beyond_pending
        if pend is None:
            return
        host, ev = pend
        if ev is not None and not ev.query():
            return                    # (not there yet: the flag is sticky, a later look finds it)
        self._beyond_pending = None
        if float(host[0]) != 0.0:
            raise RuntimeError("ShardedTableAdam: a hash grid beyond the exchanged width received a gradient -- the window "
                               "handed to the optimizer (width_source) is not the one the HashEnsemble was evaluated with")

    def _master_shard(self) -> torch.Tensor:
        return self.he.tables.data.reshape(-1)[self.lo:self.lo + self.n_local]

    # ---- compact state of the narrow phases ---------------------------------------------------------------------------
    def _full_views(self):
        """This rank's master / exp_avg / exp_avg_sq in the full layout as [local entries][2][Hp] views."""
        b = self._buffers()
        e = self.n_local // (2 * self.Hp)
        return [t[:e * 2 * self.Hp].view(e, 2, self.Hp) for t in (self._master_shard(), b["exp_avg"], b["exp_avg_sq"])]

    def _sync_compact(self) -> None:
        """Write the compact arrays back into the full-layout shard (they stay the working state)."""
        c = self._compact
        if c is None:
            return
        W = c["W"]
        for full, key in zip(self._full_views(), ("master", "exp_avg", "exp_avg_sq")):
            full[:, :, :W].copy_(c[key].view(full.shape[0], 2, W))

    def _compact_state(self, W: int) -> dict:
        """The contiguous fp32 state of grids [0, W) of this rank's entries; cut from the full layout on first use and
        handed over (written back, cut again) whenever the width changes."""
        c = self._compact
        if c is not None and c["W"] == W:
            return c
        self._sync_compact()
        views = self._full_views()
        self._compact = c = {"W": W}
        for full, key in zip(views, ("master", "exp_avg", "exp_avg_sq")):
            c[key] = full[:, :, :W].contiguous().view(-1)
        return c

    def _leave_compact(self) -> None:
        self._sync_compact()
        self._compact = None

    # ---- the HashEnsemble's compact copy (its listener) -----------------------------------------------------------------
    def compact_layouts_supported(self) -> bool:
        """Can the HashEnsemble train compact copies under this optimizer?  The exchange must be able to follow the window."""
        return self.width_source is not None and self.n_buckets > 1

    @torch.no_grad()
    def _on_compact(self, what: str) -> None:
        he = self.he
        comp = he._compact
        if comp is None:
            return
        W, per_entry = comp["width"], 2 * self.Hp
        n_entries = self.n // per_entry
        if what == "enter":
            if not self.compact_layouts_supported():
                raise RuntimeError("ShardedTableAdam: the HashEnsemble entered a compact layout, but this exchange cannot "
                                   "follow the window (no width_source, or a one-piece exchange)")
            packed = self._packed(W)
            n = n_entries * 2 * W
            packed[:n].copy_(comp["f16"].reshape(-1))          # every rank's current values (the tail stays zero)
            comp["f16"] = packed[:n].view(n_entries, 2, W)
            comp["sharded"] = True
            return
        # "sync" / "leave": the compact state goes back into the full layout -- this rank's master / moment shard, and the
        # fp16 working tables of ALL entries (every rank holds the gathered packed values)
        self._sync_compact()
        if comp.get("sharded"):
            b = self._buffers()
            self.ops.unpack_width(self._packed(W), n_entries, W, self.Hp, b["f16"])
            he.mark_half_synced()
        if what == "leave":
            self._compact = None

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
            self._mark("rs_end")
        elif early != "done":
            self._mark("rs_wait_begin")
            early.wait()                                         # the current stream waits for the collective
            self._mark("rs_wait_end")
        W = self._exchange_width()
        shard_grad = b["grad_shard"] if W == self.Hp else b["grad_shard"][:self.shard // self.Hp * W]
        self.ops.check_finite_f16(shard_grad, found_inf)
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
        self._mark("rs_end")
        self._early = "done"

    def _mark(self, what: str):
        """A timing event on the current stream (``timing`` only)."""
        if not self.timing or not self.he.tables.is_cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._events.append((what, ev))
        return ev

    def _expand_and_reduce(self, async_op: bool):
        """Dense fp16 gradient / world from the factored one, then the reduce-scatter (every rank joins, with zeros
        if it has no gradient) -- in ``n_buckets`` pieces: piece k + 1 is expanded while piece k is on the links."""
        he, b = self.he, self._buffers()
        entries = he.grad_sink.entries if he.grad_sink is not None else []
        scale = 1.0 / self.world_size
        self._raise_if_beyond()
        W = self._exchange_width()
        if W < self.Hp:
            return self._expand_and_reduce_narrow(W, entries, scale, async_op)
        if self.n_buckets == 1:
            self._mark("expand_begin")
            if not entries:
                b["grad_dense"][:self.n].zero_()
            for i, e in enumerate(entries):
                self.ops.expand_f16(he, e, b["grad_dense"], scale, i > 0)
            self._mark("expand_end")
            self._mark("rs_begin")
            return dist.reduce_scatter_tensor(b["grad_shard"], b["grad_dense"], op=dist.ReduceOp.SUM, group=self.group,
                                              async_op=async_op)
        handles = []
        for k in range(self.n_buckets):
            buf = b["buckets"][k % 2]
            if k >= 2 and handles[k - 2] is not None:
                handles[k - 2].wait()                          # (stream-side: the buffer's previous piece has left)
            self._mark("expand_begin")
            if not entries:
                buf.zero_()
            for i, e in enumerate(entries):
                self.ops.expand_f16_bucket(he, e, buf, scale, i > 0, self.shard, self.bucket, k, self.world_size)
            self._mark("expand_end")
            if k == 0:
                self._mark("rs_begin")
            out = b["grad_shard"][k * self.bucket:(k + 1) * self.bucket]
            handles.append(dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        return _Handles(handles) if async_op else None

    def _narrow_pieces(self, W: int) -> int:
        """Reduce-scatter calls of a step whose exchange is W of H grids wide (1 ... n_buckets)."""
        return max(1, (self.n_buckets * W) // self.Hp)

    def _expand_and_reduce_narrow(self, W: int, entries, scale: float, async_op: bool):
        """The bucketed exchange on the grids [0, W): the same pieces, each W / H as long."""
        he, b = self.he, self._buffers()
        per_entry = 2 * self.Hp
        # pieces stay about as long as the full-width ones: K W / H of them, each H / (K W) x as many entries -- at W = 1 the
        # whole exchange is ONE 12.6 MB reduce-scatter instead of eight of 1.6 MB, whose launch latencies (not their bytes)
        # were what the links saw.  The entries keep their order in ``grad_shard`` (piece-major = ascending either way).
        n_pieces = self._narrow_pieces(W)
        bucket = self.bucket * (self.n_buckets // n_pieces)
        be = bucket // per_entry                                 # entries per piece and rank
        n_piece = be * 2 * W
        if self._beyond is None or self._beyond.device != b["dev"]:
            self._beyond = torch.zeros((1,), dtype=torch.float32, device=b["dev"])
        # the expansion clears the planes it reads (every entry is in exactly one piece): no 1.2 GB fill in front of the next
        # backward's scatter -- while the scatter touches a minority of G's sectors (as HashTableAdam decides for its pass)
        sink = he.grad_sink
        consume = (self.consume_gradient and len(entries) == 1 and self.Hp >= 8 and W <= 16 and entries[0]["G"].is_cuda
                   and sink.is_persistent(entries[0]["G"])
                   and (0 < sink.samples_scattered * 80 < self.consume_density_limit * (entries[0]["G"].numel() // 8)
                        or entries[0]["G"].numel() * 4 <= self.consume_always_below_bytes))
        handles = []
        for k in range(n_pieces):
            buf = b["buckets"][k % 2][:self.world_size * n_piece]
            if k >= 2 and handles[k - 2] is not None:
                handles[k - 2].wait()
            self._mark("expand_begin")
            if not entries:
                buf.zero_()
            for i, e in enumerate(entries):
                if he._compact is not None and he.is_first_grid_code(e["code"]):
                    # (the compact first-grid phase runs its kernels with a [rows, 1] code of ones: the expansion reads a
                    # table over all grids)
                    e = dict(e, code=he.first_grid_code_full(e["n_rows"]))
                self.ops.expand_f16_bucket_width(he, e, buf, scale, i > 0, self.shard, bucket, k, self.world_size, W,
                                                 self._beyond, **({"consume": True} if consume else {}))
            self._mark("expand_end")
            if k == 0:
                self._mark("rs_begin")
            out = b["grad_shard"][k * n_piece:(k + 1) * n_piece]
            handles.append(dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        if consume:
            sink.mark_cleared(entries[0]["G"])           # (an event on the stream the expansions ran on)
        return _Handles(handles) if async_op else None

    consume_gradient = True
    consume_density_limit = 0.5
    consume_always_below_bytes = 1 << 29          # (a G of a few planes: HashEnsemble.first_grid_planes)

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
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(dev)
        comm = self.comm_stream
        comm.wait_stream(torch.cuda.current_stream(dev))         # codes / flags are ready on the backward's stream
        he.grad_sink.wait_scatter(comm)                          # G is complete on the scatter's stream
        with torch.cuda.stream(comm):
            self._early = self._expand_and_reduce(async_op=True)
            if self.timing:
                self._early.wait()                               # (stream-side wait: the event behind it marks completion)
                self._mark("rs_end")
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
        W = self._exchange_width()
        self._mark("adam_begin")
        if W == self.Hp:
            if he._compact is not None:
                he.leave_first_grid_phase()              # (the window has passed H / 2: back to the 32-grid layout)
            self._leave_compact()
            if self.n_local > 0:
                self.ops.adam_f16grad(b["grad_shard"], self.n_local, self._master_shard(), b["exp_avg"], b["exp_avg_sq"],
                                      b["f16"][self.lo:self.lo + self.shard], group["lr"], b1, b2, group["eps"], self._step,
                                      inv_scale, found_inf)
            self._mark("adam_end")
            dist.all_gather_into_tensor(b["f16"], b["f16"][self.lo:self.lo + self.shard], group=self.group)
        else:
            per_entry = 2 * self.Hp
            se = self.shard // per_entry                          # entries per shard
            packed = self._packed(W)
            mine = packed[self.rank * se * 2 * W:(self.rank + 1) * se * 2 * W]
            if self.n_local > 0:
                # dense Adam over the compact [entries][2][W] state; its fp16 output IS this rank's packed piece of the
                # all-gather (the full-layout working tables take every rank's values from unpack_width below)
                c = self._compact_state(W)
                n_c = (self.n_local // per_entry) * 2 * W
                self.ops.adam_f16grad(b["grad_shard"][:n_c], n_c, c["master"], c["exp_avg"], c["exp_avg_sq"], mine[:n_c],
                                      group["lr"], b1, b2, group["eps"], self._step, inv_scale, found_inf)
            self._mark("adam_end")
            dist.all_gather_into_tensor(packed, mine, group=self.group)
            comp = he._compact
            if comp is not None and comp.get("sharded"):
                if comp["width"] != W:
                    raise RuntimeError(f"ShardedTableAdam: the HashEnsemble trains a compact copy of {comp['width']} grids, "
                                       f"the exchange is {W} wide")
                # (the gathered buffer IS the compact copy's working table: nothing to unpack)
            else:
                self.ops.unpack_width(packed, self.n // per_entry, W, self.Hp, b["f16"])
            if self._beyond is not None and self._beyond_pending is None:
                if self._beyond.is_cuda:
                    if self._beyond_host is None:
                        self._beyond_host = torch.zeros((1,), dtype=torch.float32).pin_memory()
                    self._beyond_host.copy_(self._beyond, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._beyond_pending = (self._beyond_host, ev)
                else:
                    self._beyond_pending = (self._beyond.clone(), None)
        self._mark("ag_end")
        self._max_width = max(self._max_width, W)
        he.min_compact_width = self._max_width
        self._last_width, self._width = W, None
        if he.grad_sink is not None:
            he.grad_sink.clear()
        he.mark_half_synced()

    def rollback_step(self) -> None:
        """The last ``step()`` was skipped on the device (inf/NaN): it must not count (torch.optim.Adam semantics)."""
        self._step = max(0, self._step - 1)

    def comm_report(self, reset: bool = True) -> dict:
        """Mean per-step durations (ms) of the exchange's phases from the events recorded while ``timing`` was on; call
        after a device synchronisation.  ``reduce_scatter_ms`` spans the first piece's issue to the last piece's
        completion on the communication stream (the expansions of the later pieces run inside it),
        ``reduce_scatter_exposed_ms`` is what the main stream waited for it, ``bus_GBps`` prices (W - 1) / W of the
        exchanged bytes per rank against the span."""
        ev, acc, n_steps = self._events, {}, 0
        spans = {"expand_f16_ms": ("expand_begin", "expand_end"), "reduce_scatter_ms": ("rs_begin", "rs_end"),
                 "reduce_scatter_exposed_ms": ("rs_wait_begin", "rs_wait_end"), "shard_adam_ms": ("adam_begin", "adam_end"),
                 "all_gather_ms": ("adam_end", "ag_end")}
        for name, (a, b) in spans.items():
            total, open_ev = 0.0, None
            for what, e in ev:
                if what == a:
                    open_ev = e
                elif what == b and open_ev is not None:
                    total += open_ev.elapsed_time(e)
                    open_ev = None
            acc[name] = total
        n_steps = max(1, sum(1 for what, _ in ev if what == "adam_begin"))
        out = {k: v / n_steps for k, v in acc.items()}
        W = self.world_size
        # fp16 pieces leaving / arriving per rank, at the width of the LAST step's exchange
        bytes_rs = (W - 1) / W * self.shard * W * 2 * self._last_width / self.Hp
        lw = self._last_width
        out.update(steps=n_steps, buckets=self.n_buckets, world_size=W,
                   reduce_scatter_calls=self.n_buckets if (lw is None or lw >= self.Hp) else self._narrow_pieces(int(lw)),
                   exchange_width=self._last_width, grids=self.Hp,
                   reduce_scatter_bytes_per_rank=bytes_rs, all_gather_bytes_per_rank=bytes_rs,
                   reduce_scatter_bus_GBps=(bytes_rs / (out["reduce_scatter_ms"] * 1e-3) / 1e9)
                   if out["reduce_scatter_ms"] > 0 else None,
                   all_gather_bus_GBps=(bytes_rs / (out["all_gather_ms"] * 1e-3) / 1e9) if out["all_gather_ms"] > 0 else None)
        if reset:
            self._events = []
        return out

    def clear_grads(self) -> None:
        self.zero_grad(set_to_none=True)

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)
        early, self._early = self._early, None
        if early is not None and early != "done":
            early.wait()                          # a reduce-scatter nobody consumed: let it finish before G is reused
        self._width = None
        if self.he.grad_sink is not None:
            self.he.grad_sink.clear()

    # ---- checkpointing ------------------------------------------------------------------------------------------
    def _gather_shards(self, mine_local: torch.Tensor) -> torch.Tensor:
        """All ranks' shards of one fp32 per-parameter array -> the full array, shaped like ``tables``."""
        self._sync_compact()
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
        self._compact = None                         # (what is loaded replaces the full-layout state)
        self._step = int(state["step"])
        self._max_width = self.Hp if self._step > 0 else 0           # (which grids hold moments is not recorded)
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
        self._sync_compact()
        p = self.he.tables
        full = torch.zeros((self.shard * self.world_size,), dtype=torch.float32, device=p.device)
        mine = full[self.lo:self.lo + self.shard]
        mine[:self.n_local].copy_(self._master_shard())
        dist.all_gather_into_tensor(full, mine, group=self.group)
        p.data.reshape(-1).copy_(full[:self.n])
