"""Level-parallel HashEnsemble for data-parallel runs with the coarse-to-fine window open: the exchange moves SAMPLES, not
parameters.

The reference is single-GPU (scripts/train/train_nersemble.py:272-274); ``engine/sharded_adam.py`` is this package's
data-parallel contract for the hash tables: every rank holds the whole fp16 working table, the factored gradient is expanded
to a dense fp16 gradient, reduce-scattered, stepped on a 1 / W shard, and the new values are all-gathered -- 2 x (W - 1) / W x
806 MB per rank and step at the reference geometry, whatever the batch kept.  From step 80 000 of 300 001
(train_nersemble.py:77-78, 158) every grid is on and a step keeps ~10^5 samples per rank: the exchange, not the arithmetic,
bounds the step (~4.5 x at 8 GPUs against a target of 6 x).

Here the tables are partitioned by LEVEL of the multi-resolution grid -- rank r owns levels [r L / W, (r + 1) L / W) of all H
grids: a contiguous entry range of the ``[entry][f][h]`` layout -- and nothing that scales with the table ever travels:

  forward   (hash_ensemble.py:93-158: out[:, 2 l + f] = sum_h code[h] T_h[l](x)[f] -- the columns of a level depend on that
            level's entries only)
            all-gather (x, code slot) of every rank's samples (16 B / sample), each rank evaluates ITS levels for ALL samples
            with the unchanged kernels on a sub-geometry (``nsx_hash_ensemble_fwd``: [S_all, 2 L / W] fp16), all-to-all of the
            column blocks back to the samples' owners (2 B x 2 L (W - 1) / W per sample): the owner holds [S, 2 L] as before.
  backward  all-to-all of dL/dfeatures by column block (fp16: ``nsx_mlp_bwd`` produces fp16 values), every rank runs the
            unchanged backward kernel on its levels for all samples: the factored table gradient G of ITS entries is complete
            where its optimizer state lives -- no gradient exchange --, the partial dL/dx (12 B / sample) returns by
            all-to-all and is summed, the partial code gradients ([rows, H] floats) by all-reduce.
  optimizer ``nsx_adam_hash_factored`` on the rank's entry range (1 / W of the pass), no all-gather: nobody else reads these
            entries.  The gradient planes are indexed by (source rank, code row); the conditioned code rows travel with the
            samples (a few KB).

Work per rank is what it was -- (W S) samples x (L / W) levels = S x L corner gathers / scatters against whole 128-byte
entries, with a table slice (134 MB at W = 8) that stays in the 256 MB Infinity Cache -- and the bytes on the links scale with
the samples: ~(16 + 2 x 64 x (W - 1) / W + 12) B per sample and pass instead of 1.41 GB per step.

Why levels and not grids (the blend is also linear in h, the route VERDICT r04 suggested): the table gradient is factored
through the code slot, ``dL/dT[e][f][h] = sum_slot G[slot][e][f] code[slot][h]``, and G does not depend on h -- a rank that owns
grids {r, r + W, ...} needs the scatter of ALL samples into a G of ALL entries: W x the step's atomic-bound kernel on every
rank.  Partitioned by level every (sample, level, corner) item is scattered exactly once in the job.

``tests/test_sharded_gpu.py`` holds a two-rank run (two processes, one GPU, gloo) to the single process on the union batch.
"""
import ctypes as C
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _lib
from .. import functional as F
from .._lib import GridGeom, check, lib, ptr, stream
from ..field_components.hash_ensemble import HashEnsemble

MAX_ADAM_SLOTS = 192          # NSX_MAX_ADAM_SLOTS (include/nsx.h): gradient planes one optimizer pass can read


def sub_geometry(g: GridGeom, first_level: int, n_levels: int) -> GridGeom:
    """The geometry of levels [first_level, first_level + n_levels) as a grid of its own: entry offsets re-based to 0, so that
    the unchanged kernels run on a slice of the tables / gradient planes / optimizer state."""
    s = GridGeom()
    s.n_levels, s.log2_hashmap_size = n_levels, g.log2_hashmap_size
    s.base_resolution, s.per_level_scale = g.base_resolution, g.per_level_scale
    base = int(g.offset[first_level])
    for i in range(n_levels):
        l = first_level + i
        s.scale[i], s.res[i], s.size[i], s.hashed[i] = g.scale[l], g.res[l], g.size[l], g.hashed[l]
        s.offset[i] = int(g.offset[l]) - base
    s.offset[n_levels] = int(g.offset[first_level + n_levels]) - base
    return s


def level_assignment(n_levels: int, world_size: int, balanced: bool = True) -> List[List[int]]:
    """Which levels every rank owns.  ``balanced`` (the default): the levels are dealt to the ranks in snake order -- 0 .. W - 1
    forward, W .. 2 W - 1 backward, ... -- so that at W = 8 rank r holds levels r and 15 - r.  A level's cost grows with its
    resolution: the coarse levels' samples share cells (their gathers hit L2, their gradient atomics merge), the finest
    levels' do not -- measured at 8 ranks with contiguous pairs: forward 1.11 ms (levels 0, 1) against 2.20 (levels 14, 15),
    backward 2.34 against 4.39 over the benchmark window (profiles/r06_contiguous_levels/) -- and the step waits for the
    slowest rank.  The snake pairs the cheapest level with the dearest.  ``balanced=False``: contiguous blocks (round 5)."""
    n_own = n_levels // world_size
    if not balanced:
        return [list(range(r * n_own, (r + 1) * n_own)) for r in range(world_size)]
    own = [[] for _ in range(world_size)]
    for i in range(n_own):
        order = range(world_size) if i % 2 == 0 else range(world_size - 1, -1, -1)
        for k, r in enumerate(order):
            own[r].append(i * world_size + k)
    return [sorted(v) for v in own]


def sub_geometry_levels(g: GridGeom, levels: List[int]) -> GridGeom:
    """The geometry of the given levels as a grid of its own, their entries laid out one level after the other from 0: the
    unchanged kernels run on a rank's COMPACT tables / gradient planes / optimizer state."""
    s = GridGeom()
    s.n_levels, s.log2_hashmap_size = len(levels), g.log2_hashmap_size
    s.base_resolution, s.per_level_scale = g.base_resolution, g.per_level_scale
    at = 0
    for i, l in enumerate(levels):
        s.scale[i], s.res[i], s.size[i], s.hashed[i] = g.scale[l], g.res[l], g.size[l], g.hashed[l]
        s.offset[i] = at
        at += int(g.offset[l + 1]) - int(g.offset[l])
    s.offset[len(levels)] = at
    return s


def _as(t: torch.Tensor, dtype) -> torch.Tensor:
    """``t.detach().to(dtype).contiguous()`` without the three dispatches when there is nothing to do."""
    if t.dtype == dtype and t.is_contiguous() and not t.requires_grad:
        return t
    return t.detach().to(dtype).contiguous()


NATIVE_COLLECTIVES: Optional[bool] = None     # None: the library's own collectives wherever the group is an RCCL one; False:
#                                               torch.distributed everywhere (bench.py --lp-torch-collectives, the A/B switch)


class NativeLPOps:
    """The device side of the exchange: csrc/level_parallel.hip through the C ABI (include/nsx.h, "level-parallel exchange").
    Tests substitute a torch restatement to run the collectives' plumbing on CPU over gloo (tests/test_parallel_cpu.py)."""

    @staticmethod
    def layout(W: int, S_cap: int, R_cap: int, H: int, n2: int):
        lay = _lib.step_struct("nsx_lp_layout")()
        check(lib().nsx_lp_layout_make(int(W), int(S_cap), int(R_cap), int(H), int(n2), C.byref(lay)), "nsx_lp_layout_make")
        return lay

    @staticmethod
    def fwd_pack(lay, pn, slot, S, n_dev, codes, rows, payload) -> None:
        check(lib().nsx_lp_fwd_pack(C.byref(lay), ptr(pn), ptr(slot), int(S), ptr(n_dev), ptr(codes), codes.stride(0),
                                    int(rows), ptr(payload), stream()), "nsx_lp_fwd_pack")

    @staticmethod
    def fwd_run(lay, gathered, ex, tables, geom, window, send, codes_packed) -> None:
        check(lib().nsx_lp_fwd_run(C.byref(lay), ptr(gathered), ex.sizes_host, ex.rows_host, ptr(tables), C.byref(geom),
                                   ptr(window), ptr(send), ptr(codes_packed), stream()), "nsx_lp_fwd_run")

    @staticmethod
    def fwd_unpack(lay, recv, S, n_dev, feats) -> None:
        check(lib().nsx_lp_fwd_unpack(C.byref(lay), ptr(recv), int(S), ptr(n_dev), ptr(feats), stream()), "nsx_lp_fwd_unpack")

    @staticmethod
    def bwd_pack(lay, dout, pn, slot, S, n_dev, send) -> None:
        check(lib().nsx_lp_bwd_pack(C.byref(lay), ptr(dout), ptr(pn), ptr(slot), int(S), ptr(n_dev), ptr(send), stream()),
              "nsx_lp_bwd_pack")

    @staticmethod
    def bwd_run(lay, recv, gathered, ex, tables, geom, window, G, ret, nonfinite, dz) -> None:
        dev = recv.device
        check(lib().nsx_lp_bwd_run(C.byref(lay), ptr(recv), ptr(gathered), ex.sizes_host, ex.rows_host, ptr(tables),
                                   C.byref(geom), ptr(window), ptr(G), ptr(dz), ptr(F.codesum_scratch(lay.R_cap, lay.H, dev)),
                                   ptr(ret), ptr(nonfinite), stream()), "nsx_lp_bwd_run")

    @staticmethod
    def bwd_unpack(lay, ret_recv, S, n_dev, rows, dx, dcode) -> None:
        check(lib().nsx_lp_bwd_unpack(C.byref(lay), ptr(ret_recv), int(S), ptr(n_dev), int(rows), ptr(dx), ptr(dcode),
                                      stream()), "nsx_lp_bwd_unpack")

    # the same six steps with their collectives issued from C (csrc/comm.hip): one call per direction
    @staticmethod
    def forward(lay, comm, emulate_rank, x, slot, S, n_dev, codes, rows, payload, gathered, ex, tables, geom, window, send,
                codes_packed, recv, feats) -> None:
        check(lib().nsx_lp_forward(C.byref(lay), comm, int(emulate_rank), ptr(x), ptr(slot), int(S), ptr(n_dev), ptr(codes),
                                   codes.stride(0), int(rows), ptr(payload), ptr(gathered), ex.sizes_host, ex.rows_host,
                                   ptr(tables), C.byref(geom), ptr(window), ptr(send), ptr(codes_packed), ptr(recv), ptr(feats),
                                   stream()), "nsx_lp_forward")

    @staticmethod
    def backward(lay, comm, emulate_rank, dout, x, slot, S, n_dev, send, recv, gathered, ex, tables, geom, window, G, dz, ret,
                 ret_recv, nonfinite, rows, dx, dcode) -> None:
        dev = recv.device
        check(lib().nsx_lp_backward(C.byref(lay), comm, int(emulate_rank), ptr(dout), ptr(x), ptr(slot), int(S), ptr(n_dev),
                                    ptr(send), ptr(recv), ptr(gathered), ex.sizes_host, ex.rows_host, ptr(tables),
                                    C.byref(geom), ptr(window), ptr(G), ptr(dz), ptr(F.codesum_scratch(lay.R_cap, lay.H, dev)),
                                    ptr(ret), ptr(ret_recv), ptr(nonfinite), int(rows), ptr(dx), ptr(dcode), stream()),
              "nsx_lp_backward")

    @staticmethod
    def shared_columns(x, S, tables, H, geom, code, slot, window, cols) -> None:
        check(lib().nsx_hash_ensemble_fwd(ptr(x), int(S), ptr(tables), int(H), C.byref(geom), ptr(code), code.stride(0),
                                          ptr(slot), ptr(window), ptr(cols), None, stream()), "nsx_hash_ensemble_fwd")


class Exchange:
    """One forward exchange and what its backward needs again: the job's sizes (host), the layout of the payloads, the
    gathered forward payloads (the owners read every rank's code rows from them in the backward)."""
    __slots__ = ("lay", "sizes", "rows", "S", "n_rows", "S_cap", "R_cap", "sizes_host", "rows_host", "n_planes", "gathered",
                 "codes_packed", "window", "backward_done")

    def __init__(self, lay, sizes, rows, rank):
        W = len(sizes)
        self.lay, self.sizes, self.rows = lay, list(sizes), list(rows)
        self.S, self.n_rows = int(sizes[rank]), int(rows[rank])
        self.S_cap, self.R_cap = int(lay.S_cap), int(lay.R_cap)
        self.sizes_host = (C.c_int64 * W)(*[int(s) for s in sizes])
        self.rows_host = (C.c_int32 * W)(*[int(r) for r in rows])
        self.n_planes = int(sum(rows))
        self.gathered = self.codes_packed = self.window = None
        self.backward_done = False


class LevelParallel:
    """The exchange around the HashEnsemble kernels (attached as ``HashEnsemble.level_parallel``).

    One training step = ONE host-side size exchange (``exchange_sizes``: the marched sample count and the code rows of every
    rank, 16 bytes per rank through a gloo group) and FOUR device collectives: all-gather + all-to-all in the forward
    (``features``), two all-to-alls in the backward (``backward``).  The counts of valid rows, the positions, the code slots
    and the conditioned code rows travel inside those payloads (csrc/level_parallel.hip), and the partial code gradients
    return with the partial dL/dx -- every rank issues the same collectives in the same order whatever its rays produced.

    ``emulate=True``: this process is rank ``rank`` of a ``world_size``-rank job whose other ranks are REPLICAS of itself --
    the partition, the payload layouts, the per-source-rank launches and the gradient planes are those of the real job, the
    collectives run on the (one-rank) group and what they would have delivered from rank j is this rank's own block.  What a
    rank of an 8-GPU job computes and issues per step, measurable on one GPU (``bench.py --level-parallel-one-rank``); the
    feature columns of the levels this rank does not own are then copies of its own, so the numbers it trains on are not the
    model's."""

    def __init__(self, he: HashEnsemble, world_size: int, rank: int, group=None, emulate: bool = False, ops=None,
                 balanced: bool = True, native_collectives: Optional[bool] = None):
        L = int(he.geom.n_levels)
        if world_size < 2 or L % world_size != 0:
            raise ValueError(f"level-parallel tables need a world size that divides the {L} levels (got {world_size})")
        self.he, self.world_size, self.rank, self.group = he, int(world_size), int(rank), group
        self.emulate = bool(emulate)
        self.ops = ops or NativeLPOps()
        self.n_own = L // self.world_size
        # who owns which levels (round 6: balanced -- the cheapest level with the dearest; see level_assignment), and where a
        # rank's levels sit in the full [entry][f][h] tables: one entry range per level.  The rank's CURRENT values of those
        # entries -- fp16 working tables, fp32 master -- live in compact buffers of their own (the levels one after the
        # other); the full-size tensors of the HashEnsemble are stale while the mode lasts (``gather_entry_ranges``).
        self.levels_of = level_assignment(L, self.world_size, balanced)
        self.levels = self.levels_of[self.rank]
        self.level_of_flat = [l for lv in self.levels_of for l in lv]
        self.geom = sub_geometry_levels(he.geom, self.levels)
        self.ranges_of = [[(int(he.geom.offset[l]), int(he.geom.offset[l + 1])) for l in lv] for lv in self.levels_of]
        self.ranges = self.ranges_of[self.rank]
        self.n_entries = sum(b - a for a, b in self.ranges)
        self.f16 = self.master = None    # compact [n_entries][2][Hp]: made by ``pull`` (the optimizer that owns the mode)
        # sizes travel host-to-host (a device collective would make the host wait for the queue): a gloo group beside RCCL
        backend = dist.get_backend(group)
        self._a2a_native = backend == "nccl"
        self.cpu_group = group if backend == "gloo" else dist.new_group(backend="gloo")
        if self.emulate and dist.get_world_size(group) != 1:
            raise ValueError("an emulated rank runs on a one-rank process group")
        # RCCL runs: the exchange's collectives are issued by the library itself (csrc/comm.hip: an RCCL communicator of its
        # own, one C call per direction that enqueues pack -> collective -> kernels -> collective -> unpack) instead of five
        # torch.distributed calls per step with Python in between -- ~0.35 ms of host time on a rank whose step takes ~2 ms.
        # ``native_collectives=False`` keeps the torch.distributed route (the only one on gloo, where the tests' CPU ops run).
        self.comm = None
        if native_collectives is None:
            native_collectives = NATIVE_COLLECTIVES
        if native_collectives is None:
            native_collectives = backend == "nccl" and ops is None and he.tables.is_cuda
        if native_collectives:
            if backend != "nccl":
                raise ValueError("the library's own collectives are RCCL's: the process group must be an nccl one")
            self.comm = self._make_comm(he.tables.device)
        # emulated rank only: after the exchange, ONE launch of the full-geometry forward on this rank's own samples writes
        # the TRUE feature columns of all levels over the replicas' copies (this process holds the whole table): the model
        # then sees what the real job's ranks would have delivered, and -- with its parameters frozen -- the sample counts
        # of the steps stay those of the trained model while every kernel and collective of the emulated rank runs.  The
        # launch is extra work the real rank does not have; ``comm_report`` prices it (``shadow_fwd_ms``).
        self.shadow_forward = False
        self.timing = False
        self._shadow_events = []
        self.shared_inputs = False       # every rank holds the SAME positions (occupancy update): no position exchange
        self.nonfinite = None            # device float: a backward added an inf / NaN to G
        self.G = None                    # [planes][own entries][2] fp32, planes = sum of the ranks' code rows
        self.planes = 0                  # planes the step in flight writes
        self.codes_packed = None         # [planes][H] conditioned code rows, source-rank major
        self.window = None
        self.backward_calls = 0
        self.samples_scattered = 0
        self.last_exchange = None        # the most recent forward exchange (its backward follows)
        self._size_bufs = None
        self._bufs = {}                  # grow-only byte buffers of the payloads (stream-ordered reuse, step after step)
        self._g_clean = None             # event: the optimizer pass left G all zeros (nsx_adam_hash_factored_consume)
        self.stats = {"bytes_in": 0, "samples_fwd": 0, "samples_bwd": 0, "fwd_calls": 0, "bwd_calls": 0, "collectives": 0,
                      "host_exchanges": 0}

    def _make_comm(self, dev):
        """The library's RCCL communicator over the ranks of ``group``: rank 0 draws the unique id, the gloo side group
        carries it, every rank joins (collective) with its device current."""
        n, r = dist.get_world_size(self.group), dist.get_rank(self.group)
        ident = torch.zeros((_lib.NSX_COMM_ID_BYTES,), dtype=torch.uint8)
        if r == 0:
            check(lib().nsx_comm_unique_id(ident.data_ptr()), "nsx_comm_unique_id")
        if n > 1:
            dist.broadcast(ident, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.cpu_group)
        comm = C.c_void_p()
        with torch.cuda.device(dev):
            check(lib().nsx_comm_create(ident.data_ptr(), n, r, C.byref(comm)), "nsx_comm_create")
        return comm

    def close(self) -> None:
        comm, self.comm = self.comm, None
        if comm is not None:
            lib().nsx_comm_destroy(comm)

    def __del__(self):
        # (not while the interpreter shuts down: the HIP runtime may be gone, and the process' exit frees the communicator)
        try:
            import sys
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass

    # ---- collectives -------------------------------------------------------------------------------------------------
    def exchange_sizes_begin(self, S: int, rows: int):
        """Start the one host-side collective of a pass -- every rank's (sample capacity, code rows), 16 bytes per rank over
        the gloo group -- and return at once: the caller goes on issuing device work and calls ``exchange_sizes_end``."""
        n = dist.get_world_size(self.cpu_group)
        if self._size_bufs is None or self._size_bufs[1].numel() != 2 * n:
            self._size_bufs = (torch.empty((2,), dtype=torch.int64), torch.empty((2 * n,), dtype=torch.int64))
        mine, out = self._size_bufs
        mine[0], mine[1] = int(S), int(rows)
        work = dist.all_gather_into_tensor(out, mine, group=self.cpu_group, async_op=True)
        self.stats["host_exchanges"] += 1
        return work, out, n

    def exchange_sizes_end(self, pending) -> Exchange:
        work, out, n = pending
        work.wait()
        return self._exchange_from(out.view(n, 2).tolist())

    def exchange_sizes(self, S: int, rows: int) -> Exchange:
        """Blocking form of ``exchange_sizes_begin`` / ``_end``."""
        return self.exchange_sizes_end(self.exchange_sizes_begin(S, rows))

    def _exchange_from(self, got) -> Exchange:
        W = self.world_size
        if self.emulate:
            got = [got[0]] * W
        sizes, rws = [g[0] for g in got], [max(1, g[1]) for g in got]
        if sum(rws) > MAX_ADAM_SLOTS:
            raise RuntimeError(f"level-parallel HashEnsemble: {sum(rws)} code rows in the job's batch (limit {MAX_ADAM_SLOTS}: "
                               f"NSX_MAX_ADAM_SLOTS gradient planes per optimizer pass)")
        lay = self._layout(max(1, max(sizes)), max(rws))
        return Exchange(lay, sizes, rws, self.rank)

    def _layout(self, S_cap: int, R_cap: int):
        lay = self.ops.layout(self.world_size, S_cap, R_cap, self.he.n_hash_encodings, 2 * self.n_own)
        for i, l in enumerate(self.level_of_flat):
            lay.level_of[i] = l
        return lay

    collective_timeout_s = 120.0

    def collective_entry(self, what: str) -> None:
        """Called in front of a collective that user code can reach from ONE rank by accident (leaving training mode, a
        checkpoint): all ranks must arrive within ``collective_timeout_s`` -- otherwise a RuntimeError that says which ranks
        did not, instead of a hang in the first broadcast."""
        if self.emulate:
            return
        from datetime import timedelta
        try:
            dist.monitored_barrier(group=self.cpu_group, timeout=timedelta(seconds=self.collective_timeout_s))
        except RuntimeError as exc:
            raise RuntimeError(f"level-parallel HashEnsemble: {what} is a collective -- every rank has to call it (the ranks "
                               f"hold different levels of the tables); not all of them did: {exc}") from exc

    def _all_gather(self, out: torch.Tensor, mine: torch.Tensor) -> None:
        """``out`` = every rank's ``mine`` in rank order (flat byte buffers)."""
        W, n = self.world_size, mine.numel()
        if self.emulate:
            dist.all_gather_into_tensor(out[self.rank * n:(self.rank + 1) * n], mine, group=self.group)
            out.view(W, n).copy_(mine.view(1, n).expand(W, n))             # the replicas' payloads
        else:
            dist.all_gather_into_tensor(out, mine, group=self.group)
        self.stats["bytes_in"] += (W - 1) * n * mine.element_size()
        self.stats["collectives"] += 1

    def _all_to_all(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        """``out`` block j = rank j's ``inp`` block [this rank], W equal blocks.  RCCL: ``all_to_all_single``; gloo (the CPU /
        one-GPU test backends) has no device all-to-all: every rank's whole block matrix is gathered and the column picked
        -- the same result, W x the bytes (``stats`` counts the all-to-all's).  Emulated rank: the one-rank collective hands
        the blocks back as they are -- block j stands for what replica j would have sent."""
        W = self.world_size
        blk = inp.numel() // W
        if self.emulate or self._a2a_native:
            dist.all_to_all_single(out, inp, group=self.group)
        else:
            full = torch.empty((W, W, blk), dtype=inp.dtype, device=inp.device)
            dist.all_gather_into_tensor(full.view(-1), inp, group=self.group)
            out.view(W, blk).copy_(full[:, self.rank])
        self.stats["bytes_in"] += (W - 1) * blk * inp.element_size()
        self.stats["collectives"] += 1

    @contextmanager
    def shared(self):
        """Inside: every rank calls ``features`` with identical positions and codes (the occupancy-grid update draws its
        cells from a generator all ranks share, nersemble_instant_ngp.py:184-196) -- only the column blocks travel."""
        old, self.shared_inputs = self.shared_inputs, True
        try:
            yield
        finally:
            self.shared_inputs = old

    # ---- the rank's slice ---------------------------------------------------------------------------------------------
    def take_local(self, full: torch.Tensor) -> torch.Tensor:
        """The entries of this rank's levels out of a full ``[entries, ...]`` tensor, one level after the other (a copy)."""
        return torch.cat([full[a:b] for a, b in self.ranges], dim=0).contiguous()

    @torch.no_grad()
    def pull(self) -> None:
        """(Re)build the compact working tables and master from the HashEnsemble's full-size tensors (entering the mode,
        loading a checkpoint)."""
        self.he.wait_tables()
        self.f16 = self.take_local(self.he.half_tables())
        self.master = self.take_local(self.he.tables.data)

    def slice_f16(self) -> torch.Tensor:
        if self.f16 is None:
            self.pull()
        return self.f16

    def slice_master(self) -> torch.Tensor:
        if self.master is None:
            self.pull()
        return self.master

    @staticmethod
    def _bytes(n: int, dev) -> torch.Tensor:
        return torch.empty((int(n),), dtype=torch.uint8, device=dev)

    def _buf(self, name: str, n: int, dev) -> torch.Tensor:
        """``n`` bytes of the persistent buffer ``name`` (grown by a quarter beyond the need, never shrunk).  Every user
        runs on the current stream and the collectives order themselves behind and in front of it: reuse is stream-ordered.
        Not for ``Exchange.gathered``, which lives until its backward."""
        b = self._bufs.get(name)
        if b is None or b.numel() < n or b.device != torch.device(dev):
            b = self._bufs[name] = torch.empty((int(n) * 5 // 4 + 256,), dtype=torch.uint8, device=dev)
        return b[:int(n)]

    # ---- forward ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def features(self, x: torch.Tensor, code: torch.Tensor, code_index: torch.Tensor,
                 window: Optional[torch.Tensor], n_dev: Optional[torch.Tensor] = None, ex: Optional[Exchange] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``HashEnsemble.forward`` (values): x [S,3] in [0,1), ``code`` the CONDITIONED code table [rows,H], ``code_index``
        [S] -> [S, 2 L] fp16 (``out``: written there).  ``n_dev``: device count of valid rows (rows beyond it are not
        written).  ``ex``: the sizes of this pass if the caller exchanged them already.  Collective: every rank calls it the
        same number of times per step (S may be 0)."""
        he, W = self.he, self.world_size
        he.wait_tables()
        dev = x.device
        x, code, slot = _as(x, torch.float32), _as(code, torch.float32), _as(code_index, torch.int32)
        S, H, n2 = int(x.shape[0]), he.n_hash_encodings, 2 * self.n_own
        tables = self.slice_f16()
        ops = self.ops
        self.stats["fwd_calls"] += 1
        if self.shared_inputs:
            lay = self._layout(max(1, S), 1)
            cols = self._bytes(lay.feat_bytes, dev)
            if S:
                ops.shared_columns(x, S, tables, H, self.geom, code, slot, window, cols)
            allc = self._bytes(W * lay.feat_bytes, dev)
            self._all_gather(allc, cols)
            feats = out if out is not None else torch.empty((S, W * n2), dtype=torch.float16, device=dev)
            ops.fwd_unpack(lay, allc, S, None, feats)
            self.stats["samples_fwd"] += S
            self._shadow(x, code, slot, window, None, feats)
            return feats
        if ex is None:
            ex = self.exchange_sizes(S, int(code.shape[0]))
        if S != ex.S or int(code.shape[0]) > ex.R_cap:
            raise RuntimeError(f"level-parallel HashEnsemble: this pass has {S} samples / {code.shape[0]} code rows, the sizes "
                               f"exchanged for it say {ex.S} / <= {ex.R_cap}")
        lay = ex.lay
        payload = self._buf("fwd_payload", lay.fwd_bytes, dev)
        ex.gathered = self._bytes(W * lay.fwd_bytes, dev)
        send = self._buf("feat_send", W * lay.feat_bytes, dev)
        ex.codes_packed = torch.empty((ex.n_planes, H), dtype=torch.float32, device=dev)
        ex.window = window
        recv = self._buf("feat_recv", W * lay.feat_bytes, dev)
        feats = out if out is not None else torch.empty((S, W * n2), dtype=torch.float16, device=dev)
        if self.comm is not None:
            ops.forward(lay, self.comm, self.rank if self.emulate else -1, x, slot, S, n_dev, code, int(code.shape[0]), payload,
                        ex.gathered, ex, tables, self.geom, window, send, ex.codes_packed, recv, feats)
            self.stats["bytes_in"] += (W - 1) * (lay.fwd_bytes + lay.feat_bytes)
            self.stats["collectives"] += 2
        else:
            ops.fwd_pack(lay, x, slot, S, n_dev, code, int(code.shape[0]), payload)
            self._all_gather(ex.gathered, payload)
            ops.fwd_run(lay, ex.gathered, ex, tables, self.geom, window, send, ex.codes_packed)
            self._all_to_all(recv, send)
            ops.fwd_unpack(lay, recv, S, n_dev, feats)
        self.stats["samples_fwd"] += sum(ex.sizes)
        self.last_exchange = ex
        self._shadow(x, code, slot, window, n_dev, feats)
        return feats

    def _shadow(self, x, code, slot, window, n_dev, feats) -> None:
        if not (self.emulate and self.shadow_forward) or x.shape[0] == 0:
            return
        he = self.he
        ev = None
        if self.timing:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(lib().nsx_hash_ensemble_fwd(ptr(x), int(x.shape[0]), ptr(he.half_tables()), he.n_hash_encodings,
                                          C.byref(he.geom), ptr(code), code.stride(0), ptr(slot), ptr(window), ptr(feats),
                                          ptr(n_dev), stream()), "nsx_hash_ensemble_fwd (shadow)")
        if ev is not None:
            ev[1].record()
            self._shadow_events.append(ev)

    # ---- backward -----------------------------------------------------------------------------------------------------
    def begin_step(self) -> None:
        """New optimisation step: the gradient planes start from zero with its first backward."""
        self.backward_calls = 0
        self.samples_scattered = 0

    def _planes(self, n_planes: int, dev) -> torch.Tensor:
        """G for this step's planes, all zeros when the first backward of the step starts."""
        if self.nonfinite is None or self.nonfinite.device != dev:
            self.nonfinite = torch.zeros((1,), dtype=torch.float32, device=dev)
        fresh = self.backward_calls == 0
        if fresh:
            self.nonfinite.zero_()
        need = n_planes * self.n_entries * 2
        if self.G is None or self.G.numel() < need or self.G.device != dev:
            self.G = torch.zeros((need,), dtype=torch.float32, device=dev)
            self._g_clean = None
        elif fresh:
            if self._g_clean is not None:
                torch.cuda.current_stream(dev).wait_event(self._g_clean)       # left clean by the consuming optimizer pass
                self._g_clean = None
            else:
                self.G.zero_()
        elif n_planes != self.planes:
            raise RuntimeError("level-parallel HashEnsemble: the backward calls of one step disagree on the code rows")
        self.planes = n_planes
        return self.G[:need].view(n_planes, self.n_entries, 2)

    @torch.no_grad()
    def backward(self, x: torch.Tensor, code_index: torch.Tensor, dout: torch.Tensor, code: Optional[torch.Tensor] = None,
                 window: Optional[torch.Tensor] = None, n_dev: Optional[torch.Tensor] = None, need_code: bool = True,
                 need_table: bool = True, ex: Optional[Exchange] = None, dx_out: Optional[torch.Tensor] = None,
                 dcode_out: Optional[torch.Tensor] = None):
        """The backward of ``features`` for this rank's samples: returns (dL/dx [S,3] fp32, dL/dcode [rows,H] fp32); the
        table gradient of the OWNED levels over ALL ranks' samples is left in the factored planes for
        ``LevelParallelTableAdam``.  ``ex``: the forward exchange this is the backward of (default: the most recent one) --
        its sizes, its code rows and its window are used again; ``x`` may hold fewer rows than that forward (the kept
        samples), ``n_dev`` counts the valid ones on the device.  The code gradient is always formed (the exchange does not
        depend on a rank-local flag).  Collective."""
        he, W = self.he, self.world_size
        he.wait_tables()
        ex = ex if ex is not None else self.last_exchange
        if ex is None or ex.gathered is None:
            raise RuntimeError("level-parallel HashEnsemble: a backward without a forward exchange")
        dev = x.device
        x, slot, dout = _as(x, torch.float32), _as(code_index, torch.int32), _as(dout, torch.float32)
        S, H = int(x.shape[0]), he.n_hash_encodings
        if S > ex.S_cap:
            raise RuntimeError(f"level-parallel HashEnsemble: {S} samples in the backward of a forward exchange of capacity {ex.S_cap}")
        lay, ops = ex.lay, self.ops
        send = self._buf("bwd_send", W * lay.bwd_bytes, dev)
        recv = self._buf("bwd_recv", W * lay.bwd_bytes, dev)
        ret = self._buf("ret_send", W * lay.ret_bytes, dev)
        ret_recv = self._buf("ret_recv", W * lay.ret_bytes, dev)
        dz = self._buf("dz", 4 * W * lay.S_cap * lay.n2, dev)
        dx = dx_out if dx_out is not None else torch.empty((S, 3), dtype=torch.float32, device=dev)
        dcode = dcode_out if dcode_out is not None else torch.empty((ex.n_rows, H), dtype=torch.float32, device=dev)
        native = self.comm is not None
        if not native:
            ops.bwd_pack(lay, dout, x, slot, S, n_dev, send)
            self._all_to_all(recv, send)
        G = self._planes(ex.n_planes, dev) if need_table else None
        if self.nonfinite is None or self.nonfinite.device != dev:
            self.nonfinite = torch.zeros((1,), dtype=torch.float32, device=dev)
        if native:
            ops.backward(lay, self.comm, self.rank if self.emulate else -1, dout, x, slot, S, n_dev, send, recv, ex.gathered, ex,
                         self.slice_f16(), self.geom, ex.window, G, dz, ret, ret_recv, self.nonfinite, ex.n_rows, dx, dcode)
            self.stats["bytes_in"] += (W - 1) * (lay.bwd_bytes + lay.ret_bytes)
            self.stats["collectives"] += 2
        else:
            ops.bwd_run(lay, recv, ex.gathered, ex, self.slice_f16(), self.geom, ex.window, G, ret, self.nonfinite, dz)
            # partial dL/dx and partial code-row gradients back to the samples' owners, summed over the level owners
            self._all_to_all(ret_recv, ret)
            ops.bwd_unpack(lay, ret_recv, S, n_dev, ex.n_rows, dx, dcode)
        if self.backward_calls == 0 or self.codes_packed is None:
            self.codes_packed, self.window = ex.codes_packed, ex.window
        self.backward_calls += 1
        n_job = sum(min(s, ex.S_cap) for s in ex.sizes)
        self.samples_scattered += n_job
        self.stats["bwd_calls"] += 1
        self.stats["samples_bwd"] += n_job
        ex.backward_done = True
        return dx, dcode

    @torch.no_grad()
    def join_backward(self) -> None:
        """A rank whose step ran a forward exchange but no backward (nothing of its loss reached the hash features) joins
        the other ranks' backward exchange with zero valid rows -- every rank issues the same collectives in the same
        order."""
        ex = self.last_exchange
        if ex is None or ex.backward_done or ex.gathered is None:
            return
        dev = ex.gathered.device
        L2 = 2 * self.he.geom.n_levels
        self.backward(torch.zeros((1, 3), device=dev), torch.zeros((1,), dtype=torch.int32, device=dev),
                      torch.zeros((1, L2), device=dev), n_dev=torch.zeros((1,), dtype=torch.int64, device=dev), ex=ex)

    # ---- the whole table on every rank again (evaluation, checkpoints, leaving the mode) -----------------------------------
    @torch.no_grad()
    def gather_entry_ranges(self, full: torch.Tensor, mine: torch.Tensor) -> torch.Tensor:
        """``full`` [entries, 2, Hp] made current everywhere: every rank contributes the entries of ITS levels from its
        compact tensor ``mine`` [own entries, 2, Hp].  Collective; the levels differ in size, so each one is broadcast by its
        owner.  (An emulated rank has nobody to hear from: its own levels are all it can make current.)"""
        at = 0
        for a, b in self.ranges:
            full[a:b].copy_(mine[at:at + (b - a)])
            at += b - a
        if self.emulate:
            return full
        for r, ranges in enumerate(self.ranges_of):
            for a, b in ranges:
                dist.broadcast(full[a:b], src=dist.get_global_rank(self.group, r) if self.group is not None else r,
                               group=self.group)
        return full


class _LPHashFn(torch.autograd.Function):
    """``HashEnsemble.forward`` through the level-parallel exchange, differentiable w.r.t. positions and codes (the table
    gradient goes to the gradient planes of ``LevelParallel``)."""

    @staticmethod
    def forward(ctx, lp: LevelParallel, x, tables_master, code, code_index, window, precomputed):
        xx = x.detach().to(torch.float32).contiguous()
        cc = code.detach().to(torch.float32).contiguous()
        ctx.n_dev = _lib.ndev_tensor(xx.shape[0])
        out = precomputed.detach() if precomputed is not None else lp.features(xx, cc, code_index, window, n_dev=ctx.n_dev)
        ctx.lp = lp
        ctx.ex = lp.last_exchange         # (precomputed: the exchange of the pass that produced the values -- the sigma pass)
        ctx.save_for_backward(xx, code_index)
        return out

    @staticmethod
    def backward(ctx, dout):
        xx, code_index = ctx.saved_tensors
        lp = ctx.lp
        dx, dcode = lp.backward(xx, code_index, dout, n_dev=ctx.n_dev, need_table=bool(ctx.needs_input_grad[2]), ex=ctx.ex)
        return (None, (dx if ctx.needs_input_grad[1] else None), None, (dcode if ctx.needs_input_grad[3] else None), None,
                None, None)


def lp_hash_ensemble(lp: LevelParallel, x, tables_master, code, code_index, window, precomputed=None):
    return _LPHashFn.apply(lp, x, tables_master, code, code_index, window, precomputed)


class LevelParallelTableAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no amsgrad / weight decay) for ``HashEnsemble.tables`` in the level-parallel mode: every rank steps
    the entries of ITS levels from the factored gradient planes its own backward calls filled -- the optimizer needs no
    collective.  Interface of ``HashTableAdam`` / ``ShardedTableAdam`` as the trainer uses it."""
    writes_half_tables = True

    def __init__(self, hash_ensemble: HashEnsemble, lr: float = 5e-3, betas=(0.9, 0.999), eps: float = 1e-15,
                 world_size: int = 1, rank: int = 0, group=None, step: int = 0, exp_avg: Optional[torch.Tensor] = None,
                 exp_avg_sq: Optional[torch.Tensor] = None, emulate: bool = False):
        super().__init__([hash_ensemble.tables], dict(lr=lr, betas=betas, eps=eps))
        self.he = hash_ensemble
        self.world_size, self.rank, self.group = int(world_size), int(rank), group
        self.lp = LevelParallel(hash_ensemble, world_size, rank, group, emulate=emulate)
        hash_ensemble.level_parallel = self.lp
        hash_ensemble.grad_sink = None                       # (the planes of LevelParallel take the table gradient)
        p = hash_ensemble.tables
        shape = (self.lp.n_entries,) + tuple(p.shape[1:])
        if exp_avg is not None:
            self.exp_avg, self.exp_avg_sq = self.lp.take_local(exp_avg), self.lp.take_local(exp_avg_sq)
        else:
            self.exp_avg = torch.zeros(shape, dtype=torch.float32, device=p.device)
            self.exp_avg_sq = torch.zeros(shape, dtype=torch.float32, device=p.device)
        self._step = int(step)
        self.consume_density_limit = 0.5
        self.timing = False
        self._events = []
        hash_ensemble.half_tables()                          # the working copy exists and is current when the mode starts
        self.lp.pull()                                       # ... and this rank's levels move into their compact tensors

    # ---- the trainer's two phases ---------------------------------------------------------------------------------------
    reduced_nonfinite = None        # set by the trainer: the ranks' flags summed in the small gradients' bucket
    _scale = None

    def local_nonfinite(self) -> torch.Tensor:
        """This rank's flag: one of its backward calls added an inf / NaN to its gradient planes (device, fp32 [1])."""
        lp = self.lp
        if lp.nonfinite is None or not lp.backward_calls:
            dev = self.he.tables.device
            if lp.nonfinite is None or lp.nonfinite.device != dev:
                lp.nonfinite = torch.zeros((1,), dtype=torch.float32, device=dev)
            else:
                lp.nonfinite.zero_()
        return lp.nonfinite

    @torch.no_grad()
    def check_finite(self, found_inf: torch.Tensor) -> None:
        if self.reduced_nonfinite is not None:
            # the job's flags (summed over the ranks): every rank skips the step or none does
            torch.maximum(found_inf, self.reduced_nonfinite, out=found_inf)      # (a sum of 0 / 1 flags: > 0 = skip)
            return
        nf = self.lp.nonfinite
        if self.lp.backward_calls and nf is not None:
            torch.maximum(found_inf, nf.to(found_inf.dtype), out=found_inf)

    def ensure_reduce_started(self) -> None:
        """(name of ``ShardedTableAdam``'s hook) every rank takes part in the step's backward exchange, also one without
        samples."""
        if self.lp.backward_calls == 0:
            self.lp.join_backward()

    def _mark(self, what: str):
        self.lp.timing = self.timing
        if not self.timing or not self.he.tables.is_cuda:
            return
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self._events.append((what, ev))

    @torch.no_grad()
    def step(self, found_inf: Optional[torch.Tensor] = None, inv_scale: Optional[torch.Tensor] = None,
             side_stream: Optional["torch.cuda.Stream"] = None):
        he, lp = self.he, self.lp
        if lp.backward_calls == 0 or lp.planes == 0 or lp.G is None:
            lp.begin_step()
            return
        # the ranks' losses are means over THEIR rays; the job's gradient is their average (engine/parallel.py)
        if self._scale is None or self._scale.device != he.tables.device:
            self._scale = torch.empty((1,), dtype=torch.float32, device=he.tables.device)
        scale = self._scale
        if inv_scale is not None:
            torch.mul(inv_scale.reshape(1), 1.0 / self.world_size, out=scale)
        else:
            scale.fill_(1.0 / self.world_size)
        if side_stream is None or not he.tables.is_cuda:
            self._step_now(found_inf, scale)
            lp.begin_step()
            return
        he.wait_tables()
        side_stream.wait_stream(torch.cuda.current_stream(he.tables.device))
        with torch.cuda.stream(side_stream):
            self._step_now(found_inf, scale)
            done = torch.cuda.Event()
            done.record(side_stream)
        for t in (found_inf, scale, lp.codes_packed, lp.window, lp.G):
            if t is not None:
                t.record_stream(side_stream)
        he._tables_ready = done
        lp.begin_step()

    def _step_now(self, found_inf, scale):
        he, lp = self.he, self.lp
        group = self.param_groups[0]
        self._step += 1
        b1, b2 = group["betas"]
        G = lp.G[:lp.planes * lp.n_entries * 2]
        sparse = 0 < lp.samples_scattered * 80 * lp.n_own // he.geom.n_levels < self.consume_density_limit * (G.numel() // 8)
        fn = lib().nsx_adam_hash_factored_consume if sparse else lib().nsx_adam_hash_factored
        self._mark("adam_begin")
        check(fn(ptr(G), lp.planes, ptr(lp.codes_packed), lp.codes_packed.stride(0), ptr(lp.window), he.n_hash_encodings,
                 C.byref(lp.geom), ptr(lp.slice_master()), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(lp.slice_f16()),
                 group["lr"], b1, b2, group["eps"], self._step, ptr(scale), ptr(found_inf), stream()),
              "nsx_adam_hash_factored")
        self._mark("adam_end")
        if sparse:
            ev = torch.cuda.Event()
            ev.record()
            lp._g_clean = ev
        else:
            lp._g_clean = None
        he.mark_half_synced()

    def rollback_step(self) -> None:
        self._step = max(0, self._step - 1)

    def clear_grads(self) -> None:
        self.he.tables.grad = None

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=set_to_none)

    # ---- what the exchange moved ----------------------------------------------------------------------------------------
    def comm_report(self, reset: bool = True) -> dict:
        """Per step: bytes that ARRIVED at this rank over all collectives of the level-parallel exchange (algorithmic:
        all-to-all counted as such also where the gloo stand-in gathers), samples of the whole job that went through this
        rank's forward / backward kernels, the optimizer pass on the owned entry range."""
        st, lp = self.lp.stats, self.lp
        n = max(1, st["bwd_calls"])
        total, open_ev = 0.0, None
        for what, e in self._events:
            if what == "adam_begin":
                open_ev = e
            elif what == "adam_end" and open_ev is not None:
                total += open_ev.elapsed_time(e)
                open_ev = None
        n_adam = max(1, sum(1 for w, _ in self._events if w == "adam_begin"))
        out = {"exchange": "level_parallel", "world_size": self.world_size, "levels_per_rank": lp.n_own, "levels": list(lp.levels),
               "owned_entries": lp.n_entries, "steps": n, "bytes_per_rank": st["bytes_in"] / n,
               "samples_fwd_per_step": st["samples_fwd"] / n, "samples_bwd_per_step": st["samples_bwd"] / n,
               "fwd_exchanges_per_step": st["fwd_calls"] / n, "shard_adam_ms": total / n_adam, "gradient_planes": lp.planes,
               # device collectives (all-gather / all-to-all of this exchange; the occupancy update's column all-gather every
               # 16th step is among them) and host-side size exchanges per step
               "collectives_per_step": st["collectives"] / n, "host_exchanges_per_step": st["host_exchanges"] / n,
               "emulated_rank": (lp.rank if lp.emulate else None)}
        if lp.emulate and lp.shadow_forward:
            # what the emulation ADDS to a step: the full-geometry forward that replaces the replicas' feature columns
            out["shadow_fwd_ms"] = sum(a.elapsed_time(b) for a, b in lp._shadow_events) / n
        if reset:
            lp.stats = {k: 0 for k in st}
            self._events = []
            lp._shadow_events = []
        return out

    # ---- checkpointing / leaving the mode ------------------------------------------------------------------------------------
    @torch.no_grad()
    def gather_master(self) -> None:
        """The full fp32 master AND the full fp16 working tables on every rank again (collective)."""
        he = self.he
        he.wait_tables()
        self.lp.gather_entry_ranges(he.tables.data, self.lp.slice_master())
        self.lp.gather_entry_ranges(he.half_tables(), self.lp.slice_f16())
        he.mark_half_synced()

    def _gather_moment(self, mine: torch.Tensor) -> torch.Tensor:
        full = torch.zeros_like(self.he.tables.data)
        return self.lp.gather_entry_ranges(full, mine)

    @torch.no_grad()
    def table_state(self) -> dict:
        self.he.wait_tables()
        return {"step": int(self._step), "lr": float(self.param_groups[0]["lr"]),
                "exp_avg": self.he.to_tcnn_layout(self._gather_moment(self.exp_avg)),
                "exp_avg_sq": self.he.to_tcnn_layout(self._gather_moment(self.exp_avg_sq))}

    @torch.no_grad()
    def load_table_state(self, state: dict) -> None:
        self.he.wait_tables()
        self._step = int(state["step"])
        self.param_groups[0]["lr"] = float(state.get("lr", self.param_groups[0]["lr"]))
        for key, dst in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            dst.zero_()
            if state.get(key) is not None:
                dst.copy_(self.lp.take_local(self.he.from_tcnn_layout(state[key])))
        self.he._f16_version = None
        self.he.half_tables()
        self.lp.pull()                            # (the model parameters were loaded into the full-size tensors)
