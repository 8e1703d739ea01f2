"""CPU, world_size 2, gloo: the data-parallel path (gradient averaging + ray sharding) the 8-GPU bench relies on."""
import os
import socket

import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.parallel import all_reduce_gradients, SMALL_BUCKET_ELEMS
    from nersemble_amd.data.synthetic import SyntheticNeRSembleData
    torch.manual_seed(0)                                    # identical "model" on both ranks
    big = torch.nn.Parameter(torch.zeros(SMALL_BUCKET_ELEMS + 5))
    small = [torch.nn.Parameter(torch.zeros(7, 3)), torch.nn.Parameter(torch.zeros(11))]
    nograd = torch.nn.Parameter(torch.zeros(4))             # no gradient on this rank -> zeros must be contributed
    big.grad = torch.full_like(big, float(rank + 1))
    small[0].grad = torch.arange(21.).reshape(7, 3) * (rank + 1)
    small[1].grad = torch.ones(11) * (10 * rank)
    never = torch.nn.Parameter(torch.zeros(3))              # no gradient on ANY rank (a tensor that has not started yet)
    if rank == 0:
        nograd.grad = torch.ones(4) * 8
    plist = [big] + small + [nograd, never]
    counts, _ = all_reduce_gradients(plist, world)             # no history: this step's counts decide (one host read)
    ok = torch.allclose(big.grad, torch.full_like(big, 1.5))
    ok &= torch.allclose(small[0].grad, torch.arange(21.).reshape(7, 3) * 1.5)
    ok &= torch.allclose(small[1].grad, torch.ones(11) * 5)
    # a parameter SOME rank had a gradient for keeps the averaged gradient on EVERY rank (round 4 dropped it on the rank
    # that had none: that rank skipped an update the other one applied); one that no rank had a gradient for stays
    # without one (its optimizer must not count a step)
    ok &= torch.allclose(nograd.grad, torch.ones(4) * 4)
    ok &= never.grad is None and counts.tolist() == [[2.0, 2.0, 2.0, 1.0, 0.0], [0.0] * 5]
    from nersemble_amd.engine.parallel import check_gradient_presence
    check_gradient_presence(counts.tolist(), world)           # nobody dropped what somebody stepped: fine
    # with a history (the previous step's global counts): the same decisions without a host read
    for p in plist:
        p.grad = None
    big.grad = torch.full_like(big, float(rank + 1))
    small[0].grad, small[1].grad = torch.ones(7, 3), torch.ones(11)
    if rank == 1:
        nograd.grad = torch.ones(4) * 6
    counts2, _ = all_reduce_gradients(plist, world, takes_part=[c > 0 for c in counts[0].tolist()])
    ok &= torch.allclose(nograd.grad, torch.ones(4) * 3) and never.grad is None
    ok &= counts2.tolist() == [[2.0, 2.0, 2.0, 1.0, 0.0], [0.0, 0.0, 0.0, 0.0, 2.0]]
    check_gradient_presence(counts2.tolist(), world)
    # the hole that is only DETECTED: `never` starts on rank 0 in a step in which rank 1 has no gradient for it and the
    # history says it does not take part -> rank 1 drops, rank 0 steps: reported on both ranks
    for p in plist:
        p.grad = None
    big.grad, small[0].grad, small[1].grad, nograd.grad = (torch.ones_like(big), torch.ones(7, 3), torch.ones(11),
                                                           torch.ones(4))
    if rank == 0:
        never.grad = torch.ones(3)
    counts3, _ = all_reduce_gradients(plist, world, takes_part=[True, True, True, True, False])
    ok &= (never.grad is not None) == (rank == 0) and counts3[:, 4].tolist() == [1.0, 1.0]
    try:
        check_gradient_presence(counts3.tolist(), world)
        ok = False
    except RuntimeError as e:
        ok &= "parameters [4]" in str(e)
    # round 6: gradients deposited as views of ONE persistent buffer (the native step's) are reduced where they are; the
    # gradients without a slot there, the presence counts and the extra flags ride behind the buffer's fixed part -- one
    # collective with the same wire format on every rank, the views stay views.  Rank 1 plays a step that did NOT deposit
    # (its gradients live elsewhere): they are copied into their slots first.
    flat = torch.zeros(64 + 256)
    a1, a2 = torch.nn.Parameter(torch.zeros(4, 5)), torch.nn.Parameter(torch.zeros(9))
    out1 = torch.nn.Parameter(torch.zeros(6))
    slots = {id(a1): flat[3:23].view(4, 5), id(a2): flat[30:39]}

    class _Arena:
        fixed, slack = 40, 256

        def __init__(self, flat):
            self.flat = flat

        def slot_of(self, p):
            return slots.get(id(p))

    flat[23:30] = 99.0                                       # (a region between two gradients that belongs to nobody)
    if rank == 0:
        flat[3:23] = torch.arange(20.)
        flat[30:39] = 2.0
        a1.grad, a2.grad = slots[id(a1)], slots[id(a2)]
    else:
        a1.grad, a2.grad = torch.arange(20.).view(4, 5) * 2, torch.full((9,), 3.0)
    out1.grad = torch.ones(6) * 3 if rank == 0 else None
    cnt, fl = all_reduce_gradients([a1, a2, out1], world, arena=_Arena(flat), extra_flags=torch.tensor([float(rank)]))
    ok &= a1.grad.data_ptr() == flat[3:23].data_ptr() and torch.allclose(a1.grad, torch.arange(20.).view(4, 5) * 1.5)
    ok &= a2.grad.data_ptr() == flat[30:39].data_ptr() and torch.allclose(a2.grad, torch.full((9,), 2.5))
    ok &= torch.allclose(out1.grad, torch.ones(6) * 1.5)
    ok &= cnt.tolist() == [[2.0, 2.0, 1.0], [0.0, 0.0, 0.0]] and fl.tolist() == [1.0]
    # ray sharding: different rays per rank, same rig
    box = torch.tensor([[-2.5, -1.8, -2.5], [2.2, 1.8, 2.0]])
    data = SyntheticNeRSembleData(box, n_timesteps=10, n_rays=64, device="cpu", rank=rank)
    bundle, batch = data.next_train(0)
    torch.save({"ok": bool(ok), "origins": bundle.origins, "dirs": bundle.directions, "c2w": data.c2w},
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_and_ray_sharding_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["ok"] and r1["ok"]
    assert torch.equal(r0["c2w"], r1["c2w"])                         # same camera rig
    assert not torch.equal(r0["dirs"], r1["dirs"])                   # different rays per rank


# ---- sharded table optimizer: reduce-scatter -> Adam on the shard -> all-gather ------------------------------------
class _TorchTableOps:
    """torch restatement of the three libnsx kernels of engine/sharded_adam.py, so that its collective plumbing runs
    on CPU tensors under gloo (test infrastructure only)."""

    @staticmethod
    def dense(he, entry):
        H = he.n_hash_encodings
        code = entry["code"][:entry["n_rows"], :H].float()
        if entry["window"] is not None:
            code = code * entry["window"][:H]
        code = code.half().float()
        out = torch.zeros(he.tables.shape, dtype=torch.float32)
        out[:, :, :H] = torch.einsum("sef,sh->efh", entry["G"].float(), code)
        return out

    def expand_f16(self, he, entry, out, scale, accumulate):
        d = (self.dense(he, entry) * scale).reshape(-1)
        n = d.numel()
        out[:n] = (out[:n].float() + d).half() if accumulate else d.half()

    def expand_f16_bucket(self, he, entry, out, scale, accumulate, shard, bucket, k, world):
        """Piece k of every rank's shard, rank-major: what nsx_hash_grad_expand_f16_bucket writes."""
        d = (self.dense(he, entry) * scale).reshape(-1)
        padded = torch.zeros((world * shard,), dtype=torch.float32)
        padded[:d.numel()] = d
        piece = padded.view(world, shard)[:, k * bucket:(k + 1) * bucket].reshape(-1)
        out.copy_((out.float() + piece).half() if accumulate else piece.half())

    def expand_f16_bucket_width(self, he, entry, out, scale, accumulate, shard, bucket, k, world, width, beyond):
        """Piece k of every rank's shard restricted to the grids [0, width), packed [rank][entry][f][width]: what
        nsx_hash_grad_expand_f16_bucket_width writes (+ its flag for a code that is non-zero beyond the width)."""
        d = self.dense(he, entry) * scale                                  # [total, 2, Hp]
        Hp = d.shape[-1]
        H = he.n_hash_encodings
        code = entry["code"][:entry["n_rows"], :H].float()
        if entry["window"] is not None:
            code = code * entry["window"][:H]
        if beyond is not None and bool((code.half()[:, width:] != 0).any()):
            beyond.fill_(1.0)
        se, be = shard // (2 * Hp), bucket // (2 * Hp)
        padded = torch.zeros((world * se, 2, Hp), dtype=torch.float32)
        padded[:d.shape[0]] = d
        piece = padded.view(world, se, 2, Hp)[:, k * be:(k + 1) * be, :, :width].reshape(-1)
        out.copy_((out.float() + piece).half() if accumulate else piece.half())

    @staticmethod
    def adam_f16grad_width(grad, n_entries, width, Hp, master, exp_avg, exp_avg_sq, f16, packed_out, lr, b1, b2, eps, step,
                           inv_scale, found_inf):
        n, npk = n_entries * 2 * Hp, n_entries * 2 * width

        def cut(t):
            return t[:n].view(n_entries, 2, Hp)[..., :width]

        if found_inf is not None and float(found_inf) != 0:
            packed_out[:npk] = cut(f16).reshape(-1)
            return
        g = grad[:npk].float().view(n_entries, 2, width) * (float(inv_scale) if inv_scale is not None else 1.0)
        m, v, p = cut(exp_avg), cut(exp_avg_sq), cut(master)
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        p.sub_((lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps))
        cut(f16).copy_(p.half())
        packed_out[:npk] = p.half().reshape(-1)

    @staticmethod
    def unpack_width(packed, n_entries, width, Hp, f16):
        f16[:n_entries * 2 * Hp].view(n_entries, 2, Hp)[..., :width] = packed[:n_entries * 2 * width].view(n_entries, 2, width)

    @staticmethod
    def check_finite_f16(x, found_inf):
        if not torch.isfinite(x.float()).all():
            found_inf.fill_(1.0)

    @staticmethod
    def adam_f16grad(grad, n, master, exp_avg, exp_avg_sq, f16_out, lr, b1, b2, eps, step, inv_scale, found_inf):
        if found_inf is not None and float(found_inf) != 0:
            f16_out[:n] = master[:n].half()              # (skipped: the output still holds the current values)
            return
        g = grad[:n].float() * (float(inv_scale) if inv_scale is not None else 1.0)
        m, v = exp_avg[:n], exp_avg_sq[:n]
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        master.sub_((lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps))
        f16_out[:n] = master.half()


def _fake_entry(he, seed, n_rows=3, poison=False, window=None, ones=False):
    g = torch.Generator().manual_seed(seed)
    G = torch.randn((n_rows, he.geom.total_entries, 2), generator=g) * 1e-2
    G[torch.rand(G.shape, generator=g) < 0.5] = 0.0
    if poison:
        G[1, 7, 0] = float("inf")
    code = torch.randn((n_rows, he.n_hash_encodings), generator=g)
    if ones:
        code = torch.ones_like(code)              # (hash_ensemble.py:121-123: the code is replaced by ones while the window is 1)
    win = None
    if window is not None:
        from nersemble_amd.field_components.hash_ensemble import posenc_window
        win = posenc_window(window, 0, he.n_hash_encodings - 1, he.n_hash_encodings).to(torch.float32)
    return {"G": G, "code": code, "window": win, "n_rows": n_rows, "key": seed}


def _sharded_worker(rank, world, port, out_dir, n_buckets=8, log2_hashmap_size=8):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    he = HashEnsemble(HashEnsembleConfig(3, TCNNHashEncodingConfig(n_levels=3, log2_hashmap_size=log2_hashmap_size), True,
                                         True), seed=5)
    with torch.no_grad():
        he.tables.mul_(1e3)
    opt = ShardedTableAdam(he, lr=5e-3, eps=1e-15, world_size=world, rank=rank, ops=_TorchTableOps(), n_buckets=n_buckets)
    inv = torch.tensor([1.0 / 64.0])
    log = []
    small = torch.full((5,), float(rank + 1))            # stands for the small parameters' gradient bucket
    for it in range(5):
        poison = (it == 2 and rank == 1)                 # only ONE rank produces an inf: everybody must skip
        he.grad_sink.entries = [_fake_entry(he, 100 * it + rank, poison=poison)]
        he.grad_sink.nonfinite = torch.zeros(1)
        found = torch.zeros(1)
        if it == 4:
            # rank 1's rays produced no samples: its backward never completes a G, rank 0's starts the reduce-scatter from
            # inside the backward.  The trainer's order -- ensure_reduce_started(), then the all-reduce of the small
            # gradients -- keeps the collectives aligned (rank 1 joins with zeros)
            if rank == 1:
                he.grad_sink.entries = []
            else:
                he.grad_sink.expect()
                he.grad_sink.arrived()
                assert opt._early == "done"
            opt.ensure_reduce_started()
            assert opt._early == "done"
            bucket = small.clone()
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
            assert torch.equal(bucket, torch.full((5,), 3.0))
        elif it % 2 == 1:
            # the route a real backward takes: the sink announces completion and the optimizer reduce-scatters at once
            he.grad_sink.expect()
            he.grad_sink.expect()
            he.grad_sink.arrived()
            assert opt._early is None                          # one backward still pending
            he.grad_sink.arrived()
            assert opt._early == "done"
        opt.check_finite(found)
        assert opt._early is None
        dist.all_reduce(found, op=dist.ReduceOp.MAX)
        opt.step(found_inf=found, inv_scale=inv)
        if float(found) != 0:
            opt.rollback_step()                          # what the trainer does once it has read the flag
        log.append(float(found))
    f16 = he.tables_f16.detach().clone()
    opt.gather_master()
    torch.save({"log": log, "f16": f16, "master": he.tables.detach().clone(), "shard": opt.shard, "n": opt.n,
                "buckets": opt.n_buckets},
               os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


_WINDOWS = [1.0, 1.0, 1.4, 2.0, 2.0, 2.6, 1.0]          # per step (the last: a window that shrank -- the width must not)
_POISON = {0: 0, 2: 1, 3: 0}                             # step -> the rank whose gradient holds an inf: the very first step,
#                                                          the FIRST step at a new width (advisor, round 5), a later one


def _model_compact_width(w, floor, Hp):
    """HashEnsemble.compact_width without its device test (the CPU stand-in for what the model's forward decides)."""
    import math
    if w == 1:
        width = 1
    else:
        n = int(math.ceil(w))
        width = 1 << (n - 1).bit_length()
        width = width if 2 <= width < Hp else 0
    if width and floor > width:
        width = floor if floor < Hp else 0
    return width


def _window_worker(rank, world, port, out_dir, mode):
    """ShardedTableAdam over a schedule of coarse-to-fine windows: the exchange always at full width (``full``), following
    the window in the 32-grid layout (``narrow``), or following it while the HashEnsemble trains COMPACT copies of its first
    W grids, handed over 1 -> 2 -> full (``compact``: the gathered packed buffer is the copy's working table).  4 grids
    (padded), widths 1, 1, 2, 2, 2, full, full."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    he = HashEnsemble(HashEnsembleConfig(4, TCNNHashEncodingConfig(n_levels=3, log2_hashmap_size=10), True, True), seed=5)
    with torch.no_grad():
        he.tables.mul_(1e3)
    now = {"w": None}
    opt = ShardedTableAdam(he, lr=5e-3, eps=1e-15, world_size=world, rank=rank, ops=_TorchTableOps(), n_buckets=4,
                           width_source=(lambda: now["w"]) if mode != "full" else None)
    opt._buffers()
    inv = torch.tensor([1.0 / 64.0])
    log, widths, reads, layouts = [], [], [], []
    for it, w in enumerate(_WINDOWS):
        now["w"] = w
        cw = _model_compact_width(w, he.min_compact_width, 4) if mode == "compact" else 0
        # what the step's forward would READ: the compact copy's working table, or the 4-grid working tables
        if cw:
            comp = he.enter_compact(cw)
            reads.append(comp["f16"].clone())
        else:
            he.leave_first_grid_phase()
            reads.append(he.tables_f16.clone())
        layouts.append(cw)
        poison = _POISON.get(it) == rank
        entry = _fake_entry(he, 100 * it + rank, poison=poison, window=w, ones=(w == 1.0))
        if mode == "compact" and cw == 1:
            entry["code"] = he.first_grid_code(entry["n_rows"])          # (the phase's [rows, 1] code of ones)
        he.grad_sink.entries = [] if (it == 4 and rank == 1) else [entry]
        he.grad_sink.nonfinite = torch.zeros(1)
        found = torch.zeros(1)
        if it == 4:
            opt.ensure_reduce_started()                  # (rank 1 has no samples this step: it joins with zeros)
        elif it % 2 == 1:
            he.grad_sink.expect()
            he.grad_sink.arrived()
            assert opt._early == "done"
        opt.check_finite(found)
        dist.all_reduce(found, op=dist.ReduceOp.MAX)
        opt.step(found_inf=found, inv_scale=inv)
        if float(found) != 0:
            opt.rollback_step()
        log.append(float(found))
        widths.append(opt._last_width)
    he.leave_first_grid_phase()
    f16 = he.tables_f16.detach().clone()
    opt._sync_compact()
    b = opt._buffers()
    moments = (b["exp_avg"].clone(), b["exp_avg_sq"].clone())
    opt.gather_master()
    torch.save({"log": log, "widths": widths, "f16": f16, "master": he.tables.detach().clone(), "moments": moments,
                "report": opt.comm_report(), "buckets": opt.n_buckets, "reads": reads, "layouts": layouts,
                "pieces": [opt._narrow_pieces(w) for w in (1, 2)]}, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_that_follows_the_window_equals_the_full_exchange_bit_for_bit(tmp_path):
    """While ceil(window) <= W < H only the grids [0, W) are exchanged (reduce-scatter, shard Adam, all-gather on
    [entry][f][W]): working tables, master weights, both moments and the skip decisions are those of the full-width
    exchange bit for bit -- through poisoned steps (the very first one, the first one at a new width, a later one), a rank
    without samples, and a window that shrinks again.  Round 6: the same with the HashEnsemble training COMPACT copies of
    its first W grids (hand-overs 1 -> 2 -> full): what every step's forward reads is what it reads in the 32-grid layout."""
    res = {}
    for mode in ("full", "narrow", "compact"):
        out = tmp_path / mode
        out.mkdir()
        mp.spawn(_window_worker, args=(2, _free_port(), str(out), mode), nprocs=2, join=True)
        res[mode] = [torch.load(out / f"w{r}.pt") for r in range(2)]
    assert res["full"][0]["widths"] == [4] * len(_WINDOWS)
    for mode in ("narrow", "compact"):
        assert res[mode][0]["widths"] == res[mode][1]["widths"] == [1, 1, 2, 2, 2, 4, 4]
        assert res[mode][0]["report"]["exchange_width"] == 4 and res[mode][0]["report"]["grids"] == 4
        # narrow steps travel in FEWER, longer pieces (one reduce-scatter at width 1, two at width 2, four at full width)
        assert res[mode][0]["buckets"] == 4 and res[mode][0]["pieces"] == [1, 2]
        assert res[mode][0]["report"]["reduce_scatter_calls"] == 4
        for r in range(2):
            assert res["full"][r]["log"] == res[mode][r]["log"] == [1.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0]
            assert torch.equal(res["full"][r]["f16"], res[mode][r]["f16"])
            assert torch.equal(res["full"][r]["master"], res[mode][r]["master"])
            for a, b in zip(res["full"][r]["moments"], res[mode][r]["moments"]):
                assert torch.equal(a, b)
        assert torch.equal(res[mode][0]["f16"], res[mode][1]["f16"])
        # the grids the window never reached: untouched
        assert bool((res[mode][0]["moments"][0].view(-1, 4)[:, 3] == 0).all())
    # the compact copies: 1, 1, 2, 2, 2 grids wide, then the full layout; every step's forward reads the bits the 32-grid
    # layout holds for those grids at that step -- on both ranks, through the skipped steps
    assert res["compact"][0]["layouts"] == [1, 1, 2, 2, 2, 0, 0]
    for r in range(2):
        for it, (got, cw) in enumerate(zip(res["compact"][r]["reads"], res["compact"][r]["layouts"])):
            want = res["full"][r]["reads"][it]
            want = want[:, :, :cw] if cw else want
            assert torch.equal(got.reshape(want.shape), want), (r, it, cw)


def _beyond_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    he = HashEnsemble(HashEnsembleConfig(4, TCNNHashEncodingConfig(n_levels=3, log2_hashmap_size=10), True, True), seed=5)
    opt = ShardedTableAdam(he, world_size=world, rank=rank, ops=_TorchTableOps(), n_buckets=4, width_source=lambda: 1.0)
    raised = False
    for it in range(2):
        he.grad_sink.entries = [_fake_entry(he, it + rank, window=3.0)]          # evaluated with another window
        he.grad_sink.nonfinite = torch.zeros(1)
        found = torch.zeros(1)
        try:
            opt.check_finite(found)
        except RuntimeError as e:
            raised = "beyond the exchanged width" in str(e)
            break
        opt.step(found_inf=found, inv_scale=torch.ones(1))
    torch.save({"raised": raised}, os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_gradient_beyond_the_exchanged_width_is_an_error(tmp_path):
    mp.spawn(_beyond_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all(torch.load(tmp_path / f"b{r}.pt")["raised"] for r in range(2))


def test_sharded_table_adam_world2_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_sharded_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "s0.pt")
    r1 = torch.load(tmp_path / "s1.pt")
    assert r0["log"] == r1["log"] == [0.0, 0.0, 1.0, 0.0, 0.0]         # the poisoned step is skipped on BOTH ranks
    assert torch.equal(r0["f16"], r1["f16"]) and torch.equal(r0["master"], r1["master"])
    assert torch.equal(r0["f16"], r0["master"].half())
    assert r0["shard"] % 1024 == 0 and 2 * r0["shard"] >= r0["n"]
    # single-process reference: torch Adam on the fp16-rounded average of the two ranks' dense gradients
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    he = HashEnsemble(HashEnsembleConfig(3, TCNNHashEncodingConfig(n_levels=3, log2_hashmap_size=8), True, True), seed=5)
    with torch.no_grad():
        he.tables.mul_(1e3)
    opt = torch.optim.Adam([he.tables], lr=5e-3, eps=1e-15)
    ops = _TorchTableOps()
    for it in (0, 1, 3, 4):
        g = sum((ops.dense(he, _fake_entry(he, 100 * it + r)) * 0.5).half().float() for r in range(1 if it == 4 else 2))
        he.tables.grad = g.half().float() / 64.0
        opt.step()
    d = (he.tables.detach() - r0["master"]).abs().max().item()
    assert d <= 2e-5, d                                   # fp16 summation order inside the reduce-scatter


def test_bucketed_exchange_equals_the_one_piece_exchange_bit_for_bit(tmp_path):
    """The reduce-scatter in pieces (piece k of every rank's shard per collective, the expansion of piece k + 1 beside it,
    two alternating buffers) against the one-piece exchange of rounds 1-3: the same values reduced per element -- tables,
    master weights and skip decisions identical bit for bit, also through the poisoned step and the zero-sample rank."""
    res = {}
    for k in (1, 4):
        out = tmp_path / f"k{k}"
        out.mkdir()
        mp.spawn(_sharded_worker, args=(2, _free_port(), str(out), k, 10), nprocs=2, join=True)
        res[k] = [torch.load(out / f"s{r}.pt") for r in range(2)]
    assert res[1][0]["buckets"] == 1 and res[4][0]["buckets"] == 4
    assert res[4][0]["shard"] % (4 * 1024) == 0
    for r in range(2):
        assert res[1][r]["log"] == res[4][r]["log"] == [0.0, 0.0, 1.0, 0.0, 0.0]
        # (the shard sizes differ by the alignment unit: compare the table, not the padding)
        assert torch.equal(res[1][r]["f16"], res[4][r]["f16"])
        assert torch.equal(res[1][r]["master"], res[4][r]["master"])
    assert torch.equal(res[4][0]["f16"], res[4][1]["f16"])


# ---- strong scaling: one ray batch sliced over the ranks, loss denominators made global -------------------------------
def _union_batch(seed=0, R=64):
    """Per-ray / per-sample quantities of a synthetic 64-ray batch with the reference's masked-mean structure
    (models/base.py:90-249): masked rgb rays, background rays, depth rays, empty / near sample masks, per-ray distortion
    sums; the last 9 rays carry no samples (so ``ray_id.max() + 1`` < R and the second slice ends early)."""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(0, 9, (R,), generator=g)
    counts[-9:] = 0
    d = {"w_rgb": torch.rand(R, generator=g).requires_grad_(True), "m_rgb": torch.rand(R, generator=g) < 0.6,
         "w_alpha": torch.rand(R, generator=g).requires_grad_(True), "m_bg": torch.rand(R, generator=g) < 0.4,
         "w_depth": torch.rand(R, generator=g).requires_grad_(True), "m_depth": torch.rand(R, generator=g) < 0.7,
         "w_dist": torch.rand(R, generator=g).requires_grad_(True), "counts": counts}
    S = int(counts.sum())
    d["ray_of"] = torch.repeat_interleave(torch.arange(R), counts)
    d["w_empty"] = torch.rand(S, generator=g).requires_grad_(True)
    d["m_empty"] = torch.rand(S, generator=g) < 0.5
    d["w_near"] = torch.rand(S, generator=g).requires_grad_(True)
    d["m_near"] = torch.rand(S, generator=g) < 0.3
    return d


def _masked_mean(v, m):
    m = m.to(v.dtype)
    return (v * m).sum() / m.sum().clamp(min=1.0)


def _loss_terms(d, rays, samples):
    """The six terms on a slice: (terms [6], raw counts [5], raw n_eff)."""
    has = d["counts"][rays] > 0
    n_eff = (torch.nonzero(has).max() + 1).float() if has.any() else torch.zeros(())
    terms = torch.stack([_masked_mean(d["w_rgb"][rays] ** 2, d["m_rgb"][rays]),
                         _masked_mean(d["w_alpha"][rays].abs(), d["m_bg"][rays]),
                         _masked_mean(d["w_depth"][rays] ** 2, d["m_depth"][rays]),
                         _masked_mean(d["w_empty"][samples] ** 2, d["m_empty"][samples]),
                         _masked_mean(d["w_near"][samples] ** 2, d["m_near"][samples]),
                         (d["w_dist"][rays] * has).sum() / n_eff.clamp(min=1.0)])
    counts = torch.stack([d["m_rgb"][rays].sum(), d["m_bg"][rays].sum(), d["m_depth"][rays].sum(),
                          d["m_empty"][samples].sum(), d["m_near"][samples].sum()]).float()
    return terms, counts, n_eff


def _normaliser_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.parallel import all_reduce_gradients, global_normaliser_scales
    d = _union_batch()
    R = d["counts"].shape[0]
    per = R // world
    rays = torch.arange(rank * per, (rank + 1) * per)
    samples = torch.nonzero((d["ray_of"] >= rank * per) & (d["ray_of"] < (rank + 1) * per))[:, 0]
    terms, counts, n_eff = _loss_terms(d, rays, samples)
    scales = global_normaliser_scales(counts, n_eff, per, world, rank)
    loss = (terms * scales).sum()
    leaves = [d[k] for k in ("w_rgb", "w_alpha", "w_depth", "w_dist", "w_empty", "w_near")]
    loss.backward()
    all_reduce_gradients(leaves, world)                       # the trainer's gradient average
    torch.save({"loss": loss.detach(), "grads": [p.grad for p in leaves]}, os.path.join(out_dir, f"n{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_global_loss_normalisers_world2_equal_the_union_batch(tmp_path):
    """Two ranks on half-batches with ``global_normaliser_scales`` reproduce the single-process loss and gradient on the
    union batch: mean of the ranks' losses == the loss, averaged gradients == its gradient."""
    port = _free_port()
    mp.spawn(_normaliser_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = _union_batch()
    R = d["counts"].shape[0]
    terms, _, n_eff = _loss_terms(d, torch.arange(R), torch.arange(d["ray_of"].shape[0]))
    assert n_eff == R - 9
    loss = terms.sum()
    loss.backward()
    r0, r1 = torch.load(tmp_path / "n0.pt"), torch.load(tmp_path / "n1.pt")
    assert torch.allclose((r0["loss"] + r1["loss"]) / 2, loss.detach(), rtol=1e-6)
    for k, name in enumerate(("w_rgb", "w_alpha", "w_depth", "w_dist", "w_empty", "w_near")):
        assert torch.allclose(r0["grads"][k], d[name].grad, rtol=1e-5, atol=1e-8), name
        assert torch.equal(r0["grads"][k], r1["grads"][k])
    # without the re-weighting the halves do NOT add up (different mask counts per half)
    h0, c0, _ = _loss_terms(d, torch.arange(R // 2), torch.nonzero(d["ray_of"] < R // 2)[:, 0])
    h1, c1, _ = _loss_terms(d, torch.arange(R // 2, R), torch.nonzero(d["ray_of"] >= R // 2)[:, 0])
    assert not torch.allclose((h0.sum() + h1.sum()) / 2, loss.detach(), rtol=1e-3)


# ---- level-parallel exchange (engine/level_parallel.py): the collectives' plumbing on CPU tensors --------------------------
class _TorchLPOps:
    """Torch restatement of csrc/level_parallel.hip's pack / run / unpack entry points on CPU byte buffers with the layout of
    ``nsx_lp_layout_make`` (host-only: the library's own), so that the whole exchange -- sizes, payloads, counts, blocks,
    gradient planes -- runs over gloo.  The two per-source-rank KERNELS are stand-ins with a closed form the test can
    recompute from the ranks' inputs:

        feature[s][2 i + f]  = fp16( x[s][0] * (l + 1) + code[slot[s]][0] / 2 + f ),  l = the owner's i-th level
        dx partial[s][d]     = (d + 1) * (owner + 1) * sum_c dz[s][c]
        dcode partial[r][h]  = (h + 1) * sum_{s in row r} sum_c dz[s][c]
        G[plane of (j, r)][0][0] += sum_{s in row r} sum_c dz[s][c]
    """

    def __init__(self, owner_rank, levels):
        self.owner, self.levels = owner_rank, list(levels)

    @staticmethod
    def layout(W, S_cap, R_cap, H, n2):
        from nersemble_amd.engine.level_parallel import NativeLPOps
        return NativeLPOps.layout(W, S_cap, R_cap, H, n2)

    @staticmethod
    def _count(S, n_dev):
        return int(S) if n_dev is None else max(0, min(int(n_dev.item()), int(S)))

    @staticmethod
    def _f32(buf, off, n):
        return buf[off:off + 4 * n].view(torch.float32)

    @staticmethod
    def _i32(buf, off, n):
        return buf[off:off + 4 * n].view(torch.int32)

    @staticmethod
    def _f16(buf, off, n):
        return buf[off:off + 2 * n].view(torch.float16)

    def fwd_pack(self, lay, pn, slot, S, n_dev, codes, rows, payload):
        n = self._count(S, n_dev)
        payload[lay.f_count:lay.f_count + 8].view(torch.int64)[0] = n
        self._f32(payload, lay.f_pn, n * 3).copy_(pn[:n].reshape(-1))
        self._i32(payload, lay.f_slot, n).copy_(slot[:n])
        self._f32(payload, lay.f_codes, rows * lay.H).copy_(codes[:rows].reshape(-1))

    def fwd_run(self, lay, gathered, ex, tables, geom, window, send, codes_packed):
        W, n_own = lay.W, lay.n2 // 2
        base = 0
        for j in range(W):
            blk = gathered[j * lay.fwd_bytes:(j + 1) * lay.fwd_bytes]
            rows = int(ex.rows_host[j])
            codes = self._f32(blk, lay.f_codes, rows * lay.H).view(rows, lay.H)
            codes_packed[base:base + rows].copy_(codes)
            base += rows
            Sj = int(ex.sizes_host[j])
            n = min(Sj, int(blk[lay.f_count:lay.f_count + 8].view(torch.int64)[0]))
            if n == 0:
                continue
            x = self._f32(blk, lay.f_pn, n * 3).view(n, 3)
            sl = self._i32(blk, lay.f_slot, n).long()
            out = self._f16(send[j * lay.feat_bytes:(j + 1) * lay.feat_bytes], 0, n * lay.n2).view(n, n_own, 2)
            for i in range(n_own):
                for f in range(2):
                    out[:, i, f] = (x[:, 0] * (self.levels[i] + 1) + codes[sl, 0] / 2 + f).half()

    def fwd_unpack(self, lay, recv, S, n_dev, feats):
        n, n_own = self._count(S, n_dev), lay.n2 // 2
        for j in range(lay.W):
            blk = self._f16(recv[j * lay.feat_bytes:(j + 1) * lay.feat_bytes], 0, n * lay.n2).view(n, n_own, 2)
            for i in range(n_own):
                l = lay.level_of[j * n_own + i]              # where owner j's i-th level sits in the feature row
                feats[:n, 2 * l:2 * l + 2] = blk[:, i]

    def bwd_pack(self, lay, dout, pn, slot, S, n_dev, send):
        n = self._count(S, n_dev)
        for j in range(lay.W):
            blk = send[j * lay.bwd_bytes:(j + 1) * lay.bwd_bytes]
            blk[lay.b_count:lay.b_count + 8].view(torch.int64)[0] = n
            cols = [c for i in range(lay.n2 // 2) for c in (2 * lay.level_of[j * (lay.n2 // 2) + i], 2 * lay.level_of[j * (lay.n2 // 2) + i] + 1)]
            self._f16(blk, lay.b_dz, n * lay.n2).view(n, lay.n2).copy_(dout[:n][:, cols].half())
            self._f32(blk, lay.b_pn, n * 3).copy_(pn[:n].reshape(-1))
            self._i32(blk, lay.b_slot, n).copy_(slot[:n])

    def bwd_run(self, lay, recv, gathered, ex, tables, geom, window, G, ret, nonfinite, dz32=None):
        plane = 0
        for j in range(lay.W):
            rows, Sj = int(ex.rows_host[j]), int(ex.sizes_host[j])
            blk = recv[j * lay.bwd_bytes:(j + 1) * lay.bwd_bytes]
            rj = ret[j * lay.ret_bytes:(j + 1) * lay.ret_bytes]
            dcode = self._f32(rj, lay.r_dcode, lay.R_cap * lay.H).view(lay.R_cap, lay.H)
            dcode.zero_()
            n = min(Sj, max(0, int(blk[lay.b_count:lay.b_count + 8].view(torch.int64)[0])))
            if Sj > 0 and n > 0:
                dz = self._f16(blk, lay.b_dz, n * lay.n2).view(n, lay.n2).float().sum(dim=1)
                sl = self._i32(blk, lay.b_slot, n).long()
                dx = self._f32(rj, lay.r_dx, n * 3).view(n, 3)
                for d in range(3):
                    dx[:, d] = (d + 1) * (self.owner + 1) * dz
                per_row = torch.zeros((rows,)).index_add_(0, sl, dz)
                dcode[:rows] = per_row[:, None] * torch.arange(1, lay.H + 1, dtype=torch.float32)[None, :]
                if G is not None:
                    G[plane:plane + rows, 0, 0] += per_row
            plane += rows

    def bwd_unpack(self, lay, ret_recv, S, n_dev, rows, dx, dcode):
        n = self._count(S, n_dev)
        dx[:n] = 0
        dcode[:rows] = 0
        for j in range(lay.W):
            rj = ret_recv[j * lay.ret_bytes:(j + 1) * lay.ret_bytes]
            dx[:n] += self._f32(rj, lay.r_dx, n * 3).view(n, 3)
            dcode[:rows] += self._f32(rj, lay.r_dcode, lay.R_cap * lay.H).view(lay.R_cap, lay.H)[:rows]

    def shared_columns(self, x, S, tables, H, geom, code, slot, window, cols):
        n_own = geom.n_levels
        out = self._f16(cols, 0, S * 2 * n_own).view(S, n_own, 2)
        for i in range(n_own):
            for f in range(2):
                out[:, i, f] = (x[:, 0] * (self.levels[i] + 1) + code[slot.long(), 0] / 2 + f).half()


def _lp_rank_inputs(rank, world, H, L):
    """What rank ``rank`` brings to a step (every rank can rebuild everyone's): ragged sample sets, ranks 1 and world - 2
    without samples (world > 2; rank 1 of 2 in the second pass), 1 ... 3 code rows, fewer kept samples than marched ones."""
    g = torch.Generator().manual_seed(1000 + rank)
    S = 0 if (world > 2 and rank in (1, world - 2)) else 3 + 2 * rank
    rows = 1 + rank % 3
    x = torch.rand((S, 3), generator=g)
    slot = torch.randint(0, rows, (S,), generator=g, dtype=torch.int32)
    code = torch.randn((rows, H), generator=g)
    kept = S if rank % 2 == 0 else max(S - 2, 0)                   # the backward's device-side count
    dout = (torch.randn((S, 2 * L), generator=g) * 4).half().float()    # fp16-representable, as nsx_mlp_bwd emits
    return S, rows, x, slot, code, kept, dout


def _lp_plumbing_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from nersemble_amd import _lib
    from nersemble_amd.engine.level_parallel import LevelParallel, level_assignment, sub_geometry_levels
    L, H = (8 if world == 2 else 2 * world), 4
    n_own = L // world
    geom = _lib.grid_geometry(n_levels=L, per_level_scale=1.3, base_resolution=4, log2_hashmap_size=9)
    f16 = torch.zeros((geom.total_entries, 2, H), dtype=torch.float16)
    he = SimpleNamespace(geom=geom, n_hash_encodings=H, wait_tables=lambda: None, half_tables=lambda: f16,
                         tables=SimpleNamespace(data=f16.float()))
    own = level_assignment(L, world)
    lp = LevelParallel(he, world, rank, ops=_TorchLPOps(rank, own[rank]))
    # the balanced assignment: a partition of the levels, the cheapest with the dearest (rank r: levels r and L - 1 - r at two
    # levels per rank); the contiguous one of round 5 is still there
    ok = sorted(l for lv in own for l in lv) == list(range(L)) and all(len(lv) == n_own for lv in own)
    ok &= lp.levels == own[rank] and lp.n_own == n_own
    if n_own == 2:
        ok &= own[rank] == sorted([rank, L - 1 - rank])
    ok &= level_assignment(L, world, balanced=False)[rank] == list(range(rank * n_own, (rank + 1) * n_own))
    ok &= lp.ranges == [(int(geom.offset[l]), int(geom.offset[l + 1])) for l in own[rank]]
    # sub-geometry: the owned levels, one after the other from entry 0
    sg = sub_geometry_levels(geom, own[rank])
    at = 0
    for i, l in enumerate(own[rank]):
        ok &= sg.scale[i] == geom.scale[l] and sg.res[i] == geom.res[l] and sg.size[i] == geom.size[l]
        ok &= sg.hashed[i] == geom.hashed[l] and sg.offset[i] == at
        at += int(geom.offset[l + 1]) - int(geom.offset[l])
    ok &= sg.total_entries == lp.n_entries == at
    everyone = [_lp_rank_inputs(r, world, H, L) for r in range(world)]
    S, rows, x, slot, code, kept, dout = everyone[rank]
    window = torch.ones((H,))
    for a2a_native in (False, True):             # the gloo stand-in, then the all_to_all_single branch RCCL takes
        lp._a2a_native = a2a_native
        lp.begin_step()
        before = dict(lp.stats)
        # ---- forward: every level's columns of MY samples, whoever owns the level
        feats = lp.features(x, code, slot, window)
        ex = lp.last_exchange
        ok &= ex.sizes == [e[0] for e in everyone] and ex.rows == [e[1] for e in everyone]
        ok &= ex.S_cap == max(1, max(e[0] for e in everyone)) and ex.n_planes == sum(e[1] for e in everyone)
        want = torch.empty((S, 2 * L), dtype=torch.float16)
        for l in range(L):
            for f in range(2):
                want[:, 2 * l + f] = (x[:, 0] * (l + 1) + code[slot.long(), 0] / 2 + f).half()
        ok &= feats.shape == (S, 2 * L) and torch.equal(feats, want)
        ok &= torch.equal(ex.codes_packed, torch.cat([e[4] for e in everyone]))
        # ---- backward under a device-side count: partials summed over the owners, planes in (source rank, row) order
        n_dev = torch.tensor([kept], dtype=torch.int64)
        dx, dcode = lp.backward(x, slot, dout, n_dev=n_dev)
        colsum = torch.stack([sum(dout[:kept, 2 * l] + dout[:kept, 2 * l + 1] for l in own[o]) for o in range(world)], dim=1)
        want_dx = torch.stack([(d + 1) * (colsum * torch.arange(1, world + 1)[None, :]).sum(dim=1) for d in range(3)], dim=1)
        ok &= torch.allclose(dx[:kept], want_dx, rtol=1e-5, atol=1e-4)
        per_row = torch.zeros((rows,)).index_add_(0, slot[:kept].long(), dout[:kept].sum(dim=1))
        ok &= torch.allclose(dcode, per_row[:, None] * torch.arange(1, H + 1)[None, :], rtol=1e-5, atol=1e-4)
        G = lp.G[:lp.planes * lp.n_entries * 2].view(lp.planes, lp.n_entries, 2)
        plane = 0
        for r, (Sr, rr, _, slr, _, kr, dr) in enumerate(everyone):
            mine = sum(dr[:kr, 2 * l] + dr[:kr, 2 * l + 1] for l in own[rank])
            wantG = torch.zeros((rr,)).index_add_(0, slr[:kr].long(), mine)
            ok &= torch.allclose(G[plane:plane + rr, 0, 0], wantG, rtol=1e-5, atol=1e-4)
            plane += rr
        ok &= lp.planes == plane and G.abs().sum().item() == G[:, 0, 0].abs().sum().item()
        # ONE host exchange and FOUR collectives per step, whatever this rank brought
        ok &= lp.stats["host_exchanges"] - before["host_exchanges"] == 1
        ok &= lp.stats["collectives"] - before["collectives"] == 4
    # the occupancy update: identical positions everywhere, only the column blocks travel
    xs = torch.rand((7, 3), generator=torch.Generator().manual_seed(5))
    cs = torch.randn((2, H), generator=torch.Generator().manual_seed(6))
    ss = torch.tensor([0, 1, 1, 0, 1, 0, 0], dtype=torch.int32)
    before = dict(lp.stats)
    with lp.shared():
        fs = lp.features(xs, cs, ss, window)
    want = torch.empty((7, 2 * L), dtype=torch.float16)
    for l in range(L):
        for f in range(2):
            want[:, 2 * l + f] = (xs[:, 0] * (l + 1) + cs[ss.long(), 0] / 2 + f).half()
    ok &= torch.equal(fs, want) and lp.stats["collectives"] - before["collectives"] == 1
    ok &= lp.stats["host_exchanges"] == before["host_exchanges"]
    # a rank whose loss never reached the hash features joins the backward exchange with zero rows
    lp.begin_step()
    lp.features(x, code, slot, window)
    if rank == 0:
        lp.join_backward()
    else:
        lp.backward(x, slot, dout, n_dev=torch.tensor([kept], dtype=torch.int64))
    G = lp.G[:lp.planes * lp.n_entries * 2].view(lp.planes, lp.n_entries, 2)
    ok &= float(G[:everyone[0][1], 0, 0].abs().sum()) == 0.0                        # rank 0's planes: nothing arrived
    # every rank's levels become current everywhere, from the owners' compact tensors
    full = torch.zeros((geom.total_entries, 2, 4))
    lp.gather_entry_ranges(full, torch.full((lp.n_entries, 2, 4), float(rank + 1)))
    for r in range(world):
        for a, b in lp.ranges_of[r]:
            ok &= bool((full[a:b] == r + 1).all())
    ok &= lp.stats["bytes_in"] > 0
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, f"lp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_level_assignment_pairs_the_cheapest_level_with_the_dearest():
    """The balanced (snake) assignment of levels to ranks: a partition, equal counts, ascending within a rank; at two levels per
    rank, rank r holds levels r and L - 1 - r."""
    from nersemble_amd.engine.level_parallel import level_assignment
    for L, W in ((16, 8), (16, 4), (16, 2), (16, 16), (8, 2), (6, 3)):
        own = level_assignment(L, W)
        assert sorted(l for lv in own for l in lv) == list(range(L)) and all(len(lv) == L // W and lv == sorted(lv) for lv in own)
        assert level_assignment(L, W, balanced=False) == [list(range(r * (L // W), (r + 1) * (L // W))) for r in range(W)]
    assert level_assignment(16, 8) == [[k, 15 - k] for k in range(8)]
    assert level_assignment(16, 4) == [[0, 7, 8, 15], [1, 6, 9, 14], [2, 5, 10, 13], [3, 4, 11, 12]]


@pytest.mark.parametrize("world", [2, 8])
def test_level_parallel_exchange_plumbing(world, tmp_path):
    """The whole level-parallel exchange on CPU tensors over gloo with the library's own payload layout
    (``nsx_lp_layout_make``) and torch stand-ins for the device side: level ownership and sub-geometries, the one host-side
    size exchange, ragged sample sets and code rows, ranks without samples, device-side counts below the capacity, the
    all-gather of the forward payloads, the three block all-to-alls (gloo stand-in AND the all_to_all_single branch RCCL
    takes), the gradient planes in (source rank, code row) order, partial dL/dx / code gradients summed over the owners, the
    shared-input exchange of the occupancy update, a rank that joins the backward without one, the broadcast of the ranks'
    entry ranges.  world 8 = 2 levels per rank, the shape of BASELINE.json configs[4].  The kernels between the collectives
    run on the GPU (tests/test_sharded_gpu.py)."""
    port = _free_port()
    mp.spawn(_lp_plumbing_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert torch.load(tmp_path / f"lp{r}.pt")["ok"], r
