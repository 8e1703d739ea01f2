"""GPU: the data-parallel table optimizer (engine/sharded_adam.py) -- its three kernels against torch, and the whole
reduce-scatter / sharded Adam / all-gather step through a real (single-rank) RCCL process group against the
single-GPU fused optimizer.  The two-rank collective plumbing itself is covered on CPU (tests/test_parallel_cpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _he(H, cuda, seed=0):
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    cfg = HashEnsembleConfig(H, TCNNHashEncodingConfig(n_levels=6, log2_hashmap_size=11), True, True)
    he = HashEnsemble(cfg, seed=seed).to(cuda)
    with torch.no_grad():
        he.tables.mul_(3000)
    return he


@pytest.fixture(scope="module")
def single_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("H", [1, 8, 32])
def test_expand_f16_matches_fp32_expand(H, cuda):
    import ctypes as C
    from nersemble_amd._lib import check, lib, ptr, stream
    he = _he(H, cuda)
    g = torch.Generator(device=cuda).manual_seed(H)
    T = 5
    G = torch.randn((T, he.geom.total_entries, 2), device=cuda, generator=g) * 1e-2
    G[torch.rand(G.shape, device=cuda, generator=g) < 0.5] = 0
    code = torch.randn((T, H), device=cuda, generator=g)
    win = torch.rand((H,), device=cuda, generator=g)
    dense = torch.empty_like(he.tables)
    check(lib().nsx_hash_grad_expand(ptr(G), T, ptr(code), code.stride(0), ptr(win), H, C.byref(he.geom), ptr(dense), 0,
                                     stream()), "expand")
    out = torch.zeros(he.tables.numel() + 64, dtype=torch.float16, device=cuda)
    check(lib().nsx_hash_grad_expand_f16(ptr(G), T, ptr(code), code.stride(0), ptr(win), H, C.byref(he.geom), ptr(out),
                                         0.5, 0, stream()), "expand_f16")
    assert torch.equal(out[:dense.numel()], (dense.reshape(-1) * 0.5).half())
    assert out[dense.numel():].abs().max().item() == 0
    check(lib().nsx_hash_grad_expand_f16(ptr(G), T, ptr(code), code.stride(0), ptr(win), H, C.byref(he.geom), ptr(out),
                                         0.5, 1, stream()), "expand_f16 accumulate")
    want = ((dense.reshape(-1) * 0.5).half().float() + dense.reshape(-1) * 0.5).half()
    assert torch.equal(out[:dense.numel()], want)


def test_check_finite_f16_and_f16grad_adam(cuda):
    from nersemble_amd._lib import check, lib, ptr, stream
    n = 100_003
    g = torch.Generator(device=cuda).manual_seed(0)
    for pos, val in ((None, 0.0), (0, float("inf")), (n - 1, float("nan")), (54321, float("-inf"))):
        x = torch.randn(n + 5, device=cuda, generator=g).half()[:n + 5]
        x = x[:n] if x.data_ptr() % 16 == 0 else x
        if pos is not None:
            x[pos] = val
        found = torch.zeros(1, device=cuda)
        check(lib().nsx_check_finite_f16(ptr(x), n, ptr(found), stream()), "check")
        assert found.item() == (0.0 if pos is None else 1.0)
    # Adam on an fp16 gradient == torch Adam on the same gradient in fp32
    p0 = torch.randn(n, device=cuda, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=5e-3, eps=1e-15)
    master, m, v = p0.clone(), torch.zeros(n, device=cuda), torch.zeros(n, device=cuda)
    f16 = torch.empty(n, dtype=torch.float16, device=cuda)
    inv = torch.tensor([1.0 / 128], device=cuda)
    zero = torch.zeros(1, device=cuda)
    for step in range(1, 4):
        grad = (torch.randn(n, device=cuda, generator=g) * 3).half()
        ref.grad = grad.float() / 128
        opt.step()
        check(lib().nsx_adam_dense_f16grad(ptr(grad), n, ptr(master), ptr(m), ptr(v), ptr(f16), 5e-3, 0.9, 0.999, 1e-15,
                                           step, ptr(inv), ptr(zero), stream()), "adam")
        assert (master - ref.detach()).abs().max().item() <= 2e-6
        assert torch.equal(f16, master.half())
    before = master.clone()
    one = torch.ones(1, device=cuda)
    check(lib().nsx_adam_dense_f16grad(ptr(grad), n, ptr(master), ptr(m), ptr(v), ptr(f16), 5e-3, 0.9, 0.999, 1e-15, 4,
                                       ptr(inv), ptr(one), stream()), "adam skip")
    assert torch.equal(master, before)


def test_sharded_step_through_rccl_equals_fused_adam(cuda, single_rank_group):
    from nersemble_amd.engine.hash_adam import HashTableAdam
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    B, T, H = 4000, 7, 8
    g = torch.Generator(device=cuda).manual_seed(1)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    scale = 1024.0
    a, b = _he(H, cuda), _he(H, cuda)
    opt_a = HashTableAdam(a, lr=5e-3, eps=1e-15, factored=True)
    opt_b = ShardedTableAdam(b, lr=5e-3, eps=1e-15, world_size=1, rank=0)
    inv = torch.tensor([1.0 / scale], device=cuda)
    for it in range(3):
        for he, opt in ((a, opt_a), (b, opt_b)):
            found = torch.zeros(1, device=cuda)
            opt.zero_grad()
            he(x, emb, window_hash_encodings=3.0 + it, code_index=slot).backward(dout * scale)
            opt.check_finite(found)
            opt.step(found_inf=found, inv_scale=inv)
            assert found.item() == 0
        d = (a.tables - b.tables).abs()
        # the exchanged gradient is fp16: entries whose gradient is cancellation noise move by +-lr either way
        assert d.mean().item() <= 2e-6 and (d <= 1e-4).float().mean().item() >= 0.999, (it, d.max().item())
        assert torch.equal(b.half_tables(), b.tables.detach().half())
    before = b.tables.detach().clone()
    opt_b.gather_master()
    assert torch.equal(b.tables.detach(), before)


def test_training_with_sharded_table_adam(cuda, single_rank_group):
    """Trainer integration: the sharded optimizer (world of one) follows the fused single-GPU optimizer's losses."""
    from nersemble_amd.workloads import build_workload

    def run(sharded):
        torch.manual_seed(0)
        trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512, sharded_table_adam=sharded)
        losses = []
        for step in range(8):
            loss, _, _ = trainer.train_iteration(step, *data.next_train(step))
            losses.append(loss.item())
        trainer.flush_scheduler_step()
        trainer.consolidate()
        return losses

    l_s, l_f = run(True), run(False)
    assert all(np.isfinite(l_s)) and l_s[-1] < l_s[0]
    assert np.allclose(l_s, l_f, rtol=5e-3, atol=1e-5), (l_s, l_f)


def _two_rank_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)         # both ranks on cuda:0: RCCL cannot, gloo can
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)                                          # identical initial weights
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256, rank=rank, world_size=world)
    assert any(isinstance(o, ShardedTableAdam) for o in trainer.optimizers.values())
    losses = []
    for step in range(4):
        loss, _, _ = trainer.train_iteration(step, *data.next_train(step))
        losses.append(loss.item())
    trainer.flush_scheduler_step()
    model = trainer.model
    small = torch.cat([p.detach().reshape(-1).cpu() for n, p in model.named_parameters() if "tables" not in n])
    # (round 6: the ranks train the compact copy of the first grid -- its working table is the gathered packed buffer of the
    # narrow exchange; the 32-grid layout is current again after consolidate())
    compact = model.field.hash_ensemble._compact
    assert compact is not None and compact["width"] == 1 and compact.get("sharded")
    first = compact["f16"].detach().cpu().clone()
    trainer.consolidate()
    f16 = model.field.hash_ensemble.half_tables().detach().cpu()
    assert torch.equal(f16[:, :, 0:1], first)
    master = model.field.hash_ensemble.tables.detach().cpu()
    torch.save({"losses": losses, "f16": f16, "small": small, "master": master,
                "rays": data.next_train(99)[0].directions.cpu()}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_on_one_gpu(cuda, tmp_path):
    """The data-parallel step with the REAL kernels and world_size 2: two processes share the GPU (gloo moves the
    collectives).  Replicas must stay identical; the sharded fp32 master, gathered, must round to the working tables."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_two_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert all(np.isfinite(r0["losses"])) and all(np.isfinite(r1["losses"]))
    assert r0["losses"] != r1["losses"] and not torch.equal(r0["rays"], r1["rays"])      # different rays per rank
    assert torch.equal(r0["f16"], r1["f16"])                                               # identical replicas
    assert torch.equal(r0["small"], r1["small"])
    assert torch.equal(r0["master"], r1["master"]) and torch.equal(r0["master"].half(), r0["f16"])
    init = r0["master"].numel()
    assert (r0["f16"].float() - r0["master"]).abs().max().item() < 1e-3 and init > 0


# ---- strong scaling: ONE ray batch sliced over the ranks == the single-process step on the union batch ----
def _no_jitter(trainer):
    """The marcher's near-plane jitter draws from torch's global generator (one draw per ray of the local batch): switch
    it off so that the union batch and its halves march the same samples."""
    grid = trainer.model.occupancy_grid
    orig = grid.sampling

    def sampling(*args, **kw):
        kw["stratified"] = False
        return orig(*args, **kw)

    grid.sampling = sampling
    orig_count = grid.counted_march                  # (the native step drivers run the counting pass through this one)

    def counted_march(*args, **kw):
        kw["stratified"] = False
        return orig_count(*args, **kw)

    grid.counted_march = counted_march


def _union_data(n_rays_total, workload="p030_h16"):
    from nersemble_amd.data.synthetic import SyntheticNeRSembleData
    from nersemble_amd.workloads import SCENE_BOXES, WORKLOADS
    w = WORKLOADS[workload]
    box = torch.tensor(SCENE_BOXES[w["pid"]], dtype=torch.float32)
    return SyntheticNeRSembleData(box, n_timesteps=w["T"], n_rays=n_rays_total, device="cuda:0", rank=0)


def _slice_batch(bundle, batch, lo, hi):
    return bundle[lo:hi], {k: v[lo:hi] for k, v in batch.items()}


def _strong_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.workloads import build_workload
    per = 256
    bundle, batch = _union_data(per * world).next_train(0)
    res = {}
    for mode, kw in (("dense", dict(factored_table_grad=False, sharded_table_adam=False)), ("sharded", dict())):
        torch.manual_seed(19980801)
        trainer, _, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=per, rank=rank, world_size=world,
                                       global_loss_normalisers=True, **kw)
        _no_jitter(trainer)
        loss, loss_dict, _ = trainer.train_iteration(0, *_slice_batch(bundle, batch, rank * per, (rank + 1) * per))
        trainer.flush_scheduler_step()
        trainer.consolidate()
        model = trainer.model
        res[mode] = {"loss": loss.item(), "terms": {k: v.item() for k, v in loss_dict.items()},
                     # gradients of the small parameters as the optimizers saw them (unscaled, averaged over the ranks)
                     "grads": {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()
                               if "tables" not in n and p.grad is not None},
                     "tables": model.field.hash_ensemble.tables.detach().cpu(),
                     "small": torch.cat([p.detach().reshape(-1).cpu() for n, p in model.named_parameters()
                                         if "tables" not in n])}
        del trainer
        torch.cuda.empty_cache()
    torch.save(res, os.path.join(out_dir, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_half_batches_equal_one_process_on_the_union_batch(cuda, tmp_path):
    """Real kernels, world_size 2 (two processes share the GPU, gloo moves the collectives), the 512-ray batch sliced
    256 + 256 with global loss normalisers: loss, every loss term, the small parameters and the hash tables after one
    optimizer step equal the single-process step on the whole batch -- with the dense fp32 gradient all-reduce to fp32
    noise, with the sharded fp16 reduce-scatter to fp16-gradient noise."""
    import torch.multiprocessing as mp
    from nersemble_amd.workloads import build_workload
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_strong_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    bundle, batch = _union_data(512).next_train(0)
    torch.manual_seed(19980801)
    single, _, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512)
    init_tables = single.model.field.hash_ensemble.tables.detach().cpu().clone()
    _no_jitter(single)
    loss, loss_dict, _ = single.train_iteration(0, bundle, batch)
    single.flush_scheduler_step()
    single.consolidate()                    # (single-GPU default: compact first-grid phase; write grid 0 back)
    tables = single.model.field.hash_ensemble.tables.detach().cpu()
    grads = {n: p.grad.detach().float().cpu() for n, p in single.model.named_parameters()
             if "tables" not in n and p.grad is not None}
    moved = (tables - init_tables).abs() > 1e-4                    # entries the step touched (+-lr at step 1)
    assert moved.float().mean().item() > 1e-3
    for mode, tol_frac in (("dense", 0.9995), ("sharded", 0.995)):
        a, b = r0[mode], r1[mode]
        assert torch.equal(a["tables"], b["tables"]) and torch.equal(a["small"], b["small"])     # identical replicas
        # the mean of the ranks' losses is the union loss, term by term
        assert np.isclose((a["loss"] + b["loss"]) / 2, loss.item(), rtol=2e-4), (mode, a["loss"], b["loss"], loss.item())
        for k, v in loss_dict.items():
            assert np.isclose((a["terms"][k] + b["terms"][k]) / 2, v.item(), rtol=2e-3, atol=1e-9), (mode, k)
        assert a["loss"] != b["loss"]                              # the halves differ; only their mean is the loss
        d = (a["tables"] - tables).abs()
        # Adam at step 1 moves every touched entry by +-lr: an entry agrees unless its (tiny) gradient changed sign or
        # vanished in the other summation order / in fp16
        assert (d[moved] <= 1e-5).float().mean().item() >= tol_frac, (mode, (d[moved] <= 1e-5).float().mean().item())
        # small parameter groups: the averaged gradients ARE the union-batch gradients (what Adam does with them at
        # step 1 is +-lr per element, i.e. a sign test of values that can be pure summation noise -- compare upstream)
        # (the data-parallel gradient average gives parameters without a gradient -- the time codes while the hash window
        # pins them to ones -- an explicit zero; the single process leaves them at None)
        assert set(a["grads"]) >= set(grads)
        for name in set(a["grads"]) - set(grads):
            assert a["grads"][name].abs().max().item() == 0.0, (mode, name)
        for name, g_ref in grads.items():
            # (absolute floor: at initialisation the deformation gradients sit at the fp16 underflow threshold of the
            # chain's dZ -- the ranks' per-sample upstream gradients are world x larger, so a value that flushes to zero in
            # the single process can survive, ~1e-13, on the ranks)
            sc = g_ref.abs().max().item()
            err = (a["grads"][name] - g_ref).abs().max().item()
            assert err <= 2e-3 * sc + 1e-9, (mode, name, err, sc)


# ---- level-parallel exchange (engine/level_parallel.py): samples travel, parameters do not --------------------------------
OPEN_WINDOW = (-10, 1)          # the coarse-to-fine window at step 0: 1 + 15 * 10 / 11 = 14.6 of 16 grids (> H / 2)


def _level_worker(rank, world, port, out_dir, workload="p030_h16", per=256):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.workloads import build_workload
    bundle, batch = _union_data(per * world, workload).next_train(0)
    torch.manual_seed(19980801)
    trainer, data, _ = build_workload(workload, device="cuda:0", small=True, n_rays=per, rank=rank, world_size=world,
                                      global_loss_normalisers=True, window_hash=OPEN_WINDOW, table_parallel="auto")
    _no_jitter(trainer)
    loss, loss_dict, _ = trainer.train_iteration(0, *_slice_batch(bundle, batch, rank * per, (rank + 1) * per))
    trainer.flush_scheduler_step()
    opt = trainer.optimizers[trainer.group_of_tables()]
    assert isinstance(opt, LevelParallelTableAdam)             # the window is beyond H / 2: the exchange switched at step 0
    comm = opt.comm_report(reset=False)
    model = trainer.model
    assert model._native is not None                            # the step ran through the native drivers, split at the exchange
    grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters() if "tables" not in n and p.grad is not None}
    lp = opt.lp
    own = list(lp.ranges)                                        # one entry range per owned level (balanced assignment)
    stale_before = model.field.hash_ensemble.half_tables().detach().cpu().clone()
    local_f16 = lp.slice_f16().detach().cpu().clone()            # this rank's CURRENT values: compact, level after level
    trainer.consolidate()
    res = {"loss": loss.item(), "terms": {k: v.item() for k, v in loss_dict.items()}, "grads": grads,
           "tables": model.field.hash_ensemble.tables.detach().cpu(), "f16": model.field.hash_ensemble.half_tables().detach().cpu(),
           "small": torch.cat([p.detach().reshape(-1).cpu() for n, p in model.named_parameters() if "tables" not in n]),
           "comm": comm, "own": own, "stale_before": stale_before, "local_f16": local_f16}
    # two more steps on rank-local rays (weak scaling): the replicas of everything that IS replicated stay identical
    model.global_loss_normalisers = None
    losses = []
    for step in (1, 2):
        l, _, _ = trainer.train_iteration(step, *data.next_train(step))
        losses.append(l.item())
    trainer.flush_scheduler_step()
    trainer.consolidate()
    res["losses_after"] = losses
    res["tables_after"] = model.field.hash_ensemble.tables.detach().cpu()
    res["small_after"] = torch.cat([p.detach().reshape(-1).cpu() for n, p in model.named_parameters() if "tables" not in n])
    state = trainer.state_dict()                                # (collective: the moments of all level ranges)
    res["table_step"] = int(opt._step)
    res["n_opt_groups"] = len(state["optimizers"])
    torch.save(res, os.path.join(out_dir, f"l{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_level_parallel_ranks_equal_one_process_on_the_union_batch(cuda, tmp_path):
    """The level-parallel exchange with the real kernels, world_size 2 (two processes share the GPU, gloo carries the
    collectives), coarse-to-fine window open: rank r evaluates and differentiates levels [8 r, 8 r + 8) of the hash grids
    for BOTH ranks' samples, no table gradient and no table values travel.  One optimizer step on a 512-ray batch sliced
    256 + 256 (global loss normalisers) equals the single-process step on the whole batch at the bars of the reduce-scatter
    exchange: loss and every term 2e-4 / 2e-3, the small groups' averaged gradients 2e-3 of their maximum, the hash tables
    on >= 99.5 % of the touched entries; and what arrives at a rank is bounded by the samples, not by the table."""
    import torch.multiprocessing as mp
    from nersemble_amd.workloads import build_workload
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_level_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "l0.pt"), torch.load(tmp_path / "l1.pt")
    bundle, batch = _union_data(512).next_train(0)
    torch.manual_seed(19980801)
    single, _, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW)
    init_tables = single.model.field.hash_ensemble.tables.detach().cpu().clone()
    _no_jitter(single)
    loss, loss_dict, _ = single.train_iteration(0, bundle, batch)
    single.flush_scheduler_step()
    single.consolidate()
    tables = single.model.field.hash_ensemble.tables.detach().cpu()
    grads = {n: p.grad.detach().float().cpu() for n, p in single.model.named_parameters()
             if "tables" not in n and p.grad is not None}
    moved = (tables - init_tables).abs() > 1e-4
    assert moved.float().mean().item() > 1e-3 and moved[:, :, 8:].any()            # grids beyond the first are stepped
    # after consolidation both ranks hold the whole table again, identical; before it a rank held only ITS levels current
    assert torch.equal(a["tables"], b["tables"]) and torch.equal(a["small"], b["small"]) and torch.equal(a["f16"], b["f16"])
    assert torch.equal(a["tables"].half(), a["f16"])
    _ranges_partition_the_table([a["own"], b["own"]], tables.shape[0])
    for r in (a, b):
        # before consolidation a rank's current values live in its compact tensors (the full-size tables are stale); after
        # it every rank holds them in place
        assert not torch.equal(r["stale_before"], r["f16"])
        assert torch.equal(torch.cat([r["f16"][lo:hi] for lo, hi in r["own"]]), r["local_f16"])
    assert np.isclose((a["loss"] + b["loss"]) / 2, loss.item(), rtol=2e-4), (a["loss"], b["loss"], loss.item())
    for k, v in loss_dict.items():
        assert np.isclose((a["terms"][k] + b["terms"][k]) / 2, v.item(), rtol=2e-3, atol=1e-9), k
    d = (a["tables"] - tables).abs()
    frac = (d[moved] <= 1e-5).float().mean().item()
    assert frac >= 0.995, frac
    _differences_are_sign_flips(a["tables"], tables, init_tables, lr=5e-3)
    assert "time_embedding.weight" in grads                                        # the window is open: the time codes train
    for name, g_ref in grads.items():
        sc = g_ref.abs().max().item()
        for r in (a, b):
            err = (r["grads"][name] - g_ref).abs().max().item()
            assert err <= 2e-3 * sc + 1e-9, (name, err, sc)
    # the exchange: bytes arriving at a rank per step are bounded by the job's samples (2 x 64 B per sample and pass + the
    # 16-byte positions + 12-byte dL/dx), far below the 2 x (W - 1) / W x table bytes of the reduce-scatter exchange
    for r in (a, b):
        c = r["comm"]
        assert c["exchange"] == "level_parallel" and c["levels_per_rank"] == 8 and c["gradient_planes"] > 0
        n_job = c["samples_bwd_per_step"]
        assert n_job > 0 and c["bytes_per_rank"] <= (2 * 64 + 16 + 16 + 12) * max(c["samples_fwd_per_step"], n_job) + 65536
        # ONE host-side size exchange and FOUR device collectives per step (+ the occupancy update's column all-gather)
        assert c["host_exchanges_per_step"] == 1 and c["collectives_per_step"] <= 5
        # (this test's tables are 4 MB; at the reference geometry the reduce-scatter exchange moves 2 x 403 MB per rank and
        # step at W = 2 whatever the batch kept -- profiles/r05_two_ranks_one_gpu_gloo_level_parallel.json)
    # the run goes on: replicated parameters identical, tables identical after consolidation, the optimizer counted 3 steps
    assert all(np.isfinite(a["losses_after"])) and all(np.isfinite(b["losses_after"]))
    assert torch.equal(a["tables_after"], b["tables_after"]) and torch.equal(a["small_after"], b["small_after"])
    assert not torch.equal(a["tables_after"], a["tables"]) and a["table_step"] == b["table_step"] == 3


def test_four_level_parallel_ranks_with_32_grids_step_through_the_matrix_core_pass(cuda, tmp_path):
    """world_size 4 on one GPU (gloo), 32 hash grids, 4 levels per rank: every rank brings the 24 code rows of its batch, the
    owned range's optimizer pass reads 96 gradient planes -- more than NSX_MAX_SLOTS, so ``nsx_adam_hash_factored`` forms the
    gradient on the matrix cores (csrc/adam.hip) -- in plane order (source rank, code row).  One step on a 512-ray batch sliced
    4 x 128 equals the single-process step on the whole batch at the bars of the world-2 test."""
    import torch.multiprocessing as mp
    from nersemble_amd.workloads import build_workload
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 4
    mp.spawn(_level_worker, args=(world, port, str(tmp_path), "p030_h32", 128), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"l{r}.pt") for r in range(world)]
    bundle, batch = _union_data(512).next_train(0)
    torch.manual_seed(19980801)
    single, _, _ = build_workload("p030_h32", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW)
    init_tables = single.model.field.hash_ensemble.tables.detach().cpu().clone()
    _no_jitter(single)
    loss, loss_dict, _ = single.train_iteration(0, bundle, batch)
    single.flush_scheduler_step()
    single.consolidate()
    tables = single.model.field.hash_ensemble.tables.detach().cpu()
    grads = {n: p.grad.detach().float().cpu() for n, p in single.model.named_parameters()
             if "tables" not in n and p.grad is not None}
    moved = (tables - init_tables).abs() > 1e-4
    assert moved.float().mean().item() > 1e-3 and moved[:, :, 16:].any()
    bounds = [rng for r in rs for rng in r["own"]]
    _ranges_partition_the_table([r["own"] for r in rs], tables.shape[0])
    assert [r["comm"]["levels"] for r in rs] == [[0, 7, 8, 15], [1, 6, 9, 14], [2, 5, 10, 13], [3, 4, 11, 12]]
    for r in rs:
        c = r["comm"]
        assert c["exchange"] == "level_parallel" and c["levels_per_rank"] == 4
        assert 64 < c["gradient_planes"] <= 96, c["gradient_planes"]             # the matrix-core pass is the one that ran
        assert torch.equal(r["tables"], rs[0]["tables"]) and torch.equal(r["small"], rs[0]["small"])
        assert torch.equal(r["tables"].half(), r["f16"])
        assert torch.equal(r["tables_after"], rs[0]["tables_after"]) and torch.equal(r["small_after"], rs[0]["small_after"])
        assert all(np.isfinite(r["losses_after"])) and r["table_step"] == 3
    assert np.isclose(sum(r["loss"] for r in rs) / world, loss.item(), rtol=2e-4)
    for k, v in loss_dict.items():
        assert np.isclose(sum(r["terms"][k] for r in rs) / world, v.item(), rtol=2e-3, atol=1e-9), k
    d = (rs[0]["tables"] - tables).abs()
    frac = (d[moved] <= 1e-5).float().mean().item()
    assert frac >= 0.995, frac
    _differences_are_sign_flips(rs[0]["tables"], tables, init_tables, lr=5e-3)
    # ... level range by level range (a plane order that is wrong for ONE owner would hide in the average)
    for lo, hi in bounds:
        mv = moved[lo:hi]
        if mv.any():
            f = (d[lo:hi][mv] <= 1e-5).float().mean().item()
            assert f >= 0.99, (lo, hi, f)
    for name, g_ref in grads.items():
        sc = g_ref.abs().max().item()
        for r in rs:
            err = (r["grads"][name] - g_ref).abs().max().item()
            assert err <= 2e-3 * sc + 1e-9, (name, err, sc)
    assert not torch.equal(rs[0]["tables_after"], rs[0]["tables"])


def test_eight_level_parallel_ranks_on_the_p124_model_equal_one_process(cuda, tmp_path):
    """The shape of BASELINE.json configs[4] (participant 124, 475 timesteps, 32 hash grids, ray batch sharded over 8 ranks):
    world_size 8 on one GPU (gloo), 2 levels per rank, 8 x 24 = 192 gradient planes -- the limit of the optimizer pass.  One
    step on a 512-ray batch sliced 8 x 64 equals the single-process step on the whole batch at the bars of the world-2 test."""
    import torch.multiprocessing as mp
    from nersemble_amd.workloads import build_workload
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 8
    mp.spawn(_level_worker, args=(world, port, str(tmp_path), "p124_dp", 64), nprocs=world, join=True)
    rs = [torch.load(tmp_path / f"l{r}.pt") for r in range(world)]
    bundle, batch = _union_data(512, "p124_dp").next_train(0)
    torch.manual_seed(19980801)
    single, _, _ = build_workload("p124_dp", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW)
    init_tables = single.model.field.hash_ensemble.tables.detach().cpu().clone()
    _no_jitter(single)
    loss, loss_dict, _ = single.train_iteration(0, bundle, batch)
    single.flush_scheduler_step()
    single.consolidate()
    tables = single.model.field.hash_ensemble.tables.detach().cpu()
    grads = {n: p.grad.detach().float().cpu() for n, p in single.model.named_parameters()
             if "tables" not in n and p.grad is not None}
    moved = (tables - init_tables).abs() > 1e-4
    assert moved.float().mean().item() > 1e-3
    bounds = [rng for r in rs for rng in r["own"]]
    _ranges_partition_the_table([r["own"] for r in rs], tables.shape[0])
    assert [r["comm"]["levels"] for r in rs] == [[k, 15 - k] for k in range(8)]       # the cheapest level with the dearest
    for r in rs:
        c = r["comm"]
        assert c["exchange"] == "level_parallel" and c["levels_per_rank"] == 2 and c["gradient_planes"] == 192
        assert torch.equal(r["tables"], rs[0]["tables"]) and torch.equal(r["small"], rs[0]["small"])
        assert torch.equal(r["tables_after"], rs[0]["tables_after"]) and torch.equal(r["small_after"], rs[0]["small_after"])
        assert all(np.isfinite(r["losses_after"])) and r["table_step"] == 3
    assert np.isclose(sum(r["loss"] for r in rs) / world, loss.item(), rtol=2e-4)
    for k, v in loss_dict.items():
        assert np.isclose(sum(r["terms"][k] for r in rs) / world, v.item(), rtol=2e-3, atol=1e-9), k
    d = (rs[0]["tables"] - tables).abs()
    assert (d[moved] <= 1e-5).float().mean().item() >= 0.995
    _differences_are_sign_flips(rs[0]["tables"], tables, init_tables, lr=5e-3)
    for lo, hi in bounds:
        mv = moved[lo:hi]
        if mv.any():
            assert (d[lo:hi][mv] <= 1e-5).float().mean().item() >= 0.99, (lo, hi)
    for name, g_ref in grads.items():
        sc = g_ref.abs().max().item()
        for r in rs:
            err = (r["grads"][name] - g_ref).abs().max().item()
            assert err <= 2e-3 * sc + 1e-9, (name, err, sc)


def _ranges_partition_the_table(owns, total):
    """The ranks' entry ranges (one per owned level) cover [0, total) exactly once."""
    flat = sorted(rng for own in owns for rng in own)
    assert flat[0][0] == 0 and flat[-1][1] == total and all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1))


def _differences_are_sign_flips(tables, ref, init, lr):
    """What the ">= 99.5 % of the touched entries agree" bars leave out, said entry by entry.  Adam's FIRST step moves an
    entry with gradient g by -lr * g / (|g| + eps), eps = 1e-15: by +-lr when |g| >> eps, by less when the gradient is a
    cancellation residue within a few orders of eps, never by more.  Two runs whose gradients agree up to summation order
    can therefore differ at an entry only if (i) its gradient is such a residue in at least one of them -- a move strictly
    inside (-lr, lr), i.e. |g| < 1e-12 against typical gradients of 1e-7 ... 1e-4 -- or (ii) a residue's sign flipped or it
    vanished (moves on {-lr, 0, +lr}, 2 lr / lr apart).  EVERY differing entry must be one of the two; an entry whose
    gradient is robust in both runs never differs."""
    a, b = tables - init, ref - init
    differs = (tables - ref).abs() > 1e-5
    if not differs.any():
        return
    da, db = a[differs], b[differs]
    assert float(da.abs().max()) <= lr + 1e-5 and float(db.abs().max()) <= lr + 1e-5, "a move larger than a first Adam step"
    on_lattice = lambda v: ((v.abs() - lr).abs() <= 1e-5) | (v.abs() <= 1e-7)
    residue = (da.abs() < lr * (1 - 1e-3)) | (db.abs() < lr * (1 - 1e-3))
    flip = on_lattice(da) & on_lattice(db)
    assert bool((residue | flip).all()), "an entry with a robust gradient in both runs differs"


def test_emulated_level_parallel_rank_through_rccl(cuda, single_rank_group):
    """Rank r of a W-rank level-parallel job whose other ranks are replicas of itself (``LevelParallel(emulate=True)``) on a
    real one-rank RCCL group: ``all_gather_into_tensor`` / ``all_to_all_single`` run on NCCL, the size exchange on the gloo
    side group.  The own levels' columns are bit-identical to the single kernel's columns of those levels, every replica's
    gradient planes equal the direct backward on the sub-geometry, the partial dL/dx / code gradients come back summed
    over the W (identical) owners, and the valid-row count never leaves the device."""
    import ctypes as C
    from nersemble_amd import functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    from nersemble_amd.engine.level_parallel import LevelParallel
    H, W, r, B, T, kept = 8, 3, 1, 3001, 5, 2500
    he = _he(H, cuda)
    he.train()
    lp = LevelParallel(he, W, r, emulate=True)
    assert lp._a2a_native and dist.get_backend(lp.cpu_group) == "gloo" and dist.get_backend() == "nccl"
    g = torch.Generator(device=cuda).manual_seed(3)
    x = torch.rand((B, 3), device=cuda, generator=g)
    code = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    window = torch.rand((H,), device=cuda, generator=g)
    full = F._hash_ensemble_fwd_raw(x, he.half_tables(), H, he.geom, code, slot, window)
    feats = lp.features(x, code, slot, window)
    n2 = 2 * lp.n_own
    assert feats.shape == (B, W * n2)
    assert lp.levels_of == [[0, 5], [1, 4], [2, 3]] and lp.levels == [1, 4]          # balanced: cheapest with dearest
    cols = lambda lv: [c for l in lv for c in (2 * l, 2 * l + 1)]
    for j in range(W):
        # replica j's block sits where rank j's levels sit in the feature row and holds THIS rank's levels' columns
        assert torch.equal(feats[:, cols(lp.levels_of[j])], full[:, cols(lp.levels)])
    assert lp.stats["collectives"] == 2 and lp.stats["host_exchanges"] == 1
    ex = lp.last_exchange
    assert ex.sizes == [B] * W and ex.rows == [T] * W and ex.n_planes == W * T
    assert torch.equal(ex.codes_packed, code.repeat(W, 1))
    block = (torch.randn((B, n2), device=cuda, generator=g) * 3).half().float()
    n_dev = torch.tensor([kept], dtype=torch.int64, device=cuda)
    lp.begin_step()
    dout = torch.empty((B, W * n2), device=cuda)
    for j in range(W):
        dout[:, cols(lp.levels_of[j])] = block                    # every owner receives the same column block
    dx, dcode = lp.backward(x, slot, dout, n_dev=n_dev)
    assert lp.stats["collectives"] == 4
    n_e = lp.n_entries
    G_d = torch.zeros((T, n_e, 2), device=cuda)
    dcode_d = torch.empty((T, H), device=cuda)
    dx_d = torch.zeros((B, 3), device=cuda)
    check(lib().nsx_hash_ensemble_bwd_codesum(ptr(x), B, ptr(lp.slice_f16()), H, C.byref(lp.geom), ptr(code), code.stride(0), T,
                                              ptr(slot), ptr(window), ptr(block), ptr(G_d), ptr(dcode_d),
                                              ptr(F.codesum_scratch(T, H, cuda)), ptr(dx_d), None, ptr(n_dev), stream()),
          "direct backward")
    G = lp.G[:W * T * n_e * 2].view(W, T, n_e, 2)
    tol = 2e-5 * G_d.abs().max().item()
    assert G_d.abs().max().item() > 0
    for j in range(W):
        assert (G[j] - G_d).abs().max().item() <= tol
    assert torch.allclose(dx[:kept], W * dx_d[:kept], rtol=1e-5, atol=1e-6 * dx_d.abs().max().item())
    assert torch.allclose(dcode, W * dcode_d, rtol=1e-5, atol=1e-6 * dcode_d.abs().max().item())
    # rows beyond the device-side count were neither packed nor unpacked
    dx2 = torch.full((B, 3), 7.0, device=cuda)
    lp.begin_step()
    lp.backward(x, slot, dout, n_dev=n_dev, dx_out=dx2)
    assert bool((dx2[kept:] == 7.0).all()) and torch.allclose(dx2[:kept], dx[:kept])


def test_bucket_tail_in_one_launch_equals_the_copies(cuda, single_rank_group):
    """``all_reduce_gradients`` with the step's gradient arena on the device: the bucket's tail (gradients without a slot, presence
    counts, flags) goes in and comes back by ``nsx_bucket_pack`` / ``nsx_bucket_unpack`` -- one launch each -- instead of a copy
    per piece (the route the gloo CPU tests hold).  Same gradients, counts and flags; the averaging factor of a W-rank job is
    applied once to the buffer's fixed part and once to every returned piece; also through the library's own communicator."""
    from nersemble_amd.engine import parallel
    from nersemble_amd.engine.level_parallel import LevelParallel

    def run(native, world, comm=None):
        flat = torch.zeros((64 + 256,), device=cuda)
        a1, a2 = torch.nn.Parameter(torch.zeros((4, 5), device=cuda)), torch.nn.Parameter(torch.zeros((9,), device=cuda))
        out1, out2 = torch.nn.Parameter(torch.zeros((6,), device=cuda)), torch.nn.Parameter(torch.zeros((3, 7), device=cuda))
        absent = torch.nn.Parameter(torch.zeros((5,), device=cuda))
        slots = {id(a1): flat[3:23].view(4, 5), id(a2): flat[30:39]}

        class _Arena:
            fixed, slack = 40, 256

            def __init__(self):
                self.flat = flat

            def slot_of(self, p):
                return slots.get(id(p))

        flat[23:30] = 99.0
        flat[3:23] = torch.arange(20., device=cuda)
        flat[30:39] = 2.0
        a1.grad, a2.grad = slots[id(a1)], slots[id(a2)]
        out1.grad = torch.full((6,), 3.0, device=cuda)
        out2.grad = torch.arange(21., device=cuda).view(3, 7)
        old = parallel._NATIVE_PIECES
        parallel._NATIVE_PIECES = old if native else 0
        try:
            cnt, fl = parallel.all_reduce_gradients([a1, a2, out1, out2, absent], world, force=True, arena=_Arena(),
                                                    extra_flags=torch.tensor([1.0, 0.0], device=cuda), native_comm=comm)
        finally:
            parallel._NATIVE_PIECES = old
        torch.cuda.synchronize()
        assert a1.grad.data_ptr() == flat[3:23].data_ptr() and absent.grad is None
        return [a1.grad.clone(), a2.grad.clone(), out1.grad.clone(), out2.grad.clone(), cnt.clone(), fl.clone(), flat[23:30].clone()]

    lp = LevelParallel(_he(8, cuda), 2, 0, emulate=True)          # (its communicator: the bucket's sum through the library)
    try:
        for world in (1, 4):
            want = run(False, world)
            for got in (run(True, world), run(True, world, lp.comm)):
                for a, b in zip(got, want):
                    assert torch.equal(a, b)
            assert torch.equal(want[2], torch.full((6,), 3.0 / world, device=cuda))
            assert want[4].tolist() == [[1.0, 1.0, 1.0, 1.0, 0.0], [0.0] * 5] and want[5].tolist() == [1.0, 0.0]
            assert bool((want[6] == 99.0 / world).all())             # (a region of the fixed part that is nobody's is averaged with it)
    finally:
        lp.close()


def test_collectives_issued_by_the_library_equal_the_torch_distributed_route(cuda, single_rank_group):
    """``LevelParallel`` on an RCCL group issues the exchange's collectives from C (csrc/comm.hip: ``nsx_lp_forward`` /
    ``nsx_lp_backward`` on the library's own communicator, one call per direction); ``native_collectives=False`` keeps the five
    torch.distributed calls.  Same feature rows bit for bit, same planes / dL/dx / code gradients up to the order of the
    atomics, same collective count; the communicator reports the group's size."""
    import ctypes as C
    from nersemble_amd._lib import check, lib, ptr, stream
    from nersemble_amd.engine.level_parallel import LevelParallel
    H, B, T, kept = 32, 2777, 6, 2100
    he2 = _he(H, cuda)
    he2.train()
    g = torch.Generator(device=cuda).manual_seed(5)
    x = torch.rand((B, 3), device=cuda, generator=g)
    code = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    window = torch.rand((H,), device=cuda, generator=g)
    n_dev = torch.tensor([kept], dtype=torch.int64, device=cuda)
    # (six levels in the test model: 2 and 3 ranks divide them)
    for W, r in ((3, 1), (2, 0)):
        res = []
        for native in (True, False):
            lp = LevelParallel(he2, W, r, emulate=True, native_collectives=native)
            assert (lp.comm is not None) == native
            if native:
                assert lib().nsx_comm_world_size(lp.comm) == 1 and lib().nsx_comm_rank(lp.comm) == 0
                # the sum over ONE rank is the identity; a W-rank layout on a one-rank communicator is refused unless the
                # call says it emulates (block counts would not match the peers)
                v = torch.arange(7, dtype=torch.float32, device=cuda)
                check(lib().nsx_comm_all_reduce_sum(lp.comm, ptr(v), 7, stream()), "nsx_comm_all_reduce_sum")
                assert torch.equal(v, torch.arange(7, dtype=torch.float32, device=cuda))
                lay = lp._layout(B, T)
                rc = lib().nsx_lp_forward(C.byref(lay), lp.comm, -1, *([None] * 2), 0, *([None] * 2), 0, 1, *([None] * 5),
                                          C.byref(lp.geom), *([None] * 6))
                assert rc != 0 and b"the communicator has 1" in lib().nsx_last_error()
            feats = lp.features(x, code, slot, window, n_dev=n_dev)
            assert lp.stats["collectives"] == 2 and lp.stats["host_exchanges"] == 1
            n2 = 2 * lp.n_own
            dout = (torch.randn((B, W * n2), device=cuda, generator=torch.Generator(device=cuda).manual_seed(9)) * 3).half().float()
            lp.begin_step()
            dx, dcode = lp.backward(x, slot, dout, n_dev=n_dev)
            assert lp.stats["collectives"] == 4
            torch.cuda.synchronize()
            res.append((feats[:kept].clone(), dx[:kept].clone(), dcode.clone(),
                        lp.G[:W * T * lp.n_entries * 2].clone(), dict(lp.stats)))
            lp.close()
            assert lp.comm is None
        (f1, dx1, dc1, G1, st1), (f0, dx0, dc0, G0, st0) = res
        assert torch.equal(f1, f0)
        assert float(G0.abs().max()) > 0 and float((G1 - G0).abs().max()) <= 2e-5 * float(G0.abs().max())
        assert torch.allclose(dx1, dx0, rtol=1e-5, atol=1e-6 * float(dx0.abs().max()))
        assert torch.allclose(dc1, dc0, rtol=1e-4, atol=2e-6 * float(dc0.abs().max()))
        assert st1["bytes_in"] == st0["bytes_in"] and st1["samples_bwd"] == st0["samples_bwd"]


@pytest.mark.parametrize("H,table_grad", [(32, True), (8, True), (32, False)])
def test_all_source_ranks_in_one_launch_equal_one_launch_per_source_rank(H, table_grad, cuda):
    """``nsx_lp_fwd_run`` / ``nsx_lp_bwd_run`` with NSX_OPT_LP_ONE_LAUNCH (grid.y = source rank) against the per-source-rank
    launches of the unchanged kernels, on RAGGED sources: different sample counts, device-side counts below the capacity,
    different numbers of code rows, one source without samples.  Forward columns bit for bit; gradient planes, dL/dx and
    code-row gradients up to the order of the additions."""
    import ctypes as C
    from nersemble_amd import _lib, functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    from nersemble_amd.engine.level_parallel import NativeLPOps, sub_geometry_levels
    he = _he(H, cuda)
    levels = [1, 4]
    geom = sub_geometry_levels(he.geom, levels)
    n_e = int(geom.offset[2])
    n2 = 4
    cut = torch.cat([he.half_tables()[int(he.geom.offset[l]):int(he.geom.offset[l + 1])] for l in levels]).contiguous()
    assert cut.shape[0] == n_e
    W, R_cap = 5, 7
    sizes = [2900, 0, 3001, 17, 1500]              # capacities (the marcher's counts)
    counts = [2500, 0, 3001, 9, 1]                 # valid rows on the device
    rows = [5, 1, 7, 2, 3]
    S_cap = max(sizes)
    lay = NativeLPOps.layout(W, S_cap, R_cap, H, n2)
    g = torch.Generator(device=cuda).manual_seed(11)
    window = torch.rand((H,), device=cuda, generator=g)
    gathered = torch.zeros((W * lay.fwd_bytes,), dtype=torch.uint8, device=cuda)
    recv = torch.zeros((W * lay.bwd_bytes,), dtype=torch.uint8, device=cuda)
    tmp = torch.zeros((W * lay.bwd_bytes,), dtype=torch.uint8, device=cuda)
    src = []
    for j in range(W):
        S = sizes[j]
        x = torch.rand((max(S, 1), 3), device=cuda, generator=g)
        x[: S // 2] = x[:1] + 1e-3 * torch.rand((S // 2, 3), device=cuda, generator=g)     # neighbours share cells
        x.clamp_(0, 0.999)
        slot = torch.randint(0, rows[j], (max(S, 1),), device=cuda, generator=g, dtype=torch.int32)
        code = torch.randn((rows[j], H), device=cuda, generator=g)
        dout = (torch.randn((max(S, 1), W * n2), device=cuda, generator=g) * 3).half().float()
        n_dev = torch.tensor([counts[j]], dtype=torch.int64, device=cuda)
        NativeLPOps.fwd_pack(lay, x, slot, S, n_dev, code, rows[j], gathered[j * lay.fwd_bytes:])
        NativeLPOps.bwd_pack(lay, dout, x, slot, S, n_dev, tmp)              # block 0 of `tmp`: what owner 0 receives
        recv[j * lay.bwd_bytes:(j + 1) * lay.bwd_bytes] = tmp[:lay.bwd_bytes]
        src.append((x, slot, code, dout, n_dev))

    class Ex:
        sizes_host = (C.c_int64 * W)(*sizes)
        rows_host = (C.c_int32 * W)(*rows)
    planes = sum(rows)

    def run(one_launch):
        check(lib().nsx_set_option(_lib.NSX_OPT_LP_ONE_LAUNCH, one_launch), "option")
        try:
            send = torch.full((W * lay.feat_bytes,), 0x5A, dtype=torch.uint8, device=cuda)
            codes_packed = torch.zeros((planes, H), device=cuda)
            NativeLPOps.fwd_run(lay, gathered, Ex, cut, geom, window, send, codes_packed)
            G = torch.zeros((planes, n_e, 2), device=cuda) if table_grad else None
            ret = torch.full((W * lay.ret_bytes,), 0x5A, dtype=torch.uint8, device=cuda)
            nonfinite = torch.zeros((1,), device=cuda)
            dz = torch.zeros((W * S_cap * n2,), device=cuda)
            NativeLPOps.bwd_run(lay, recv, gathered, Ex, cut, geom, window, G, ret, nonfinite, dz)
            torch.cuda.synchronize()
            return send, codes_packed, G, ret
        finally:
            check(lib().nsx_set_option(_lib.NSX_OPT_LP_ONE_LAUNCH, 1), "option")

    send1, cp1, G1, ret1 = run(1)
    send0, cp0, G0, ret0 = run(0)
    assert torch.equal(cp1, cp0)
    at = 0
    for j in range(W):
        n = counts[j]
        f1 = send1[j * lay.feat_bytes:(j + 1) * lay.feat_bytes].view(torch.float16)[: n * n2]
        f0 = send0[j * lay.feat_bytes:(j + 1) * lay.feat_bytes].view(torch.float16)[: n * n2]
        assert torch.equal(f1, f0), f"forward columns of source {j}"
        if n:
            x, slot, code, dout, n_dev = src[j]
            want = F._hash_ensemble_fwd_raw(x[:n], cut, H, geom, code, slot[:n], window)
            assert torch.equal(f1.view(n, n2), want)
        r1 = ret1[j * lay.ret_bytes:(j + 1) * lay.ret_bytes]
        r0 = ret0[j * lay.ret_bytes:(j + 1) * lay.ret_bytes]
        dx1 = r1[lay.r_dx:lay.r_dx + n * 12].view(torch.float32)
        dx0 = r0[lay.r_dx:lay.r_dx + n * 12].view(torch.float32)
        dc1 = r1[lay.r_dcode:lay.r_dcode + rows[j] * H * 4].view(torch.float32)
        dc0 = r0[lay.r_dcode:lay.r_dcode + rows[j] * H * 4].view(torch.float32)
        if n:
            assert torch.allclose(dx1, dx0, rtol=1e-5, atol=1e-6 * float(dx0.abs().max()))
            assert torch.allclose(dc1, dc0, rtol=1e-4, atol=2e-6 * float(dc0.abs().max()) + 1e-30)
            assert float(dc0.abs().max()) > 0
        else:
            assert float(dc1.abs().max()) == 0 and float(dc0.abs().max()) == 0
        if table_grad:
            a, b = G1[at:at + rows[j]], G0[at:at + rows[j]]
            assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-30)
            assert (n == 0) == (float(b.abs().max()) == 0)
        at += rows[j]


def test_emulated_rank_7_of_8_trains_through_the_native_step(cuda, single_rank_group):
    """``NeRSembleTrainer(level_parallel_emulation=(8, 7))``: the training step of the finest levels' owner of an 8-rank job on
    one GPU -- native step drivers split at the exchange, four device collectives + one host-side size exchange per step on
    the one-rank RCCL / gloo groups, 8 x 24 gradient planes through the matrix-core optimizer pass."""
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(0)
    trainer, data, _ = build_workload("p030_h32", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW,
                                      level_parallel_emulation=(8, 7))
    opt = trainer.optimizers[trainer.group_of_tables()]
    assert isinstance(opt, LevelParallelTableAdam) and opt.lp.emulate and opt.lp.n_own == 2 and opt.lp.levels == [7, 8]
    trainer.train_iteration(0, *data.next_train(0))              # (step 0 refreshes the occupancy grid: one more exchange)
    assert opt.lp.stats["collectives"] == 5
    opt.comm_report()
    losses = []
    for step in range(1, 4):                                    # (no occupancy update in these steps: step % 16 != 0)
        loss, _, _ = trainer.train_iteration(step, *data.next_train(step))
        losses.append(loss.item())
        assert int(trainer.model.occupancy_grid.last_n_marched) > 0
    trainer.flush_scheduler_step()
    assert all(np.isfinite(losses))
    assert trainer.model._native is not None
    st = opt.lp.stats
    assert st["bwd_calls"] == 3 and st["fwd_calls"] == 3 and st["collectives"] == 12 and st["host_exchanges"] == 3
    assert opt.lp.planes == 8 * 24 and opt._step == 4
    c = opt.comm_report()
    assert c["collectives_per_step"] == 4 and c["host_exchanges_per_step"] == 1 and c["gradient_planes"] == 192


def test_level_parallel_step_keeps_gradscaler_skip_semantics(cuda, single_rank_group):
    """The level-parallel twin of ``test_early_table_step_keeps_gradscaler_skip_semantics``: an emulated rank of 8 through the native
    step and the library's collectives; the owners' non-finite flags ride in the small gradients' bucket.  When the scaled
    gradients overflow the step is skipped as a whole -- the owned levels' master / working tables and moments, the fused MLPs,
    the deformation field and every step count untouched, the scale halves -- and the next (finite) step goes through."""
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(4)
    trainer, data, _ = build_workload("p030_h32", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW,
                                      level_parallel_emulation=(8, 7))
    model = trainer.model
    opt = trainer.optimizers[trainer.group_of_tables()]
    assert isinstance(opt, LevelParallelTableAdam) and opt.lp.comm is not None
    small = trainer.optimizers["fields"]
    trainer.train_iteration(0, *data.next_train(0))                     # a normal step first (moments exist)
    trainer.flush_scheduler_step()
    torch.cuda.synchronize()
    model.field.hash_ensemble.wait_tables()
    snap = lambda: {"master": opt.lp.slice_master().detach().clone(), "f16": opt.lp.slice_f16().clone(),
                    "m": opt.exp_avg.clone(), "v": opt.exp_avg_sq.clone(),
                    "base": model.field.mlp_base.params.detach().clone(),
                    "deform": model.deformation_field.se3_field.mlp_stem.layers[0].weight.detach().clone(),
                    "emb": model.time_embedding.weight.detach().clone()}
    before = snap()
    assert opt._step == 1 and small.step_count == 1
    trainer.grad_scaler._scale.fill_(2.0 ** 60)                          # every fp16 gradient overflows
    trainer.train_iteration(1, *data.next_train(1))
    trainer.flush_scheduler_step()
    torch.cuda.synchronize()
    model.field.hash_ensemble.wait_tables()
    after = snap()
    for k in before:
        assert torch.equal(after[k], before[k]), k
    assert opt._step == 1 and small.step_count == 1
    assert trainer.grad_scaler.get_scale() == 2.0 ** 59
    trainer.grad_scaler._scale.fill_(65536.0)
    trainer.train_iteration(2, *data.next_train(2))
    trainer.flush_scheduler_step()
    torch.cuda.synchronize()
    model.field.hash_ensemble.wait_tables()
    moved = snap()
    assert opt._step == 2 and small.step_count == 2
    for k in ("master", "f16", "m", "base", "deform"):
        assert not torch.equal(moved[k], before[k]), k


def test_a_trained_model_goes_on_as_a_frozen_emulated_rank(cuda, single_rank_group):
    """``NeRSembleTrainer.become_emulated_level_parallel_rank`` (what ``bench.py --level-parallel-one-rank`` prices in steady
    state): a single-GPU run continues as rank 3 of 8 with frozen parameters, the replicas' feature columns replaced by the
    true ones (one full-geometry forward per pass).  The model then sees exactly what it saw before: the loss of a step on a
    given batch is the loss the un-emulated model has on it, bit for bit, no parameter moves -- while the level-parallel
    kernels, the 192 gradient planes and the four collectives of the emulated rank run."""
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.workloads import build_workload

    torch.manual_seed(0)
    trainer, data, _ = build_workload("p030_h32", device="cuda:0", small=True, n_rays=512, window_hash=OPEN_WINDOW,
                                      compact_first_grid=False)
    _no_jitter(trainer)
    for step in range(3):
        trainer.train_iteration(step, *data.next_train(step))
    trainer.flush_scheduler_step()
    for o in trainer.optimizers.values():               # frozen from here on (two separate trainings would differ by the
        for g in o.param_groups:                        # order of their gradient atomics: ONE model is stepped twice)
            g["lr"] = 0.0
    batches = [data.next_train(step) for step in range(3, 6)]

    def steps():
        out = []
        for step, (bundle, batch) in zip(range(3, 6), batches):
            loss, _, metrics = trainer.train_iteration(step, bundle, batch)
            out.append((loss.item(), int(metrics["num_samples_per_batch"])))
        trainer.flush_scheduler_step()
        return out

    def params():
        trainer.consolidate()
        return torch.cat([p.detach().reshape(-1).float().cpu() for p in trainer.model.parameters()])

    before = params()
    l_s = steps()                                        # the single-GPU step on the three batches
    assert torch.equal(params(), before)
    trainer.become_emulated_level_parallel_rank(8, 3)
    opt = trainer.optimizers[trainer.group_of_tables()]
    assert isinstance(opt, LevelParallelTableAdam) and opt.lp.emulate and opt.lp.shadow_forward and opt.lp.levels == [3, 12]
    l_e = steps()                                        # the same batches through the emulated rank
    assert opt.lp.stats["bwd_calls"] == 3 and opt.lp.planes == 8 * 24 and opt.lp.stats["collectives"] == 12
    assert l_e == l_s, (l_e, l_s)                        # the same losses and kept-sample counts, bit for bit
    assert torch.equal(params(), before)


def _empty_rank_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256, rank=rank, world_size=world,
                                      window_hash=OPEN_WINDOW, table_parallel="level")
    losses, marched = [], []
    for step in range(0, 3):                                    # (step 0 refreshes the occupancy grid on both ranks)
        bundle, batch = data.next_train(step)
        if rank == 1 and step == 1:
            bundle.directions = -bundle.directions              # every ray leaves the scene: nothing is marched on this rank
        loss, _, metrics = trainer.train_iteration(step, bundle, batch)
        losses.append(loss.item())
        marched.append(int(trainer.model.occupancy_grid.last_n_marched))
    trainer.flush_scheduler_step()
    opt = trainer.optimizers[trainer.group_of_tables()]
    stats = dict(opt.lp.stats)
    trainer.consolidate()
    model = trainer.model
    torch.save({"losses": losses, "marched": marched, "stats": stats,
                "tables": model.field.hash_ensemble.tables.detach().cpu(),
                "small": torch.cat([p.detach().reshape(-1).cpu() for n, p in model.named_parameters() if "tables" not in n])},
               os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_a_rank_that_marches_nothing_issues_the_same_collectives(cuda, tmp_path):
    """Two level-parallel ranks on one GPU (gloo); in the second step every ray of rank 1 points away from the scene, so
    its marcher counts zero samples, the native drivers decline and the per-kernel path runs the reference's one-fake-sample
    step (nersemble_volumetric_sampler.py:109-115).  The collective sequence does not depend on it: one size exchange, an
    all-gather and three all-to-alls per step on BOTH ranks, no hang, replicas identical afterwards."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_empty_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "e0.pt"), torch.load(tmp_path / "e1.pt")
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    assert b["marched"][1] <= 0 < a["marched"][1]
    for r in (a, b):
        # 3 steps x (one size exchange, all-gather + three all-to-alls); (the occupancy update of step 0 runs in front of
        # the hand-over to the level-parallel exchange)
        assert r["stats"]["host_exchanges"] == 3 and r["stats"]["collectives"] == 12
    assert torch.equal(a["tables"], b["tables"]) and torch.equal(a["small"], b["small"])


def _switch_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nersemble_amd.engine.level_parallel import LevelParallelTableAdam
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(19980801)
    # window: 1 at step 0, 6 at step 1, 11 at step 2 (> H / 2 = 8: the hand-over), 16 from step 3
    trainer, data, _ = build_workload("p030_h16", device="cuda:0", small=True, n_rays=256, rank=rank, world_size=world,
                                      window_hash=(0, 3))
    kinds, losses = [], []
    for step in range(5):
        loss, _, _ = trainer.train_iteration(step, *data.next_train(step))
        kinds.append(type(trainer.optimizers[trainer.group_of_tables()]).__name__)
        losses.append(loss.item())
    trainer.flush_scheduler_step()
    state = trainer.state_dict()
    key = [k for k in state["optimizers"] if k == "fields"][0]
    st = state["optimizers"][key]["state"]
    first = min(st)
    torch.save({"kinds": kinds, "losses": losses, "tables": trainer.model.field.hash_ensemble.tables.detach().cpu(),
                "exp_avg": st[first]["exp_avg"].cpu(), "step": float(st[first]["step"])}, os.path.join(out_dir, f"w{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_is_handed_from_reduce_scatter_to_level_parallel_when_the_window_opens(cuda, tmp_path):
    """``table_parallel="auto"`` (the trainer's default): the narrow fp16 reduce-scatter exchange while the window is below
    H / 2, the level-parallel exchange from the step at which it passes H / 2 -- master, both moments and the step count are
    handed over (a checkpoint written afterwards holds 5 steps of moments for every entry)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_switch_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "w0.pt"), torch.load(tmp_path / "w1.pt")
    assert a["kinds"] == b["kinds"] == ["ShardedTableAdam"] * 2 + ["LevelParallelTableAdam"] * 3
    assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
    assert torch.equal(a["tables"], b["tables"]) and torch.equal(a["exp_avg"], b["exp_avg"])
    assert a["step"] == b["step"] == 5.0 and a["exp_avg"].abs().max().item() > 0


# ---- the exchange that follows the coarse-to-fine window (grids [0, W) only) -------------------------------------------
@pytest.mark.parametrize("H,W", [(3, 2), (8, 1), (8, 2), (16, 8), (32, 1), (32, 4), (32, 8), (32, 16)])
def test_narrow_exchange_kernels(H, W, cuda):
    """nsx_hash_grad_expand_f16_bucket_width / nsx_adam_dense_f16grad_width / nsx_tables_unpack_width: the packed piece is
    the full-width piece's first W grids, the packed Adam is the dense Adam where the other grids have neither gradient
    nor moments (bit for bit), unpack scatters back; a code that is non-zero beyond W raises the flag."""
    from nersemble_amd.engine.sharded_adam import NativeTableOps
    ops = NativeTableOps()
    he = _he(H, cuda)
    Hp = int(he.tables.shape[-1])
    g = torch.Generator(device=cuda).manual_seed(7 * H + W)
    T = 5
    G = torch.randn((T, he.geom.total_entries, 2), device=cuda, generator=g) * 1e-2
    G[torch.rand(G.shape, device=cuda, generator=g) < 0.5] = 0
    code = torch.randn((T, H), device=cuda, generator=g)
    win = torch.rand((H,), device=cuda, generator=g) + 0.1
    win[W:] = 0
    entry = {"G": G, "code": code, "window": win, "n_rows": T}
    world, nb, per_entry = 2, 2, 2 * Hp
    unit = 1024 * nb
    shard = ((he.tables.numel() + world - 1) // world + unit - 1) // unit * unit
    bucket = shard // nb
    be = bucket // per_entry
    beyond = torch.zeros(1, device=cuda)
    for k in range(nb):
        full = torch.zeros(world * bucket, dtype=torch.float16, device=cuda)
        ops.expand_f16_bucket(he, entry, full, 0.5, False, shard, bucket, k, world)
        packed = torch.full((world * be * 2 * W,), 7.0, dtype=torch.float16, device=cuda)
        ops.expand_f16_bucket_width(he, entry, packed, 0.5, False, shard, bucket, k, world, W, beyond)
        want = full.view(world * be, 2, Hp)
        assert torch.equal(packed.view(world * be, 2, W), want[..., :W])
        assert want[..., W:].abs().max().item() == 0 or W == Hp
        ops.expand_f16_bucket(he, entry, full, 0.5, True, shard, bucket, k, world)
        ops.expand_f16_bucket_width(he, entry, packed, 0.5, True, shard, bucket, k, world, W, beyond)
        assert torch.equal(packed.view(world * be, 2, W), full.view(world * be, 2, Hp)[..., :W])
    if Hp >= 8 and W <= 16 and W < Hp:
        # round 6: the variant that clears the planes it reads -- the same bits, and once every piece has been expanded G is
        # all zeros (no fill in front of the next backward's scatter)
        consumed = dict(entry, G=G.clone())
        for k in range(nb):
            plain = torch.full((world * be * 2 * W,), 7.0, dtype=torch.float16, device=cuda)
            eaten = torch.full((world * be * 2 * W,), 7.0, dtype=torch.float16, device=cuda)
            ops.expand_f16_bucket_width(he, entry, plain, 0.5, False, shard, bucket, k, world, W, beyond)
            ops.expand_f16_bucket_width(he, consumed, eaten, 0.5, False, shard, bucket, k, world, W, beyond, consume=True)
            assert torch.equal(plain, eaten)
        assert float(consumed["G"].abs().max()) == 0.0 and float(G.abs().max()) > 0
    assert beyond.item() == 0
    if W < H:
        win2 = win.clone()
        win2[W] = 0.25
        ops.expand_f16_bucket_width(he, dict(entry, window=win2), packed, 0.5, False, shard, bucket, 0, world, W, beyond)
        assert beyond.item() == 1
    # Adam on the packed gradient == the dense kernel on the same gradient with zeros (and zero moments) beyond W
    ne = 3000
    n = ne * per_entry
    gp = (torch.randn((ne, 2, W), device=cuda, generator=g) * 30).half()
    gd = torch.zeros((ne, 2, Hp), dtype=torch.float16, device=cuda)
    gd[..., :W] = gp
    state = {}
    for name in ("narrow", "dense"):
        gg = torch.Generator(device=cuda).manual_seed(99)
        master = torch.randn((n,), device=cuda, generator=gg)
        m = torch.randn((ne, 2, Hp), device=cuda, generator=gg) * 1e-2
        v = torch.rand((ne, 2, Hp), device=cuda, generator=gg) * 1e-3
        m[..., W:] = 0
        v[..., W:] = 0
        state[name] = [master, m.reshape(-1).contiguous(), v.reshape(-1).contiguous(), master.half()]
    inv, found = torch.tensor([1.0 / 64], device=cuda), torch.zeros(1, device=cuda)
    out = torch.zeros(ne * 2 * W, dtype=torch.float16, device=cuda)
    for step in (1, 2):
        a, d = state["narrow"], state["dense"]
        ops.adam_f16grad_width(gp.reshape(-1), ne, W, Hp, a[0], a[1], a[2], a[3], out, 5e-3, 0.9, 0.999, 1e-15, step, inv,
                               found)
        ops.adam_f16grad(gd.reshape(-1), n, d[0], d[1], d[2], d[3], 5e-3, 0.9, 0.999, 1e-15, step, inv, found)
        for x, y in zip(a, d):
            assert torch.equal(x, y)
        assert torch.equal(out.view(ne, 2, W), a[3].view(ne, 2, Hp)[..., :W])
    # a skipped step moves nothing and still fills the packed buffer with the current values
    found.fill_(1)
    before = [t.clone() for t in state["narrow"]]
    out.zero_()
    a = state["narrow"]
    ops.adam_f16grad_width(gp.reshape(-1), ne, W, Hp, a[0], a[1], a[2], a[3], out, 5e-3, 0.9, 0.999, 1e-15, 3, inv, found)
    assert all(torch.equal(x, y) for x, y in zip(a, before))
    assert torch.equal(out.view(ne, 2, W), a[3].view(ne, 2, Hp)[..., :W])
    # unpack: the first W grids of every (entry, f) row are replaced, the others stay
    tab = torch.randn((ne, 2, Hp), device=cuda, generator=g).half()
    keep = tab.clone()
    ops.unpack_width(gp.reshape(-1), ne, W, Hp, tab)
    assert torch.equal(tab[..., :W], gp) and torch.equal(tab[..., W:], keep[..., W:])


def test_sharded_step_follows_the_window(cuda, single_rank_group):
    """ShardedTableAdam with a window schedule (1, 1, 1.5, 2.5, 6): the exchange widens 1, 1, 2, 4, full and the tables are
    those of the optimizer that always exchanges every grid, bit for bit.  Round 6: a third run trains COMPACT copies of the
    first 1 / 2 / 4 grids with the H = 1 / 2 / 4 kernels (``compact_first_grid``: the gathered packed buffer of the narrow
    exchange is the copy's working table, handed over at every doubling) -- the same tables again."""
    from nersemble_amd.engine.sharded_adam import ShardedTableAdam
    B, T, H = 4000, 7, 8
    g = torch.Generator(device=cuda).manual_seed(2)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    scale = 1024.0
    now = {"w": None}
    a, b, c = _he(H, cuda), _he(H, cuda), _he(H, cuda)
    opt_a = ShardedTableAdam(a, lr=5e-3, eps=1e-15, world_size=1, rank=0)
    opt_b = ShardedTableAdam(b, lr=5e-3, eps=1e-15, world_size=1, rank=0, width_source=lambda: now["w"])
    opt_c = ShardedTableAdam(c, lr=5e-3, eps=1e-15, world_size=1, rank=0, width_source=lambda: now["w"])
    c.compact_first_grid = True
    inv = torch.tensor([1.0 / scale], device=cuda)
    widths, layouts = [], []
    for w in (1.0, 1.0, 1.5, 2.5, 6.0):
        now["w"] = w
        for he, opt in ((a, opt_a), (b, opt_b), (c, opt_c)):
            found = torch.zeros(1, device=cuda)
            opt.zero_grad()
            he(x, emb, window_hash_encodings=w, code_index=slot).backward(dout * scale)
            if he is c:
                layouts.append(c._compact["width"] if c._compact is not None else 0)
            opt.check_finite(found)
            opt.step(found_inf=found, inv_scale=inv)
            assert found.item() == 0
        widths.append(opt_b._last_width)
        assert opt_c._last_width == opt_b._last_width
        # (the scatter's fp32 atomics run in another order in the two runs: equal up to that, as two runs of ONE optimizer are)
        d = (a.half_tables().float() - b.half_tables().float()).abs()
        assert (d == 0).float().mean().item() >= 0.999, (w, d.max().item())
        c.sync_first_grid()                                   # (the compact copy written back; it stays the working state)
        d = (a.half_tables().float() - c.half_tables().float()).abs()
        assert (d == 0).float().mean().item() >= 0.999, ("compact", w, d.max().item())
    assert widths == [1, 1, 2, 4, 8] and layouts == [1, 1, 2, 4, 0]
    ba, bb = opt_a._buffers(), opt_b._buffers()
    opt_c.gather_master()
    dm = (a.tables.detach() - c.tables.detach()).abs()
    assert (dm <= 1e-6).float().mean().item() >= 0.999
    bc = opt_c._buffers()
    diff_c = (ba["exp_avg"] - bc["exp_avg"]).abs()
    assert diff_c.max().item() <= 2e-3 * ba["exp_avg"].abs().max().item() and (bc["exp_avg"].view(-1, 8)[:, 6:] == 0).all()
    assert (bb["exp_avg"].view(-1, 8)[:, 6:] == 0).all()              # grids the window never reached
    # (the two runs' scatters add their fp32 atomics in different orders, so single elements of the fp16 gradient the Adam
    # kernels read differ by one fp16 ulp: the moments agree to that -- relative to their scale, not element by element, where
    # five steps' contributions may cancel.  Bit-identity of the two exchanges on identical gradients is the CPU test's,
    # tests/test_parallel_cpu.py::test_exchange_that_follows_the_window_equals_the_full_exchange_bit_for_bit.)
    diff = (ba["exp_avg"] - bb["exp_avg"]).abs()
    assert diff.max().item() <= 2e-3 * ba["exp_avg"].abs().max().item(), diff.max().item()
    assert (diff <= 1e-3 * ba["exp_avg"].abs() + 1e-7).float().mean().item() >= 0.999
