"""GPU parity: native hash-table Adam (factored + dense) vs torch.optim.Adam on the materialised gradient, and the
GradScaler skip semantics."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _he(H, cuda, seed=0):
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    cfg = HashEnsembleConfig(H, TCNNHashEncodingConfig(n_levels=6, log2_hashmap_size=11), True, True)
    he = HashEnsemble(cfg, seed=seed).to(cuda)
    with torch.no_grad():
        he.tables.mul_(3000)          # non-trivial values
    return he


@pytest.mark.parametrize("H", [1, 4, 32])
@pytest.mark.parametrize("first_window", ["half", "one"])
def test_factored_adam_equals_torch_adam(H, first_window, cuda):
    """``first_window`` "one": the schedule's start (one grid on, then 1.3, 1.6 -> a second grid opens on the way)."""
    from nersemble_amd.engine.hash_adam import HashTableAdam
    B, T = 4000, 7
    g = torch.Generator(device=cuda).manual_seed(1)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    scale = 1024.0
    # reference: dense gradient + torch Adam (+ manual unscale)
    ref = _he(H, cuda)
    opt_ref = torch.optim.Adam([ref.tables], lr=5e-3, eps=1e-15)
    nat = _he(H, cuda)
    opt_nat = HashTableAdam(nat, lr=5e-3, eps=1e-15, factored=True)
    inv = torch.tensor([1.0 / scale], device=cuda)
    found = torch.zeros(1, device=cuda)
    for it in range(3):
        win = (0.5 * H if first_window == "half" else 1.0) + it * 0.3
        opt_ref.zero_grad()
        ref(x, emb, window_hash_encodings=win, code_index=slot).backward(dout * scale)
        ref.tables.grad.mul_(1.0 / scale)
        opt_ref.step()
        opt_nat.zero_grad()
        nat(x, emb, window_hash_encodings=win, code_index=slot).backward(dout * scale)
        assert nat.tables.grad is None and len(nat.grad_sink.entries) == 1
        opt_nat.check_finite(found)
        opt_nat.step(found_inf=found, inv_scale=inv)
        assert found.item() == 0
        d = (nat.tables - ref.tables).abs()
        # Adam with eps = 1e-15 is scale-free: on entries whose gradient is pure cancellation noise the update is
        # lr * (noise ratio), so the atomics' summation order shows up at ~1e-5 (lr = 5e-3); everything else ~1e-7.
        # (Counted, not bounded by the maximum: where the noise flips the SIGN of a sum the two differ by up to 2 lr per
        # step -- seen once, on one of 403 M entries, in tests/test_full_size_gpu.py.)
        assert int((d > 5e-5).sum().item()) <= 3 and d.max().item() <= 2 * 5e-3 * (it + 1) + 1e-6, (it, d.max().item())
        assert d.mean().item() <= 1e-7
        # the fp16 working copy is refreshed by the kernel
        assert torch.equal(nat.half_tables(), nat.tables.detach().half())
        ref._f16_version = None


def test_adam_skips_on_inf_and_dense_path(cuda):
    from nersemble_amd.engine.hash_adam import HashTableAdam, NativeGradScaler
    H = 4
    he = _he(H, cuda)
    opt = HashTableAdam(he, lr=5e-3, eps=1e-15, factored=False)          # dense (data-parallel) mode
    ref = _he(H, cuda)
    opt_ref = torch.optim.Adam([ref.tables], lr=5e-3, eps=1e-15)
    g = torch.Generator(device=cuda).manual_seed(2)
    grad = torch.randn(he.tables.shape, device=cuda, generator=g)
    he.tables.grad = grad.clone()
    ref.tables.grad = grad.clone()
    found = torch.zeros(1, device=cuda)
    opt.check_finite(found)
    opt.step(found_inf=found, inv_scale=None)
    opt_ref.step()
    assert (he.tables - ref.tables).abs().max().item() <= 1e-6
    before = he.tables.detach().clone()
    he.tables.grad = grad.clone()
    he.tables.grad[5, 1, 2] = float("inf")
    found = torch.zeros(1, device=cuda)
    opt.check_finite(found)
    assert found.item() == 1
    opt.step(found_inf=found, inv_scale=None)
    assert torch.equal(he.tables.detach(), before)                       # skipped
    sc = NativeGradScaler(cuda)
    s0 = sc.get_scale()
    sc.update([found])
    assert sc.get_scale() == s0 * 0.5
    sc.update([torch.zeros(1, device=cuda)])
    assert sc.get_scale() == s0 * 0.5


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_factored_backward_flags_nonfinite_gradient(bad, cuda):
    """GradScaler's inf check on the table gradient: the backward kernel flags a non-finite value it adds to G, and the
    optimizer skips the step exactly like a check over the whole buffer would."""
    from nersemble_amd.engine.hash_adam import HashTableAdam
    from nersemble_amd._lib import check, lib, ptr, stream
    B, T, H = 3000, 5, 8
    g = torch.Generator(device=cuda).manual_seed(2)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    he = _he(H, cuda)
    opt = HashTableAdam(he, lr=5e-3, eps=1e-15, factored=True)
    for poisoned in (False, True, False):          # the flag must clear again on the next clean step
        d = dout.clone()
        if poisoned:
            d[1234, 7] = bad
        before = he.tables.detach().clone()
        opt.zero_grad()
        he(x, emb, window_hash_encodings=4.0, code_index=slot).backward(d)
        G = he.grad_sink.entries[0]["G"]
        brute = torch.zeros(1, device=cuda)
        check(lib().nsx_check_finite(ptr(G), G.numel(), ptr(brute), stream()), "nsx_check_finite")
        found = torch.zeros(1, device=cuda)
        opt.check_finite(found)
        assert found.item() == brute.item() == (1.0 if poisoned else 0.0)
        opt.step(found_inf=found, inv_scale=None)
        assert torch.equal(he.tables.detach(), before) == poisoned


def test_placement_calibration_keeps_values_and_installs_state(cuda):
    """engine/placement.py: after trying candidate placements the tables, their fp16 copy and the (zero) Adam moments
    hold exactly what they held before; the optimizer steps from the chosen buffers like an uncalibrated twin."""
    from nersemble_amd.engine.hash_adam import HashTableAdam
    from nersemble_amd.engine.placement import calibrate_table_placement
    H, B, T = 8, 4000, 5
    he, twin = _he(H, cuda), _he(H, cuda)
    opt, opt_twin = HashTableAdam(he, lr=5e-3, eps=1e-15), HashTableAdam(twin, lr=5e-3, eps=1e-15)
    before, before16 = he.tables.detach().clone(), he.half_tables().clone()
    assert calibrate_table_placement(he, opt) is None                      # small tables: nothing to gain, untouched
    rep = calibrate_table_placement(he, opt, candidates=4, iters=2, min_params=0)
    assert rep is not None and len(rep["candidate_ms"]) == 4 and rep["chosen_ms"] == min(rep["candidate_ms"])
    assert torch.equal(he.tables.detach(), before) and torch.equal(he.half_tables(), before16)
    st = opt.state[he.tables]
    assert st["step"] == 0 and not st["exp_avg"].any() and not st["exp_avg_sq"].any()
    assert st["exp_avg"].data_ptr() != st["exp_avg_sq"].data_ptr() != he.tables.data_ptr()
    assert calibrate_table_placement(he, opt, min_params=0) is None        # optimizer already holds state

    g = torch.Generator(device=cuda).manual_seed(1)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    for it in range(3):
        for h, o in ((he, opt), (twin, opt_twin)):
            o.zero_grad()
            h(x, emb, window_hash_encodings=4.0, code_index=slot).backward(dout)
            o.step()
    assert opt.state[he.tables]["step"] == 3 and opt.state[he.tables]["exp_avg"].any()
    # (fp32 atomics accumulate G in arbitrary order: equal up to that, see test_factored_adam_equals_torch_adam)
    # (an entry whose summed gradient is zero up to that order flips the sign of m / sqrt(v): at most a few, each <= lr/step)
    d = (he.tables - twin.tables).abs()
    assert int((d > 5e-5).sum().item()) <= 3 and d.max().item() <= 2 * 5e-3 * 3 + 1e-6, d.max().item()
    assert (he.tables.detach() - before).abs().max().item() > 1e-3          # it did train
    assert torch.equal(he.half_tables(), he.tables.detach().half())


def test_small_group_adam_equals_torch_adam_with_gradscaler_semantics(cuda):
    """engine/small_adam.py (two launches for all small groups) against torch.optim.Adam per group + GradScaler's
    unscale / inf check / per-group skip (nersemble_trainer.py:185-203): identical parameters and moments to fp32
    rounding, a poisoned group is skipped and its step count does not advance, tensors without a gradient are left
    alone."""
    from nersemble_amd.engine.small_adam import SmallGroupAdam, adam_groups, unscale_and_check_groups
    g = torch.Generator(device=cuda).manual_seed(0)
    shapes = [[(64, 32), (16, 64), (64, 64)], [(100, 32), (100, 128)], [(128, 173), (128,), (3, 128), (3,), (7, 5, 3)]]
    lrs, eps = [5e-3, 5e-3, 1e-3], 1e-15

    def make():
        gg = torch.Generator(device=cuda).manual_seed(1)
        return [[torch.nn.Parameter(torch.randn(s, device=cuda, generator=gg)) for s in grp] for grp in shapes]

    ref_params, nat_params = make(), make()
    ref_opts = [torch.optim.Adam(p, lr=lr, eps=eps) for p, lr in zip(ref_params, lrs)]
    nat_opts = [SmallGroupAdam(p, lr=lr, eps=eps) for p, lr in zip(nat_params, lrs)]
    scale = 1024.0
    inv = torch.tensor([1.0 / scale], device=cuda)
    for it in range(5):
        poisoned = 1 if it == 2 else -1
        skip_tensor = (2, 4)                                     # group 2's last tensor gets no gradient at all
        late_tensor = (1, 1)                                     # its first gradient arrives at iteration 2 (in a step
        #                                                          that is then skipped): torch starts its `step` there
        for gi, (rp, np_) in enumerate(zip(ref_params, nat_params)):
            for ti, (a, b) in enumerate(zip(rp, np_)):
                if (gi, ti) == skip_tensor or ((gi, ti) == late_tensor and it < 2):
                    a.grad = b.grad = None
                    continue
                grad = torch.randn(a.shape, device=cuda, generator=g)
                if gi == poisoned and ti == 0:
                    grad.view(-1)[3] = float("inf")
                a.grad = grad.clone()                            # reference: unscaled gradient
                b.grad = grad * scale                            # native: scaled, as backward() of the scaled loss leaves it
        found = torch.zeros(3, device=cuda)
        table = unscale_and_check_groups(nat_opts, [0, 1, 2], 3, found, inv)
        assert found.tolist() == [1.0 if k == poisoned else 0.0 for k in range(3)]
        adam_groups(nat_opts, [0, 1, 2], 3, table, found)
        for k, opt in enumerate(ref_opts):
            if k != poisoned:
                opt.step()
            else:
                nat_opts[k].rollback_step()                      # what the trainer does once it has read the flag
        for gi, (rp, np_) in enumerate(zip(ref_params, nat_params)):
            for ti, (a, b) in enumerate(zip(rp, np_)):
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-7), (it, gi, ti, (a - b).abs().max().item())
                if (gi, ti) != skip_tensor and gi != poisoned and a.grad is not None:
                    assert torch.allclose(b.grad, a.grad, rtol=1e-6)         # unscaled in place
    assert [o.step_count for o in nat_opts] == [5, 4, 5]
    # per-parameter step counts, as torch keeps them: the late tensor has taken 2 steps (iterations 3 and 4), with the
    # bias corrections of steps 1 and 2 -- the parameter comparison above holds it to torch's lazily created state
    late = nat_params[1][1]
    assert nat_opts[1].steps[late] == 2 and int(ref_opts[1].state[ref_params[1][1]]["step"]) == 2
    assert nat_opts[1].steps[nat_params[1][0]] == 4
    assert nat_params[2][4] not in nat_opts[2].steps
    for ro, no in zip(ref_opts, nat_opts):
        for (pa, sa), (pb, sb) in zip(ro.state.items(), no.state.items()):
            assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6)
            assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    # torch-shaped state dict round trip
    sd = nat_opts[0].state_dict()
    assert int(sd["state"][0]["step"]) == 5 and "exp_avg" in sd["state"][0]
    fresh = SmallGroupAdam(make()[0], lr=5e-3, eps=eps)
    fresh.load_state_dict(sd)
    assert fresh.step_count == 5
    # per-parameter steps survive the round trip; a parameter that never had a gradient has no state (as in torch)
    sd1, sd2 = nat_opts[1].state_dict(), nat_opts[2].state_dict()
    assert [int(sd1["state"][i]["step"]) for i in (0, 1)] == [4, 2]
    assert 4 not in sd2["state"] and set(sd2["state"]) == {0, 1, 2, 3}
    again = SmallGroupAdam(make()[1], lr=5e-3, eps=eps)
    again.load_state_dict(sd1)
    assert [again.steps[p] for p in again._params()] == [4, 2]


def test_a_parameter_no_rank_had_a_gradient_for_is_left_alone(cuda):
    """Data-parallel rule of ``nsx_multi_adam_present`` (advisor, round 5): every rank joins the gradient all-reduce with zeros
    for a parameter it has no gradient for; when NO rank had one, the averaged `gradient` is those zeros and a single process
    (torch.optim.Adam: ``grad is None``) leaves the parameter, its moments and its step count alone.  The counts are on the
    device: the kernel skips the tensor, the host takes the step count back when the counts arrive."""
    from nersemble_amd.engine.small_adam import SmallGroupAdam, adam_groups, unscale_and_check_groups
    g = torch.Generator(device=cuda).manual_seed(2)
    shapes = [(64, 32), (16,), (33, 7)]

    def make():
        gg = torch.Generator(device=cuda).manual_seed(3)
        return [torch.nn.Parameter(torch.randn(s, device=cuda, generator=gg)) for s in shapes]

    ref, nat = make(), make()
    ref_opt, nat_opt = torch.optim.Adam(ref, lr=1e-2, eps=1e-15), SmallGroupAdam(nat, lr=1e-2, eps=1e-15)
    index = {id(p): i for i, p in enumerate(nat)}
    for it in range(4):
        idle = 1 if it in (1, 2) else -1                       # tensor 1: a gradient in step 0, none anywhere in steps 1, 2
        counts = torch.tensor([0.0 if i == idle else 2.0 for i in range(3)], device=cuda)
        for i, (a, b) in enumerate(zip(ref, nat)):
            grad = torch.randn(a.shape, device=cuda, generator=g)
            a.grad = None if i == idle else grad.clone()
            b.grad = torch.zeros_like(b) if i == idle else grad.clone()      # (the zeros the ranks joined the collective with)
        found = torch.zeros(1, device=cuda)
        table = unscale_and_check_groups([nat_opt], [0], 1, found, None)
        adam_groups([nat_opt], [0], 1, table, found, counts, index)
        ref_opt.step()
        if idle >= 0:
            nat_opt.rollback_params([nat[idle]])                # (the trainer, once the counts have reached the host)
        for a, b in zip(ref, nat):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
    assert [nat_opt.steps[p] for p in nat] == [4, 2, 4]
    assert [int(ref_opt.state[p]["step"]) for p in ref] == [4, 2, 4]
    for a, b in zip(ref, nat):
        assert torch.allclose(ref_opt.state[a]["exp_avg"], nat_opt.state[b]["exp_avg"], rtol=1e-5, atol=1e-7)
        assert torch.allclose(ref_opt.state[a]["exp_avg_sq"], nat_opt.state[b]["exp_avg_sq"], rtol=1e-5, atol=1e-7)
    # without the presence vector the same call steps the tensor with g = 0 (what the rule exists to prevent)
    before = nat[1].detach().clone()
    nat[1].grad = torch.zeros_like(nat[1])
    found = torch.zeros(1, device=cuda)
    adam_groups([nat_opt], [0], 1, unscale_and_check_groups([nat_opt], [0], 1, found, None), found)
    assert not torch.equal(nat[1].detach(), before)


def test_optimizer_pass_consumes_the_factored_gradient(cuda):
    """nsx_adam_hash_factored_consume: same update as nsx_adam_hash_factored, G all zeros afterwards -- also when the
    step is skipped -- and the next backward adds to that buffer without a fill of its own."""
    from nersemble_amd.engine.hash_adam import HashTableAdam
    H, B, T = 32, 3000, 5
    g = torch.Generator(device=cuda).manual_seed(4)
    x = torch.rand((B, 3), device=cuda, generator=g)
    emb = torch.randn((T, H), device=cuda, generator=g)
    slot = torch.randint(0, T, (B,), device=cuda, generator=g, dtype=torch.int32)
    dout = torch.randn((B, 12), device=cuda, generator=g).half()
    inv = torch.tensor([1.0], device=cuda)
    a, b = _he(H, cuda), _he(H, cuda)
    opt_a = HashTableAdam(a, lr=5e-3, eps=1e-15, factored=True)
    opt_b = HashTableAdam(b, lr=5e-3, eps=1e-15, factored=True)
    opt_a.consume_density_limit = float("inf")          # (the toy table is written densely: take the pass anyway)
    opt_b.consume_gradient = False
    for it in range(3):
        for he, opt in ((a, opt_a), (b, opt_b)):
            opt.zero_grad()
            he(x, emb, window_hash_encodings=None, code_index=slot).backward(dout * (1.0 + it))
            found = torch.zeros(1, device=cuda)
            opt.check_finite(found)
            G = he.grad_sink.entries[0]["G"]
            assert G.abs().sum().item() > 0
            if he is a and it > 0:
                assert G.data_ptr() == last_G.data_ptr()                  # the persistent buffer, handed out again
            opt.step(found_inf=found, inv_scale=inv)
            if he is a:
                last_G = G
                assert not G.any().item()                                  # consumed
                assert he.grad_sink.pre_cleared is not None and he.grad_sink.pre_cleared[0] is G
            else:
                assert G.any().item()
        # (two runs of the atomic scatter: each is within 5e-5 of the exact sum on the entries whose gradient is pure
        # cancellation noise -- see test_factored_adam_equals_torch_adam -- so the pair is compared at twice that)
        d = (a.tables - b.tables).abs()
        outliers = int((d > 1e-4).sum().item())             # (sign flips of m / sqrt(v) on pure-noise entries)
        assert outliers <= 3 and d.max().item() <= 2 * 5e-3 * (it + 1) + 1e-6 and d.mean().item() <= 2e-7, \
            (it, d.max().item(), d.mean().item(), outliers)
    # a skipped step leaves the parameters alone and still hands back a clean buffer
    before = a.tables.detach().clone()
    opt_a.zero_grad()
    a(x, emb, window_hash_encodings=None, code_index=slot).backward(dout)
    G = a.grad_sink.entries[0]["G"]
    opt_a.step(found_inf=torch.ones(1, device=cuda), inv_scale=inv)
    assert torch.equal(a.tables.detach(), before)
    assert not G.any().item()


def test_native_scale_update_is_torchs_amp_update(cuda):
    """nsx_grad_scaler_update against torch._amp_update_scale_ over a sequence of clean and poisoned steps (growth every 3
    clean steps, back-off on a flag of ANY group): scale and growth tracker bit for bit; the flags are copied out, the next
    step's buffer is cleared, 1 / scale and the mirrors of the scale follow."""
    from nersemble_amd.engine.hash_adam import NativeGradScaler
    a = NativeGradScaler(cuda, init_scale=1024.0, growth_interval=3)
    b_scale = torch.full((), 1024.0, device=cuda)
    b_track = torch.zeros((), dtype=torch.int32, device=cuda)
    g = a.loss_grad_vector(8, 5)                              # a cached one-hot loss gradient: must follow the scale
    assert g[5].item() == 1024.0 and g.sum().item() == 1024.0
    slots = [torch.zeros(3, device=cuda), torch.zeros(3, device=cuda)]
    invs = [torch.ones(1, device=cuda), torch.ones(1, device=cuda)]
    copy = torch.full((3,), -1.0, device=cuda)
    pattern = [0, 0, 0, 0, 1, 0, 0, 2, 0, 0, 0, 0, 0, 0]       # group that raises a flag (0: none)
    turn = 0
    for step, bad in enumerate(pattern):
        cur, nxt = slots[turn], slots[1 - turn]
        assert cur.abs().sum().item() == 0                     # cleared by the previous update
        if bad:
            cur[bad] = 1.0
        nxt.fill_(7.0)                                         # (stale content the update must clear)
        want_flags = cur.clone()
        a.update_native(cur, copy, nxt, invs[1 - turn])
        torch._amp_update_scale_(b_scale, b_track, want_flags.sum().reshape(()), 2.0, 0.5, 3)
        assert a._scale.item() == b_scale.item() and a._growth_tracker.item() == b_track.item(), step
        assert torch.equal(copy, want_flags) and torch.equal(cur, want_flags)      # this step's flags stay readable
        assert nxt.abs().sum().item() == 0
        assert invs[1 - turn].item() == 1.0 / b_scale.item()
        assert a.loss_grad_vector(8, 5) is g and g[5].item() == b_scale.item() and g.sum().item() == b_scale.item()
        turn = 1 - turn
    assert a.get_scale() == b_scale.item() != 1024.0
    # a torch-side write to the scale is noticed: the cached gradient is rebuilt
    a.load_state_dict({"scale": 64.0, "_growth_tracker": 1})
    g2 = a.loss_grad_vector(8, 5)
    assert g2 is not g and g2[5].item() == 64.0


@pytest.mark.parametrize("planes,H,entries", [(192, 32, 4099), (150, 30, 1000), (65, 32, 37)])
def test_many_plane_pass_forms_the_gradient_on_the_matrix_cores(planes, H, entries, cuda):
    """More than NSX_MAX_SLOTS gradient planes (level-parallel runs: one per (source rank, code row)): the pass expands
    G x code with bf16 MFMAs on a three-way split of G -- fp32 accuracy.  Held to an fp64 expansion + the Adam formula, to
    the VALU pass on <= 64 planes (the other planes zero), and the consuming variant to its contract."""
    import ctypes as C
    from nersemble_amd._lib import lib, ptr, stream, check, GridGeom
    g = torch.Generator(device=cuda).manual_seed(planes)
    geom = GridGeom()
    geom.n_levels = 1
    geom.offset[0], geom.offset[1] = 0, entries
    G = torch.randn((planes, entries, 2), device=cuda, generator=g)
    G = G * torch.exp2(torch.randint(-20, 22, G.shape, device=cuda, generator=g).float())     # loss-scaled sums: any magnitude
    G = torch.where(torch.rand(G.shape, device=cuda, generator=g) < 0.6, torch.zeros_like(G), G)
    # (fp16 codes and a window in sixteenths: their product is exact in fp32, so its rounding to fp16 is the same single rounding
    # wherever it is formed -- a product that is itself rounded first can land one fp16 ulp away, once in ~ 2^13 values)
    code = torch.randn((planes, 32), device=cuda, generator=g)[:, :H].half().float().contiguous()
    window = torch.randint(1, 17, (H,), device=cuda, generator=g).float() / 16.0
    window[0] = 1.0
    inv = torch.tensor([1.0 / 512], device=cuda)
    zero = torch.zeros(1, device=cuda)

    def state():
        gs = torch.Generator(device=cuda).manual_seed(7)
        p = torch.randn((entries, 2, 32), device=cuda, generator=gs)
        m = torch.randn((entries, 2, 32), device=cuda, generator=gs) * 1e-2
        v = torch.rand((entries, 2, 32), device=cuda, generator=gs) * 1e-3
        return p, m, v, torch.zeros((entries, 2, 32), device=cuda, dtype=torch.float16)

    def run(fn, Gt, n, found=zero):
        p, m, v, h = state()
        check(fn(ptr(Gt), n, ptr(code), code.stride(0), ptr(window), H, C.byref(geom), ptr(p), ptr(m), ptr(v), ptr(h),
                 5e-3, 0.9, 0.999, 1e-15, 3, ptr(inv), ptr(found), stream()), "adam")
        torch.cuda.synchronize()
        return p, m, v, h

    p, m, v, h = run(lib().nsx_adam_hash_factored, G, planes)
    cw = (code * window).half().double()                                      # the kernels' fp16-rounded windowed codes
    grad = torch.einsum("sef,sh->efh", G.double(), cw) * float(inv.item())
    mag = torch.einsum("sef,sh->efh", G.double().abs(), cw.abs()) * float(inv.item())
    p0, m0, v0, _ = state()
    # (the kernels form 1 - beta in fp32 from the fp32 betas of the C-ABI: 1.3e-5 away from the double's 0.001)
    import numpy as np
    omb1, b2 = float(np.float32(1) - np.float32(0.9)), float(np.float32(0.999))
    omb2 = float(np.float32(1) - np.float32(0.999))
    m_ref = m0[..., :H].double() + (grad - m0[..., :H].double()) * omb1
    v_ref = v0[..., :H].double() * b2 + omb2 * grad * grad
    # the first moment is linear in the gradient: its error IS the expansion's error -- fp32 accumulation of <= 192 terms
    err = (m[..., :H].double() - m_ref).abs()
    bound = 0.1 * 4e-6 * mag + 3e-7 * (m0[..., :H].abs().double() + 0.1 * grad.abs()) + 1e-30
    assert bool((err <= bound).all()), (err / bound).max().item()
    err_g = (4e-6 * mag + 3e-7 * grad.abs())
    assert bool(((v[..., :H].double() - v_ref).abs() <= 0.001 * (2 * grad.abs() + err_g) * err_g + 3e-7 * v_ref).all())
    bc1, bc2 = 1 - 0.9 ** 3, (1 - 0.999 ** 3) ** 0.5
    p_ref = p0[..., :H].double() - (5e-3 / bc1) * (m_ref / (v_ref.sqrt() / bc2 + 1e-15))
    upd = (p_ref - p0[..., :H].double()).abs()
    # (the first moment's bound carried through m / denom, + the fp32 rounding of the parameter and of the quotient)
    carried = (5e-3 / bc1) * 2.0 * bound / (v_ref.sqrt() / bc2 + 1e-15)
    assert bool(((p[..., :H].double() - p_ref).abs() <= 2e-6 * (1.0 + upd) + carried).all())
    assert torch.equal(h[..., :H], p[..., :H].half())
    if H < 32:                                                                # padded grids: never touched
        assert torch.equal(p[..., H:], p0[..., H:]) and torch.equal(m[..., H:], m0[..., H:])
    # the VALU pass on the first 64 planes == this pass on all planes with the others zero (up to the sums' rounding)
    G64 = G.clone()
    G64[64:] = 0
    pa, ma, va, _ = run(lib().nsx_adam_hash_factored, G64, planes)
    pb, mb, vb, _ = run(lib().nsx_adam_hash_factored, G64[:64].contiguous(), 64)
    mag64 = torch.einsum("sef,sh->efh", G64.double().abs(), cw.abs()) * float(inv.item())
    assert bool(((ma - mb)[..., :H].abs().double() <= 0.1 * 8e-6 * mag64 + 3e-7 * ma[..., :H].abs().double() + 1e-30).all())
    assert bool(((pa - pb).abs().double()[..., :H] <= 2e-6 + 2e-5 * (pa - p0).abs().double()[..., :H]).all())
    # consuming variant: the same bits as the plain pass, G all zeros afterwards; a skipped step leaves the state alone and clears
    Gc = G.clone()
    pc, mc, vc, hc = run(lib().nsx_adam_hash_factored_consume, Gc, planes)
    assert torch.equal(pc, p) and torch.equal(mc, m) and torch.equal(vc, v) and torch.equal(hc, h)
    assert not Gc.any().item()
    Gc = G.clone()
    ps, ms, vs, _ = run(lib().nsx_adam_hash_factored_consume, Gc, planes, found=torch.ones(1, device=cuda))
    assert torch.equal(ps, p0) and torch.equal(ms, m0) and torch.equal(vs, v0) and not Gc.any().item()
    Gk = G.clone()
    ps, ms, vs, _ = run(lib().nsx_adam_hash_factored, Gk, planes, found=torch.ones(1, device=cuda))
    assert torch.equal(ps, p0) and torch.equal(Gk, G)
