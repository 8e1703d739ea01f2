"""GPU, REFERENCE GEOMETRY (16 levels x 2^19 entries, H = 32 -> 403 M table parameters): the backward and the optimizer
at BASELINE.json's full size, the 2^20-sample chunk boundary of ``max_n_samples_per_batch``, and full-size training
steps of the configurations that otherwise only run inside bench.py (configs[2] p030_h32, configs[3] p097_dense with its
two chunks, configs[4]'s model p124_dp)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from oracle import hashgrid as ohg
from tests.helpers import REF_GEOM_KW

pytestmark = pytest.mark.gpu
FP16 = 2.0 ** -10


def test_full_size_backward_h32_vs_oracle(cuda):
    """H = 32 factored backward (gather + scatter halves) on 2^20 samples at the reference geometry: dx / dcode / table
    gradient against the C oracle on a 3000-sample sub-batch run on its own (the gradient is a sum over samples, so the
    sub-batch is a complete problem), and a conservation property over the full batch: every level's table gradient
    sums to the sum of its upstream feature gradients (trilinear weights sum to one)."""
    from nersemble_amd import _lib, functional as F
    from nersemble_amd._lib import check, lib, ptr, stream
    H, B, T = 32, 1 << 20, 24
    gn = _lib.grid_geometry(**REF_GEOM_KW)
    go = oracle.grid_geometry(**REF_GEOM_KW)
    gen = torch.Generator(device=cuda).manual_seed(11)
    total = gn.total_entries
    f16 = (torch.rand((total, 2, H), device=cuda, generator=gen) - 0.5).half()
    x = torch.rand((B, 3), device=cuda, generator=gen)
    table = (torch.randn((T, H), device=cuda, generator=gen) * 0.5)
    slot = torch.randint(0, T, (B,), device=cuda, generator=gen, dtype=torch.int32)
    dout = torch.randn((B, 32), device=cuda, generator=gen)

    def backward(n):
        G = torch.zeros((T, total, 2), device=cuda)
        dcode = torch.empty((n, H), device=cuda)
        dx = torch.empty((n, 3), device=cuda)
        check(lib().nsx_hash_ensemble_bwd_scatter(ptr(x), n, C.byref(gn), T, ptr(slot), ptr(dout), ptr(G), None, 8,
                                                  None,
                                                  stream()), "scatter")
        check(lib().nsx_hash_ensemble_bwd_factored(ptr(x), n, ptr(f16), H, C.byref(gn), ptr(table), table.stride(0), T,
                                                   ptr(slot), None, ptr(dout), None, ptr(dcode), ptr(dx), None, None, stream()),
              "gather")
        return G, dcode, dx

    # (1) conservation over the full batch, per level and feature
    G, dcode, dx = backward(B)
    for l in (0, 4, 5, 15):
        lo, hi = int(gn.offset[l]), int(gn.offset[l + 1])
        got = G[:, lo:hi, :].double().sum(dim=(0, 1)).cpu().numpy()
        want = dout[:, 2 * l:2 * l + 2].double().sum(dim=0).cpu().numpy()
        assert np.allclose(got, want, rtol=1e-4, atol=1e-2), (l, got, want)
    assert torch.isfinite(dx).all() and torch.isfinite(dcode).all()
    # (2) a 3000-sample sub-batch against the oracle
    n = 3000
    Gs, dcs, dxs = backward(n)
    dtab = torch.empty((total, 2, H), device=cuda)
    check(lib().nsx_hash_grad_expand(ptr(Gs), T, ptr(table), table.stride(0), None, H, C.byref(gn), ptr(dtab), 0, stream()),
          "expand")
    tc = F.tables_to_tcnn(f16.float(), H, gn).cpu().numpy().astype(np.float16)
    codes = table[slot[:n].long()].cpu().numpy()
    dt_o, dc_o, dx_o = ohg.ensemble_bwd(x[:n].cpu().numpy(), tc.view(np.uint16), H, go, codes, dout[:n].cpu().numpy())
    assert np.abs(dxs.cpu().numpy() - dx_o).max() <= 2e-5 * np.abs(dx_o).max()
    assert np.abs(dcs.cpu().numpy() - dc_o).max() <= 2e-5 * np.abs(dc_o).max()
    got_t = F.tables_to_tcnn(dtab, H, gn).cpu().numpy()
    # the expansion multiplies G by the fp16-rounded code, as the forward does
    assert np.abs(got_t - dt_o).max() <= 2e-3 * np.abs(dt_o).max()


def test_full_size_table_adam_equals_torch_adam(cuda):
    """403 M parameters, two steps: native factored Adam against torch.optim.Adam on the expanded dense gradient."""
    from nersemble_amd.engine.hash_adam import HashTableAdam
    from nersemble_amd.field_components.hash_ensemble import HashEnsemble, HashEnsembleConfig, TCNNHashEncodingConfig
    H, B, T = 32, 200_000, 24

    def make():
        he = HashEnsemble(HashEnsembleConfig(H, TCNNHashEncodingConfig(), True, True), seed=3).to(cuda)
        with torch.no_grad():
            he.tables.mul_(3000)
        return he

    gen = torch.Generator(device=cuda).manual_seed(1)
    x = torch.rand((B, 3), device=cuda, generator=gen)
    emb = torch.randn((T, H), device=cuda, generator=gen)
    slot = torch.randint(0, T, (B,), device=cuda, generator=gen, dtype=torch.int32)
    dout = torch.randn((B, 32), device=cuda, generator=gen).half()
    ref, nat = make(), make()
    assert ref.tables.numel() == 6299960 * 2 * 32
    opt_ref = torch.optim.Adam([ref.tables], lr=5e-3, eps=1e-15)
    opt_nat = HashTableAdam(nat, lr=5e-3, eps=1e-15, factored=True)
    scale = 256.0
    inv = torch.tensor([1.0 / scale], device=cuda)
    found = torch.zeros(1, device=cuda)
    for it in range(2):
        opt_ref.zero_grad()
        ref(x, emb, window_hash_encodings=None, code_index=slot).backward(dout * scale)
        ref.tables.grad.mul_(1.0 / scale)
        opt_ref.step()
        ref._f16_version = None
        opt_nat.zero_grad()
        nat(x, emb, window_hash_encodings=None, code_index=slot).backward(dout * scale)
        opt_nat.check_finite(found)
        opt_nat.step(found_inf=found, inv_scale=inv)
        d = (nat.tables - ref.tables).abs()
        # both gradients are sums of fp32 atomics in an order that changes from run to run; where a summed gradient is
        # zero up to that rounding, m / sqrt(v) is a coin flip of size lr in EACH implementation.  Such entries are a
        # handful in 403 M (this assertion was `max <= 1e-4` and tripped once in ~10 runs with one entry at 1.6e-4): bound
        # their number and their size (|update| <= lr per step), and keep the mean, which any real error would move
        outliers = int((d > 1e-4).sum().item())
        assert d.mean().item() <= 1e-7 and outliers <= 40 and d.max().item() <= 2.0 * 5e-3 * (it + 1) + 1e-6, \
            (it, d.max().item(), d.mean().item(), outliers)
    assert torch.equal(nat.half_tables(), nat.tables.detach().half())


def test_chunk_boundary_of_max_n_samples_per_batch(cuda):
    """More than 2^20 samples in one batch (the dense-march configuration): the fields walk them in chunks of
    ``max_n_samples_per_batch`` (reference: util/chunker.py via nersemble_nerfacto_field.py:259-265, deformation_field.py:
    152-155).  Chunked and un-chunked evaluation agree bit for bit, also in the samples either side of the boundary."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(0)
    trainer, data, _ = build_workload("p097_dense", device="cuda:0", small=True, n_rays=4096)
    model = trainer.model.eval()
    bundle, _ = data.next_train(0)
    outs = {}
    for chunk in (2 ** 20, -1, 300_000):
        model.field.max_n_samples_per_batch = chunk
        model.deformation_field.max_n_samples_per_batch = chunk
        with torch.no_grad():
            o = model.get_outputs(bundle)
        outs[chunk] = (o["rgb"].clone(), o["depth"].clone(), o["num_samples_per_ray"].clone())
    n = int(outs[-1][2].sum())
    assert n > 2 ** 20, n                                           # the boundary is crossed
    for chunk in (2 ** 20, 300_000):
        assert torch.equal(outs[chunk][2], outs[-1][2])
        assert torch.equal(outs[chunk][0], outs[-1][0]) and torch.equal(outs[chunk][1], outs[-1][1]), chunk
    # and one full training step over the two chunks
    model.train()
    model.field.max_n_samples_per_batch = model.deformation_field.max_n_samples_per_batch = 2 ** 20
    loss, _, metrics = trainer.train_iteration(0, *data.next_train(1))
    assert torch.isfinite(loss) and int(metrics["num_samples_per_batch"]) > 2 ** 20


@pytest.mark.parametrize("name", ["p030_h32", "p124_dp", "p097_dense"])
def test_full_size_training_steps(name, cuda, golden_dir):
    """Reference-size tables, 4096 rays: a few complete training iterations (march, sigma pass, fields, compositing, all
    losses, backward, GradScaler, Adam on every group) -- finite, learning, sample counts in the expected range."""
    from nersemble_amd.workloads import WORKLOADS, build_workload
    torch.manual_seed(19980801)
    trainer, data, info = build_workload(name, device="cuda:0", small=False)
    assert info["params"] > 403_000_000
    losses, samples = [], []
    for step in range(8):
        loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step))
        losses.append(loss.item())
        samples.append(int(metrics["num_samples_per_batch"]))
    trainer.flush_scheduler_step()
    assert all(np.isfinite(losses)) and min(losses[4:]) < losses[0]
    if WORKLOADS[name]["disable_occ"]:
        assert min(samples) > 2 ** 20                               # dense march: two chunks per step
    else:
        assert 2e5 < max(samples) <= 4096 * 700
    he = trainer.model.field.hash_ensemble
    assert trainer.grad_scaler.get_scale() == 65536.0              # no overflow skipped a step
    if name == "p030_h32":
        _check_against_reference_manifest(trainer, golden_dir)
    trainer.consolidate()                                           # (the compact first-grid phase holds grid 0 apart)
    assert torch.equal(he.half_tables(), he.tables.detach().half())


def _check_against_reference_manifest(trainer, golden_dir):
    """SURVEY 8 f3 at the reference's size (H = 32, 403 M parameters): the trained model's ``state_dict()``, the trainer's
    parameter groups and the checkpoint it writes equal tests/golden/state_manifest.json -- the reference's own module tree
    -- key for key, shape for shape, group member for group member; the optimizer state of the natively stepped tables
    arrives as the 8 tcnn encodings' flat moments with torch's per-parameter step."""
    import json
    from nersemble_amd.util.checkpoint import nerfstudio_checkpoint_from_model
    with open(f"{golden_dir}/state_manifest.json") as f:
        entry = json.load(f)["configs"]["H32"]
    model = trainer.model
    ckpt = nerfstudio_checkpoint_from_model(model, 8, trainer=trainer)
    got = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in ckpt["pipeline"].items()}
    want = {"_model." + e["key"]: (e["shape"], e["dtype"]) for e in entry["state_dict"]}
    assert got == want, sorted(set(got) ^ set(want))
    assert {g: len(v) for g, v in trainer.group_layout.items()} == {g: len(v) for g, v in entry["param_groups"].items()}
    fields = ckpt["optimizers"]["fields"]
    names = entry["param_groups"]["fields"]
    assert fields["param_groups"][0]["params"] == list(range(len(names))) and len(names) == 12
    for i, n in enumerate(names):
        if n.endswith(("direction_encoding.params", "position_encoding.params")):
            assert i not in fields["state"]
        else:
            st = fields["state"][i]
            assert int(st["step"]) == 8 and st["exp_avg"].dtype == torch.float32
            assert list(st["exp_avg"].shape) == dict((e["key"], e["shape"]) for e in entry["state_dict"])[n]
    assert 0 not in ckpt["optimizers"]["deformation_field"]["state"]            # the frozen aabb
    assert len(ckpt["optimizers"]["deformation_field"]["state"]) == 16
