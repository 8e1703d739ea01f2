"""GPU: the native step drivers (csrc/step.hip behind engine/native_step.py) against the per-kernel path
(``fused_train_forward`` issuing every native call from Python): the SAME kernels with the same arguments in the same
order, so everything the forward produces is equal bit for bit and the gradients agree up to the order of the fp32
atomics.  Phases of the reference's schedule (train_nersemble.py:77-78): one grid on (compact first-grid copy / full
layout), every grid on (time codes trained)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(phase, native, seed=6, n_rays=512, workload="p030_h16"):
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(seed)
    trainer, data, _ = build_workload(workload, device="cuda:0", small=True, n_rays=n_rays,
                                      window_hash=(0, 1) if phase == "open" else None,
                                      compact_first_grid=(phase == "compact"))
    model = trainer.model
    model.native_step = native
    if phase == "open":                                           # open from step 0 on (the ramp is over before it)
        model.sched_window_hash_encodings.begin_step, model.sched_window_hash_encodings.end_step = -2, -1
    return trainer, data, model


@pytest.mark.parametrize("phase", ["compact", "closed", "open"])
def test_native_forward_equals_the_per_kernel_forward(phase, cuda):
    """One forward of a training step: loss vector, metrics and every entry of the ``get_outputs`` dict."""
    res = {}
    for native in (False, True):
        trainer, data, model = _build(phase, native)
        for cb in trainer.callbacks:
            cb.run(0)
        model.train()
        bundle, batch = data.next_train(0)
        torch.manual_seed(70)
        with torch.autocast(device_type="cuda", dtype=torch.float16, cache_enabled=False):
            out = model.fused_train_forward(bundle, batch)
        assert out is not None
        loss_dict, metrics, outputs = out
        assert (model._native is not None) == native
        n = int(metrics["num_samples_per_batch"])
        o = {k: outputs[k] for k in ("rgb", "accumulation", "depth", "deformation", "num_samples_per_ray")}
        o["weights"] = outputs["weights"][0].reshape(-1)[:n]
        o["ray_indices"] = outputs["ray_indices"][0][:n]
        fr = outputs["ray_samples"][0].frustums
        o.update(starts=fr.starts.reshape(-1)[:n], ends=fr.ends.reshape(-1)[:n], offsets=fr.offsets[:n],
                 origins=fr.origins[:n], directions=fr.directions[:n])
        # (the defined entries of the loss vector: terms, total, metrics, sample sums, denominators)
        res[native] = (loss_dict.fused.detach()[:19].clone(), n, {k: v.detach().clone() for k, v in o.items()},
                       {k: float(v.detach()) for k, v in loss_dict.items()})
    (f_p, n_p, o_p, t_p), (f_n, n_n, o_n, t_n) = res[False], res[True]
    assert n_p == n_n and n_p > 1000
    assert torch.equal(f_p, f_n), (f_p, f_n)
    assert t_p == t_n
    for k in o_p:
        assert torch.equal(o_p[k], o_n[k]), k


@pytest.mark.parametrize("phase", ["compact", "closed", "open"])
def test_native_training_steps_equal_the_per_kernel_steps(phase, cuda):
    res = {}
    for native in (False, True):
        trainer, data, model = _build(phase, native)
        calls = []
        orig = model.fused_train_forward

        def counted(*a, _orig=orig, _calls=calls, **k):
            r = _orig(*a, **k)
            _calls.append(r is not None)
            return r

        model.fused_train_forward = counted
        losses, terms, grads, tables = [], None, None, None
        for step in range(5):
            torch.manual_seed(70 + step)                          # same near-plane jitter on both sides
            loss, loss_dict, metrics = trainer.train_iteration(step, *data.next_train(step))
            losses.append(loss.item())
            if step == 0:
                terms = {k: v.item() for k, v in loss_dict.items()}
                terms.update({"m:" + k: float(v) for k, v in metrics.items()})
                grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
                trainer.consolidate()                             # (the compact phase holds grid 0 apart)
                model.field.hash_ensemble.wait_tables()
                tables = model.field.hash_ensemble.tables.detach().clone()           # after ONE optimizer step
        trainer.flush_scheduler_step()
        assert all(calls) and len(calls) == 5
        assert (model._native is not None) == native
        res[native] = (losses, terms, grads, tables, trainer.grad_scaler.get_scale())
    (l_p, t_p, g_p, tab_p, sc_p), (l_n, t_n, g_n, tab_n, sc_n) = res[False], res[True]
    assert l_p[0] == l_n[0] and t_p == t_n, (l_p[0], l_n[0], t_p, t_n)                 # forward: bit for bit
    assert set(g_p) == set(g_n) and sc_p == sc_n == 65536.0
    assert ("time_embedding.weight" in g_n) == (phase == "open")
    for name in g_p:
        sc = g_p[name].abs().max().item()
        assert (g_p[name] - g_n[name]).abs().max().item() <= 1e-4 * sc + 1e-12, name
    assert np.allclose(l_p, l_n, rtol=2e-3), (l_p, l_n)
    assert l_n[-1] < l_n[0]
    # one Adam step (+-lr per touched entry): equal unless a cancelling gradient changed sign with the atomics' order;
    # later steps diverge chaotically from there, which is why the run is compared through the loss
    d = (tab_p - tab_n).abs()
    assert (d <= 1e-5).float().mean().item() >= 0.9995


def test_native_step_with_the_march_counted_a_step_ahead(cuda):
    """The trainer's ``next_ray_bundle`` hand-over (the traversal's counting pass on a side stream, one step ahead) feeds
    the native sampler as it feeds the per-kernel one: same losses as marching in place."""
    runs = {}
    for ahead in (False, True):
        trainer, data, model = _build("compact", True, seed=11)
        batches = [data.next_train(s) for s in range(7)]
        losses, used = [], []
        for step in range(6):
            torch.manual_seed(500 + step)
            nxt = batches[step + 1][0] if ahead else None
            if ahead and step + 1 < 6:
                pass
            loss, _, metrics = trainer.train_iteration(step, *batches[step], next_ray_bundle=nxt)
            losses.append((loss.item(), int(metrics["num_samples_per_batch"])))
            used.append(model.occupancy_grid.last_march_prefetched)
        trainer.flush_scheduler_step()
        runs[ahead] = (losses, used)
    assert not any(runs[False][1])
    assert sum(runs[True][1]) >= 3                    # (a step that follows a grid update marches on its own)
    # the jitter of a prefetched pass is drawn one step earlier: the runs see different random near planes, so they
    # agree statistically, not bit for bit
    a, b = np.array(runs[False][0]), np.array(runs[True][0])
    assert np.allclose(a[:, 0], b[:, 0], rtol=0.1) and np.allclose(a[:, 1], b[:, 1], rtol=0.1)


@pytest.mark.parametrize("phase", ["compact", "open"])
def test_the_stash_of_the_counting_pass_is_the_second_march(phase, cuda):
    """``OccGridEstimator.march_stash``: the counting pass a step ahead keeps the samples' starts and the native sampler copies
    them (``nsx_step_sample.march_stash`` -> ``nsx_march_fill_from_stash``) instead of marching a second time -- the step sees the
    same samples (``tests/test_march_gpu.py`` holds the copy to the oracle's march bit for bit): the runs agree as two runs of one
    configuration do (from step 1 on they differ by the order of the gradient atomics of the steps before)."""
    from nersemble_amd.nerfacc import OccGridEstimator
    runs = {}
    for stash in (False, True):
        trainer, data, model = _build(phase, True, seed=13)
        grid = model.occupancy_grid
        grid.march_stash = stash
        batches = [data.next_train(s) for s in range(6)]
        rec, used, kept = [], [], []
        for step in range(5):
            torch.manual_seed(900 + step)
            loss, loss_dict, metrics = trainer.train_iteration(step, *batches[step], next_ray_bundle=batches[step + 1][0])
            rec.append((loss.item(), int(metrics["num_samples_per_batch"])))
            used.append(bool(grid.last_march_prefetched))
            kept.append(grid.last_march_stash is not None)
        trainer.flush_scheduler_step()
        runs[stash] = (rec, used, kept)
    assert runs[False][1] == runs[True][1] and sum(runs[True][1]) >= 3
    assert not any(runs[False][2])
    assert runs[True][2] == runs[True][1]                     # every prefetched pass handed its stash on (no ray beyond the cap)
    assert runs[True][0][0] == runs[False][0][0]                  # step 0 marches in place on both sides: bit for bit
    a, b = np.array(runs[False][0]), np.array(runs[True][0])
    assert np.allclose(a[:, 0], b[:, 0], rtol=5e-2) and np.allclose(a[:, 1], b[:, 1], rtol=2e-2)
    assert OccGridEstimator.march_stash is True                    # (the class default: on)


def test_native_step_falls_back_outside_its_configuration(cuda):
    """Dense configuration (occupancy grid off: nothing to reuse from a sigma pass) and per-sample-count read-backs stay on
    the per-kernel path; the step still runs."""
    from nersemble_amd.workloads import build_workload
    torch.manual_seed(3)
    trainer, data, _ = build_workload("p097_dense", device="cuda:0", small=True, n_rays=128)
    model = trainer.model
    assert model.native_step
    loss, _, metrics = trainer.train_iteration(0, *data.next_train(0))
    assert torch.isfinite(loss) and model._native is not None       # asked, declined
    trainer2, data2, model2 = _build("compact", True)
    model2.device_sample_counts = False
    loss2, _, _ = trainer2.train_iteration(0, *data2.next_train(0))
    assert torch.isfinite(loss2)


def test_leaf_gradients_are_the_steps_buffer_and_still_accumulate(cuda):
    """The native backward leaves the gradients of the leaf parameters (fused MLPs, deformation tensors) in its persistent
    buffer and makes the views their ``.grad`` (no clone per tensor).  A second backward without clearing must still ADD,
    as ``AccumulateGrad`` would: the earlier sum is moved out of the buffer before stage 0 clears it."""
    trainer, data, model = _build("compact", True, seed=3)
    for cb in trainer.callbacks:
        cb.run(0)
    model.train()
    bundle, batch = data.next_train(0)

    def one_backward():
        torch.manual_seed(70)
        with torch.autocast(device_type="cuda", dtype=torch.float16, cache_enabled=False):
            loss_dict, _, _ = model.fused_train_forward(bundle, batch)
        loss_dict.total.backward()

    one_backward()
    flat = model._native._grad_buffers[next(iter(model._native._grad_buffers))].flat
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    leaves = {n: p for n, p in model.named_parameters()
              if p.grad is not None and "embedding" not in n            # (the code lookups are autograd's)
              and ("mlp_base" in n or "mlp_head" in n or "deformation_field" in n)}
    assert len(leaves) >= 10
    inside = [n for n, p in leaves.items() if lo <= p.grad.data_ptr() < hi]
    assert len(inside) == len(leaves), set(leaves) - set(inside)
    first = {n: p.grad.detach().clone() for n, p in leaves.items()}
    model.field.hash_ensemble.grad_sink.clear()                     # (the table gradient is not this test's subject)
    one_backward()                                                  # same batch, same jitter: the gradient doubles
    for n, p in leaves.items():
        sc = first[n].abs().max().item()
        assert (p.grad - 2 * first[n]).abs().max().item() <= 2e-4 * sc + 1e-12, n
