"""CPU, world_size 2, gloo: the data-parallel path (gradient averaging + ray sharding) the 8-GPU bench relies on."""
import os
import socket

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
    if rank == 0:
        nograd.grad = torch.ones(4) * 8
    all_reduce_gradients([big] + small + [nograd], world)
    ok = torch.allclose(big.grad, torch.full_like(big, 1.5))
    ok &= torch.allclose(small[0].grad, torch.arange(21.).reshape(7, 3) * 1.5)
    ok &= torch.allclose(small[1].grad, torch.ones(11) * 5)
    ok &= torch.allclose(nograd.grad, torch.ones(4) * 4)
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
