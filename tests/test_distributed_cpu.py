"""CPU, world_size 2, gloo: the data-parallel wrapper bench.py uses averages gradients like one big-batch step, keeps
BatchNorm statistics per rank (the reference's no-SyncBN semantics) and times with max-over-ranks."""
import os
import socket

import torch
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _toy():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
                         nn.Linear(8, 5))


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mrla_amd import distributed as D
    r, l, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    net = D.wrap_data_parallel(_toy())
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(4, 3, 6, 6, generator=g)
    y = torch.randint(0, 5, (4,), generator=g)
    loss = nn.functional.cross_entropy(net(x), y)
    loss.backward()
    grads = [p.grad.clone() for p in net.parameters()]
    t = D.max_over_ranks(1.0 + rank)
    D.barrier()
    out[rank] = dict(grads=grads, t=t, rm=net.module[1].running_mean.clone(), x=x, y=y)
    torch.distributed.destroy_process_group()


def test_ddp_gloo_world2_matches_manual_average():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    assert res[0]["t"] == res[1]["t"] == 2.0                       # max over ranks
    # reference: each rank's local-BN gradient on its own shard, averaged
    want = None
    for r in range(world):
        net = _toy()
        nn.functional.cross_entropy(net(res[r]["x"]), res[r]["y"]).backward()
        gs = [p.grad for p in net.parameters()]
        want = gs if want is None else [a + b for a, b in zip(want, gs)]
        assert torch.allclose(net[1].running_mean, res[r]["rm"], atol=1e-6)   # statistics stayed local to the rank
    want = [g / world for g in want]
    for r in range(world):
        for a, b in zip(res[r]["grads"], want):
            assert torch.allclose(a, b, atol=1e-6)
    assert not torch.allclose(res[0]["rm"], res[1]["rm"])
