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


def _flat_worker(rank, world, port, out, bucket_mb, overlap, stages=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mrla_amd import distributed as D
    D.init_from_env("gloo")
    torch.manual_seed(7 + rank)                         # ranks start from DIFFERENT weights: the constructor broadcasts rank 0's
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 4, 1),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(4, 5), nn.Linear(5, 5))
    net = net.to(memory_format=torch.channels_last)
    for p in net[7].parameters():                        # a parameter that never receives a gradient
        p.requires_grad_(True)
    ex = D.FlatGradientExchange(net.parameters(), bucket_mb=bucket_mb, overlap=overlap)
    assert (len(ex.buckets) == 1) if (bucket_mb > 1 or not overlap) else (len(ex.buckets) >= 4)
    w0 = [p.detach().clone() for p in net.parameters()]
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(100 + rank)
    steps = []
    for _ in range(2):                                   # two steps: `.grad` is re-pointed at the flat views every step
        x = torch.randn(4, 3, 6, 6, generator=g).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 5, (4,), generator=g)
        opt.zero_grad(set_to_none=True)
        nn.functional.cross_entropy(net[:7](x), y).backward()
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        if stages:                                       # what bench.py replays from two HIP graphs around an eager all-reduce
            ex.gather()
            ex.allreduce_flat()
            ex.adopt()
        else:
            ex.reduce()
        for p, v in zip(ex.params, ex.views):
            assert p.grad is v and p.grad.stride() == p.stride()
        steps.append(dict(local=local, avg=[p.grad.clone() for p in net.parameters()]))
        opt.step()
    out[rank] = dict(w0=w0, steps=steps, w=[p.detach().clone() for p in net.parameters()])
    torch.distributed.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("bucket_mb,overlap,stages", [(25, True, False), (2e-4, True, False), (25, False, False), (25, False, True)],
                         ids=["one-bucket", "many-buckets", "no-overlap", "no-overlap-staged"])
def test_flat_gradient_exchange_gloo_world2(bucket_mb, overlap, stages):
    """The exchange bench.py captures into the HIP graph at N > 1: initial weights broadcast from rank 0, gradients =
    the average of the ranks' local gradients (zeros for a parameter without one), `.grad` = views of the flat buffer with
    the parameters' own (channels_last) strides, identical weights on both ranks after two optimizer steps.  With tiny
    buckets the per-bucket hooks send several asynchronous all-reduces during backward (one bucket holds only parameters
    that never receive a gradient: reduce() sends it).  `staged`: the three stateless stages gather() / allreduce_flat() /
    adopt() that bench.py puts into and between two HIP graphs when the collective cannot be captured."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_flat_worker, args=(world, port, out, bucket_mb, overlap, stages), nprocs=world, join=True)
    a, b = out[0], out[1]
    for x, y in zip(a["w0"], b["w0"]):
        assert torch.equal(x, y)
    for s in range(2):
        for i, (la, lb) in enumerate(zip(a["steps"][s]["local"], b["steps"][s]["local"])):
            want = torch.zeros_like(a["steps"][s]["avg"][i]) if la is None else (la + lb) / 2
            assert torch.allclose(a["steps"][s]["avg"][i], want, atol=1e-7), (s, i)
            assert torch.equal(a["steps"][s]["avg"][i], b["steps"][s]["avg"][i])
    for x, y in zip(a["w"], b["w"]):
        assert torch.equal(x, y)


def test_flat_exchange_refuses_a_second_backward_before_reduce():
    """overlap=True sends a bucket when its last gradient of ONE backward has arrived; a second backward before reduce()
    (gradient accumulation) would be dropped silently -- the hook raises instead; overlap=False accumulates fine (it gathers
    inside reduce()).  Single process, no process group: the exchange degenerates to the gather."""
    from mrla_amd import distributed as D
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    x = torch.randn(4, 6)
    ex = D.FlatGradientExchange(net.parameters(), overlap=True)
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match="second backward"):
        net(x).sum().backward()
    ex.remove_hooks()
    for p in net.parameters():
        p.grad = None
    ex2 = D.FlatGradientExchange(net.parameters(), overlap=False)
    net(x).sum().backward()
    net(x).sum().backward()                                # accumulates into .grad; gathered by reduce()
    want = [p.grad.clone() for p in net.parameters()]
    ex2.reduce()
    for p, w in zip(net.parameters(), want):
        assert torch.equal(p.grad, w)


def _sync_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mrla_amd import distributed as D
    D.init_from_env("gloo")
    torch.manual_seed(5)                                 # the same weights on both ranks
    net = nn.Sequential(nn.Linear(6, 5), nn.ReLU(), nn.Linear(5, 3))
    same = D.replicas_in_sync(list(net.parameters()))
    if rank == 1:
        with torch.no_grad():
            net[2].bias[1] += 1e-7                       # one weight off by one ulp-ish on one rank
    differ = D.replicas_in_sync(list(net.parameters()))
    out[rank] = (same, differ)
    torch.distributed.destroy_process_group()


def test_replicas_in_sync_detects_a_single_diverged_weight():
    """bench.py's `config.replicas_in_sync`: per-parameter float64 sums, all-reduced with MIN and MAX."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sync_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0] == (True, False) and out[1] == (True, False)


def _rank0_first_worker(rank, world, port, out):
    """bench.py's rank-0-first section (MIOpen's solver search) and its votes, on two gloo ranks: rank 0's body has finished
    before any other rank's starts, nobody deadlocks, the votes agree on every rank, and the per-rank timing spread is
    reported as (fastest, slowest)."""
    import sys
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from mrla_amd import distributed as D
    D.init_from_env("gloo")
    stamps = {}

    def body():
        stamps["start"] = time.time()
        time.sleep(0.5 if rank == 0 else 0.05)
        stamps["end"] = time.time()
    bench.rank0_first(body, rank, world, "unit")
    lo, hi = D.min_max_over_ranks(10.0 + rank)
    yes = bench.all_ranks_ok(True, "unit/yes", rank, world)
    no = bench.all_ranks_ok(rank != 1, "unit/no", rank, world)          # one rank says no: every rank hears no
    D.barrier()
    out[rank] = dict(stamps=stamps, lo=lo, hi=hi, yes=yes, no=no)
    torch.distributed.destroy_process_group()


def test_rank0_first_votes_and_rank_spread_on_two_gloo_ranks():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank0_first_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert a["stamps"]["end"] <= b["stamps"]["start"] + 1e-3         # rank 0's section is over before rank 1's begins
    assert (a["lo"], a["hi"]) == (b["lo"], b["hi"]) == (10.0, 11.0)
    assert a["yes"] is True and b["yes"] is True and a["no"] is False and b["no"] is False
