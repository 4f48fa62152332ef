"""Child process of tests/test_ddp_gpu.py: one DDP rank of a real MRLA model on ONE GPU (both ranks on cuda:0, gloo).

Started with `python tests/ddp_worker.py <arch> <rank> <world> <port> <outdir> <batch>` as a fresh process (it has not
touched the GPU before it is started; the pytest process is never re-executed).  Writes `<outdir>/rank<r>.pt` with the
averaged gradients, the logits and the bn_mrla running statistics of this rank.  TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build(arch):
    """The model with deterministic weights (oracle/detgen.py: no RNG involved, identical in every process)."""
    import torch
    from mrla_amd import models
    from oracle import detgen
    net = getattr(models, arch)()
    vals = detgen.fill_state_dict(net.state_dict())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return net.cuda().train()


def rank_batch(rank, batch):
    import torch
    from oracle import detgen
    x = detgen.normalish((batch, 3, 224, 224), detgen.seed_of(f"ddp/img/{rank}"))
    y = (torch.arange(batch) * 37 + 11 * rank) % 1000
    return torch.from_numpy(x).cuda(), y.cuda()


def step(net, x, y, amp=False):
    """One forward/backward; amp: under bf16 autocast, the configuration bench.py times (1x1 convolutions on the MFMA GEMM)."""
    import torch
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        logits = net(x)
    torch.nn.functional.cross_entropy(logits.float(), y).backward()
    return logits.detach().float()


def stats_of(module):
    return {k: v.detach().float().cpu().clone() for k, v in module.state_dict().items()
            if "bn_mrla.running" in k or k in ("bn1.running_mean", "bn1.running_var")}


def main():
    arch, rank, world, port, outdir, batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    amp = len(sys.argv) > 7 and sys.argv[7] == "amp"
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import torch
    from mrla_amd import distributed as D
    torch.cuda.set_device(0)
    D.init_from_env("gloo")
    net = D.wrap_data_parallel(build(arch), device_ids=[0])
    assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
    x, y = rank_batch(rank, batch)
    logits = step(net, x, y, amp)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().float().cpu().clone() for k, p in net.module.named_parameters()}
    torch.save(dict(grads=grads, logits=logits.float().cpu(), stats=stats_of(net.module)),
               os.path.join(outdir, f"rank{rank}.pt"))
    D.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
