"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

The MRLA path shards by image: every kernel is independent per sample and `bn_mrla` statistics stay local to a GPU
(the reference keeps BatchNorm unsynchronised, resnet/models/resnet_mrla_light.py:58-60), so the only exchange
step is the gradient all-reduce of the training step (resnet/train.py:174, deit/main.py:308).
"""
import os

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun / torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_from_env(backend=None):
    """Initialise the default process group when WORLD_SIZE > 1.  Returns (rank, local_rank, world_size)."""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, local, world


def wrap_data_parallel(model, device_ids=None, bucket_cap_mb=32, force=False):
    """DistributedDataParallel tuned for this workload: gradients are views into the flat buckets (no extra copy),
    buckets large enough that the 103 MB of resnet50_mrlal gradients go out as a handful of RCCL all-reduces that
    overlap the rest of backward, static graph (every parameter is used in every step), no buffer broadcast per
    step (BatchNorm statistics are per-GPU by design).  force: wrap even in a one-rank group (diagnostics: the reducer's
    hooks, bucket views and RCCL calls then run exactly as at N > 1, with nobody to exchange with)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return model
    ddp = torch.nn.parallel.DistributedDataParallel(
        model, device_ids=device_ids, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb, static_graph=True,
        broadcast_buffers=False)
    # Without a communication hook the reducer divides EVERY parameter's gradient by the world size with a kernel of its
    # own before the bucket goes out (161 launches and ~0.5 ms of GPU time per resnet50_mrlal step, rocprofv3 trace in
    # profiles/r03_notes.md).  With a hook the averaging is the hook's business: one division of the flat bucket.
    how = os.environ.get("MRLA_DDP_HOOK", "builtin")          # (A/B switch for measurements: none | python | builtin)
    if how == "builtin":
        # the C++ all-reduce hook: divides the flat bucket once, launches the all-reduce, no Python in the backward pass
        ddp._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
    elif how == "python":
        ddp.register_comm_hook(None, _allreduce_avg_hook)
    return ddp


def _allreduce_avg_hook(state, bucket):
    buf = bucket.buffer()
    if dist.get_backend() == "nccl":
        fut = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True).get_future()
    else:
        buf.div_(dist.get_world_size())
        fut = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True).get_future()
    return fut.then(lambda f: f.value()[0])


def max_over_ranks(seconds, device=None):
    """The contract's timing rule: the slowest rank's wall time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
