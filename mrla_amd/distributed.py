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
    # profiles/r03_notes.md).  With a hook the averaging is the hook's business: the C++ all-reduce hook divides the flat
    # bucket once and launches the all-reduce, no Python in the backward pass (a Python averaging hook measured slower
    # than no hook at all: profiles/r03_notes.md).
    ddp._register_builtin_comm_hook(dist.BuiltinCommHookType.ALLREDUCE)
    return ddp


class FlatGradientExchange:
    """The gradient average of a data-parallel step in a form that a HIP-graph-captured step can hold
    (resnet/train.py:174's DistributedDataParallel does the same averaging from its reducer).

    Why not DistributedDataParallel inside the graph: its reducer copies every parameter's gradient into the bucket view
    with a kernel of its own (161 launches for resnet50_mrlal) and joins its streams per bucket; captured, that step
    replays no faster than the eager launches (profiles/r03_notes.md section 4), so the N > 1 points would carry the eager
    launch gaps (~7 % of the step) that the graph-replayed N = 1 point does not.  Here
      * `.grad` is None before backward, so autograd's AccumulateGrad adopts the incoming gradient tensors (no kernel);
      * the parameters are laid out in ONE flat buffer in REVERSE registration order (the order backward produces their
        gradients) and cut into a few buckets (`bucket_mb`); when the last gradient of a bucket has arrived (a
        post-accumulate hook per parameter counts them down) ONE `_foreach_copy_` gathers the bucket and ONE asynchronous
        all-reduce (RCCL `avg`) sends it off -- it overlaps the rest of backward, as DDP's buckets do;
      * `reduce()` (after backward) waits for the collectives and re-points every `.grad` at its view of the flat buffer
        (same sizes and strides as the parameter: the optimizer's layout contract).
    2 launches per bucket, all capturable.  overlap=False: one copy and one all-reduce for everything inside reduce().

    usage per step:  opt.zero_grad(set_to_none=True); loss.backward(); exchange.reduce(); opt.step()
    ONE backward per reduce(): with overlap=True a bucket goes out the moment its last gradient of that backward has
    arrived, so a second backward before reduce() (gradient accumulation, two losses) would add to gradients that are
    already on the wire; the hook raises instead of averaging the first micro-batch only.  Accumulate with
    overlap=False (everything is gathered and sent inside reduce())."""

    def __init__(self, params, group=None, broadcast=True, bucket_mb=25, overlap=True):
        self.params = [p for p in params if p.requires_grad][::-1]          # backward's order
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = group
        dev, dt = self.params[0].device, self.params[0].dtype
        for p in self.params:
            if p.device != dev or p.dtype != dt:
                raise ValueError("FlatGradientExchange expects all trainable parameters on one device in one dtype")
            if not _is_dense(p):
                raise ValueError(f"parameter of shape {tuple(p.shape)} / stride {p.stride()} is not dense")
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=dt, device=dev)
        self.views, self.bucket_of, bounds, off = [], [], [0], 0
        limit = max(1, int(bucket_mb * (1 << 20)) // self.flat.element_size()) if overlap else self.flat.numel() + 1
        for p in self.params:
            n = p.numel()
            if off + n - bounds[-1] > limit and off > bounds[-1]:
                bounds.append(off)
            self.bucket_of.append(len(bounds) - 1)
            self.views.append(self.flat[off:off + n].as_strided(p.size(), p.stride()))   # the parameter's own memory order
            off += n
        bounds.append(off)
        self.buckets = [self.flat[bounds[k]:bounds[k + 1]] for k in range(len(bounds) - 1)]
        self.members = [[i for i, kb in enumerate(self.bucket_of) if kb == k] for k in range(len(self.buckets))]
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._avg = dist.is_initialized() and dist.get_backend(group) == "nccl"
        if self._avg:                            # (an RCCL build without ncclAvg: pre-divide and sum instead)
            try:
                dist.all_reduce(torch.zeros(1, dtype=dt, device=dev), op=dist.ReduceOp.AVG, group=group)
            except (RuntimeError, ValueError):
                self._avg = False
        self._pending = [len(m) for m in self.members]
        self._works = [None] * len(self.buckets)
        self._sent = [False] * len(self.buckets)
        self._gathered = [False] * len(self.buckets)
        self._hooks = []
        if overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._arrived(self.bucket_of[i])))
        if broadcast and dist.is_initialized() and self.world > 1:         # what DistributedDataParallel does when it is built
            with torch.no_grad():
                for p in self.params[::-1]:
                    dist.broadcast(p, src=0, group=group)

    def _arrived(self, k):
        def hook(_p):
            if self._sent[k]:
                raise RuntimeError("FlatGradientExchange(overlap=True): a gradient arrived for a bucket that has already been "
                                   "sent -- a second backward before reduce(); accumulate with overlap=False")
            self._pending[k] -= 1
            if self._pending[k] == 0:
                self._send(k)
        return hook

    def _gather(self, k):
        """Bucket k's gradients -> its slice of the flat buffer (one `_foreach_copy_`; no collective)."""
        dst, src = [], []
        for i in self.members[k]:
            g, v = self.params[i].grad, self.views[i]
            if g is None:                        # a parameter that took no part in this step contributes zeros
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g)                    # (any strides: copy_ semantics; the usual case is the parameter's own)
        if dst:
            torch._foreach_copy_(dst, src)
        self._gathered[k] = True

    def _allreduce(self, k):
        if dist.is_initialized():
            buf = self.buckets[k]
            if self._avg:
                self._works[k] = dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            else:
                if self.world > 1:
                    buf.div_(self.world)
                self._works[k] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _send(self, k):
        self._gather(k)
        self._allreduce(k)
        self._sent[k] = True

    # The stages of reduce() on their own, for a step that is replayed from HIP graphs AROUND a collective that cannot be
    # captured (gloo; an RCCL build whose collectives do not survive stream capture):
    #   graph 1: forward, loss, backward, gather()   |   allreduce_flat(), launched eagerly   |   graph 2: adopt(), optimizer step
    # gather() and allreduce_flat() keep no state between calls (a replayed graph does not re-run the Python around it).
    def gather(self):
        """After backward: every bucket's gradients into the flat buffer (capturable: copies only)."""
        for k in range(len(self.buckets)):
            self._gather(k)
            self._gathered[k] = False            # (stateless: reduce() / exchange() after it would simply gather again)

    def allreduce_flat(self):
        """All-reduce (average) every bucket of the flat buffer as it stands, and wait."""
        for k in range(len(self.buckets)):
            self._allreduce(k)
        for k, w in enumerate(self._works):
            if w is not None:
                w.wait()
            self._works[k] = None

    def exchange(self):
        """Stateful form used by reduce(): send what the hooks have not sent yet, wait for everything."""
        for k in range(len(self.buckets)):
            if not self._sent[k]:
                if not self._gathered[k]:
                    self._gather(k)
                self._allreduce(k)
                self._sent[k] = True
        for k, w in enumerate(self._works):
            if w is not None:
                w.wait()
            self._works[k] = None

    def adopt(self):
        """Every `p.grad` becomes its view of the flat buffer (no kernel); the bookkeeping is reset for the next step."""
        for k in range(len(self.buckets)):
            self._sent[k] = self._gathered[k] = False
            self._pending[k] = len(self.members[k])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def reduce(self):
        """After backward: finish the exchange; afterwards every `p.grad` is a view of the flat buffer holding the average."""
        self.exchange()
        self.adopt()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def _is_dense(t):
    """Non-overlapping and dense in some dimension order (contiguous or channels_last parameters)."""
    if t.numel() == 0:
        return True
    dims = sorted((d for d in range(t.dim()) if t.size(d) > 1), key=lambda d: t.stride(d))
    expect = 1
    for d in dims:
        if t.stride(d) != expect:
            return False
        expect *= t.size(d)
    return True


def replicas_in_sync(params, group=None):
    """True when every rank holds bit-identical parameters (per-parameter float64 sums, all-reduced with MIN and MAX): what
    data-parallel training guarantees after any number of steps if -- and only if -- every rank applied the same averaged
    gradients to the same starting weights.  BatchNorm buffers are not compared: they stay per-GPU by design."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return True
    sums = torch.stack([p.detach().double().sum() for p in params])
    if dist.get_backend(group) != "nccl":
        sums = sums.cpu()
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    return bool(torch.equal(lo, hi))


def max_over_ranks(seconds, device=None):
    """The contract's timing rule: the slowest rank's wall time."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_max_over_ranks(seconds):
    """(fastest, slowest) rank's value of a per-rank wall time (not inside any timed region: one all-gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([seconds], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(every, mine)
    vals = [float(t.item()) for t in every]
    return min(vals), max(vals)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
