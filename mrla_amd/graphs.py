"""Replaying a whole training step from ONE HIP graph, and proving that the replay computes what eager launches compute.

The reference launches its step eagerly (resnet/train.py:387-409: forward, criterion, zero_grad, backward, optimizer
step).  On MI355X the step of this package is launch-order static -- no host synchronisation, no allocation outside the
graph pool, every C-ABI entry point capturable -- so the same kernels on the same buffers can be captured once and
replayed, which removes the host's launch gaps (resnet50_mrlal b = 256: 30.7 -> 30.2 ms per step since the sequence entry
points of ABI 4 -- worth 1 - 2 %; round 4's per-pass calls: 34.1 -> 30.4).  `graphed_step` is the
recipe bench.py measures with, packaged for a training loop:

    step = mrla_amd.graphed_step(model, optimizer, criterion, (images, target))      # after model.cuda().train()
    for images, target in loader:
        loss = step(images, target)          # copies the batch into the static buffers, replays; `step.output` = logits
                                             # (a batch of ANOTHER shape -- the tail of an epoch -- is stepped eagerly on
                                             # the tensors it was given: `step.eager(images, target)`, `step.last_launch`)

What the recipe takes care of (each item was a bug or a trap at some point of this repository's history):
  * a few eagerly launched steps first, on a side stream: the optimizer allocates its state, MIOpen finishes its solver
    search (`torch.backends.cudnn.benchmark`), the model's WeightBank builds its pointer table (a host-to-device copy: not
    capturable) -- then ONE captured step;
  * `optimizer.zero_grad(set_to_none=True)` INSIDE the step: the gradients live in the graph's private pool and autograd
    adopts them instead of accumulating into stale ones;
  * stochastic depth draws its table with `torch.rand` inside the step; PyTorch registers the generator with the graph, so
    every replay draws fresh masks (and the same ones as an eager step started from the same generator state);
  * once a refresh of the WeightBank has been captured the bank re-casts on every later forward (functional.WeightBank);
  * `verify=K` (default 2): BEFORE the step is handed out, K replays are compared with K eagerly launched steps from the
    same weights, optimizer state, BatchNorm buffers, inputs and generator state (`replay_matches_eager`).  A library kernel
    that misbehaves under replay -- MIOpen's split-K 3x3 weight gradient did at small batches, finite but wrong,
    profiles/r04_notes.md section 10 -- is caught here instead of training on garbage: `graphed_step` raises GraphReplayMismatch
    (or, with `on_mismatch="eager"`, hands out the eagerly launched step); `deterministic_fallback=True` tries once more
    with `torch.backends.cudnn.deterministic = True` first, which makes MIOpen leave those solvers out.
  * the capture's warm-up steps, the self-check's steps and the timing steps of the deterministic retry are REAL optimizer
    steps on the example batch: `graphed_step` snapshots (model, optimizer, scaler, generator) on entry and writes the
    snapshot back in place before it returns (`restore_state=True`), so a run resumed from a checkpoint starts exactly
    where resnet/train.py's would (optimizer state that did not exist yet -- SGD's momentum buffers -- is zeroed, which is
    what a first step starts from unless `dampening` is used);
  * with a process group initialised, every rank takes the SAME decision (the verdict of the self-check is all-reduced
    with MIN before anyone raises, retries or falls back to eager launches -- a lone rank doing either hangs the others at
    the next collective), and the cudnn.deterministic retry (extra eager steps = extra collectives) is off.
The optimizer must be capturable (torch.optim.SGD in its foreach / fused forms is; Adam needs `capturable=True`).
`scaler=torch.amp.GradScaler(...)` (deit/engine.py:37,51: fp16 autocast + timm's NativeScaler): the scaled backward, the
unscale + inf check and the scale update go into the graph as well; that needs an optimizer whose step takes the scaler's
`grad_scale` / `found_inf` tensors on the device (the `fused=True` forms of SGD / Adam / AdamW; Adam / AdamW also need
`capturable=True`) -- no host decision remains.
"""
import torch

from ._lib import MrlaHipError


class GraphReplayMismatch(MrlaHipError):
    pass


def capture_step(step, warmup=3, distributed=None, pool=None):
    """PyTorch's whole-network-capture recipe: `warmup` eager calls of `step()` on a side stream, then one captured call.
    Returns the torch.cuda.CUDAGraph.  distributed (default: is a torch.distributed process group initialised?): RCCL's
    watchdog thread may query the events of earlier, eagerly launched collectives while this thread captures -- whether or not
    a collective is enqueued inside; in the default (global) capture-error mode that query invalidates the capture
    (hipErrorStreamCaptureUnsupported; seen with the two-graph tier), so such captures run in thread-local mode."""
    if distributed is None:
        import torch.distributed as dist
        distributed = dist.is_available() and dist.is_initialized()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    kw = {"capture_error_mode": "thread_local"} if distributed else {}
    if pool is not None:
        kw["pool"] = pool
    with torch.cuda.graph(graph, **kw):
        step()
    return graph


# ----------------------------------------------------------------------------------------------------------------------
# training state: everything a step reads and writes besides its inputs
# ----------------------------------------------------------------------------------------------------------------------
def _scaler_tensors(scaler):
    """The device-side state of a torch.amp.GradScaler (created lazily by its first scale())."""
    out = []
    if scaler is not None:
        for key in ("_scale", "_growth_tracker"):
            v = getattr(scaler, key, None)
            if isinstance(v, torch.Tensor):
                out.append(("optim:scaler" + key, v))
    return out


def _state_tensors(model, optimizer, scaler=None):
    """[(name, tensor)] of the parameters, the buffers (BatchNorm statistics and counters), the optimizer's state tensors and
    the loss scaler's."""
    out = [("param:" + k, p) for k, p in model.named_parameters()]
    out += [("buffer:" + k, b) for k, b in model.named_buffers()]
    if optimizer is not None:
        names = {id(p): k for k, p in model.named_parameters()}
        for p, st in optimizer.state.items():
            for key, v in st.items():
                if isinstance(v, torch.Tensor):
                    out.append((f"optim:{key}:{names.get(id(p), hex(id(p)))}", v))
    return out + _scaler_tensors(scaler)


class TrainingState:
    """A snapshot of (model, optimizer) that can be written back IN PLACE (the graph holds the tensors' addresses)."""

    def __init__(self, model, optimizer=None, scaler=None):
        self.model, self.optimizer, self.scaler = model, optimizer, scaler
        self.entries = [(k, t, t.detach().clone()) for k, t in _state_tensors(model, optimizer, scaler)]
        # compared exactly, not as float64 vectors: integer / bool tensors (BatchNorm counters, step counts)
        self.exact_keys = {k for k, t, _ in self.entries if not (t.is_floating_point() or t.is_complex())}
        self.rng = torch.cuda.get_rng_state()

    def restore(self, zero_new_state=False):
        """Write the snapshot back in place.  zero_new_state: optimizer / scaler state tensors created SINCE the snapshot
        (momentum buffers of a first step, Adam's moments and step counts) are zeroed -- their addresses may already be
        in a captured graph, so they cannot be dropped; a first step from zeroed state is a first step."""
        with torch.no_grad():
            for _, t, saved in self.entries:
                t.copy_(saved)
            if zero_new_state:
                known = {id(t) for _, t, _ in self.entries}
                for k, t in _state_tensors(self.model, self.optimizer, self.scaler):
                    if id(t) not in known:
                        # (a GradScaler creates its tensors at the first scale(): back to what that call starts from)
                        t.fill_(float(self.scaler._init_scale)) if k == "optim:scaler_scale" else t.zero_()
        torch.cuda.set_rng_state(self.rng)

    def now(self):
        """{name: float64 copy on the device} of the live tensors."""
        return {k: t.detach().double().clone() for k, t, _ in self.entries}

    def initial(self):
        return {k: saved.double() for k, _, saved in self.entries}


def _vec_norms(diff_of, keys):
    """sqrt(sum over keys of |diff_of(k)|^2) with ONE device-to-host transfer."""
    sq = torch.stack([diff_of(k).pow(2).sum() for k in keys])
    return float(sq.sum().sqrt()), sq.sqrt()


def compare_states(got, want, start, exact=()):
    """How far `got` is from `want` (both {name: float64 tensor}, reached from `start` by the same step):
      weights_rel_l2   |got - want| / |want| over ALL parameters taken as one vector
      update_rel_l2    |got - want| / |want - start| over all parameters as one vector: relative to what the step changed, so a
                       wrong gradient cannot hide behind large weights
      update_rel_l2_worst_parameter   the same per parameter, the maximum, and `worst_parameter` (noise-limited for the tiny,
                       cancelling Wq / Wk sums and for zero-initialised BatchNorm weights)
      buffers_rel_l2   float buffers (BatchNorm running statistics) as one vector, relative to their own change
      optim_rel_l2     the optimizer's state tensors (momentum buffers) as one vector, |got - want| / |want|
    plus `nonfinite` / `counter_mismatch` (names) when something is not a number / an integer tensor differs (`exact`: the
    names that were integer or bool before TrainingState.now() widened everything to float64 -- BatchNorm counters, step
    counts; `num_batches_tracked` always)."""
    res = {}
    exact = {k for k in want if k in exact or "num_batches_tracked" in k}
    params = [k for k in want if k.startswith("param:")]
    bufs = [k for k in want if k.startswith("buffer:") and k not in exact]
    optim = [k for k in want if k.startswith("optim:") and k not in exact]
    for k in exact:
        if not torch.equal(got[k], want[k]):
            res["counter_mismatch"] = k
    finite = torch.stack([torch.isfinite(got[k]).all() for k in params + bufs + optim])
    if not bool(finite.all()):
        res["nonfinite"] = (params + bufs + optim)[int((~finite).nonzero()[0])]
    err, err_k = _vec_norms(lambda k: got[k] - want[k], params)
    upd, upd_k = _vec_norms(lambda k: want[k] - start[k], params)
    wn, _ = _vec_norms(lambda k: want[k], params)
    ratio = torch.where(upd_k > 0, err_k / upd_k.clamp_min(1e-300), torch.zeros_like(err_k))
    worst = int(torch.nan_to_num(ratio, nan=float("inf")).argmax())
    res.update(weights_rel_l2=err / wn if wn > 0 else err, update_rel_l2=err / upd if upd > 0 else err,
               update_rel_l2_worst_parameter=float(ratio[worst]), worst_parameter=params[worst][6:])
    if bufs:
        e, _ = _vec_norms(lambda k: got[k] - want[k], bufs)
        u, _ = _vec_norms(lambda k: want[k] - start[k], bufs)
        res["buffers_rel_l2"] = e / u if u > 0 else e
    else:
        res["buffers_rel_l2"] = 0.0
    if optim:
        e, _ = _vec_norms(lambda k: got[k] - want[k], optim)
        u, _ = _vec_norms(lambda k: want[k], optim)
        res["optim_rel_l2"] = e / u if u > 0 else e
    else:
        res["optim_rel_l2"] = 0.0
    return res


_MEASURES = ("weights_rel_l2", "update_rel_l2", "update_rel_l2_worst_parameter", "buffers_rel_l2", "optim_rel_l2")


def replay_matches_eager(eager_step, replay, model, optimizer, steps=3, replay_loss=None, tol=1e-2, scaler=None):
    """Do `steps` consecutive replays compute what eagerly launched steps compute?  Step by step, from the CURRENT state of
    (model, optimizer): at step k the same state s_k (weights, optimizer state, BatchNorm buffers, generator state -- so all
    legs drop the same images) is stepped three times -- eagerly (e), eagerly again (e': the run-to-run noise floor; MIOpen
    accumulates its weight gradients with atomics) and by the k-th replay of the graph (r) -- and r is compared with e
    relative to what the step changed; then training continues from e.  One-step comparisons keep the chaos of bf16 training
    out of the measurement (after four free-running steps two EAGER runs differ by 6 % of their update on resnet50_mrlal),
    while the replays still come one after the other, which is what exposes state that a replay carries over (MIOpen's
    split-K weight gradient was right on the first replay and garbage from the second on).
    Returns the maxima over the steps of compare_states() for replay vs eager, `noise_*` = the same for eager vs eager,
    `loss_eager` / `loss_replay` (when eager_step() returns its loss tensor / `replay_loss` is the captured step's static
    loss tensor), and `ok`: everything finite, no counter mismatch, and each of weights_rel_l2, update_rel_l2,
    buffers_rel_l2, optim_rel_l2 at most max(tol, 4 x its noise floor)."""
    torch.cuda.synchronize()
    res = {k: 0.0 for k in _MEASURES}
    res.update({"noise_" + k: 0.0 for k in _MEASURES})
    res.update(steps=steps, tol=tol, worst_parameter=None)
    l_e, l_r = [], []

    def run(fn, state, static_loss=None):
        state.restore()
        out = fn()
        loss = static_loss if static_loss is not None else out
        val = float(loss.detach().float()) if isinstance(loss, torch.Tensor) and loss.numel() == 1 else None
        torch.cuda.synchronize()
        return state.now(), val

    for _ in range(steps):
        s_k = TrainingState(model, optimizer, scaler)
        start = s_k.initial()
        e1, le = run(eager_step, s_k)
        e2, _ = run(eager_step, s_k)
        r, lr = run(replay, s_k, replay_loss)
        noise, cmp_ = compare_states(e2, e1, start, s_k.exact_keys), compare_states(r, e1, start, s_k.exact_keys)
        for k in _MEASURES:
            if cmp_[k] != cmp_[k] or cmp_[k] > res[k]:
                res[k] = cmp_[k] if cmp_[k] == cmp_[k] else float("inf")
                if k == "update_rel_l2_worst_parameter":
                    res["worst_parameter"] = cmp_["worst_parameter"]
            res["noise_" + k] = max(res["noise_" + k], noise[k])
        for flag in ("nonfinite", "counter_mismatch"):
            if flag in cmp_:
                res[flag] = cmp_[flag]
        if le is not None:
            l_e.append(round(le, 5))
        if lr is not None:
            l_r.append(round(lr, 5))
        with torch.no_grad():                   # training continues from the first eager result
            for k, t, _s in s_k.entries:
                t.copy_(e1[k].to(t.dtype))
    if l_e or l_r:
        res.update(loss_eager=l_e, loss_replay=l_r)
    res["ok"] = bool("nonfinite" not in res and "counter_mismatch" not in res
                     and all(res[k] == res[k] and res[k] <= max(tol, 4 * res["noise_" + k])
                             for k in ("weights_rel_l2", "update_rel_l2", "buffers_rel_l2", "optim_rel_l2")))
    return res


# ----------------------------------------------------------------------------------------------------------------------
def _all_ranks(flag):
    """AND of `flag` over the ranks of the default process group (True alone): decisions that change what a rank launches
    next -- raise, retry, fall back to eager steps -- must be the same everywhere or the next collective hangs."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return bool(flag)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


class GraphedStep:
    """See graphed_step()."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, autocast=torch.bfloat16, exchange=None, warmup=3,
                 verify=2, tol=1e-2, on_mismatch="raise", deterministic_fallback=False, scaler=None, restore_state=True):
        if scaler is not None and scaler.is_enabled() and not getattr(optimizer, "_step_supports_amp_scaling", False):
            raise MrlaHipError("graphed_step(scaler=...): GradScaler.step() decides on the HOST whether to skip the step unless the "
                               "optimizer takes the scaler's grad_scale / found_inf tensors itself -- build the optimizer with "
                               "fused=True (torch.optim.SGD / Adam / AdamW), or keep the fp16 recipe on eager launches")
        if not isinstance(example_inputs, (tuple, list)) or not example_inputs:
            raise MrlaHipError("example_inputs: a tuple (model input, *loss_fn arguments)")
        if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in example_inputs):
            raise MrlaHipError("graphed_step: every example input must be a CUDA tensor (the graph reads static device buffers)")
        self.model, self.optimizer, self.loss_fn, self.exchange, self.scaler = model, optimizer, loss_fn, exchange, scaler
        self.autocast = autocast
        self.static = [t.detach().clone() for t in example_inputs]
        self.loss = self.output = None           # of the latest step (static tensors while the graph is in use)
        self.graph = None
        self.report = None
        self.last_launch = None                  # "graph" | "eager" | "eager (other shape)": how the latest step ran
        import torch.distributed as dist
        dist_on = exchange is not None or (dist.is_available() and dist.is_initialized())
        self.miopen_deterministic = bool(torch.backends.cudnn.deterministic)
        # everything below takes REAL optimizer steps on the example batch; the caller's training state is put back at the end
        entry = TrainingState(model, optimizer, scaler) if restore_state else None
        try:
            self._build(warmup, verify, tol, on_mismatch, deterministic_fallback and not dist_on, dist_on)
        finally:
            if entry is not None:
                torch.cuda.synchronize()
                entry.restore(zero_new_state=True)

    def _build(self, warmup, verify, tol, on_mismatch, deterministic_fallback, dist_on):
        model, optimizer = self.model, self.optimizer
        for attempt in (0, 1):
            self.graph = capture_step(self.eager, warmup=max(1, warmup), distributed=dist_on)
            self._static = (self.loss, self.output)  # what the captured step wrote: every replay overwrites these two
            if not verify:
                break
            self.report = replay_matches_eager(self.eager, self.graph.replay, model, optimizer, steps=verify,
                                               replay_loss=self._static[0], tol=tol, scaler=self.scaler)
            ok_here = self.report["ok"]
            self.report["ok"] = _all_ranks(ok_here)          # one verdict for every rank
            self.report["ok_on_this_rank"] = ok_here
            if self.report["ok"]:
                break
            msg = ("the replayed HIP graph of the training step does not reproduce the eagerly launched step"
                   + ("" if not ok_here else " on another rank (it does on this one)") + ": "
                   + ", ".join(f"{k} {self.report[k]:.3g}" for k in ("weights_rel_l2", "update_rel_l2", "noise_update_rel_l2"))
                   + f", worst parameter {self.report['worst_parameter']}"
                   + "".join(f", {k}: {self.report[k]}" for k in ("nonfinite", "counter_mismatch") if k in self.report))
            import warnings
            if attempt == 0 and deterministic_fallback and not torch.backends.cudnn.deterministic:
                # MIOpen's atomically accumulating (split-K) weight-gradient solvers are right when launched eagerly and
                # garbage from the second replay of a graph on; cudnn.deterministic leaves them out (resnet/train.py:107-110
                # sets it with --seed).  One more attempt with it: warm-up (MIOpen searches again), capture, check.
                # (Single process only: these are extra steps, i.e. extra collectives, that the other ranks would not take.)
                before = self._eager_ms()
                torch.backends.cudnn.deterministic = True
                for _ in range(max(1, warmup)):
                    self.eager()                     # (MIOpen searches again)
                after = self._eager_ms()
                if after <= 1.5 * before:
                    warnings.warn(msg + "; trying once more with torch.backends.cudnn.deterministic = True")
                    self.miopen_deterministic = True
                    self.graph = None
                    continue
                # MIOpen's deterministic solver list can be catastrophically slow (resnet50_mrlal b = 256: 7.3 s per step
                # against 30 ms): a fallback that costs more than the graph can win is no fallback
                torch.backends.cudnn.deterministic = False
                self.eager()
                msg += (f"; torch.backends.cudnn.deterministic was tried and switched off again ({after:.0f} ms per eager step "
                        f"against {before:.0f} ms)")
            if on_mismatch != "eager":
                raise GraphReplayMismatch(msg)
            warnings.warn(msg + "; launching the step eagerly instead")
            self.graph = None
            break

    def _eager_ms(self, n=2):
        import time
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            self.eager()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    def eager(self, *inputs):
        """One step launched kernel by kernel -- on `inputs` = (x, *rest) of ANY batch shape (the tail batch of an epoch), or,
        without arguments, on the static buffers (this is also what gets captured)."""
        if inputs:
            if len(inputs) != len(self.static):
                raise MrlaHipError(f"graphed step takes {len(self.static)} tensors, got {len(inputs)}")
            x, rest = inputs[0], inputs[1:]
        else:
            x, rest = self.static[0], self.static[1:]
        if self.autocast is not None:
            with torch.autocast("cuda", dtype=self.autocast):
                out = self.model(x)
                loss = self.loss_fn(out.float() if out.dtype != torch.float32 else out, *rest)
        else:
            out = self.model(x)
            loss = self.loss_fn(out, *rest)
        self.optimizer.zero_grad(set_to_none=True)
        if self.scaler is not None:
            # deit/engine.py:51 (timm's NativeScaler): scale, backward, unscale + inf check + step, update -- with a fused
            # optimizer all of it stays on the device (found_inf gates the update inside the optimizer's kernel)
            self.scaler.scale(loss).backward()
            if self.exchange is not None:
                self.exchange.reduce()
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            loss.backward()
            if self.exchange is not None:
                self.exchange.reduce()
            self.optimizer.step()
        # detached aliases: holding the loss itself would keep this step's autograd graph -- and with it the parameters'
        # AccumulateGrad nodes and the stream they were created on -- alive into the next step, which breaks a later capture
        self.loss, self.output = loss.detach(), out.detach()
        self.last_launch = "eager"
        return self.loss

    def __call__(self, *inputs):
        if inputs:
            if len(inputs) != len(self.static):
                raise MrlaHipError(f"graphed step takes {len(self.static)} tensors, got {len(inputs)}")
            if any(t.shape != s.shape for s, t in zip(self.static, inputs)):
                # the last, smaller batch of an epoch (resnet/train.py's loader does not drop it): this batch, stepped
                # eagerly on the tensors given -- NOT the static buffers, which still hold the previous batch
                if inputs[0].shape[1:] != self.static[0].shape[1:]:
                    raise MrlaHipError(f"graphed step was captured for inputs of shape {tuple(self.static[0].shape)}, got "
                                       f"{tuple(inputs[0].shape)}: only the batch dimension may differ (such a batch is stepped "
                                       "eagerly on the tensors given)")
                loss = self.eager(*inputs)
                self.last_launch = "eager (other shape)"
                return loss
            for s, t in zip(self.static, inputs):
                if t is not s:
                    s.copy_(t, non_blocking=True)
        if self.graph is None:
            return self.eager()
        self.graph.replay()
        self.loss, self.output = self._static
        self.last_launch = "graph"
        return self.loss


def graphed_step(model, optimizer, loss_fn, example_inputs, **kw):
    """Capture `loss = loss_fn(model(x), *rest); optimizer.zero_grad(); loss.backward(); [exchange.reduce();] optimizer.step()`
    (resnet/train.py:397-409) into one HIP graph.  example_inputs = (x, *rest): CUDA tensors of the batch shape the loop will
    feed.  Returns a callable `step(x, *rest) -> loss` (a static tensor, overwritten by every replay; `step.output`: the
    logits); `step.report` holds the replay-vs-eager comparison made before it was handed out.
    A call with another BATCH size (the tail of an epoch) steps that batch eagerly on the tensors given (`step.last_launch`).
    Keywords: autocast (dtype or None, default torch.bfloat16), exchange (a distributed.FlatGradientExchange for N > 1),
    scaler (a torch.amp.GradScaler for the fp16 recipe of deit/engine.py:37,51; needs a fused=True optimizer),
    restore_state (default True: weights, BatchNorm buffers, optimizer / scaler state and the generator are put back to
    what they were on entry -- the warm-up and self-check steps leave no trace),
    warmup (eager steps before the capture, default 3), verify (steps of the replay-vs-eager check, 0 = skip, default 2),
    tol (its bound on the weights' relative L2 difference), on_mismatch ("raise" | "eager"), deterministic_fallback (default
    False; True: if the check fails, switch torch.backends.cudnn.deterministic on -- MIOpen then leaves out its atomically
    accumulating solvers, one known cause -- and capture + check once more before giving up, unless the eager step then runs
    > 1.5 x slower: MIOpen's deterministic solver list took 7.3 s per resnet50_mrlal step at b = 256 on MI355X, against 30 ms;
    `step.miopen_deterministic` says what is on; never with a process group initialised).  With N > 1 ranks every rank
    reaches the same verdict (`step.report["ok"]` is the AND over the ranks)."""
    return GraphedStep(model, optimizer, loss_fn, example_inputs, **kw)
