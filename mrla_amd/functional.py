"""torch.autograd wrappers around the C ABI (include/mrla_hip.h).

PyTorch is plumbing here: it owns the device buffers and the stream and records the autograd edge; all
arithmetic of the MRLA path happens in libmrla_hip.so.  There is deliberately no fallback: CPU tensors or
a missing library raise.
"""
import contextlib
import ctypes
import threading
import weakref
from math import log

import torch

from . import _lib as L

_DT = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


def k_size_for(c):
    """Conv1d tap count of Wq / Wk (reference: mrla_light_module.py:40-42)."""
    t = int(abs((log(c, 2) + 1) / 2.0))
    return t if t % 2 else t + 1


class KernelTimer:
    """Optional HIP-event timing of selected C-ABI launches on the stream they are launched on (used by
    bench.py for the roofline figure).  Events are read back only after the caller synchronises."""

    def __init__(self, names=None):
        self.names = None if names is None else set(names)       # None: every C-ABI launch
        self.records = []          # (name, nbytes, start_event, end_event, alg_bytes, path_bytes)

    def summary(self):
        """Per label: launches, ms, `bytes` (what the launch is built to move), `bytes_alg` (SURVEY.md section 8(d)'s
        algorithmic count for that launch where it differs) and `bytes_path` (its share of the section-8(d) compulsory bytes
        of the whole MRLA block, attributed to the pass that completes a direction; 0 for the other passes)."""
        out = {}
        for name, nbytes, e0, e1, alg, path in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, bytes=0, bytes_alg=0, bytes_path=0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["bytes"] += nbytes
            d["bytes_alg"] += nbytes if alg is None else alg
            d["bytes_path"] += path
        return out


TIMER = None      # set to a KernelTimer to time launches
SEQUENCES = True  # a tail / BatchNorm direction = ONE call of a sequence entry point (mrla_light_tail_fwd, mrla_bn_bwd, ...;
#                   include/mrla_hip.h ABI 4) instead of one call per pass.  Same kernels, same buffers, bit-identical
#                   results; per-pass calls are used while a KernelTimer is on (events around every kernel) or when False.


LEAN = False      # True: the fused training tail does not store x_t where the library can re-form it (mrla_light_lean_supported:
#                   13N instead of 15N elements of HBM traffic per block and step, 1N less activation memory, bit-identical
#                   results).  Not the default: inside the resnet50_mrlal step the two forms take the same time on MI355X
#                   (29.96 - 29.99 ms lean, 29.80 - 29.95 ms stored: the backward statistics pass pays for re-forming x_t
#                   what the forward one saves, profiles/r06_notes.md section 3) -- switch it on for the memory.


def _seq():
    return SEQUENCES and TIMER is None


def _seq_call(name, *args):
    L.call(name, *args)


def _call(name, nbytes, *args, entry=None, alg=None, path=0):
    """Launch the C-ABI entry point `entry` (default: `name`); `name` is the label the optional timer records it under,
    with `nbytes` / `alg` / `path` as KernelTimer.summary describes them."""
    t = TIMER
    if t is not None and (t.names is None or name in t.names):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call(entry or name, *args)
        e1.record()
        t.records.append((name, nbytes, e0, e1, alg, path))
    else:
        L.call(entry or name, *args)


def _ptr(t):
    """Device address as a plain int (ctypes converts it for a void* parameter; None is NULL) -- a c_void_p object per
    argument costs ~0.3 us on a path that passes ~6000 pointers per training step."""
    return t.data_ptr() if t is not None else None


# ------------------------------------------------------------------------------------------------------
# Per-forward batching of the tiny bookkeeping kernels: 85 `num_batches_tracked += 1` launches and 16 x 4 stochastic-depth
# mask kernels per resnet50_mrlal step become one _foreach_add_ and one mask table.
# ------------------------------------------------------------------------------------------------------
class _Bookkeeping:
    def __init__(self, n_drop_paths, bank=None):
        self.counters, self.n_dp, self.table, self.key, self.next = [], int(n_drop_paths), None, None, 0
        self.bank = bank           # WeightBank of the model this forward belongs to (or None)

    def drop_path_row(self, batch, drop_prob, device):
        """floor(keep + U[0,1)) / keep for one block: a row of a table drawn once per forward."""
        key = (batch, float(drop_prob), str(device))
        if self.table is None or self.key != key or self.next >= self.n_dp:
            keep = 1.0 - drop_prob
            self.table = torch.floor(keep + torch.rand((self.n_dp, batch), dtype=torch.float32, device=device)) / keep
            self.key, self.next = key, 0
        row = self.table[self.next]
        self.next += 1
        return row


_TLS = threading.local()


def current_bookkeeping():
    return getattr(_TLS, "book", None)


@contextlib.contextmanager
def batched_bookkeeping(n_drop_paths=0, bank=None):
    """Inside: train-mode BatchNorm counters handled by bn_act / the fused tails are collected and bumped with ONE
    torch._foreach_add_ on exit, layers.drop_path_scale serves stochastic-depth rows from one table, and conv_bn_act
    takes the bf16 working copies of its weights from `bank` (a refreshed WeightBank)."""
    prev, book = current_bookkeeping(), _Bookkeeping(n_drop_paths, bank)
    _TLS.book = book
    try:
        yield book
    finally:
        _TLS.book = prev
        if book.counters:
            torch._foreach_add_(book.counters, 1)


def bump_batch_counter(bn):
    """num_batches_tracked += 1 for a train-mode BatchNorm; returns the momentum to use (handles momentum=None)."""
    momentum = bn.momentum
    if bn.num_batches_tracked is not None:
        book = current_bookkeeping()
        if book is not None and momentum is not None:
            book.counters.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
            if momentum is None:
                momentum = 1.0 / float(bn.num_batches_tracked)
    return momentum


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_graph_task = getattr(torch._C, "_current_graph_task_id", None)


def _graph_task_id():
    """Id of the autograd engine's running backward pass, -1 outside one (or when this torch has no such query)."""
    return _graph_task() if _graph_task is not None else -1


def _stream():
    """The current stream of the current device as the C ABI takes it (the raw-handle query: torch.cuda.current_stream()
    builds a Stream object, ~10 us on a path that launches ~600 kernels per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _f32(t):
    """Parameters / statistics as contiguous float32 (no copy when they already are)."""
    if t is None:
        return None
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _grad_like(g, shape, stride):
    """`g` (any dense tensor with shape.numel() elements in row-major order of `shape`) as a gradient with the
    parameter's own shape AND strides -- autograd's / DDP's gradient layout contract.  Free (a re-striding view) when the
    parameter's strides enumerate its elements in row-major order once size-1 dimensions are ignored (contiguous
    weights, and channels_last [n, k, 1, 1] / [c, 1, 3, 3] ones); any other dense layout gets a strided copy."""
    g = g.contiguous()
    dims = [(n, s) for n, s in zip(shape, stride) if n > 1]
    expect, row_major = 1, True
    for n, s in reversed(dims):
        row_major &= s == expect
        expect *= n
    if row_major:
        return g.as_strided(shape, stride)
    return torch.empty_strided(shape, stride, dtype=g.dtype, device=g.device).copy_(g.view(shape))


def _on_device(fn):
    """Run a Function.forward / backward body with the tensor argument's device current: the C ABI launches on the current
    HIP device and on the stream handed over, while the buffers live on x.device (a model on cuda:1 in a process whose
    current device is cuda:0 would otherwise launch on the wrong device)."""
    import functools

    @functools.wraps(fn)
    def wrapped(ctx, x, *args):
        if isinstance(x, torch.Tensor) and x.is_cuda and x.device.index != torch.cuda.current_device():
            with torch.cuda.device(x.device):
                return fn(ctx, x, *args)
        return fn(ctx, x, *args)
    return wrapped


class _RunningStats:
    """running_mean / running_var as the contiguous float32 buffers the kernels read and (train mode) update in place.
    Buffers of another dtype or layout (model.half() / .bfloat16(), as the reference tolerates) go through float32 copies
    that are written back after the update; a wrong size raises."""

    def __init__(self, rm, rv, c, what):
        for t in (rm, rv):
            if t is None or not t.is_cuda or t.numel() != c:
                raise L.MrlaHipError(f"{what}: running statistics must be CUDA tensors with {c} elements")
        self.src = (rm, rv)
        ok = all(t.dtype == torch.float32 and t.is_contiguous() for t in (rm, rv))
        self.rm, self.rv = (rm, rv) if ok else (rm.detach().float().contiguous(), rv.detach().float().contiguous())
        self.copy_back = not ok

    def finish(self, updated):
        if self.copy_back and updated:
            with torch.no_grad():
                self.src[0].copy_(self.rm)
                self.src[1].copy_(self.rv)


def _require_cuda(x, what):
    if not x.is_cuda:
        raise L.MrlaHipError(f"{what}: got a {x.device} tensor. mrla_amd runs only on an AMD GPU through libmrla_hip.so; "
                             "the CPU restatement lives in oracle/ and is test infrastructure, not a fallback.")
    if x.dtype not in _DT:
        raise L.MrlaHipError(f"{what}: unsupported dtype {x.dtype}")


_CL = torch.channels_last


def _layout_of(x, want=None):
    """(layout enum, tensor to hand to the kernels).  NCHW-contiguous is the reference's contract; channels_last
    (NHWC) tensors run on the NHWC kernels without any conversion.  `want` forces the layout of a second operand."""
    if want is None:
        if x.is_contiguous():
            return L.NCHW, x
        if x.dim() == 4 and x.is_contiguous(memory_format=_CL):
            return L.NHWC, x
        return L.NCHW, x.contiguous()
    if want == L.NHWC:
        return L.NHWC, x.contiguous(memory_format=_CL)
    return L.NCHW, x.contiguous()


class _DeferredBnBox:
    """Hand-over between the fused MRLA producer and the deferred BatchNorm in front of it (bn_act(defer=True)): the MRLA
    backward apply pass forms dpre, the gradient of that BatchNorm's output, and can take the two per-channel sums its
    backward needs on the way (mrla_light_apply_bwd: pre / pre_tmom).  It leaves them here, tagged with the gradient tensor
    they belong to; the BatchNorm's backward uses them when exactly that tensor arrives, and runs its own statistics pass
    otherwise (another consumer added to the gradient, a layout conversion, a path without the fused sums)."""
    __slots__ = ("ptr", "shape", "tmom", "rows", "center", "ref", "version")

    def __init__(self):
        self.ptr = self.shape = self.tmom = self.rows = self.ref = self.version = None
        self.center = None         # that BatchNorm's saved mean [c]: the sums are taken about it

    def put(self, dpre, tmom, rows):
        if (dpre.numel() // dpre.shape[1]) % rows:       # the BatchNorm backward takes rows of equally many pixels: fold them
            tmom, rows = tmom.sum(0, keepdim=True, dtype=torch.float64).float(), 1
        self.ptr, self.shape, self.tmom, self.rows = dpre.data_ptr(), tuple(dpre.shape), tmom, rows
        self.ref, self.version = weakref.ref(dpre), dpre._version

    def take(self, dy):
        """The sums belong to `dy` only if it is the very tensor the producer wrote, unmodified: the producer's tensor is
        still alive (so nothing else can have been allocated at its address), `dy` sits at that address with that shape,
        and nothing has been written into it since (autograd accumulates a second consumer's gradient IN PLACE into the
        first-arrived buffer when it owns it; a tensor hook may edit it: both bump the version counter)."""
        alive = self.ref() if self.ref is not None else None
        ok = (self.tmom is not None and alive is not None and dy.data_ptr() == self.ptr and tuple(dy.shape) == self.shape
              and dy._version == self.version)
        tmom, rows = self.tmom, self.rows
        self.ptr = self.shape = self.tmom = self.rows = self.ref = self.version = None
        return (tmom, rows) if ok else None


class LightConfig:
    """Static configuration of one MRLA-light call.  fuse: the first tensor argument is the block's pre-activation
    and x_t = relu(pre + o_prev) is formed inside the statistics kernel (resnet_mrla_light.py:113-114 folded in)."""
    __slots__ = ("d", "bn_mode", "momentum", "eps", "res", "act", "fuse", "pre_affine", "pre_box", "infer")

    def __init__(self, d, bn_mode=L.BN_NONE, momentum=0.1, eps=1e-5, res=0, act=L.ACT_NONE, fuse=False, pre_affine=None,
                 pre_box=None):
        self.d, self.bn_mode, self.momentum, self.eps, self.res, self.act = d, bn_mode, momentum, eps, res, act
        self.pre_affine = pre_affine     # (scale[c], shift[c]) fp32 of a deferred BatchNorm in front of the fused producer
        self.pre_box = pre_box           # its _DeferredBnBox: where the backward leaves that BatchNorm's gradient sums
        self.infer = False               # nothing will be differentiated (set by mrla_light from the autograd state)
        self.fuse = fuse


class _LightFn(torch.autograd.Function):
    """out = res*x + dp[b] * BN( a[b,g] * act(dwconv3x3(x, wv)) + lam * o_prev )

    Covers mrla_light_layer (o_prev = lam = BN = dp = None, res = 0), the light mrla_module (+ lam*o_prev)
    and the fused block tail of resnet_mrla_light.py:116 (BN + DropPath + residual).
    """

    @staticmethod
    @_on_device
    def forward(ctx, x, o_prev, wq, wk, wv, lam, gamma, beta, running_mean, running_var, dp, cfg):
        _require_cuda(x, "mrla light forward")
        layout, xc = _layout_of(x)
        b, c, h, w = xc.shape
        d = cfg.d
        # NCHW is the reference's contract for any map size (mmdet: 800x1333); the NCHW slab kernels need a plane row to
        # fit one wave and a slab to fit the LDS.  Shapes beyond that run on the NHWC kernels through ONE internal
        # channels_last conversion, and the result goes back to NCHW-contiguous memory.
        via_nhwc = layout == L.NCHW and L.load().mrla_light_wgrad_rows(b, c, h, w, _DT[xc.dtype], L.NCHW) == L.EUNSUPPORTED
        if via_nhwc:
            layout, xc = L.NHWC, xc.contiguous(memory_format=_CL)
        if c % d:
            raise L.MrlaHipError(f"channels ({c}) not divisible by dim_perhead ({d})")
        oc = None
        if o_prev is not None:
            if o_prev.shape != x.shape or o_prev.dtype != x.dtype:
                raise L.MrlaHipError("o_prev must match x in shape and dtype")
            oc = _layout_of(o_prev, layout)[1]
        dt = _DT[xc.dtype]
        dev = xc.device
        wq32, wk32 = _f32(wq).reshape(-1), _f32(wk).reshape(-1)
        wv32 = _f32(wv).reshape(c, 9)
        lam32 = _f32(lam).reshape(-1) if lam is not None else None
        dp32 = _f32(dp).reshape(-1) if dp is not None else None
        ks = wq32.numel()
        G = c // d
        st = _stream()

        # (the statistics passes leave one record per strip range and fold them into mom[0]; > 1 range only for few, large images)
        msplits = L.load().mrla_light_mom_splits(b, c, h, w, dt, layout)
        L.check(min(msplits, 0), "mrla_light_mom_splits")
        mom = torch.empty((msplits, b, c, L.FWD_MOMENTS), dtype=torch.float32, device=dev)
        if cfg.fuse and (oc is None or cfg.act != L.ACT_NONE):
            raise L.MrlaHipError("the fused relu(pre + o_prev) producer needs o_prev and no activation on V")
        # inference form: nothing will be differentiated and bn_mrla does not need the batch statistics of m, so x_t is
        # neither materialised nor saved (pooling pass + an apply pass that re-forms x_t: 5N instead of 6N elements)
        if (cfg.fuse and layout == L.NHWC and cfg.bn_mode != L.BN_TRAIN and c % 64 == 0
                and cfg.infer):
            psc, psh = cfg.pre_affine if cfg.pre_affine is not None else (None, None)
            rows = L.load().mrla_bn_moment_rows(b, c, h, w, layout)
            part = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
            _call("mrla_light_pool_fused", xc.numel() * xc.element_size() * 2, _ptr(xc), _ptr(psc), _ptr(psh), _ptr(oc),
                  _ptr(part), _ptr(mom), b, c, h, w, dt, layout, st)
            gate = torch.empty((b, G), dtype=torch.float32, device=dev)
            _call("mrla_light_gate_fwd", 0, _ptr(mom), _ptr(wq32), _ptr(wk32), ks, _ptr(gate), b, c, h * w, d, st)
            bnbuf = None
            if cfg.bn_mode != L.BN_NONE:
                gamma32, beta32 = _f32(gamma), _f32(beta)
                rs = _RunningStats(running_mean, running_var, c, "mrla light forward")
                bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)
                _call("mrla_light_bn_fwd", 0, _ptr(mom), _ptr(gate), _ptr(lam32), _ptr(gamma32), _ptr(beta32),
                       _ptr(rs.rm), _ptr(rs.rv), cfg.bn_mode, float(cfg.momentum), float(cfg.eps),
                       _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(bnbuf[2]), _ptr(bnbuf[3]), b, c, h * w, d, st)
            out = torch.empty_like(xc)
            _call("mrla_light_apply_fwd_fused", xc.numel() * xc.element_size() * 3, _ptr(xc), _ptr(psc), _ptr(psh),
                  _ptr(oc), _ptr(wv32), _ptr(gate), _ptr(bnbuf[0]) if bnbuf is not None else None,
                  _ptr(bnbuf[1]) if bnbuf is not None else None, _ptr(lam32), _ptr(dp32), _ptr(out), b, c, h, w, d,
                  cfg.res, dt, layout, st, path=xc.numel() * xc.element_size() * 3)
            return out.contiguous() if via_nhwc else out
        pre = None
        # x_t = relu(bn3(pre) + o_prev) is re-formed by every pass that needs it instead of being stored (ABI 5)
        lean = bool(LEAN and cfg.fuse and not via_nhwc and L.load().mrla_light_lean_supported(b, c, h, w, dt, layout) == 1)
        nb = xc.numel() * xc.element_size()
        if _seq():
            psc = psh = None
            if cfg.fuse:
                pre, xc = xc, (None if lean else torch.empty_like(xc))
                psc, psh = cfg.pre_affine if cfg.pre_affine is not None else (None, None)
            gate = torch.empty((b, G), dtype=torch.float32, device=dev)
            bnbuf = gamma32 = beta32 = rs = None
            if cfg.bn_mode != L.BN_NONE:
                gamma32, beta32 = _f32(gamma), _f32(beta)
                rs = _RunningStats(running_mean, running_var, c, "mrla light forward")
                bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)       # sc, sh, save_mean, save_inv
            out = torch.empty_like(oc if lean else xc)
            _seq_call("mrla_light_tail_fwd", _ptr(pre if cfg.fuse else xc), _ptr(psc), _ptr(psh), _ptr(oc), _ptr(wq32),
                      _ptr(wk32), ks, _ptr(wv32), _ptr(lam32), _ptr(gamma32), _ptr(beta32),
                      _ptr(rs.rm) if rs is not None else None, _ptr(rs.rv) if rs is not None else None, cfg.bn_mode,
                      float(cfg.momentum), float(cfg.eps), _ptr(dp32), _ptr(mom), _ptr(xc) if (cfg.fuse and not lean) else None,
                      _ptr(gate), _ptr(bnbuf), _ptr(out), b, c, h, w, d, cfg.res, (2 if lean else int(cfg.fuse)), dt, layout,
                      cfg.act, st)
            if rs is not None:
                rs.finish(cfg.bn_mode == L.BN_TRAIN)
        else:
            if cfg.fuse:
                pre, xc = xc, (None if lean else torch.empty_like(xc))
                psc, psh = cfg.pre_affine if cfg.pre_affine is not None else (None, None)
                _call("mrla_light_stats_fwd_fused", nb * (2 if lean else 3), _ptr(pre), _ptr(psc), _ptr(psh),
                      _ptr(oc), _ptr(wv32), _ptr(mom), _ptr(xc), b, c, h, w, dt, layout, st)
            else:
                _call("mrla_light_stats_fwd", xc.numel() * xc.element_size() * (2 if oc is not None else 1), _ptr(xc),
                      _ptr(oc), _ptr(wv32), _ptr(mom), b, c, h, w, dt, layout, cfg.act, st)
            gate = torch.empty((b, G), dtype=torch.float32, device=dev)
            _call("mrla_light_gate_fwd", 0, _ptr(mom), _ptr(wq32), _ptr(wk32), ks, _ptr(gate), b, c, h * w, d, st)
            bnbuf = gamma32 = None
            if cfg.bn_mode != L.BN_NONE:
                gamma32, beta32 = _f32(gamma), _f32(beta)
                rs = _RunningStats(running_mean, running_var, c, "mrla light forward")
                bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)       # sc, sh, save_mean, save_inv
                _call("mrla_light_bn_fwd", 0, _ptr(mom), _ptr(gate), _ptr(lam32), _ptr(gamma32), _ptr(beta32),
                       _ptr(rs.rm), _ptr(rs.rv), cfg.bn_mode, float(cfg.momentum), float(cfg.eps),
                       _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(bnbuf[2]), _ptr(bnbuf[3]), b, c, h * w, d, st)
                rs.finish(cfg.bn_mode == L.BN_TRAIN)
            if lean:
                out = torch.empty_like(oc)
                _call("mrla_light_apply_fwd_fused", nb * 3, _ptr(pre), _ptr(psc), _ptr(psh), _ptr(oc), _ptr(wv32), _ptr(gate),
                      _ptr(bnbuf[0]) if bnbuf is not None else None, _ptr(bnbuf[1]) if bnbuf is not None else None,
                      _ptr(lam32), _ptr(dp32), _ptr(out), b, c, h, w, d, cfg.res, dt, layout, st, path=nb * 3)
            else:
                out = torch.empty_like(xc)
                _call("mrla_light_apply_fwd", nb * (3 if oc is not None else 2), _ptr(xc), _ptr(oc), _ptr(wv32), _ptr(gate),
                      _ptr(bnbuf[0]) if bnbuf is not None else None, _ptr(bnbuf[1]) if bnbuf is not None else None,
                      _ptr(lam32), _ptr(dp32), _ptr(out), b, c, h, w, d, cfg.res, dt, layout, cfg.act, st,
                      path=nb * (3 if oc is not None else 2))

        ctx.cfg, ctx.layout, ctx.ks = cfg, layout, ks
        ctx.shapes = (wq.shape, wk.shape, wv.shape, lam.shape if lam is not None else None)
        ctx.wv_stride = wv.stride()          # gradient layout contract of DDP: same strides as the parameter
        ctx.pdtypes = (wq.dtype, wk.dtype, wv.dtype, lam.dtype if lam is not None else None,
                       gamma.dtype if gamma is not None else None)
        # conv3's raw output, when the BatchNorm behind it was deferred and its backward sums can ride in apply_bwd
        keep_pre = (cfg.fuse and cfg.pre_box is not None and not via_nhwc
                    and L.load().mrla_light_apply_bwd_pre_sums(b, c, h, w, dt, layout) == 1)
        ctx.lean, ctx.pre_sums = lean, keep_pre
        psc, psh = cfg.pre_affine if (lean and cfg.pre_affine is not None) else (None, None)
        ctx.save_for_backward(xc, oc, wq32, wk32, wv32, lam32, gamma32, dp32, mom, gate, bnbuf,
                              pre if (keep_pre or lean) else None, psc, psh)
        return out.contiguous() if via_nhwc else out

    @staticmethod
    @_on_device
    def backward(ctx, dout):
        xc, oc, wq32, wk32, wv32, lam32, gamma32, dp32, mom, gate, bnbuf, pre, psc, psh = ctx.saved_tensors
        cfg, layout, ks, lean = ctx.cfg, ctx.layout, ctx.ks, ctx.lean
        like = oc if lean else xc                    # (lean: x_t was never stored; it is re-formed from pre, psc, psh, oc)
        b, c, h, w = like.shape
        d = cfg.d
        dt = _DT[like.dtype]
        dev = like.device
        st = _stream()
        nb = like.numel() * like.element_size()
        if dout.dtype != like.dtype:
            dout = dout.to(like.dtype)
        dout = _layout_of(dout, layout)[1]

        has_bn = cfg.bn_mode != L.BN_NONE
        # (partial records of the backward statistics pass: > 1 only for few, large images -- detection batches)
        splits = L.load().mrla_light_bmom_splits(b, c, h, w, dt, layout)
        L.check(min(splits, 0), "mrla_light_bmom_splits")
        if _seq():
            bmom = torch.empty((splits, b, c, L.BWD_MOMENTS), dtype=torch.float32, device=dev)
            small = torch.empty((11, c), dtype=torch.float32, device=dev)     # cb[c,4] | dgamma | dbeta | dlam | cb_lo[c,4]
            dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
            dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
            rows = L.load().mrla_light_wgrad_rows(b, c, h, w, dt, layout)
            L.check(min(rows, 0), "mrla_light_wgrad_rows")
            dwv_part = torch.empty((rows, c * 9), dtype=torch.float32, device=dev)
            dx = torch.empty_like(like)
            do = torch.empty_like(oc) if oc is not None else None
            pre_tmom = None
            if ctx.pre_sums and pre is not None:
                pre_tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
            elif not lean:
                pre = None
            wsum = torch.empty((c * 9 + 2 * ks,), dtype=torch.float32, device=dev)
            _seq_call("mrla_light_tail_bwd", _ptr(dout), _ptr(xc), _ptr(oc), _ptr(wq32), _ptr(wk32), ks, _ptr(wv32),
                      _ptr(lam32), _ptr(gamma32) if has_bn else None, _ptr(dp32), _ptr(mom), _ptr(gate),
                      _ptr(bnbuf) if has_bn else None, cfg.bn_mode, _ptr(bmom), _ptr(small), _ptr(dyx), _ptr(dwqk_part),
                      _ptr(dwv_part), rows, _ptr(dx), _ptr(do), _ptr(pre), _ptr(psc), _ptr(psh),
                      _ptr(cfg.pre_box.center) if pre_tmom is not None else None, _ptr(pre_tmom), _ptr(wsum), b, c, h, w, d,
                      cfg.res, int(cfg.fuse), dt, layout, cfg.act, st)
            if pre_tmom is not None:
                cfg.pre_box.put(dx, pre_tmom, rows)
        else:
            bmom = torch.empty((splits, b, c, L.BWD_MOMENTS), dtype=torch.float32, device=dev)
            if lean:
                _call("mrla_light_stats_bwd_fused", nb * 3, _ptr(dout), _ptr(pre), _ptr(psc), _ptr(psh), _ptr(oc), _ptr(wv32),
                      _ptr(mom), _ptr(bmom), b, c, h, w, dt, layout, st)
            else:
                _call("mrla_light_stats_bwd", nb * (3 if oc is not None else 2), _ptr(dout), _ptr(xc), _ptr(oc), _ptr(wv32),
                      _ptr(mom), _ptr(bmom), b, c, h, w, dt, layout, cfg.act, st)
            small = torch.empty((11, c), dtype=torch.float32, device=dev)     # cb[c,4] | dgamma | dbeta | dlam | cb_lo[c,4]
            cb, cb_lo = small[:4].view(c, 4), small[7:].view(c, 4)
            has_bn = cfg.bn_mode != L.BN_NONE
            _call("mrla_light_bn_bwd", 0, _ptr(mom), _ptr(bmom), _ptr(gate), _ptr(lam32), _ptr(gamma32) if has_bn else None,
                   _ptr(dp32), _ptr(bnbuf[2]) if has_bn else None, _ptr(bnbuf[3]) if has_bn else None, cfg.bn_mode,
                   _ptr(cb), _ptr(cb_lo), _ptr(small[4]) if has_bn else None, _ptr(small[5]) if has_bn else None,
                   _ptr(small[6]) if lam32 is not None else None, b, c, h * w, d, st)
            dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
            dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
            _call("mrla_light_gate_bwd", 0, _ptr(mom), _ptr(bmom), _ptr(gate), _ptr(cb), _ptr(cb_lo), _ptr(dp32), _ptr(wq32),
                   _ptr(wk32), ks, _ptr(dyx), _ptr(dwqk_part), b, c, h * w, d, st)
            rows = L.load().mrla_light_wgrad_rows(b, c, h, w, dt, layout)
            L.check(min(rows, 0), "mrla_light_wgrad_rows")
            dwv_part = torch.empty((rows, c * 9), dtype=torch.float32, device=dev)
            dx = torch.empty_like(like)
            do = torch.empty_like(oc) if oc is not None else None
            # the deferred bn3's backward sums (sum dpre, sum dpre*y3) ride in this pass: one more row fetch, no 2N pass
            pre_tmom = None
            if ctx.pre_sums and pre is not None:
                pre_tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
            elif not lean:
                pre = None
            if lean:
                # dOut, y3, o_prev in; dx, do out: section 8(d)'s five passes exactly (x_t is re-formed, bn3's sums ride along)
                _call("mrla_light_apply_bwd", nb * 5, _ptr(dout), _ptr(pre), _ptr(psc), _ptr(psh), _ptr(oc), _ptr(wv32),
                      _ptr(gate), _ptr(cb), _ptr(lam32), _ptr(dp32), _ptr(dyx), _ptr(dx), _ptr(do), _ptr(dwv_part),
                      _ptr(cfg.pre_box.center) if pre_tmom is not None else None, _ptr(pre_tmom), b, c, h, w, d, cfg.res, dt,
                      layout, st, entry="mrla_light_apply_bwd_fused", alg=nb * 5, path=nb * 5)
            else:
                _call("mrla_light_apply_bwd", nb * ((5 if oc is not None else 3) + (pre is not None)),
                      _ptr(dout), _ptr(xc), _ptr(oc), _ptr(wv32), _ptr(gate), _ptr(cb), _ptr(lam32),
                      _ptr(dp32), _ptr(dyx), _ptr(dx), _ptr(do), _ptr(dwv_part), _ptr(pre),
                      _ptr(cfg.pre_box.center) if pre is not None else None, _ptr(pre_tmom), b, c, h, w, d, cfg.res,
                      int(cfg.fuse), dt, layout, cfg.act, st,
                      # section 8(d): dOut, x_t, o_prev in; dx, do out (the y3 row read for bn3's folded sums is not in that count)
                      alg=nb * (5 if oc is not None else 3), path=nb * (5 if oc is not None else 3))
            if pre_tmom is not None:
                cfg.pre_box.put(dx, pre_tmom, rows)
            wsum = torch.empty((c * 9 + 2 * ks,), dtype=torch.float32, device=dev)
            _call("mrla_reduce_rows2", 0, _ptr(dwv_part), _ptr(wsum), rows, c * 9, _ptr(dwqk_part), _ptr(wsum[c * 9:]), b, 2 * ks, st)

        sq, sk, sv, sl = ctx.shapes
        tq, tk, tv, tl, tg = ctx.pdtypes
        dwv = _grad_like(wsum[:c * 9].to(tv), sv, ctx.wv_stride)
        dwq = wsum[c * 9:c * 9 + ks].view(sq).to(tq)
        dwk = wsum[c * 9 + ks:].view(sk).to(tk)
        dlam = small[6].view(sl).to(tl) if lam32 is not None else None
        dgamma = small[4].to(tg) if has_bn else None
        dbeta = small[5].to(tg) if has_bn else None
        return dx, do, dwq, dwk, dwv, dlam, dgamma, dbeta, None, None, None, None


def mrla_light(x, wq, wk, wv, d, o_prev=None, lam=None, bn=None, dp=None, res=False, act_gelu=False,
               pre_activation=False):
    """Functional entry point.

    bn: None or dict(weight, bias, running_mean, running_var, training, momentum, eps).
    dp: per-sample drop-path multiplier [b] (mask / keep_prob) or None.
    pre_activation: `x` is the block's pre-activation; x_t = relu(x + o_prev) is formed inside the kernels.  When `x`
    came out of `bn_act(..., defer=True)` its BatchNorm affine is applied there too (x then aliases the conv output).
    """
    pre_affine, pre_box = getattr(x, "_mrla_affine", None), getattr(x, "_mrla_bn_box", None)
    if pre_affine is not None and not pre_activation:
        raise L.MrlaHipError("a deferred BatchNorm output can only feed the fused producer (pre_activation=True)")
    tensors = [t for t in (x, o_prev, wq, wk, wv, lam) if t is not None]
    if bn is not None:
        tensors += [bn["weight"], bn["bias"]]
    infer = not (torch.is_grad_enabled() and any(t.requires_grad for t in tensors))
    if bn is None:
        cfg = LightConfig(d, L.BN_NONE, res=int(res), act=L.ACT_GELU if act_gelu else L.ACT_NONE, fuse=pre_activation,
                          pre_affine=pre_affine, pre_box=pre_box)
        cfg.infer = infer
        return _LightFn.apply(x, o_prev, wq, wk, wv, lam, None, None, None, None, dp, cfg)
    cfg = LightConfig(d, L.BN_TRAIN if bn["training"] else L.BN_EVAL, bn.get("momentum", 0.1), bn.get("eps", 1e-5),
                      int(res), L.ACT_GELU if act_gelu else L.ACT_NONE, fuse=pre_activation, pre_affine=pre_affine,
                      pre_box=pre_box)
    cfg.infer = infer
    return _LightFn.apply(x, o_prev, wq, wk, wv, lam, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"],
                          dp, cfg)


# ======================================================================================================
# MRLA-base
# ======================================================================================================
class BaseStage:
    """Stage-resident K/V history of MRLA-base on the device (replaces the reference's torch.cat growth).

    One instance lives for one forward(+backward) pass of one network stage: the `init_cell` layer creates
    it, every later layer of the stage appends one slot.  Gradients with respect to the history do not
    travel along autograd edges; they are accumulated in the dA / dK rings, which is valid because layer
    t+1's backward always runs before layer t's (x_{t+1} depends on out_t through the block chain).
    """

    def __init__(self, b, c, h, w, d, dtype, device, capacity, layout=L.NCHW):
        self.b, self.c, self.h, self.w, self.d = b, c, h, w, d
        self.dtype, self.device, self.layout = dtype, device, layout
        self.T = max(1, int(capacity))
        self.t = 0
        self.bwd_started = False
        self.bwd_task = -1        # autograd graph-task id of the backward pass the dA / dK rings currently belong to
        self.bwd_last_t = 0       # layer index of the previous backward call (calls of one pass come in decreasing t)
        self.bwd_top = 0          # deepest layer that took part in the current backward pass
        self.V = torch.empty(self._vshape(self.T), dtype=dtype, device=device)
        self.K = torch.empty((b, self.T, c), dtype=torch.float32, device=device)
        self.P = torch.empty((b, c // d, self.T, self.T), dtype=torch.float32, device=device)
        self.dA = self.dK = None

    @staticmethod
    def layout_for(x, d):
        """MRLA_NHWC when `x` is a channels_last tensor the slot-major NHWC kernels handle, else MRLA_NCHW."""
        if x.dim() == 4 and x.dtype in _DT:
            b, c, h, w = x.shape
            lib = L.load()
            nhwc_ok = lib.mrla_base_tile_rows(b, c, h, w, _DT[x.dtype], L.NHWC) > 0
            if not x.is_contiguous() and x.is_contiguous(memory_format=_CL) and nhwc_ok:
                return L.NHWC
            # channel-contiguous views that are not dense (DeiT's map tokens x[:, 1:] seen as [b, c, 14, 14]): one dense
            # NHWC copy is cheaper than the transposing copy the NCHW kernels would need
            if nhwc_ok and c > 1 and x.stride(1) == 1 and not x.is_contiguous():
                return L.NHWC
            # NCHW tensors the slab kernels cannot take (plane rows wider than a wave, slabs beyond the LDS) go through one
            # internal channels_last conversion per layer instead of failing (the reference accepts any map size)
            if nhwc_ok and lib.mrla_light_wgrad_rows(b, c, h, w, _DT[x.dtype], L.NCHW) == L.EUNSUPPORTED:
                return L.NHWC
        return L.NCHW

    def _vshape(self, T):
        """NCHW: [b, T, c, h, w] (the reference's V layout).  NHWC: slot-major [T, b, h, w, c]."""
        if self.layout == L.NHWC:
            return (T, self.b, self.h, self.w, self.c)
        return (self.b, T, self.c, self.h, self.w)

    def slot(self, ring, j):
        return ring[j] if self.layout == L.NHWC else ring[:, j]

    def reserve_slot(self):
        """Make room for one more layer (amortised doubling when the capacity hint was too small)."""
        if self.t < self.T:
            return
        if self.bwd_started:
            raise L.MrlaHipError("MRLA-base history grown after its backward pass started")
        T2, t = 2 * self.T, self.t
        V = torch.empty(self._vshape(T2), dtype=self.dtype, device=self.device)
        K = torch.empty((self.b, T2, self.c), dtype=torch.float32, device=self.device)
        P = torch.empty((self.b, self.c // self.d, T2, T2), dtype=torch.float32, device=self.device)
        if self.layout == L.NHWC:
            V[:t].copy_(self.V[:t])
        else:
            V[:, :t].copy_(self.V[:, :t])
        K[:, :t].copy_(self.K[:, :t])
        P[:, :, :t, :t].copy_(self.P[:, :, :t, :t])
        self.V, self.K, self.P, self.T = V, K, P, T2

    def begin_layer_backward(self, t):
        """Called by layer t's backward.  Returns True on the first call of a backward PASS: then the dK ring is re-zeroed
        and `bwd_top` (the deepest layer whose dA slot this pass fills) is reset.  A pass is identified by the autograd
        engine's graph-task id (a partial pass -- autograd.grad down to layer 3 -- followed by one whose deepest layer is
        shallower is a new pass although t decreased); outside an engine-driven backward (id -1) a pass is recognised
        by t not decreasing: a repeated backward over the same graph starts again at the top."""
        if self.dA is None:
            self.dA = torch.empty_like(self.V)
            self.dK = torch.empty_like(self.K)
        task = _graph_task_id()
        if task != -1:
            first = (not self.bwd_started) or task != self.bwd_task
        else:
            first = (not self.bwd_started) or t >= self.bwd_last_t
        self.bwd_started = True
        self.bwd_task = task
        self.bwd_last_t = t
        if first:
            self.bwd_top = t
        return first

    def views(self):
        """(K[b,t,c], V[b,t,c,h,w]) as the reference returns them -- views of the rings."""
        K = self.K[:, :self.t]
        V = self.V[:self.t].permute(1, 0, 4, 2, 3) if self.layout == L.NHWC else self.V[:, :self.t]
        K._mrla_stage = V._mrla_stage = self
        return K, V


class BaseConfig:
    __slots__ = ("d", "bn_mode", "momentum", "eps", "tail", "fuse", "pre_affine", "pre_box")

    def __init__(self, d, bn_mode=L.BN_NONE, momentum=0.1, eps=1e-5, tail=False, fuse=False, pre_affine=None, pre_box=None):
        self.d, self.bn_mode, self.momentum, self.eps, self.tail, self.fuse = d, bn_mode, momentum, eps, tail, fuse
        self.pre_affine = pre_affine     # deferred bn3 affine (NHWC stages only), see bn_act(defer=True)
        self.pre_box = pre_box           # its _DeferredBnBox: where the value backward leaves that BatchNorm's gradient sums


class _BaseFn(torch.autograd.Function):
    """tail:  out = x + dp[b]*relu(BN(attn)),  attn = sum_{j<=t} softmax_j(<q_t,k_j>/sqrt(d)) * v_j
    no tail: out = attn  (bare mrla_base_layer)."""

    @staticmethod
    @_on_device
    def forward(ctx, x, identity, wq, wk, wv, gamma, beta, running_mean, running_var, dp, stage, cfg):
        _require_cuda(x, "mrla base forward")
        layout, xc = _layout_of(x, stage.layout)    # the stage's rings fix the layout (NHWC rings are slot-major)
        b, c, h, w = xc.shape
        d = cfg.d
        if (b, c, h, w, d) != (stage.b, stage.c, stage.h, stage.w, stage.d) or xc.dtype != stage.dtype:
            raise L.MrlaHipError("input does not match the stage's K/V history (shape/dtype); is init_cell set on the "
                                 "first block of the stage?")
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        wq32, wk32 = _f32(wq).reshape(-1), _f32(wk).reshape(-1)
        wv32 = _f32(wv).reshape(c, 9)
        dp32 = _f32(dp).reshape(-1) if dp is not None else None
        ks = wq32.numel()
        stage.reserve_slot()
        t, T = stage.t + 1, stage.T

        # ([splits, b, c, 8]: mrla_light_stats_fwd* -- the NCHW path's pooling pass below -- may leave a record per strip range)
        mom = torch.empty((max(1, L.load().mrla_light_mom_splits(b, c, h, w, dt, layout)), b, c, L.FWD_MOMENTS), dtype=torch.float32,
                          device=dev)
        nhwc = layout == L.NHWC
        es = xc.element_size()
        if nhwc and _seq():     # the whole layer + tail: one call (mrla_base_layer_fwd)
            idc = pre = None
            if cfg.fuse:
                idc = _layout_of(identity, layout)[1]
                pre, xc = xc, torch.empty_like(xc)
            psc, psh = cfg.pre_affine if cfg.pre_affine is not None else (None, None)
            q = torch.empty((b, c), dtype=torch.float32, device=dev)
            attn = torch.empty_like(xc)
            arows = L.load().mrla_base_tile_rows(b, c, h, w, dt, layout)
            L.check(min(arows, 0), "mrla_base_tile_rows")
            amom = torch.empty((arows, c, 2), dtype=torch.float32, device=dev)
            bnbuf = gamma32 = beta32 = rs = None
            out = attn
            if cfg.tail:
                gamma32, beta32 = _f32(gamma), _f32(beta)
                rs = _RunningStats(running_mean, running_var, c, "mrla base forward")
                bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)
                out = torch.empty_like(xc)
            _seq_call("mrla_base_layer_fwd", _ptr(pre if cfg.fuse else xc), _ptr(psc), _ptr(psh), _ptr(idc), _ptr(wq32),
                      _ptr(wk32), ks, _ptr(wv32), _ptr(gamma32), _ptr(beta32), _ptr(rs.rm) if rs is not None else None,
                      _ptr(rs.rv) if rs is not None else None, cfg.bn_mode if cfg.tail else L.BN_NONE, float(cfg.momentum),
                      float(cfg.eps), _ptr(dp32), _ptr(mom), _ptr(xc) if cfg.fuse else None, _ptr(stage.V), _ptr(stage.K),
                      _ptr(stage.P), _ptr(q), _ptr(attn), _ptr(amom), arows, _ptr(bnbuf), _ptr(out) if cfg.tail else None,
                      int(cfg.tail), b, c, h, w, d, T, t, dt, st)
            if rs is not None:
                rs.finish(cfg.bn_mode == L.BN_TRAIN)
            stage.t = t
        else:
            if nhwc:                # one pass: pooling moments, V_t -> ring slot (and x_t = relu(x + identity) when fused)
                idc = pre = None
                if cfg.fuse:
                    idc = _layout_of(identity, layout)[1]
                    pre, xc = xc, torch.empty_like(xc)
                psc, psh = cfg.pre_affine if cfg.pre_affine is not None else (None, None)
                _call("mrla_base_pool_value_fwd", xc.numel() * es * (4 if cfg.fuse else 2), _ptr(pre if cfg.fuse else xc),
                      _ptr(psc), _ptr(psh), _ptr(idc), _ptr(wv32), _ptr(mom), _ptr(xc) if cfg.fuse else None,
                      _ptr(stage.V[t - 1]), b, c, h, w, dt, layout, st, path=xc.numel() * es * 2)
            elif cfg.fuse:          # x is the pre-activation: x_t = relu(x + identity) formed by the pooling pass
                idc = _layout_of(identity, L.NCHW)[1]
                pre, xc = xc, torch.empty_like(xc)
                _call("mrla_light_stats_fwd_fused", xc.numel() * xc.element_size() * 3, _ptr(pre), None, None, _ptr(idc),
                      _ptr(wv32), _ptr(mom), _ptr(xc), b, c, h, w, dt, layout, st)
            else:
                _call("mrla_light_stats_fwd", xc.numel() * xc.element_size(), _ptr(xc), None, _ptr(wv32), _ptr(mom), b, c, h,
                      w, dt, layout, L.ACT_NONE, st)
            q = torch.empty((b, c), dtype=torch.float32, device=dev)
            _call("mrla_base_gate_fwd", 0, _ptr(mom), _ptr(wq32), _ptr(wk32), ks, _ptr(stage.K), _ptr(stage.P), _ptr(q), b, c,
                   h * w, d, T, t, st)
            attn = torch.empty_like(xc)
            arows = L.load().mrla_base_tile_rows(b, c, h, w, dt, layout)      # rows of the (sum, sum^2) partials
            L.check(min(arows, 0), "mrla_base_tile_rows")
            amom = torch.empty((arows, c, 2), dtype=torch.float32, device=dev)
            _call("mrla_base_attend_fwd", xc.numel() * es * ((t + 1) if nhwc else (t + 2)), None if nhwc else _ptr(xc),
                  _ptr(wv32), _ptr(stage.V), _ptr(stage.P), _ptr(attn), _ptr(amom), b, c, h, w, d, T, t, dt, layout, st,
                  path=xc.numel() * es * (t - 1 if nhwc else t + 1))      # section 8(d): (t + 2) N per layer with the value pass + tail
            stage.t = t
            bnbuf = gamma32 = None
            out = attn
            if cfg.tail:
                gamma32, beta32 = _f32(gamma), _f32(beta)
                rs = _RunningStats(running_mean, running_var, c, "mrla base forward")
                bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)
                _call("mrla_bn_stats_fwd", 0, _ptr(amom), None, _ptr(gamma32), _ptr(beta32), _ptr(rs.rm), _ptr(rs.rv),
                       cfg.bn_mode, float(cfg.momentum), float(cfg.eps), _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(bnbuf[2]),
                       _ptr(bnbuf[3]), arows, c, b * h * w // arows, st)
                rs.finish(cfg.bn_mode == L.BN_TRAIN)
                out = torch.empty_like(xc)
                _call("mrla_base_tail_fwd", xc.numel() * xc.element_size() * 3, _ptr(xc), _ptr(attn), _ptr(bnbuf[0]),
                      _ptr(bnbuf[1]), _ptr(dp32), _ptr(out), b, c, h, w, dt, layout, st, path=xc.numel() * xc.element_size())
        ctx.cfg, ctx.layout, ctx.ks, ctx.stage, ctx.t = cfg, layout, ks, stage, t
        ctx.shapes = (wq.shape, wk.shape, wv.shape)
        ctx.wv_stride = wv.stride()
        ctx.pdtypes = (wq.dtype, wk.dtype, wv.dtype, gamma.dtype if gamma is not None else None)
        # conv3's raw output, when the BatchNorm behind it was deferred and its backward sums can ride in the value backward
        keep_pre = (nhwc and cfg.fuse and cfg.pre_box is not None
                    and L.load().mrla_base_value_bwd_pre_sums(b, c, h, w, dt, layout) == 1)
        ctx.save_for_backward(xc, attn if cfg.tail else None, wq32, wk32, wv32, gamma32, dp32, mom, q, bnbuf,
                              pre if keep_pre else None)
        if nhwc and x.is_contiguous() and not x.is_contiguous(memory_format=_CL):
            return out.contiguous()          # an NCHW caller of an NHWC stage gets NCHW-contiguous memory back
        return out

    @staticmethod
    @_on_device
    def backward(ctx, dout):
        xc, attn, wq32, wk32, wv32, gamma32, dp32, mom, q, bnbuf, pre = ctx.saved_tensors
        cfg, layout, ks, stage, t = ctx.cfg, ctx.layout, ctx.ks, ctx.stage, ctx.t
        b, c, h, w = xc.shape
        d, T = cfg.d, stage.T
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        if dout.dtype != xc.dtype:
            dout = dout.to(xc.dtype)
        dout = _layout_of(dout, layout)[1]
        nhwc = layout == L.NHWC
        first = stage.begin_layer_backward(t)
        Tc = stage.bwd_top            # later layers that received no gradient in this pass left no dA slot behind
        es = xc.element_size()

        if nhwc and _seq():     # the whole layer + tail backward: one call (mrla_base_layer_bwd)
            dgamma = dbeta = small = tmom = None
            trows = 1
            if cfg.tail:
                trows = L.load().mrla_bn_moment_rows(b, c, h, w, layout)
                tmom = torch.empty((trows, c, 2), dtype=torch.float32, device=dev)
                small = torch.empty((5, c), dtype=torch.float32, device=dev)         # cb[c,3] | dgamma | dbeta
            pmom = torch.empty((b, c, t), dtype=torch.float32, device=dev)
            prows = L.load().mrla_base_pmom_rows(b, c, h, w, dt, layout)
            L.check(min(prows, 0), "mrla_base_pmom_rows")
            ppart = torch.empty((prows, t, c), dtype=torch.float32, device=dev)
            dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
            dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
            rows = L.load().mrla_light_wgrad_rows(b, c, h, w, dt, layout)
            L.check(min(rows, 0), "mrla_light_wgrad_rows")
            dwv_part = torch.empty((rows, c * 9), dtype=torch.float32, device=dev)
            dx = torch.empty_like(xc)
            dv = torch.empty((b, h, w, c), dtype=xc.dtype, device=dev)
            res = int(cfg.tail) | (2 if cfg.fuse else 0)
            pre_tmom = None
            if pre is not None:
                pre_tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
            else:
                pre = None
            wsum = torch.empty((c * 9 + 2 * ks,), dtype=torch.float32, device=dev)
            _seq_call("mrla_base_layer_bwd", _ptr(dout), _ptr(xc), _ptr(attn), _ptr(wq32), _ptr(wk32), ks, _ptr(wv32),
                      _ptr(gamma32), _ptr(dp32), _ptr(mom), _ptr(q), _ptr(bnbuf), cfg.bn_mode if cfg.tail else L.BN_NONE,
                      _ptr(stage.V), _ptr(stage.dA), _ptr(stage.K), _ptr(stage.dK), _ptr(stage.P), _ptr(tmom), trows,
                      _ptr(small), _ptr(ppart), prows, _ptr(pmom), _ptr(dyx), _ptr(dwqk_part), _ptr(dv), _ptr(dwv_part), rows,
                      _ptr(dx), _ptr(pre), _ptr(cfg.pre_box.center) if pre is not None else None, _ptr(pre_tmom), _ptr(wsum),
                      int(cfg.tail), int(first), res, b, c, h, w, d, T, t, Tc, dt, st)
            if pre_tmom is not None:
                cfg.pre_box.put(dx, pre_tmom, rows)
            if cfg.tail:
                dgamma, dbeta = small[3].to(ctx.pdtypes[3]), small[4].to(ctx.pdtypes[3])
        else:
            cb = dgamma = dbeta = None
            if cfg.tail:
                trows = L.load().mrla_bn_moment_rows(b, c, h, w, layout) if nhwc else b
                tmom = torch.empty((trows, c, 2), dtype=torch.float32, device=dev)
                center = bnbuf[2] if nhwc else None      # sums about the saved batch mean (the NCHW slab kernel keeps raw sums)
                _call("mrla_base_tail_stats_bwd", xc.numel() * es * 2, _ptr(dout), _ptr(attn), _ptr(bnbuf[0]), _ptr(bnbuf[1]),
                      _ptr(center), _ptr(dp32), _ptr(tmom), b, c, h, w, dt, layout, st)
                small = torch.empty((5, c), dtype=torch.float32, device=dev)         # cb[c,3] | dgamma | dbeta
                cb = small[:3].view(c, 3)
                _call("mrla_bn_stats_bwd", 0, _ptr(tmom), _ptr(gamma32), _ptr(bnbuf[2]), _ptr(bnbuf[3]), cfg.bn_mode,
                       int(center is not None), _ptr(cb), _ptr(small[3]), _ptr(small[4]), trows, c, b * h * w // trows, st)
                dgamma, dbeta = small[3].to(ctx.pdtypes[3]), small[4].to(ctx.pdtypes[3])
            pmom = torch.empty((b, c, t), dtype=torch.float32, device=dev)
            prows = L.load().mrla_base_pmom_rows(b, c, h, w, dt, layout)
            ppart = torch.empty((prows, t, c), dtype=torch.float32, device=dev) if nhwc else pmom
            # (t + 3) N of streams + the fp32 partial rows of <dA_t, V_j> (one row of t x c floats per 16-pixel tile: 2 / 16 of the
            # V bytes it reads -- the "9 %" the PMC pass sees above (t + 3) N; mrla_base_pmom_reduce reads them back)
            _call("mrla_base_attend_bwd", xc.numel() * es * (t + 3) + (ppart.numel() * 4 if nhwc else 0), _ptr(dout), _ptr(attn),
                  _ptr(bnbuf[0]) if cfg.tail else None, _ptr(bnbuf[1]) if cfg.tail else None, _ptr(dp32), _ptr(cb),
                  _ptr(stage.V), _ptr(stage.dA), _ptr(ppart), b, c, h, w, T, t, dt, layout, st,
                  alg=xc.numel() * es * (t + 3),
                  path=xc.numel() * es * (t + 1))                         # section 8(d): ~(2t + 3) N per layer, backward
            if nhwc:
                _call("mrla_base_pmom_reduce", (ppart.numel() + pmom.numel()) * 4, _ptr(ppart), _ptr(pmom), b, c, t, prows, st)
            dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
            dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
            _call("mrla_base_gate_bwd", 0, _ptr(mom), _ptr(pmom), _ptr(stage.P), _ptr(q), _ptr(stage.K), _ptr(stage.dK),
                   _ptr(wq32), _ptr(wk32), ks, _ptr(dyx), _ptr(dwqk_part), b, c, h * w, d, T, t, int(first), st)
            rows = L.load().mrla_light_wgrad_rows(b, c, h, w, dt, layout)
            L.check(min(rows, 0), "mrla_light_wgrad_rows")
            dwv_part = torch.empty((rows, c * 9), dtype=torch.float32, device=dev)
            dx = torch.empty_like(xc)
            res = int(cfg.tail) | (2 if cfg.fuse else 0)
            if nhwc:                # dV_t from the dA slots t..Tc, then the transposed 3x3 pass
                dv = torch.empty((b, h, w, c), dtype=xc.dtype, device=dev)
                _call("mrla_base_dv_combine", xc.numel() * es * (Tc - t + 2), _ptr(stage.dA), _ptr(stage.P), _ptr(dv),
                      b, c, h, w, d, T, t, Tc, dt, layout, st, path=xc.numel() * es * (Tc - t + 1))
                # the deferred bn3's backward sums (sum dpre, sum dpre*(y3 - mean)) ride in this pass: one more row fetch, no 2N pass
                pre_tmom = None
                if pre is not None:
                    pre_tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
                else:
                    pre = None
                _call("mrla_base_value_bwd_dv", xc.numel() * es * (4 + (pre is not None)), _ptr(dout), _ptr(xc), _ptr(wv32),
                      _ptr(dv), _ptr(dyx), _ptr(dx), _ptr(dwv_part), _ptr(pre),
                      _ptr(cfg.pre_box.center) if pre is not None else None, _ptr(pre_tmom), b, c, h, w, res, dt, layout, st,
                      alg=xc.numel() * es * 4, path=xc.numel() * es)
                if pre_tmom is not None:
                    cfg.pre_box.put(dx, pre_tmom, rows)
            else:
                _call("mrla_base_value_bwd", xc.numel() * es * (Tc - t + 4), _ptr(dout), _ptr(xc), _ptr(wv32),
                      _ptr(stage.dA), _ptr(stage.P), _ptr(dyx), _ptr(dx), _ptr(dwv_part), b, c, h, w, d, T, t, Tc, res, dt,
                      layout, st, path=xc.numel() * es * (Tc - t + 2))
            wsum = torch.empty((c * 9 + 2 * ks,), dtype=torch.float32, device=dev)
            _call("mrla_reduce_rows2", 0, _ptr(dwv_part), _ptr(wsum), rows, c * 9, _ptr(dwqk_part), _ptr(wsum[c * 9:]), b, 2 * ks, st)
        sq, sk, sv = ctx.shapes
        tq, tk, tv, _ = ctx.pdtypes
        return (dx, dx if cfg.fuse else None, wsum[c * 9:c * 9 + ks].view(sq).to(tq), wsum[c * 9 + ks:].view(sk).to(tk),
                _grad_like(wsum[:c * 9].to(tv), sv, ctx.wv_stride), dgamma, dbeta, None, None, None, None, None)


def mrla_base(x, wq, wk, wv, d, stage, bn=None, dp=None, identity=None):
    """One MRLA-base layer on `stage` (a BaseStage).  bn: None (bare layer, returns attn) or the dict of
    mrla_light(); with bn the block tail x + dp*relu(BN(attn)) is fused in.  identity: when given, `x` is the
    bottleneck's pre-activation and x_t = relu(x + identity) is formed inside the pooling pass."""
    fuse = identity is not None
    pre_affine, pre_box = getattr(x, "_mrla_affine", None), getattr(x, "_mrla_bn_box", None)
    if pre_affine is not None and not (fuse and stage.layout == L.NHWC):
        raise L.MrlaHipError("a deferred BatchNorm output can only feed the fused producer of an NHWC MRLA-base stage")
    if bn is None:
        return _BaseFn.apply(x, identity, wq, wk, wv, None, None, None, None, None, stage,
                             BaseConfig(d, fuse=fuse, pre_affine=pre_affine, pre_box=pre_box))
    cfg = BaseConfig(d, L.BN_TRAIN if bn["training"] else L.BN_EVAL, bn.get("momentum", 0.1), bn.get("eps", 1e-5), True,
                     fuse, pre_affine, pre_box)
    return _BaseFn.apply(x, identity, wq, wk, wv, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"], dp,
                         stage, cfg)


# ======================================================================================================
# MRLA-light on token sequences (DeiT)
# ======================================================================================================
class _TokenLightFn(torch.autograd.Function):
    """out = res*x + cat(LN_x(x)[:, :1], a*gelu(dwconv3x3(LN_x(x) map)) + lam*LN_o(o_prev)[:, 1:])
    (deit_mrla_light.py:194-209 and the block residual :234)."""

    @staticmethod
    @_on_device
    def forward(ctx, x, o_prev, lnx_w, lnx_b, lno_w, lno_b, wq, wk, wv, lam, d, eps, res):
        _require_cuda(x, "mrla token forward")
        if x.dim() != 3 or o_prev.shape != x.shape or o_prev.dtype != x.dtype:
            raise L.MrlaHipError("expected x and o_prev of identical shape [b, n, c] and dtype")
        xc, oc = x.contiguous(), o_prev.contiguous()
        b, n, c = xc.shape
        side = int(round((n - 1) ** 0.5))
        if side * side != n - 1 or c % d:
            raise L.MrlaHipError(f"token count {n} is not 1 + a perfect square, or c={c} not divisible by d={d}")
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        p32 = [_f32(t).reshape(-1) for t in (lnx_w, lnx_b, lno_w, lno_b, wq, wk, lam)]
        wxw, wxb, wow, wob, wq32, wk32, lam32 = p32
        wv32 = _f32(wv).reshape(c, 9)
        ks = wq32.numel()
        stats = torch.empty((b, n, 4), dtype=torch.float32, device=dev)
        mom = torch.empty((b, c, L.FWD_MOMENTS), dtype=torch.float32, device=dev)
        gate = torch.empty((b, c // d), dtype=torch.float32, device=dev)
        out = torch.empty_like(xc)
        if _seq():
            _seq_call("mrla_token_light_fwd", _ptr(xc), _ptr(oc), _ptr(wxw), _ptr(wxb), _ptr(wow), _ptr(wob), _ptr(wq32),
                      _ptr(wk32), ks, _ptr(wv32), _ptr(lam32), float(eps), _ptr(stats), _ptr(mom), _ptr(gate), _ptr(out), b, n,
                      c, d, int(res), dt, st)
        else:
            _call("mrla_token_norm_pool", 0, _ptr(xc), _ptr(oc), _ptr(wxw), _ptr(wxb), float(eps), _ptr(stats), _ptr(mom), b, n,
                   c, dt, st)
            _call("mrla_light_gate_fwd", 0, _ptr(mom), _ptr(wq32), _ptr(wk32), ks, _ptr(gate), b, c, n - 1, d, st)
            _call("mrla_token_apply_fwd", xc.numel() * xc.element_size() * 3, _ptr(xc), _ptr(oc), _ptr(stats), _ptr(wxw),
                  _ptr(wxb), _ptr(wow), _ptr(wob), _ptr(wv32), _ptr(gate), _ptr(lam32), _ptr(out), b, n, c, d, int(res), dt, st,
                  path=xc.numel() * xc.element_size() * 3)
        ctx.d, ctx.res, ctx.ks = d, int(res), ks
        ctx.meta = [(t.shape, t.dtype) for t in (lnx_w, lnx_b, lno_w, lno_b, wq, wk, wv, lam)]
        ctx.save_for_backward(xc, oc, wxw, wxb, wow, wob, wq32, wk32, wv32, lam32, stats, mom, gate)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, dout):
        xc, oc, wxw, wxb, wow, wob, wq32, wk32, wv32, lam32, stats, mom, gate = ctx.saved_tensors
        b, n, c = xc.shape
        d, res, ks = ctx.d, ctx.res, ctx.ks
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        if dout.dtype != xc.dtype:
            dout = dout.to(xc.dtype)
        dout = dout.contiguous()
        es = xc.element_size()
        # one pass over the map (dxn without the pooled-descriptor term dy, bmom, the parameter partials); the gate backward
        # turns bmom into dy and completes the LayerNorm partials with it; the LayerNorm backward adds dy as it reads dxn
        bmom = torch.empty((b, c, L.BWD_MOMENTS), dtype=torch.float32, device=dev)
        dxn = torch.empty((b, n, c), dtype=torch.float32, device=dev)
        prow = L.load().mrla_token_part_rows(b, n, c, dt)
        L.check(min(prow, 0), "mrla_token_part_rows")
        part = torch.empty((prow, c * L.TOKEN_PARTIALS), dtype=torch.float32, device=dev)
        dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
        dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
        dx, do = torch.empty_like(xc), torch.empty_like(oc)
        sums = torch.empty((c * L.TOKEN_PARTIALS + 2 * ks,), dtype=torch.float32, device=dev)
        if _seq():
            _seq_call("mrla_token_light_bwd", _ptr(dout), _ptr(xc), _ptr(oc), _ptr(stats), _ptr(wxw), _ptr(wxb), _ptr(wow),
                      _ptr(wob), _ptr(wq32), _ptr(wk32), ks, _ptr(wv32), _ptr(gate), _ptr(lam32), _ptr(mom), _ptr(dxn),
                      _ptr(part), prow, _ptr(bmom), _ptr(dyx), _ptr(dwqk_part), _ptr(dx), _ptr(do), _ptr(sums), b, n, c, d, res,
                      dt, st)
        else:
            _call("mrla_token_apply_bwd", xc.numel() * es * 3 + dxn.numel() * 4, _ptr(dout), _ptr(xc), _ptr(oc), _ptr(stats),
                  _ptr(wxw), _ptr(wxb), _ptr(wow), _ptr(wob), _ptr(wv32), _ptr(gate), _ptr(lam32), _ptr(dxn), _ptr(part),
                  _ptr(bmom), b, n, c, d, dt, st, path=xc.numel() * es * 5)      # section 8(d) backward: 5 N s for the block
            _call("mrla_token_gate_bwd", 0, _ptr(mom), _ptr(bmom), _ptr(gate), _ptr(wq32), _ptr(wk32), ks, _ptr(dyx),
                   _ptr(dwqk_part), _ptr(part), b, n, c, d, dt, st)
            _call("mrla_token_ln_bwd", xc.numel() * es * 5 + dxn.numel() * 4, _ptr(dout), _ptr(xc), _ptr(oc), _ptr(dxn),
                  _ptr(dyx), _ptr(stats), _ptr(wxw), _ptr(wow), _ptr(lam32), _ptr(dx), _ptr(do), b, n, c, res, dt, st)
            _call("mrla_reduce_rows2", 0, _ptr(part), _ptr(sums), prow, c * L.TOKEN_PARTIALS, _ptr(dwqk_part),
                   _ptr(sums[c * L.TOKEN_PARTIALS:]), b, 2 * ks, st)
        pc = sums[:c * L.TOKEN_PARTIALS].view(c, L.TOKEN_PARTIALS)
        dwqk = sums[c * L.TOKEN_PARTIALS:]
        raw = (pc[:, 10], pc[:, 11], pc[:, 12], pc[:, 13], dwqk[:ks], dwqk[ks:], pc[:, :9], pc[:, 9])
        grads = [g.reshape(shape).to(dtype) for g, (shape, dtype) in zip(raw, ctx.meta)]
        return (dx, do, *grads, None, None, None)


def mrla_token_light(x, o_prev, lnx_w, lnx_b, lno_w, lno_b, wq, wk, wv, lam, d, eps=1e-6, res=False):
    return _TokenLightFn.apply(x, o_prev, lnx_w, lnx_b, lno_w, lno_b, wq, wk, wv, lam, d, eps, res)


# ======================================================================================================
# MRLA-base on token sequences (DeiT)
# ======================================================================================================
def token_base_supported(x, d):
    """True when the fused token module of MRLA-base (LayerNorm on load, map rows written in place) takes `x` [b, n, c]."""
    if x.dim() != 3 or not x.is_cuda or x.dtype not in _DT:
        return False
    b, n, c = x.shape
    return c % d == 0 and L.load().mrla_token_base_supported(b, n, c, _DT[x.dtype]) == 1


class _TokenBaseFn(torch.autograd.Function):
    """out = cat(LN_x(x)[:, :1], mrla_base_layer(LN_x(x)[:, 1:] as a 14x14 map)) without materialising LN_x(x), the map view
    or the cat (deit/deit_mrla_base.py:224-243).  The K / V history lives in `stage` (slot-major NHWC rings)."""

    @staticmethod
    @_on_device
    def forward(ctx, x, lnx_w, lnx_b, wq, wk, wv, stage, d, eps):
        _require_cuda(x, "mrla token-base forward")
        xc = x.contiguous()
        b, n, c = xc.shape
        side = int(round((n - 1) ** 0.5))
        if (b, c, side, side, d) != (stage.b, stage.c, stage.h, stage.w, stage.d) or xc.dtype != stage.dtype \
                or stage.layout != L.NHWC:
            raise L.MrlaHipError("input does not match the stage's K/V history (shape/dtype/layout); is init_cell set on "
                                 "the first block?")
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        wxw, wxb, wq32, wk32 = (_f32(t).reshape(-1) for t in (lnx_w, lnx_b, wq, wk))
        wv32 = _f32(wv).reshape(c, 9)
        ks = wq32.numel()
        es = xc.element_size()
        stats = torch.empty((b, n, 4), dtype=torch.float32, device=dev)
        mom = torch.empty((b, c, L.FWD_MOMENTS), dtype=torch.float32, device=dev)
        _call("mrla_token_norm_pool", 0, _ptr(xc), None, _ptr(wxw), _ptr(wxb), float(eps), _ptr(stats), _ptr(mom), b, n, c,
               dt, st)
        stage.reserve_slot()
        t, T = stage.t + 1, stage.T
        _call("mrla_token_base_value_fwd", xc.numel() * es * 2, _ptr(xc), _ptr(stats), _ptr(wxw), _ptr(wxb), _ptr(wv32),
              _ptr(stage.V[t - 1]), b, n, c, dt, st, path=xc.numel() * es * 2)
        q = torch.empty((b, c), dtype=torch.float32, device=dev)
        _call("mrla_base_gate_fwd", 0, _ptr(mom), _ptr(wq32), _ptr(wk32), ks, _ptr(stage.K), _ptr(stage.P), _ptr(q), b, c,
               n - 1, d, T, t, st)
        out = torch.empty_like(xc)
        arows = L.load().mrla_base_tile_rows(b, c, side, side, dt, L.NHWC)
        L.check(min(arows, 0), "mrla_base_tile_rows")
        amom = torch.empty((arows, c, 2), dtype=torch.float32, device=dev)
        _call("mrla_token_base_attend_fwd", xc.numel() * es * (t + 1), _ptr(stage.V), _ptr(stage.P), _ptr(xc), _ptr(stats),
              _ptr(wxw), _ptr(wxb), _ptr(out), _ptr(amom), b, n, c, d, T, t, dt, st, path=xc.numel() * es * t)
        stage.t = t
        ctx.stage, ctx.t, ctx.d, ctx.ks, ctx.side = stage, t, d, ks, side
        ctx.meta = [(p.shape, p.dtype) for p in (lnx_w, lnx_b, wq, wk, wv)]
        ctx.wv_stride = wv.stride()
        ctx.save_for_backward(xc, wxw, wxb, wq32, wk32, wv32, stats, mom, q)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, dout):
        xc, wxw, wxb, wq32, wk32, wv32, stats, mom, q = ctx.saved_tensors
        stage, t, d, ks, side = ctx.stage, ctx.t, ctx.d, ctx.ks, ctx.side
        b, n, c = xc.shape
        T = stage.T
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        if dout.dtype != xc.dtype:
            dout = dout.to(xc.dtype)
        dout = dout.contiguous()
        es = xc.element_size()
        first = stage.begin_layer_backward(t)
        Tc = stage.bwd_top
        pmom = torch.empty((b, c, t), dtype=torch.float32, device=dev)
        prows = L.load().mrla_base_pmom_rows(b, c, side, side, dt, L.NHWC)
        L.check(min(prows, 0), "mrla_base_pmom_rows")
        ppart = torch.empty((prows, t, c), dtype=torch.float32, device=dev)
        _call("mrla_token_base_attend_bwd", xc.numel() * es * (t + 2), _ptr(dout), _ptr(stage.V), _ptr(stage.dA), _ptr(ppart),
              b, n, c, T, t, dt, st, path=xc.numel() * es * (t + 1))
        _call("mrla_base_pmom_reduce", 0, _ptr(ppart), _ptr(pmom), b, c, t, prows, st)
        dv = torch.empty((b, side, side, c), dtype=xc.dtype, device=dev)
        _call("mrla_base_dv_combine", xc.numel() * es * (Tc - t + 2), _ptr(stage.dA), _ptr(stage.P), _ptr(dv), b, c, side,
              side, d, T, t, Tc, dt, L.NHWC, st, path=xc.numel() * es * (Tc - t + 1))
        dxn = torch.empty((b, n, c), dtype=torch.float32, device=dev)
        prow = L.load().mrla_token_part_rows(b, n, c, dt)
        L.check(min(prow, 0), "mrla_token_part_rows")
        part = torch.empty((prow, c * L.TOKEN_PARTIALS), dtype=torch.float32, device=dev)
        _call("mrla_token_base_value_bwd", xc.numel() * es * 2 + dxn.numel() * 4, _ptr(dout), _ptr(xc), _ptr(stats),
              _ptr(wxw), _ptr(wxb), _ptr(wv32), _ptr(dv), _ptr(dxn), _ptr(part), b, n, c, dt, st, path=xc.numel() * es)
        dyx = torch.empty((b, c), dtype=torch.float32, device=dev)
        dwqk_part = torch.empty((b, 2 * ks), dtype=torch.float32, device=dev)
        _call("mrla_token_base_gate_bwd", 0, _ptr(mom), _ptr(pmom), _ptr(stage.P), _ptr(q), _ptr(stage.K), _ptr(stage.dK),
               _ptr(wq32), _ptr(wk32), ks, _ptr(dyx), _ptr(dwqk_part), _ptr(part), b, n, c, d, T, t, int(first), dt, st)
        dx = torch.empty_like(xc)
        _call("mrla_token_ln_bwd", xc.numel() * es * 3 + dxn.numel() * 4, _ptr(dout), _ptr(xc), None, _ptr(dxn), _ptr(dyx),
              _ptr(stats), _ptr(wxw), None, None, _ptr(dx), None, b, n, c, 0, dt, st)
        sums = torch.empty((c * L.TOKEN_PARTIALS + 2 * ks,), dtype=torch.float32, device=dev)
        _call("mrla_reduce_rows2", 0, _ptr(part), _ptr(sums), prow, c * L.TOKEN_PARTIALS, _ptr(dwqk_part),
               _ptr(sums[c * L.TOKEN_PARTIALS:]), b, 2 * ks, st)
        pc = sums[:c * L.TOKEN_PARTIALS].view(c, L.TOKEN_PARTIALS)
        dwqk = sums[c * L.TOKEN_PARTIALS:]
        raw = (pc[:, 10], pc[:, 11], dwqk[:ks], dwqk[ks:])
        grads = [g.reshape(shape).to(dtype) for g, (shape, dtype) in zip(raw, ctx.meta)]
        sv, tv = ctx.meta[4]
        return (dx, *grads, _grad_like(pc[:, :9].to(tv), sv, ctx.wv_stride), None, None, None)


def mrla_token_base(x, lnx_w, lnx_b, wq, wk, wv, d, stage, eps=1e-6):
    return _TokenBaseFn.apply(x, lnx_w, lnx_b, wq, wk, wv, stage, d, eps)


# ======================================================================================================
# fused BatchNorm2d (+ReLU)  -- the producer-side epilogue in front of the MRLA tail
# ======================================================================================================
class _BnActFn(torch.autograd.Function):
    """y = relu?(BatchNorm2d(x)): one statistics pass + one elementwise pass per direction.
    defer: skip the forward elementwise pass; the result aliases x and (scale, shift) are returned beside it for the
    consumer (the fused MRLA producer) to apply.  The backward is the same either way."""

    @staticmethod
    @_on_device
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps, relu, defer=False,
                pre_moments=None, box=None):
        _require_cuda(x, "fused bn/act forward")
        ctx.set_materialize_grads(False)       # (no zero-filled gradient tensor for the non-differentiable second output)
        layout, xc = _layout_of(x)
        b, c, h, w = xc.shape
        if layout == L.NHWC and c % (16 // xc.element_size()):
            layout, xc = _layout_of(x, L.NCHW)           # the NHWC passes move 16-byte channel vectors
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        gamma32, beta32 = _f32(gamma), _f32(beta)
        rs = _RunningStats(running_mean, running_var, c, "fused bn/act forward")
        bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)       # sc, sh, save_mean, save_inv
        rows = L.load().mrla_bn_moment_rows(b, c, h, w, layout)           # partial-sum rows (b, or b*nsplit for NHWC)
        frows, pivot = rows, None
        records = training and pre_moments is not None
        if records:                # the producer (the 1x1 convolution GEMM) already took them: pivoted records per row
            amom, frows = pre_moments, pre_moments.shape[0]
            if tuple(amom.shape) != (frows, c, L.GEMM_MOMENTS) or amom.dtype != torch.float32:
                raise L.MrlaHipError("pre_moments must be float32 [rows, c, MRLA_GEMM_MOMENTS] records")
        else:
            amom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
            if training:           # sums about a per-channel pivot (a sample of the channel): robust for |mean| >> sigma
                pivot = torch.empty((c,), dtype=torch.float32, device=dev)
        if defer and relu:
            raise L.MrlaHipError("a deferred BatchNorm cannot carry a ReLU")
        y = None if defer else torch.empty_like(xc)
        if _seq():
            _seq_call("mrla_bn_fwd", _ptr(xc), _ptr(amom) if records else None, frows, None if records else _ptr(amom),
                      _ptr(pivot), rows, _ptr(gamma32), _ptr(beta32), _ptr(rs.rm), _ptr(rs.rv),
                      L.BN_TRAIN if training else L.BN_EVAL, float(momentum), float(eps), _ptr(bnbuf), int(relu), _ptr(y),
                      b, c, h, w, dt, layout, st)
        else:
            if training and not records:
                _call("mrla_bn_plane_moments", xc.numel() * xc.element_size(), _ptr(xc), _ptr(amom), _ptr(pivot), b, c, h, w,
                      dt, layout, st)
            if records:
                _call("mrla_bn_stats_fwd_rows", 0, _ptr(amom), _ptr(gamma32), _ptr(beta32), _ptr(rs.rm), _ptr(rs.rv), L.BN_TRAIN,
                       float(momentum), float(eps), _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(bnbuf[2]), _ptr(bnbuf[3]), frows, c, st)
            else:
                _call("mrla_bn_stats_fwd", 0, _ptr(amom), _ptr(pivot), _ptr(gamma32), _ptr(beta32), _ptr(rs.rm), _ptr(rs.rv),
                       L.BN_TRAIN if training else L.BN_EVAL, float(momentum), float(eps), _ptr(bnbuf[0]), _ptr(bnbuf[1]),
                       _ptr(bnbuf[2]), _ptr(bnbuf[3]), frows, c, b * h * w // frows, st)
            if y is not None:
                _call("mrla_bn_act_fwd", xc.numel() * xc.element_size() * 2, _ptr(xc), _ptr(bnbuf[0]), _ptr(bnbuf[1]), int(relu),
                      _ptr(y), b, c, h, w, dt, layout, st)
        rs.finish(training)
        ctx.training, ctx.relu, ctx.gdtype, ctx.layout, ctx.rows = training, int(relu), gamma.dtype, layout, rows
        ctx.box = box if defer else None
        if ctx.box is not None:
            ctx.box.center = bnbuf[2]
        ctx.save_for_backward(xc, gamma32, bnbuf)
        if defer:
            ctx.mark_non_differentiable(bnbuf)
            return xc.detach(), bnbuf
        return y

    @staticmethod
    @_on_device
    def backward(ctx, dy, _dbuf=None):
        if dy is None:
            return (None,) * 12
        xc, gamma32, bnbuf = ctx.saved_tensors
        b, c, h, w = xc.shape
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        if dy.dtype != xc.dtype:
            dy = dy.to(xc.dtype)
        layout, rows = ctx.layout, ctx.rows
        handed = ctx.box.take(dy) if ctx.box is not None else None      # (sum dz, sum dz*x) taken by the producer of dy
        dy = _layout_of(dy, layout)[1]
        es = xc.element_size()
        if handed is not None:
            tmom, rows = handed
        else:
            tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
        small = torch.empty((5, c), dtype=torch.float32, device=dev)          # cb[c,3] | dgamma | dbeta
        dx = torch.empty_like(xc)
        mode = L.BN_TRAIN if ctx.training else L.BN_EVAL
        if _seq():
            _seq_call("mrla_bn_bwd", _ptr(dy), _ptr(xc), _ptr(gamma32), _ptr(bnbuf), _ptr(tmom), rows, int(handed is not None),
                      mode, ctx.relu, _ptr(small), _ptr(dx), b, c, h, w, dt, layout, st)
        else:
            if handed is None:
                _call("mrla_bn_plane_dmoments", xc.numel() * es * 2, _ptr(dy), _ptr(xc), _ptr(bnbuf[0]), _ptr(bnbuf[1]),
                      _ptr(bnbuf[2]), ctx.relu, _ptr(tmom), b, c, h, w, dt, layout, st)
            cb = small[:3].view(c, 3)
            _call("mrla_bn_stats_bwd", 0, _ptr(tmom), _ptr(gamma32), _ptr(bnbuf[2]), _ptr(bnbuf[3]), mode, 1, _ptr(cb),
                  _ptr(small[3]), _ptr(small[4]), rows, c, b * h * w // rows, st)
            _call("mrla_bn_act_bwd", xc.numel() * es * 3, _ptr(dy), _ptr(xc), _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(cb), ctx.relu,
                  _ptr(dx), b, c, h, w, dt, layout, st)
        return dx, small[3].to(ctx.gdtype), small[4].to(ctx.gdtype), None, None, None, None, None, None, None, None, None


def bn_act(x, bn, relu, defer=False, pre_moments=None):
    """relu?(bn(x)) for an nn.BatchNorm2d module `bn` on the fused HIP passes; any other norm layer (or a layout /
    device the kernels do not handle) runs as the caller's module followed by torch.relu.
    defer=True (relu must be False): only the statistics are taken; the returned tensor aliases x and carries the
    per-channel affine as `._mrla_affine` for `mrla_light(..., pre_activation=True)` to apply in its first pass."""
    if (type(bn) is torch.nn.BatchNorm2d and bn.affine and bn.track_running_stats and x.is_cuda and x.dim() == 4
            and x.dtype in _DT and (x.is_contiguous() or x.is_contiguous(memory_format=_CL))):
        training = bn.training
        momentum = bump_batch_counter(bn) if training else bn.momentum
        if defer:
            box = _DeferredBnBox()
            y, buf = _BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum or 0.0,
                                    bn.eps, False, True, pre_moments, box)
            y._mrla_affine = (buf[0], buf[1])
            y._mrla_bn_box = box
            return y
        return _BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum or 0.0, bn.eps,
                              relu, False, pre_moments)
    y = bn(x)
    return torch.relu(y) if relu else y


class _BnReluPoolFn(torch.autograd.Function):
    """maxpool3x3/s2/p1(relu(BatchNorm2d(x))) on a channels_last tensor without the full-size intermediate
    (mrla_bn_relu_pool_*; resnet_mrla_light.py:220-222).  The statistics pass and the running-stat update are
    _BnActFn's; the window maximum follows ATen's first-maximum rule on the rounded values."""

    @staticmethod
    @_on_device
    def forward(ctx, x, gamma, beta, running_mean, running_var, training, momentum, eps):
        _require_cuda(x, "fused bn/relu/maxpool forward")
        layout, xc = _layout_of(x, L.NHWC)
        b, c, h, w = xc.shape
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        gamma32, beta32 = _f32(gamma), _f32(beta)
        rs = _RunningStats(running_mean, running_var, c, "fused bn/relu/maxpool forward")
        bnbuf = torch.empty((4, c), dtype=torch.float32, device=dev)       # sc, sh, save_mean, save_inv
        rows = L.load().mrla_bn_moment_rows(b, c, h, w, layout)
        amom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
        pivot = torch.empty((c,), dtype=torch.float32, device=dev) if training else None
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        out = torch.empty((b, c, ho, wo), dtype=xc.dtype, device=dev, memory_format=_CL)
        mode = L.BN_TRAIN if training else L.BN_EVAL
        if _seq():
            _seq_call("mrla_stem_fwd", _ptr(xc), _ptr(amom), _ptr(pivot), rows, _ptr(gamma32), _ptr(beta32), _ptr(rs.rm),
                      _ptr(rs.rv), mode, float(momentum), float(eps), _ptr(bnbuf), _ptr(out), b, c, h, w, dt, layout, st)
        else:
            if training:
                _call("mrla_bn_plane_moments", xc.numel() * xc.element_size(), _ptr(xc), _ptr(amom), _ptr(pivot), b, c, h, w, dt,
                      layout, st)
            _call("mrla_bn_stats_fwd", 0, _ptr(amom), _ptr(pivot), _ptr(gamma32), _ptr(beta32), _ptr(rs.rm), _ptr(rs.rv),
                   mode, float(momentum), float(eps), _ptr(bnbuf[0]), _ptr(bnbuf[1]), _ptr(bnbuf[2]), _ptr(bnbuf[3]), rows, c,
                   b * h * w // rows, st)
            _call("mrla_bn_relu_pool_fwd", (xc.numel() + out.numel()) * xc.element_size(), _ptr(xc), _ptr(bnbuf[0]),
                  _ptr(bnbuf[1]), _ptr(out), b, c, h, w, dt, layout, st)
        rs.finish(training)
        ctx.training, ctx.gdtype = training, gamma.dtype
        ctx.save_for_backward(xc, gamma32, bnbuf)
        return out

    @staticmethod
    @_on_device
    def backward(ctx, dp):
        xc, gamma32, bnbuf = ctx.saved_tensors
        b, c, h, w = xc.shape
        dt, dev, st = _DT[xc.dtype], xc.device, _stream()
        if dp.dtype != xc.dtype:
            dp = dp.to(xc.dtype)
        dp = _layout_of(dp, L.NHWC)[1]
        es = xc.element_size()
        rows = L.load().mrla_bn_pool_rows(b, c, h, w, dt, L.NHWC)
        L.check(min(rows, 0), "mrla_bn_pool_rows")
        tmom = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
        small = torch.empty((5, c), dtype=torch.float32, device=dev)          # cb[c,3] | dgamma | dbeta
        dx = torch.empty_like(xc) if ctx.needs_input_grad[0] else None
        mode = L.BN_TRAIN if ctx.training else L.BN_EVAL
        if _seq():
            _seq_call("mrla_stem_bwd", _ptr(dp), _ptr(xc), _ptr(gamma32), _ptr(bnbuf), _ptr(tmom), rows, mode, _ptr(small),
                      _ptr(dx), b, c, h, w, dt, L.NHWC, st)
        else:
            _call("mrla_bn_relu_pool_dmoments", (xc.numel() + dp.numel()) * es, _ptr(dp), _ptr(xc), _ptr(bnbuf[0]), _ptr(bnbuf[1]),
                  _ptr(bnbuf[2]), _ptr(tmom), b, c, h, w, dt, L.NHWC, st)
            cb = small[:3].view(c, 3)
            _call("mrla_bn_stats_bwd", 0, _ptr(tmom), _ptr(gamma32), _ptr(bnbuf[2]), _ptr(bnbuf[3]), mode, 1, _ptr(cb),
                  _ptr(small[3]), _ptr(small[4]), rows, c, b * h * w // rows, st)
            if dx is not None:
                _call("mrla_bn_relu_pool_bwd", (2 * xc.numel() + dp.numel()) * es, _ptr(dp), _ptr(xc), _ptr(bnbuf[0]),
                      _ptr(bnbuf[1]), _ptr(cb), _ptr(dx), b, c, h, w, dt, L.NHWC, st)
        return dx, small[3].to(ctx.gdtype), small[4].to(ctx.gdtype), None, None, None, None, None


def bn_relu_maxpool(x, bn, pool):
    """pool(relu(bn(x))) for the ResNet stem (nn.BatchNorm2d, then nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
    resnet_mrla_light.py:220-222): one fused pass per direction when x is a channels_last CUDA tensor with c % 64 == 0;
    any other configuration runs `pool(bn_act(x, bn, relu=True))`."""
    def _pair(v):
        return tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    if (type(bn) is torch.nn.BatchNorm2d and bn.affine and bn.track_running_stats and type(pool) is torch.nn.MaxPool2d
            and _pair(pool.kernel_size) == (3, 3) and _pair(pool.stride) == (2, 2) and _pair(pool.padding) == (1, 1)
            and _pair(pool.dilation) == (1, 1) and not pool.ceil_mode and not pool.return_indices
            and x.is_cuda and x.dim() == 4 and x.dtype in _DT and x.is_contiguous(memory_format=_CL)
            and x.data_ptr() % 16 == 0
            and L.load().mrla_bn_pool_rows(x.shape[0], x.shape[1], x.shape[2], x.shape[3], _DT[x.dtype], L.NHWC) > 0):
        training = bn.training
        momentum = bump_batch_counter(bn) if training else bn.momentum
        return _BnReluPoolFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, momentum or 0.0, bn.eps)
    return pool(bn_act(x, bn, relu=True))


# ======================================================================================================
# 1x1 stride-1 convolution as an MFMA GEMM with the BatchNorm statistics in its epilogue
# ======================================================================================================
class WeightBank:
    """bf16 working copies [n, k] (and transposes [k, n]) of a model's fp32 1x1-convolution weights, refreshed by ONE
    launch of mrla_weight_bank_refresh per training step: what torch.autocast does with one cast kernel per convolution
    and forward, and what the input-gradient GEMM needed one transposing copy per call for.  Built lazily for the
    eligible convolutions of a model (1x1, stride 1, no bias, fp32 weight, both channel counts multiples of 64);
    `refresh()` re-launches whenever gradients are enabled (a training forward: the optimizer will have moved the masters,
    and one launch costs 0.02 ms), whenever a HIP graph is being captured, and ALWAYS once a refresh has been captured:
    replays of that graph update the masters without touching any Python-side version counter, so after them the
    counters prove nothing.  Only a grad-free forward outside any capture history (validation loops) skips the launch
    while every weight's version counter stands still; writes the counters cannot see (`p.data.copy_(...)`) need
    `invalidate()`."""

    def __init__(self, convs):
        self.convs = [c for c in convs if self.eligible(c)]
        self.key = self.sig = None
        self.captured = False          # sticky: a refresh is part of some HIP graph
        self.entries = {}

    def invalidate(self):
        """Force the next refresh() to re-cast (after weight updates that bypass the version counters)."""
        self.key = None

    @staticmethod
    def eligible(conv):
        # (strided 1x1 convolutions -- the downsample branch of a stage's first block -- run the same GEMM on the
        # subsampled input: conv_bn_act)
        return (type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1)
                and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
                and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0)

    def _build(self, live):
        dev = live[0].weight.device
        total = sum(c.weight.numel() for c in live)
        self.flat = torch.empty((2, total), dtype=torch.bfloat16, device=dev)
        rows, off, self.entries, self.max_tiles = [], 0, {}, 1
        for c in live:
            n, k = c.out_channels, c.in_channels
            w16, w16t = self.flat[0, off:off + n * k].view(n, k), self.flat[1, off:off + n * k].view(k, n)
            rows.append([c.weight.data_ptr(), w16.data_ptr(), w16t.data_ptr(), (n << 32) | k])
            self.entries[id(c)] = (w16, w16t)
            self.max_tiles = max(self.max_tiles, (n // 64) * (k // 64))
            off += n * k
        self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        self.n = len(rows)

    def refresh(self):
        live = [c for c in self.convs if c.weight.is_cuda and c.weight.dtype == torch.float32
                and (c.weight.is_contiguous() or c.weight.is_contiguous(memory_format=_CL))]
        if not live:
            self.entries = {}
            return self
        sig = tuple((id(c), c.weight.data_ptr()) for c in live)
        capturing = torch.cuda.is_current_stream_capturing()
        if sig != self.sig:                       # first use, or the parameters moved (.to(), load with assign=...)
            if capturing:                         # (the table is built with a host-to-device copy: not capturable)
                raise L.MrlaHipError("WeightBank: first use (or moved parameters) inside a HIP graph capture; run one "
                                     "forward of the model eagerly before capturing it")
            self._build(live)
            self.sig, self.key = sig, None
        key = tuple(c.weight._version for c in live)
        self.captured = self.captured or capturing
        if key != self.key or self.captured or torch.is_grad_enabled():
            with torch.cuda.device(self.table.device):
                L.call("mrla_weight_bank_refresh", _ptr(self.table), self.n, self.max_tiles, _stream())
            self.key = key
        return self

    def get(self, conv):
        """(w bf16 [n, k], w^T bf16 [k, n]) of `conv`, or None when it is not in the bank."""
        return self.entries.get(id(conv))


class _Conv1x1Fn(torch.autograd.Function):
    """y = conv2d(x, w) for a bias-free 1x1 stride-1 convolution of a channels_last bf16 tensor, plus the partial
    moment-record rows of y the following BatchNorm needs (mrla_conv1x1_fwd: MRLA_GEMM_MOMENTS records, one row per
    workgroup pixel range -- the resident-weight kernels and the K-streaming kernel of the wide reductions, k >= 512, both
    write them; `mrla_bn_stats_fwd_rows` merges them).  Shapes the GEMMs do not take (mrla_conv1x1_rows < 0) run the stock
    convolution and return no rows; the BatchNorm then takes its own statistics pass.  Backward: dX through the same GEMM on w^T where it applies, dW
    through the split-M GEMM mrla_conv1x1_wgrad; what neither takes stays on the stock convolution backward."""

    @staticmethod
    @_on_device
    def forward(ctx, x, w, want_moments, passthrough=False, w16=None, w16t=None):
        """w16 / w16t: bf16 working copies [n, k] / [k, n] of an fp32 master weight `w` (WeightBank); the weight gradient
        is then returned in the master's dtype straight from the reduction kernel."""
        ctx.set_materialize_grads(False)       # (the moment rows carry no gradient; the shortcut's may be absent)
        b, k, h, wd = x.shape
        n = w.shape[0]
        m = b * h * wd
        dev, st = x.device, _stream()
        ctx.wshape, ctx.wstride, ctx.wdtype = w.shape, w.stride(), w.dtype   # [n, k] or the module's [n, k, 1, 1]
        if w16 is not None:
            w = w16
        else:
            w = w.reshape(n, k)                            # (a view: the 1x1 taps carry no data)
            if not w.is_contiguous():
                w = w.contiguous()
        part = None
        rows = L.load().mrla_conv1x1_rows(m, k, n, _DT[x.dtype])      # > 0: supported, that many moment-record rows; < 0: not taken
        if rows >= 0:
            y = torch.empty((b, n, h, wd), dtype=x.dtype, device=dev, memory_format=_CL)
            if want_moments and rows > 0:
                part = torch.empty((rows, n, L.GEMM_MOMENTS), dtype=torch.float32, device=dev)
            _call("mrla_conv1x1_fwd", (x.numel() + y.numel()) * x.element_size(), _ptr(x), _ptr(w), _ptr(y), _ptr(part), m, k,
                  n, _DT[x.dtype], st)
        else:
            y = torch.nn.functional.conv2d(x, w.view(n, k, 1, 1))
        ctx.save_for_backward(x, w, w16t)
        if part is None:
            part = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(part)
        if passthrough:             # x itself as a third output: its gradient (the shortcut's) comes back to backward()
            return y, part, x
        return y, part

    @staticmethod
    @_on_device
    def backward(ctx, dy, _dpart=None, d_through=None):
        x, w, w16t = ctx.saved_tensors
        if dy is None:             # only the shortcut carried a gradient
            return (d_through if ctx.needs_input_grad[0] else None), None, None, None, None, None
        dy = dy.contiguous(memory_format=_CL)
        n, k = w.shape
        b, _, h, wd = x.shape
        m = b * h * wd
        lib, dt = L.load(), _DT[x.dtype]
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = gw = None
        same = dy.dtype == x.dtype and dy.data_ptr() % 16 == 0
        # the input gradient is the same GEMM with the transposed weight: dX[m, k] = sum_n dY[m, n] * W^T[k, n]
        # (no memset of dX, as MIOpen's backward-data solver needs); shapes the kernel does not take stay on MIOpen
        if d_through is not None and (d_through.dtype != x.dtype or not d_through.is_contiguous(memory_format=_CL)
                                      or d_through.data_ptr() % 16):
            d_through = d_through.to(x.dtype).contiguous(memory_format=_CL)
        if need_x and same and lib.mrla_conv1x1_rows(m, n, k, dt) >= 0:
            gx = torch.empty_like(x)
            wt = w16t if w16t is not None else w.t().contiguous()
            if d_through is not None and lib.mrla_conv1x1_add_supported(m, n, k, dt) == 1:
                # ... + the shortcut's gradient in the GEMM epilogue (fp32 sum, one rounding) instead of a separate
                # accumulation pass over the block input's gradient
                _call("mrla_conv1x1_bwd_data", (dy.numel() + 2 * gx.numel()) * x.element_size(), _ptr(dy), _ptr(wt),
                      _ptr(d_through), _ptr(gx), m, n, k, dt, _stream(), entry="mrla_conv1x1_fwd_add")
                d_through = None
            else:
                _call("mrla_conv1x1_bwd_data", (dy.numel() + gx.numel()) * x.element_size(), _ptr(dy), _ptr(wt), _ptr(gx),
                      None, m, n, k, dt, _stream(), entry="mrla_conv1x1_fwd")
            need_x = False
        # the weight gradient dW[n, k] = sum_m dY[m, n] * X[m, k]: one pass over both activations, per-workgroup partial
        # tiles summed by a second kernel (MIOpen: memset + atomics into fp32 + a cast kernel)
        if need_w and same and w.dtype == x.dtype and ctx.wdtype in (x.dtype, torch.float32):
            rows = lib.mrla_conv1x1_wgrad_rows(m, k, n, dt)
            if rows > 0:
                part = torch.empty((rows, n, k), dtype=torch.float32, device=x.device)
                gw = torch.empty((n, k), dtype=ctx.wdtype, device=x.device)       # the master's dtype, no cast kernel behind
                _call("mrla_conv1x1_wgrad", (dy.numel() + x.numel()) * x.element_size(), _ptr(dy), _ptr(x), _ptr(part),
                      _ptr(gw), m, k, n, dt, _DT[ctx.wdtype], _stream())
                need_w = False
        if need_x or need_w:
            gx2, gw2, _ = torch.ops.aten.convolution_backward(dy, x, w.view(n, k, 1, 1), None, (1, 1), (0, 0), (1, 1), False,
                                                              (0, 0), 1, [need_x, need_w, False])
            gx = gx2 if need_x else gx
            gw = gw2.reshape(n, k) if need_w else gw
        if d_through is not None and ctx.needs_input_grad[0]:
            gx = gx + d_through
        if gw is not None:          # the weight's own dtype, shape AND strides (autograd's / DDP's gradient layout contract)
            gw = _grad_like(gw.to(ctx.wdtype), ctx.wshape, ctx.wstride)
        return gx, gw, None, None, None, None


class _SubsampleFn(torch.autograd.Function):
    """x[:, :, ::sh, ::sw] as a dense channels_last tensor, with a backward that stays channels_last: zeros everywhere, the
    incoming gradient at the sampled pixels (autograd's own slice backward answers in NCHW memory, which would cost the
    consumer -- conv1's input-gradient GEMM takes it as the shortcut's gradient -- a full-size layout conversion)."""

    @staticmethod
    def forward(ctx, x, sh, sw):
        ctx.shape, ctx.s = x.shape, (sh, sw)
        # (the memory format of the input is kept: NCHW models stay NCHW)
        ctx.fmt = _CL if x.is_contiguous(memory_format=_CL) and not x.is_contiguous() else torch.contiguous_format
        return x[:, :, ::sh, ::sw].contiguous(memory_format=ctx.fmt)

    @staticmethod
    def backward(ctx, g):
        sh, sw = ctx.s
        dx = torch.empty(ctx.shape, dtype=g.dtype, device=g.device, memory_format=ctx.fmt).zero_()
        dx[:, :, ::sh, ::sw] = g
        return dx, None, None


def _strided_1x1(conv):
    """nn.Conv2d 1x1, no padding, stride > 1 (resnet_mrla_light.py:196-199: the downsample branch of a stage's first block)."""
    return (type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and conv.stride != (1, 1)
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None)


def conv1x1_applies(conv, x, strided=False):
    """True when `conv(x)` belongs on the HIP GEMMs: nn.Conv2d 1x1 / stride 1 / no bias, channels_last bf16 input, and a
    shape the forward kernel (mrla_conv1x1_rows) or the weight-gradient kernel (mrla_conv1x1_wgrad_rows) takes.
    strided: the question is asked for a strided 1x1 convolution, which runs as the stride-1 GEMM on the subsampled input
    (`x` is the full-size input; the pixel count is the subsampled one)."""
    if not (type(conv) is torch.nn.Conv2d and conv.kernel_size == (1, 1) and (strided or conv.stride == (1, 1))
            and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None):
        return False
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=_CL)
            and x.data_ptr() % 16 == 0):               # (the kernels move 16-byte fragments)
        return False
    b, k, h, w = x.shape
    if k != conv.in_channels:
        return False
    if strided:
        h, w = (h + conv.stride[0] - 1) // conv.stride[0], (w + conv.stride[1] - 1) // conv.stride[1]
    lib, m, n = L.load(), b * h * w, conv.out_channels
    if lib.mrla_conv1x1_rows(m, k, n, L.BF16) >= 0:
        return True
    return torch.is_grad_enabled() and conv.weight.requires_grad and lib.mrla_conv1x1_wgrad_rows(m, k, n, L.BF16) > 0


def conv_bn_act(x, conv, bn, relu, defer=False, passthrough=False):
    """relu?(bn(conv(x))) -- resnet_mrla_light.py:93-94,100-101.  Eligible 1x1 convolutions run on the HIP GEMM, whose
    epilogue hands the train-mode BatchNorm its statistics (the moments pass over the output disappears); everything
    else is `bn_act(conv(x), ...)` with the stock convolution.
    passthrough=True returns (result, x'), x' being x routed through the convolution's autograd node: a consumer that
    uses x' as the block's shortcut (resnet_mrla_light.py:91,110-114) gets the shortcut gradient added inside the
    convolution's input-gradient GEMM instead of by a separate accumulation pass."""
    fused_bn = (type(bn) is torch.nn.BatchNorm2d and bn.affine and bn.track_running_stats)
    # A strided 1x1 convolution is the stride-1 one on the subsampled input: one strided copy (a quarter of the pixels), then
    # the same GEMMs -- forward with the BatchNorm statistics in its epilogue, input gradient, weight gradient.  Not for speed
    # alone: MIOpen's input gradient of exactly these convolutions is right when launched eagerly and garbage from the second
    # replay of a HIP graph on (every mode: immediate, find, deterministic; scripts/miopen_bwd_graph_probe.py,
    # profiles/r05_notes.md section 2) -- with them on the GEMMs the whole step replays correctly.
    strided = _strided_1x1(conv) and not passthrough and x.dim() == 4 and conv1x1_applies(conv, x, True)
    if strided or conv1x1_applies(conv, x):
        if strided:
            x = _SubsampleFn.apply(x, conv.stride[0], conv.stride[1])
        wt, w16, w16t = conv.weight, None, None
        if wt.dtype != x.dtype:
            book = current_bookkeeping()
            held = book.bank.get(conv) if (book is not None and book.bank is not None and wt.dtype == torch.float32) else None
            if held is not None:
                w16, w16t = held                     # the step's bf16 working copy and its transpose (WeightBank)
            else:
                wt = wt.to(x.dtype)                  # what autocast does for the stock convolution (differentiable)
        if passthrough and torch.is_grad_enabled() and x.requires_grad:
            y, part, through = _Conv1x1Fn.apply(x, wt, bool(fused_bn and bn.training), True, w16, w16t)
            return bn_act(y, bn, relu, defer, pre_moments=part if part.numel() else None), through
        y, part = _Conv1x1Fn.apply(x, wt, bool(fused_bn and bn.training), False, w16, w16t)
        out = bn_act(y, bn, relu, defer, pre_moments=part if part.numel() else None)
        return (out, x) if passthrough else out
    if _strided_1x1(conv) and not passthrough and x.dim() == 4 and x.is_cuda:
        # every other dtype / layout (resnet/train.py itself trains in fp32, :397-409): still the stride-1 convolution on the
        # subsampled input -- a plain GEMM for MIOpen, whose input gradient needs no zero-fill + scatter; the zero-fill and
        # the strided copy of _SubsampleFn are ordinary captured kernels.  (F.conv2d: autocast casts as for the module.)
        xs = _SubsampleFn.apply(x, conv.stride[0], conv.stride[1])
        return bn_act(torch.nn.functional.conv2d(xs, conv.weight), bn, relu, defer)
    out = bn_act(conv(x), bn, relu, defer)
    return (out, x) if passthrough else out
