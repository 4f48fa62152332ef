"""GPU: the training tail WITHOUT a stored x_t (ABI 5: mrla_light_stats_fwd_fused(x_out = NULL), mrla_light_apply_fwd_fused,
mrla_light_stats_bwd_fused, mrla_light_apply_bwd_fused; resnet_mrla_light.py:100-116) computes exactly what the x_t-storing
passes compute.  x_t = relu(round(round(psc*y3 + psh) + o)) is re-formed in every pass by the formula the fused forward
statistics pass stored it with, so every output, every input gradient, every parameter gradient, bn3's folded backward sums
and every running statistic must be BIT-IDENTICAL between `functional.LEAN = True` (13N elements per block and step)
and `False` (15N: the path the suite pinned against the oracle in rounds 1 - 5, and the default: in the training step the two
forms take the same time on MI355X -- profiles/r06_notes.md section 3 -- so the lean form is what it is for: 1N of
activation memory less per block).  Shapes: whole and ragged strips, one to eight
strips, bf16 and fp16, with and without the deferred bn3 in front, both call paths (sequence / per pass)."""
import contextlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def lean(on):
    from mrla_amd import functional as Fm
    was = Fm.LEAN
    Fm.LEAN = on
    try:
        yield
    finally:
        Fm.LEAN = was


def _calls(fn):
    from mrla_amd import _lib as L
    names, orig = [], L.call
    L.call = lambda name, *a: (names.append(name), orig(name, *a))[1]
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        L.call = orig
    return out, names


def _same(a, b, noisy=()):
    """Bit-identical -- except the keys in `noisy`: results that differ between two runs of the SAME path (MIOpen accumulates
    some gradients -- and, at small batches, some forward products -- with atomics): those to that noise's level."""
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape, k
        if k in noisy:
            assert ((a[k].float() - b[k].float()).norm() / b[k].float().norm().clamp_min(1e-20)).item() < 1e-2, k
        else:
            assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))


@pytest.mark.parametrize("seq", [True, False], ids=["sequence", "per-pass"])
@pytest.mark.parametrize("shape", [(4, 64, 56, 56), (3, 128, 28, 28), (5, 256, 14, 14), (6, 512, 7, 7), (2, 64, 20, 10), (2, 64, 9, 33)],
                         ids=lambda s: "x".join(map(str, s)))
def test_bottleneck_without_stored_xt_is_bit_identical(shape, seq):
    """One MRLA_Bottleneck (conv1 .. conv3, DEFERRED bn3 -> the fused MRLA tail with bn3's backward sums folded into its apply
    pass), bf16 autocast, train mode, stochastic depth pinned: forward + backward with and without a stored x_t."""
    from mrla_amd import functional as Fm, resnet
    b, planes, h, w = shape
    c = planes * 4
    torch.manual_seed(11)
    blk = resnet.MRLA_Bottleneck(c, planes, drop_path=0.2).cuda().to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        blk.bn3.weight.uniform_(0.3, 1.2)                                  # (zero_init_last_bn would switch the branch off)
        blk.bn3.bias.uniform_(-0.2, 0.2)
        blk.mrla.lambda_t.normal_()
    state = {k: v.clone() for k, v in blk.state_dict().items()}
    g = torch.Generator(device="cuda").manual_seed(3)
    # (bf16, as the block's input is inside the network: the output of the previous block under autocast)
    x0 = torch.randn(b, c, h, w, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gup = torch.randn(b, c, h, w, device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)

    def run():
        blk.load_state_dict(state)
        blk.zero_grad(set_to_none=True)
        torch.manual_seed(5)                                                # the same images dropped in both legs
        xt = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(xt)
        out.backward(gup)
        r = dict(out=out.detach(), dx=xt.grad)
        r.update({"grad:" + k: p.grad for k, p in blk.named_parameters()})
        r.update({"buf:" + k: v.clone() for k, v in blk.named_buffers()})
        return r
    was, det = Fm.SEQUENCES, torch.backends.cudnn.deterministic
    Fm.SEQUENCES = seq
    torch.backends.cudnn.deterministic = True      # MIOpen's deterministic solvers for the stock convolutions around the tail
    try:
        with lean(True):
            a, ca = _calls(run)
        with lean(False):
            s, cs = _calls(run)
            s2, _ = _calls(run)              # each path against ITSELF: what is not bit-reproducible on this shape anyway
        with lean(True):
            a2, _ = _calls(run)
    finally:
        Fm.SEQUENCES = was
        torch.backends.cudnn.deterministic = det
    assert Fm._DT[torch.bfloat16] is not None
    if seq:
        assert "mrla_light_tail_fwd" in ca and "mrla_light_tail_bwd" in ca
    else:       # the lean passes really ran: the x_t-free entry points, and no x_t-reading ones
        assert {"mrla_light_stats_fwd_fused", "mrla_light_apply_fwd_fused", "mrla_light_stats_bwd_fused",
                "mrla_light_apply_bwd_fused"} <= set(ca), ca
        assert not {"mrla_light_apply_fwd", "mrla_light_stats_bwd", "mrla_light_apply_bwd"} & set(ca), ca
        assert {"mrla_light_apply_fwd", "mrla_light_stats_bwd", "mrla_light_apply_bwd"} <= set(cs), cs
    # (the stock 3x3 convolution is not run-to-run reproducible at every shape -- b = 3 at 128 x 28 x 28: 704 of 1.2 M outputs of the
    # block differ between two runs of the storing path -- and everything downstream inherits that)
    noisy = {k for k in s if not (torch.equal(s[k], s2[k]) and torch.equal(a[k], a2[k]))}
    if "out" in noisy:       # the trunk itself is not reproducible at this shape: which elements flip is random, every key inherits it
        noisy = set(s)
    _same(a, s, noisy=noisy)
    assert all(torch.isfinite(v.float()).all() for v in a.values())
    assert float(a["grad:bn3.weight"].abs().max()) > 0 and float(a["grad:mrla.lambda_t"].abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("bn_mode", ["train", "eval", "nobn"])
def test_functional_tail_on_a_plain_pre_activation(dtype, bn_mode):
    """mrla_light(pre, ..., pre_activation=True) on a plain tensor (no deferred BatchNorm in front: x_t = relu(pre + o), no bn3
    sums to fold): the lean passes in their other template instances, fp16 included."""
    from mrla_amd.functional import mrla_light
    from tests import cases
    b, c, h, w, d = 3, 128, 14, 12, 32
    x, o, gup = cases.light_inputs("lean-plain", b, c, h, w)
    P = cases.block_params(c, 33)
    dev = lambda a, dt=torch.float32: torch.from_numpy(a).to("cuda", dt)          # noqa: E731

    def run():
        xt = dev(x, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        ot = dev(o, dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        prm = {k: dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
        rm, rv = dev(P["bn_mrla.running_mean"]), dev(P["bn_mrla.running_var"])
        bn = None if bn_mode == "nobn" else dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"], running_mean=rm,
                                                 running_var=rv, training=(bn_mode == "train"), momentum=0.1, eps=1e-5)
        out = mrla_light(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d, o_prev=ot,
                         lam=prm["mrla.lambda_t"], bn=bn, res=bn is not None, pre_activation=True)
        out.backward(dev(gup, dtype).contiguous(memory_format=torch.channels_last))
        r = dict(out=out.detach(), dx=xt.grad, do=ot.grad, rm=rm, rv=rv)
        r.update({k: v.grad for k, v in prm.items() if v.grad is not None})
        return r
    with lean(True):
        a, ca = _calls(run)
    with lean(False):
        s, _ = _calls(run)
    assert "mrla_light_tail_fwd" in ca
    _same(a, s)


def test_where_the_lean_passes_do_not_exist_the_stored_form_runs():
    """fp32 activations (LDS: three fp32 row rings do not fit beside the others), NCHW memory, channel counts off the 64-lane
    grid: mrla_light_lean_supported says 0 and the x_t-storing passes run -- same API, nothing raised."""
    from mrla_amd import _lib as L
    lib = L.load()
    assert lib.mrla_light_lean_supported(8, 256, 56, 56, L.BF16, L.NHWC) == 1
    assert lib.mrla_light_lean_supported(8, 256, 56, 56, L.F16, L.NHWC) == 1
    assert lib.mrla_light_lean_supported(8, 256, 56, 56, L.F32, L.NHWC) == 0
    assert lib.mrla_light_lean_supported(8, 256, 56, 56, L.BF16, L.NCHW) == 0
    assert lib.mrla_light_lean_supported(8, 96, 56, 56, L.BF16, L.NHWC) == 0
    assert lib.mrla_light_lean_supported(0, 256, 56, 56, L.BF16, L.NHWC) == L.EINVAL
    # x_out = NULL is refused where the later passes could not re-form x_t
    assert lib.mrla_light_stats_fwd_fused(1, None, None, 1, 1, 1, None, 2, 96, 8, 8, L.BF16, L.NHWC, None) == L.EUNSUPPORTED
    from mrla_amd.functional import mrla_light
    from tests import cases
    b, c, h, w, d = 2, 64, 8, 8, 32
    x, o, gup = cases.light_inputs("lean-f32", b, c, h, w)
    P = cases.block_params(c, 34)
    xt = torch.from_numpy(x).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ot = torch.from_numpy(o).cuda().contiguous(memory_format=torch.channels_last)
    prm = {k: torch.from_numpy(v).cuda() for k, v in P.items()}
    with lean(True):
        (out, names) = _calls(lambda: mrla_light(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"],
                                                 d, o_prev=ot, lam=prm["mrla.lambda_t"], pre_activation=True))
    assert torch.isfinite(out).all() and "mrla_light_tail_fwd" in names
