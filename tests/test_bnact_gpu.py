"""GPU parity of the fused BatchNorm2d(+ReLU) passes vs torch's own BatchNorm2d + relu (fp64 reference)."""
import numpy as np
import pytest
import torch

from tests.test_light_gpu import relmax

pytestmark = pytest.mark.gpu

SHAPES = [(3, 64, 112, 112), (4, 64, 56, 56), (2, 128, 28, 28), (3, 256, 14, 14), (5, 512, 7, 7), (2, 24, 5, 3)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_bn_act_matches_torch(shape, relu, training, cl):
    from mrla_amd.functional import bn_act
    torch.manual_seed(0)
    b, c, h, w = shape
    fmt = torch.channels_last if cl else torch.contiguous_format
    x = (torch.randn(shape, device="cuda") * 1.7 + 0.4).contiguous(memory_format=fmt)
    g = torch.randn(shape, device="cuda").contiguous(memory_format=fmt)
    bn = torch.nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.8, 1.2)
    ref = torch.nn.BatchNorm2d(c).cuda().double()
    ref.load_state_dict(bn.state_dict())
    bn.train(training); ref.train(training)
    xa = x.clone().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    y = bn_act(xa, bn, relu)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    y.backward(g); yr.backward(g.double())
    # vs torch's own fp32 BatchNorm2d on the same GPU: both sides round in fp32 (measured <= 4.8e-7)
    assert relmax(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 2e-6
    assert relmax(xa.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 2e-6
    assert relmax(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 2e-6
    assert relmax(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 2e-6
    assert relmax(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 2e-6
    assert relmax(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 2e-6
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


def test_bn_act_bf16():
    from mrla_amd.functional import bn_act
    torch.manual_seed(1)
    x = torch.randn(4, 256, 56, 56, device="cuda").bfloat16()
    g = torch.randn_like(x)
    bn = torch.nn.BatchNorm2d(256).cuda()
    ref = torch.nn.BatchNorm2d(256).cuda().double()
    xa = x.clone().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    y = bn_act(xa, bn, True)
    yr = torch.relu(ref(xr))
    y.backward(g); yr.backward(g.double())
    want = yr.detach().float().bfloat16().float().cpu().numpy()
    got = y.detach().float().cpu().numpy()
    assert (np.abs(got - want) <= np.abs(want) * 2.0 ** -7 + 1e-6).all()
    gw = xr.grad.float().cpu().numpy()
    gg = xa.grad.float().cpu().numpy()
    bad = np.abs(gg - gw) > 2.0 ** -7 * (np.abs(gw) + 0.05 * np.abs(gw).max())
    assert bad.mean() < 1e-4        # a y that rounds across zero flips its ReLU mask


DEFER_SHAPES = [(2, 64, 9, 6), (3, 72, 7, 7), (2, 128, 5, 61), (1, 256, 12, 75), (4, 256, 56, 56)]


@pytest.mark.parametrize("shape", DEFER_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_deferred_bn3_affine_is_bit_identical_to_the_separate_pass(shape, training, cl, dtype):
    """bn3's elementwise pass folded into the first MRLA pass (resnet_mrla_light.py:101-102,113-116): outputs and every
    gradient must equal, bit for bit, what the stand-alone BatchNorm pass followed by the same fused tail produces."""
    from mrla_amd.functional import bn_act, mrla_light
    b, c, h, w = shape
    if not cl and w > 64:
        pytest.skip("NCHW kernels: rows wider than one wave are served by the channels_last path")
    torch.manual_seed(5)
    mk = lambda *s: torch.randn(*s, device="cuda")
    conv_out, idn, g = mk(b, c, h, w).to(dtype), mk(b, c, h, w).to(dtype), mk(b, c, h, w).to(dtype)
    if cl:
        conv_out, idn, g = (t.contiguous(memory_format=torch.channels_last) for t in (conv_out, idn, g))
    k = 3 if c == 64 else 5
    wq, wk, wv, lam = mk(1, 1, k) * 0.5, mk(1, 1, k) * 0.5, mk(c, 1, 3, 3) * 0.3, mk(c, 1, 1)
    results = []
    for defer in (False, True):
        bn3, bnm = torch.nn.BatchNorm2d(c).cuda(), torch.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            for bn in (bn3, bnm):
                bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
                bn.running_mean.copy_(torch.linspace(-0.2, 0.2, c)); bn.running_var.copy_(torch.linspace(0.8, 1.2, c))
        bn3.train(training); bnm.train(training)
        xin, oin = conv_out.clone().requires_grad_(True), idn.clone().requires_grad_(True)
        prm = [p.clone().requires_grad_(True) for p in (wq, wk, wv, lam)]
        pre = bn_act(xin, bn3, relu=False, defer=defer)
        assert (getattr(pre, "_mrla_affine", None) is not None) == defer
        out = mrla_light(pre, prm[0], prm[1], prm[2], 8 if c % 32 else 32, o_prev=oin, lam=prm[3],
                         bn=dict(weight=bnm.weight, bias=bnm.bias, running_mean=bnm.running_mean,
                                 running_var=bnm.running_var, training=training, momentum=0.1, eps=1e-5),
                         res=True, pre_activation=True)
        out.backward(g)
        torch.cuda.synchronize()
        results.append([out.detach(), xin.grad, oin.grad, bn3.weight.grad, bn3.bias.grad, bn3.running_var.clone(),
                        bnm.weight.grad, bnm.running_mean.clone()] + [p.grad for p in prm])
    # With the affine deferred on the channels_last row pipeline (c % 64 == 0, 16-bit types), bn3's backward sums -- sum dpre
    # and sum dpre * (y3 - mean) -- are taken inside mrla_light_apply_bwd, of dpre as it is stored, in that kernel's
    # summation order: bn3's parameter gradients and the constants of its input gradient then differ from the separate
    # pass by fp32 summation order only; everything else stays bit-identical.
    from mrla_amd import _lib as L
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    # (partial rows that do not divide the pixel count are folded by the hand-over box: _DeferredBnBox.put)
    fused_sums = cl and L.load().mrla_light_apply_bwd_pre_sums(b, c, h, w, dt, L.NHWC) == 1
    for i, (a, bb) in enumerate(zip(*results)):
        if fused_sums and i in (3, 4):          # bn3.weight.grad, bn3.bias.grad
            tol = 2e-5
            assert ((a - bb).abs().max() / bb.abs().max()).item() < tol, i
        elif fused_sums and i == 1:             # the gradient wrt conv3's output: e*dpre + f*y3 + h with those constants
            af, bf_ = a.float(), bb.float()
            unit = (2.0 ** -7 if dtype == torch.bfloat16 else 1e-5) * (bf_.abs() + 0.05 * bf_.abs().max())
            assert ((af - bf_).abs() <= unit).all(), i
            assert (af != bf_).float().mean().item() < 1e-3, i          # (a constant moved in its last fp32 bits)
        else:
            assert torch.equal(a, bb), i


def test_deferred_bn_output_is_refused_outside_the_fused_producer():
    from mrla_amd.functional import bn_act, mrla_light
    from mrla_amd._lib import MrlaHipError
    bn = torch.nn.BatchNorm2d(64).cuda()
    x = torch.randn(2, 64, 8, 8, device="cuda")
    pre = bn_act(x, bn, relu=False, defer=True)
    with pytest.raises(MrlaHipError):
        mrla_light(pre, torch.randn(1, 1, 3).cuda(), torch.randn(1, 1, 3).cuda(), torch.randn(64, 1, 3, 3).cuda(), 32)


@pytest.mark.parametrize("shape", [(2, 64, 9, 6), (2, 128, 5, 61), (1, 256, 12, 75), (4, 256, 56, 56), (3, 2048, 7, 7)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("with_bn3", [True, False], ids=["bn3-deferred", "plain-pre"])
def test_inference_form_of_the_fused_tail_matches_the_training_form(shape, dtype, with_bn3):
    """no_grad + eval BatchNorm on channels_last: x_t is not materialised (pooling pass + an apply pass that re-forms it).
    Same result as the two-pass form that saves x_t, up to the summation order of the pooled sums."""
    from mrla_amd.functional import bn_act, mrla_light
    b, c, h, w = shape
    torch.manual_seed(11)
    mk = lambda *s: torch.randn(*s, device="cuda")
    cl = lambda t: t.to(dtype).contiguous(memory_format=torch.channels_last)
    conv_out, idn = cl(mk(b, c, h, w)), cl(mk(b, c, h, w))
    k = 3 if c == 64 else (7 if c == 2048 else 5)
    wq, wk, wv, lam = mk(1, 1, k) * 0.5, mk(1, 1, k) * 0.5, mk(c, 1, 3, 3) * 0.3, mk(c, 1, 1)
    bn3, bnm = torch.nn.BatchNorm2d(c).cuda().eval(), torch.nn.BatchNorm2d(c).cuda().eval()
    with torch.no_grad():
        for bn in (bn3, bnm):
            bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
            bn.running_mean.copy_(torch.linspace(-0.2, 0.2, c)); bn.running_var.copy_(torch.linspace(0.8, 1.2, c))
    bnargs = dict(weight=bnm.weight, bias=bnm.bias, running_mean=bnm.running_mean, running_var=bnm.running_var,
                  training=False, momentum=0.1, eps=1e-5)

    def tail(xin):
        pre = bn_act(xin, bn3, relu=False, defer=True) if with_bn3 else xin
        return mrla_light(pre, wq, wk, wv, 32, o_prev=idn, lam=lam, bn=bnargs, res=True, pre_activation=True)

    from mrla_amd import functional as Fm
    Fm.TIMER = timer = Fm.KernelTimer(["mrla_light_pool_fused", "mrla_light_apply_fwd_fused", "mrla_light_stats_fwd_fused"])
    try:
        with torch.no_grad():
            got = tail(conv_out)
        names = {rec[0] for rec in timer.records}
    finally:
        Fm.TIMER = None
    assert names == {"mrla_light_pool_fused", "mrla_light_apply_fwd_fused"}, names      # the inference form did run
    want = tail(conv_out.clone().requires_grad_(True)).detach()          # needs a gradient: the x_t-saving form
    torch.cuda.synchronize()
    a, r = got.float(), want.float()
    if dtype == torch.float32:
        assert (a - r).abs().max().item() <= 2e-5 * r.abs().max().item()
    else:
        bad = (a - r).abs() > 2.0 ** -7 * (r.abs() + 0.05 * r.abs().max())
        assert bad.float().mean().item() < 1e-4


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("shape", [(8, 64, 28, 28), (4, 256, 56, 56), (6, 24, 5, 3)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("ratio", [100.0, 1000.0])
def test_batchnorm_statistics_stay_accurate_when_the_mean_dwarfs_sigma(cl, shape, ratio):
    """Channels with |mean| / sigma ~ 1e3 (the one-pass E[x^2] - E[x]^2 on raw fp32 sums would lose ~ eps * ratio^2 = 6 %
    of the variance; torch uses Welford): the moments pass accumulates about a per-channel pivot, so the batch variance,
    the running statistics and the gradients keep fp32 accuracy."""
    from mrla_amd.functional import bn_act
    torch.manual_seed(3)
    b, c, h, w = shape
    fmt = torch.channels_last if cl else torch.contiguous_format
    sigma = 0.5 + torch.rand(1, c, 1, 1, device="cuda")
    sign = torch.where(torch.arange(c, device="cuda") % 2 == 0, 1.0, -1.0).view(1, c, 1, 1)
    x = (sigma * (torch.randn(shape, device="cuda") + ratio * sign)).contiguous(memory_format=fmt)
    g = torch.randn(shape, device="cuda").contiguous(memory_format=fmt)
    bn = torch.nn.BatchNorm2d(c).cuda()
    ref = torch.nn.BatchNorm2d(c).cuda().double()
    xa, xr = x.clone().requires_grad_(True), x.double().requires_grad_(True)
    y, yr = bn_act(xa, bn, False), ref(xr)
    y.backward(g); yr.backward(g.double())
    # variance to 1e-5 (raw sums: off by percents at ratio 1e3); the mean is trivially right either way
    assert relmax(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 1e-5
    assert relmax(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 1e-6
    # y = sc*x + sh is evaluated in fp32 on x ~ ratio*sigma: its rounding is eps*ratio of a unit-variance output
    tol = 4e-7 * ratio
    assert relmax(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < tol
    assert relmax(xa.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 4 * tol
    assert relmax(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 4 * tol
