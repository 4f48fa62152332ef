"""GPU parity of the fused stem tail maxpool3x3/s2/p1(relu(bn1(x))) (mrla_bn_relu_pool_*; resnet_mrla_light.py:220-222)
against the stock modules nn.BatchNorm2d -> relu -> nn.MaxPool2d on the same tensors: outputs, running statistics and every
gradient, train and eval mode, odd / even / ragged spatial sizes, fp32 and bf16."""
import pytest
import torch

from oracle import detgen

pytestmark = pytest.mark.gpu


def _modules(c, seed):
    bn = torch.nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(1 + 0.3 * detgen.uniform((c,), seed)))
        bn.bias.copy_(torch.from_numpy(0.2 * detgen.uniform((c,), seed + 1)))
    return bn, torch.nn.MaxPool2d(kernel_size=3, stride=2, padding=1)


# (b, c, h, w): even, odd, ragged last strip (wo % 4 != 0), several bands, two channel groups, tiny
SHAPES = [(2, 64, 16, 16), (3, 64, 15, 17), (2, 128, 9, 22), (4, 64, 112, 112), (1, 64, 2, 2), (2, 64, 7, 5), (1, 192, 30, 34)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_fused_stem_tail_matches_stock_modules(shape, training, dtype):
    from mrla_amd import functional as Fm
    b, c, h, w = shape
    bn, pool = _modules(c, 3)
    bn_r, _ = _modules(c, 3)
    bn.train(training); bn_r.train(training)
    if not training:
        with torch.no_grad():
            for m in (bn, bn_r):
                m.running_mean.copy_(torch.from_numpy(0.1 * detgen.uniform((c,), 7)))
                m.running_var.copy_(torch.from_numpy(1 + 0.2 * detgen.uniform((c,), 8)))
    x0 = torch.from_numpy(detgen.normalish((b, c, h, w), detgen.seed_of(f"stem/{shape}"))).cuda().to(dtype)
    x = x0.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    xr = x0.clone().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    Fm.TIMER = timer = Fm.KernelTimer(["mrla_bn_relu_pool_fwd", "mrla_bn_relu_pool_bwd"])
    try:
        out = Fm.bn_relu_maxpool(x, bn, pool)
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        assert tuple(out.shape) == (b, c, ho, wo) and out.dtype == dtype
        g = torch.from_numpy(detgen.normalish((b, c, ho, wo), 17)).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
        out.backward(g)
        torch.cuda.synchronize()
    finally:
        Fm.TIMER = None
    assert {n for n, *_ in timer.records} == {"mrla_bn_relu_pool_fwd", "mrla_bn_relu_pool_bwd"}     # the fused path ran
    # stock route on the same values; for bf16 through the product's own BatchNorm+ReLU pass (identical rounding), so that
    # the comparison isolates the pooling and its gradient routing
    if dtype == torch.float32:
        a = torch.relu(bn_r(xr))
    else:
        a = Fm.bn_act(xr, bn_r, relu=True)
    out_r = pool(a)
    out_r.backward(g)
    if dtype == torch.float32:
        assert torch.allclose(out, out_r, rtol=1e-5, atol=1e-5)
        tol = 2e-4
    else:
        assert torch.equal(out, out_r)
        tol = 2e-2
    for got, want in ((x.grad, xr.grad), (bn.weight.grad, bn_r.weight.grad), (bn.bias.grad, bn_r.bias.grad)):
        rel = ((got.float() - want.float()).norm() / want.float().norm().clamp_min(1e-20)).item()
        assert rel < tol, rel
    assert torch.allclose(bn.running_mean, bn_r.running_mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(bn.running_var, bn_r.running_var, rtol=1e-4, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(bn_r.num_batches_tracked)


def test_other_pool_configurations_keep_the_two_step_route():
    from mrla_amd import functional as Fm
    bn, _ = _modules(64, 5)
    x = torch.randn(2, 64, 12, 12, device="cuda").contiguous(memory_format=torch.channels_last)
    for pool in (torch.nn.MaxPool2d(2, 2), torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True), torch.nn.MaxPool2d(3, 1, 1)):
        Fm.TIMER = timer = Fm.KernelTimer(["mrla_bn_relu_pool_fwd"])
        try:
            y = Fm.bn_relu_maxpool(x, bn, pool)
        finally:
            Fm.TIMER = None
        assert not timer.records and torch.allclose(y, pool(torch.relu(bn(x))), atol=1e-5)
    xn = torch.randn(2, 64, 12, 12, device="cuda")                        # NCHW: bn_act + the module's own pooling
    y = Fm.bn_relu_maxpool(xn, bn, torch.nn.MaxPool2d(3, 2, 1))
    assert y.shape == (2, 64, 6, 6)
