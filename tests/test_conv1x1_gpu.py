"""GPU parity of the 1x1-convolution MFMA GEMM (mrla_conv1x1_fwd; resnet_mrla_light.py:93-94 conv1/bn1, :100-101 conv3/bn3)
through the C ABI: outputs vs a float64 matrix product of the same bf16-rounded operands rounded once to bf16 (<= 1 bf16
ulp), the BatchNorm moment partials vs float64 sums of the ROUNDED outputs, gradients vs the stock convolution's, and the
conv+BatchNorm(+ReLU) composite vs the stock modules."""
import numpy as np
import pytest
import torch

from oracle import detgen
from tests.test_light_gpu import assert_bf16_close, bf16_round, relmax

pytestmark = pytest.mark.gpu

# (b, h, w, k, n): every ResNet-50 shape class the kernel takes (scaled-down batch) + a ragged pixel count + tiny N
SHAPES = [(2, 56, 56, 64, 256), (2, 56, 56, 256, 64), (2, 56, 56, 64, 64), (2, 56, 56, 256, 128), (3, 28, 28, 128, 512),
          (4, 14, 14, 256, 1024), (1, 4, 8, 64, 64), (5, 12, 16, 128, 192), (3, 7, 7, 64, 128), (1, 1, 1, 256, 64),
          (1, 5, 7, 64, 256), (3, 7, 7, 256, 512), (2, 9, 9, 128, 256), (1, 1, 1, 256, 256)]   # wide form, ragged pixel counts


def raw_sums(part):
    """[n, 2] float64 (sum y, sum y^2) from the GEMM epilogue's moment records [rows, n, 4] = (S1, S2, pivot, count)."""
    r = part.double()
    s1, s2, p, cnt = r[..., 0], r[..., 1], r[..., 2], r[..., 3]
    return torch.stack([(s1 + cnt * p).sum(0), (s2 + 2 * p * s1 + cnt * p * p).sum(0)], dim=1)


def _operands(b, h, w, k, n, salt=0):
    s = detgen.seed_of(f"conv1x1/{b}/{h}/{k}/{n}/{salt}")
    x = bf16_round(detgen.normalish((b, h, w, k), s))
    wt = bf16_round(detgen.normalish((n, k), s + 1) * (2.0 / k) ** 0.5)
    return x, wt


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_gemm_outputs_and_moment_partials(shape):
    from mrla_amd import _lib as L, functional as Fm
    b, h, w, k, n = shape
    m = b * h * w
    rows = L.load().mrla_conv1x1_rows(m, k, n, L.BF16)
    assert rows > 0 and m % rows == 0
    x, wt = _operands(b, h, w, k, n)
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2)            # [b, k, h, w], channels_last memory
    assert xt.is_contiguous(memory_format=torch.channels_last)
    wtt = torch.from_numpy(wt).cuda().bfloat16()
    y, part = Fm._Conv1x1Fn.apply(xt, wtt, True)
    torch.cuda.synchronize()
    assert y.is_contiguous(memory_format=torch.channels_last) and tuple(part.shape) == (rows, n, L.GEMM_MOMENTS)
    want = x.reshape(m, k).astype(np.float64) @ wt.astype(np.float64).T
    got = y.permute(0, 2, 3, 1).reshape(m, n).float().cpu().numpy()
    assert_bf16_close(got, want, "y")
    # the statistics are those of the stored (rounded) tensor, as the stand-alone moments pass would read them back:
    # records (sum (y-p), sum (y-p)^2, p, n) per workgroup row -> raw sums in float64
    g64 = got.astype(np.float64)
    s = raw_sums(part).cpu().numpy()
    assert part[:, :, 3].double().sum(0).eq(m).all()             # every pixel counted once per channel
    assert relmax(s[:, 0], g64.sum(0)) < 1e-5 and relmax(s[:, 1], (g64 * g64).sum(0)) < 1e-5
    # no moments requested: same outputs
    y2, p2 = Fm._Conv1x1Fn.apply(xt, wtt, False)
    assert p2.numel() == 0 and torch.equal(y, y2)


def test_unsupported_shapes_are_reported_not_run():
    from mrla_amd import _lib as L
    lib = L.load()
    assert lib.mrla_conv1x1_rows(64, 96, 64, L.BF16) == L.EUNSUPPORTED        # k not in {64, 128, 256} and < 512
    assert lib.mrla_conv1x1_rows(64, 1024, 256, L.BF16) == 1                   # the K-streaming kernel: one record row per pixel tile
    assert lib.mrla_conv1x1_rows(64, 512, 128, L.BF16) == 1 and lib.mrla_conv1x1_rows(4 * 3136, 512, 128, L.BF16) > 1
    assert lib.mrla_conv1x1_rows(64, 512, 64, L.BF16) == L.EUNSUPPORTED        # ... needs n % 128 == 0
    assert lib.mrla_conv1x1_rows(64, 544, 128, L.BF16) == 1 and lib.mrla_conv1x1_rows(64, 520, 128, L.BF16) == L.EUNSUPPORTED
    assert lib.mrla_conv1x1_fwd(None, None, None, None, 64, 1024, 256, L.BF16, None) == L.EINVAL
    assert lib.mrla_conv1x1_rows(48, 64, 64, L.BF16) > 0                       # ragged last pixel block
    assert lib.mrla_conv1x1_rows(1, 256, 64, L.BF16) == 1                       # a single pixel: one workgroup, one row
    assert lib.mrla_conv1x1_rows(64, 64, 96, L.BF16) == L.EUNSUPPORTED        # n % 64
    assert lib.mrla_conv1x1_rows(64, 64, 64, L.F32) == L.EUNSUPPORTED
    assert lib.mrla_conv1x1_rows(0, 64, 64, L.BF16) == L.EINVAL


@pytest.mark.parametrize("shape", [(4, 14, 14, 256, 64, True), (4, 14, 14, 64, 256, False), (3, 28, 28, 128, 512, False)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_bn_act_composite_matches_stock_modules(shape):
    """conv_bn_act (GEMM + its moment partials + fused BatchNorm(+ReLU)) vs nn.Conv2d -> nn.BatchNorm2d -> relu in fp32 on
    the same bf16 operands: outputs, running statistics, and every gradient (conv backward is the stock one)."""
    from mrla_amd import functional as Fm
    b, h, w, k, n, relu = shape
    x, wt = _operands(b, h, w, k, n, salt=1)
    conv = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(n).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(wt).view(n, k, 1, 1))
        bn.weight.copy_(torch.from_numpy(1 + 0.2 * detgen.uniform((n,), 5)))
        bn.bias.copy_(torch.from_numpy(0.1 * detgen.uniform((n,), 6)))
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    assert Fm.conv1x1_applies(conv, xt)
    out = Fm.conv_bn_act(xt, conv, bn, relu=relu)
    gup = torch.from_numpy(bf16_round(detgen.normalish((b, n, h, w), 9))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    out.backward(gup)
    # reference: fp32 modules on the same operands; the convolution output rounded to bf16 where the product stores it
    conv_r = torch.nn.Conv2d(k, n, 1, bias=False).cuda()
    bn_r = torch.nn.BatchNorm2d(n).cuda()
    conv_r.load_state_dict(conv.state_dict()); bn_r.load_state_dict({k_: v for k_, v in bn.state_dict().items()})
    bn_r.running_mean.zero_(); bn_r.running_var.fill_(1.0); bn_r.num_batches_tracked.zero_()
    xr = xt.detach().float().requires_grad_(True)
    yr = conv_r(xr).bfloat16().float()
    zr = bn_r(yr)
    if relu:
        zr = torch.relu(zr)
    zr.backward(gup.float())
    a, r = out.detach().float(), zr.detach()
    bad = (a - r).abs() > 2.0 ** -7 * (r.abs() + 0.05 * r.abs().max())
    assert bad.float().mean().item() < 1e-4
    assert torch.allclose(bn.running_mean, bn_r.running_mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(bn.running_var, bn_r.running_var, rtol=1e-4, atol=1e-6)
    for got, want, tol in ((bn.weight.grad, bn_r.weight.grad, 2e-2), (bn.bias.grad, bn_r.bias.grad, 2e-2),
                           (conv.weight.grad, conv_r.weight.grad, 3e-2), (xt.grad.float(), xr.grad, 3e-2)):
        assert ((got.float() - want).norm() / want.norm()).item() < tol


@pytest.mark.parametrize("shape", [(4, 14, 14, 256, 64), (4, 14, 14, 64, 256), (2, 28, 28, 64, 64), (3, 7, 7, 128, 512)],
                         ids=lambda s: "x".join(map(str, s)))
def test_input_gradient_is_the_gemm_with_the_transposed_weight(shape):
    """dX = dY * W (resnet backward of conv1 / conv3) through mrla_conv1x1_fwd on W^T where the kernel takes the shape
    (c_out in {64, 128, 256}), else through the stock backward: both must equal a float64 product rounded once to bf16."""
    from mrla_amd import _lib as L, functional as Fm
    b, h, w, k, n = shape
    m = b * h * w
    x, wt = _operands(b, h, w, k, n, salt=2)
    dy = bf16_round(detgen.normalish((b, h, w, n), detgen.seed_of(f"conv1x1/dy/{k}/{n}")))
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    wtt = torch.from_numpy(wt).cuda().bfloat16().requires_grad_(True)
    y, _ = Fm._Conv1x1Fn.apply(xt, wtt, False)
    y.backward(torch.from_numpy(dy).cuda().bfloat16().permute(0, 3, 1, 2))
    own = L.load().mrla_conv1x1_rows(m, n, k, L.BF16) >= 0
    assert own == (n in (64, 128, 256) or (n >= 512 and k % 128 == 0))
    want_dx = dy.reshape(m, n).astype(np.float64) @ wt.astype(np.float64)
    got_dx = xt.grad.permute(0, 2, 3, 1).reshape(m, k).float().cpu().numpy()
    assert_bf16_close(got_dx, want_dx, "dx")
    want_dw = dy.reshape(m, n).astype(np.float64).T @ x.reshape(m, k).astype(np.float64)
    assert relmax(wtt.grad.float().cpu().numpy(), want_dw) < 2.0 ** -7


# (b, h, w, k, n): every tile class (64/128/256 on either side), several tiles per extent, ragged pixel counts (not a
# multiple of the 32-pixel chunk), fewer chunks than workgroups, a single pixel
WGRAD_SHAPES = [(2, 56, 56, 64, 64), (2, 56, 56, 64, 256), (2, 56, 56, 256, 64), (2, 28, 28, 256, 128), (2, 28, 28, 512, 128),
                (2, 28, 28, 128, 512), (2, 14, 14, 1024, 256), (2, 14, 14, 256, 1024), (3, 7, 7, 2048, 512),
                (3, 7, 7, 512, 2048), (1, 5, 7, 128, 64), (1, 1, 1, 64, 128), (5, 12, 16, 192, 320), (2, 9, 9, 128, 128)]


@pytest.mark.parametrize("shape", WGRAD_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_weight_gradient_gemm(shape):
    """dW[n,k] = sum_m dY[m,n] X[m,k] (mrla_conv1x1_wgrad through the C ABI) vs a float64 product of the same bf16 operands
    rounded once to bf16; workspace rows as the library reports them; poisoned workspace and output (no memset needed)."""
    import ctypes
    from mrla_amd import _lib as L
    b, h, w, k, n = shape
    m = b * h * w
    lib = L.load()
    rows = lib.mrla_conv1x1_wgrad_rows(m, k, n, L.BF16)
    assert rows > 0
    x, _ = _operands(b, h, w, k, n, salt=3)
    dy = bf16_round(detgen.normalish((b, h, w, n), detgen.seed_of(f"conv1x1/wgrad/dy/{m}/{k}/{n}")))
    xt = torch.from_numpy(x).cuda().bfloat16().reshape(m, k)
    dyt = torch.from_numpy(dy).cuda().bfloat16().reshape(m, n)
    part = torch.full((rows, n, k), float("nan"), dtype=torch.float32, device="cuda")
    dw = torch.full((n, k), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.call("mrla_conv1x1_wgrad", ctypes.c_void_p(dyt.data_ptr()), ctypes.c_void_p(xt.data_ptr()),
           ctypes.c_void_p(part.data_ptr()), ctypes.c_void_p(dw.data_ptr()), m, k, n, L.BF16, L.BF16, st)
    torch.cuda.synchronize()
    want = dy.reshape(m, n).astype(np.float64).T @ x.reshape(m, k).astype(np.float64)
    got = dw.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert_bf16_close(got, want, "dw")
    # the partial tiles themselves: their fp64 sum is the product to fp32 accuracy
    psum = part.double().sum(0).cpu().numpy()
    assert relmax(psum, want) < 1e-5


def test_weight_gradient_unsupported_shapes_are_reported():
    from mrla_amd import _lib as L
    lib = L.load()
    assert lib.mrla_conv1x1_wgrad_rows(64, 96, 64, L.BF16) == L.EUNSUPPORTED
    assert lib.mrla_conv1x1_wgrad_rows(64, 64, 32, L.BF16) == L.EUNSUPPORTED
    assert lib.mrla_conv1x1_wgrad_rows(64, 64, 64, L.F32) == L.EUNSUPPORTED
    assert lib.mrla_conv1x1_wgrad_rows(1 << 24, 128, 64, L.BF16) == L.EUNSUPPORTED      # 32-bit buffer offsets
    assert lib.mrla_conv1x1_wgrad_rows(0, 64, 64, L.BF16) == L.EINVAL
    assert lib.mrla_conv1x1_wgrad_rows(1, 64, 64, L.BF16) == 1
    assert lib.mrla_conv1x1_wgrad_rows(256 * 56 * 56, 64, 256, L.BF16) == 256           # one workgroup per CU


def test_wide_reduction_convolution_takes_the_weight_gradient_gemm():
    """A 1x1 convolution with a wide reduction (c_in = 1024): forward on the K-streaming GEMM (no statistics epilogue: the
    BatchNorm takes its own moments pass), weight gradient on mrla_conv1x1_wgrad, input gradient (reduction 256) on the
    wide-output GEMM: outputs and gradients agree with the stock modules'."""
    from mrla_amd import functional as Fm
    b, h, w, k, n = 4, 14, 14, 1024, 256
    x, wt = _operands(b, h, w, k, n, salt=4)
    conv = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(n).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(wt).view(n, k, 1, 1))
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    assert Fm.conv1x1_applies(conv, xt)
    with torch.no_grad():
        assert Fm.conv1x1_applies(conv, xt)                      # (the forward GEMM applies on its own now)
    Fm.TIMER = timer = Fm.KernelTimer(["mrla_conv1x1_wgrad", "mrla_conv1x1_fwd"])
    try:
        out = Fm.conv_bn_act(xt, conv, bn, relu=True)
        gup = torch.from_numpy(bf16_round(detgen.normalish((b, n, h, w), 19))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
        out.backward(gup)
        torch.cuda.synchronize()
    finally:
        Fm.TIMER = None
    assert set(timer.summary()) == {"mrla_conv1x1_wgrad", "mrla_conv1x1_fwd"}
    conv_r = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn_r = torch.nn.BatchNorm2d(n).cuda()
    conv_r.load_state_dict(conv.state_dict())
    xr = xt.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_r = Fm.bn_act(conv_r(xr), bn_r, relu=True)
    out_r.backward(gup)
    assert ((out.float() - out_r.float()).norm() / out_r.float().norm()).item() < 1e-2       # (the solver may differ)
    assert ((conv.weight.grad - conv_r.weight.grad).norm() / conv_r.weight.grad.norm()).item() < 1e-2
    assert ((xt.grad.float() - xr.grad.float()).norm() / xr.grad.float().norm()).item() < 1e-2


@pytest.mark.parametrize("shape", [(2, 56, 56, 64, 256), (3, 28, 28, 128, 512), (4, 14, 14, 256, 1024), (1, 5, 7, 64, 256),
                                   (2, 9, 9, 256, 256)], ids=lambda s: "x".join(map(str, s)))
def test_gemm_with_addend_epilogue(shape):
    """y = x w^T + addend (mrla_conv1x1_fwd_add): fp32 sum rounded once, vs a float64 product + addend; in place
    (addend aliasing y) gives the same bits."""
    import ctypes
    from mrla_amd import _lib as L
    b, h, w, k, n = shape
    m = b * h * w
    lib = L.load()
    assert lib.mrla_conv1x1_add_supported(m, k, n, L.BF16) == 1
    assert lib.mrla_conv1x1_add_supported(m, k, 64, L.BF16) == L.EUNSUPPORTED
    x, wt = _operands(b, h, w, k, n, salt=5)
    add = bf16_round(detgen.normalish((m, n), detgen.seed_of(f"conv1x1/add/{m}/{k}/{n}")))
    xt = torch.from_numpy(x).cuda().bfloat16().reshape(m, k)
    wtt = torch.from_numpy(wt).cuda().bfloat16()
    at = torch.from_numpy(add).cuda().bfloat16()
    y = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    L.call("mrla_conv1x1_fwd_add", P(xt), P(wtt), P(at), P(y), m, k, n, L.BF16, L.BF16, st)
    torch.cuda.synchronize()
    want = x.reshape(m, k).astype(np.float64) @ wt.astype(np.float64).T + add.astype(np.float64)
    assert_bf16_close(y.float().cpu().numpy(), want, "y")
    inplace = at.clone()
    L.call("mrla_conv1x1_fwd_add", P(xt), P(wtt), P(inplace), P(inplace), m, k, n, L.BF16, L.BF16, st)
    torch.cuda.synchronize()
    assert torch.equal(inplace, y)


@pytest.mark.parametrize("shape", [(4, 14, 14, 256, 64), (2, 28, 28, 512, 128), (3, 7, 7, 2048, 512)],
                         ids=lambda s: "x".join(map(str, s)))
def test_shortcut_gradient_joins_the_input_gradient_gemm(shape):
    """conv_bn_act(..., passthrough=True): the second result is x routed through the convolution's autograd node; a
    gradient arriving there (the shortcut's, resnet_mrla_light.py:110-114) must come out as dX + that gradient, whether
    the GEMM epilogue adds it (c_in % 256 == 0, c_out <= 256) or the fallback does."""
    from mrla_amd import _lib as L, functional as Fm
    b, h, w, k, n = shape
    m = b * h * w
    x, wt = _operands(b, h, w, k, n, salt=6)
    conv = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(n).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(wt).view(n, k, 1, 1))
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    g1 = torch.from_numpy(bf16_round(detgen.normalish((b, n, h, w), 29))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    g2 = torch.from_numpy(bf16_round(detgen.normalish((b, k, h, w), 31))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    Fm.TIMER = timer = Fm.KernelTimer(["mrla_conv1x1_bwd_data"])
    try:
        out, through = Fm.conv_bn_act(xt, conv, bn, relu=True, passthrough=True)
        assert through.data_ptr() == xt.data_ptr() and through.grad_fn is not None
        torch.autograd.backward([out, through], [g1, g2])
        torch.cuda.synchronize()
    finally:
        Fm.TIMER = None
    fused = L.load().mrla_conv1x1_add_supported(m, n, k, L.BF16) == 1
    assert fused == (n <= 256)
    assert len(timer.records) == 1            # (n = 512: the K-streaming GEMM, the shortcut gradient added behind it)
    conv_r = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn_r = torch.nn.BatchNorm2d(n).cuda()
    conv_r.load_state_dict(conv.state_dict())
    xr = xt.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out_r = Fm.bn_act(conv_r(xr), bn_r, relu=True)
    torch.autograd.backward([out_r, xr * 1.0], [g1, g2])
    assert ((xt.grad.float() - xr.grad.float()).norm() / xr.grad.float().norm()).item() < 1e-2
    assert ((conv.weight.grad - conv_r.weight.grad).norm() / conv_r.weight.grad.norm()).item() < 1e-2
    # without autograd (or for a convolution off the GEMM path) the second result is x itself
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        o2, t2 = Fm.conv_bn_act(xt, conv, bn, relu=True, passthrough=True)
    assert t2 is xt


def test_weight_bank_refresh_and_fp32_weight_gradient():
    """mrla_weight_bank_refresh: every fp32 1x1-convolution weight -> its bf16 working copy and the transpose, in one
    launch, bit-equal to torch's `.bfloat16()` (what autocast's cast kernel produces); refreshed by every training
    forward, by grad-free forwards only when a weight changed -- and always once a refresh has been captured into a HIP graph;
    and mrla_conv1x1_wgrad with dw_dtype = MRLA_F32 writes the fp32 sum whose bf16 rounding is the bf16 result."""
    import ctypes
    from mrla_amd import _lib as L, functional as Fm
    torch.manual_seed(3)
    convs = [torch.nn.Conv2d(k, n, 1, bias=False).cuda() for k, n in ((64, 256), (256, 64), (512, 2048), (1024, 256), (128, 128))]
    convs[1].to(memory_format=torch.channels_last)
    convs.append(torch.nn.Conv2d(64, 64, 1, stride=2, bias=False).cuda())      # strided 1x1: the same GEMM on the subsampled input
    odd = [torch.nn.Conv2d(64, 96, 1, bias=False).cuda(), torch.nn.Conv2d(64, 64, 1, padding=1, bias=False).cuda(),
           torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).cuda()]
    bank = Fm.WeightBank(convs + odd).refresh()
    torch.cuda.synchronize()
    assert all(bank.get(c) is None for c in odd)
    for c in convs:
        w16, w16t = bank.get(c)
        want = c.weight.detach().reshape(c.out_channels, c.in_channels).bfloat16()
        assert torch.equal(w16, want) and torch.equal(w16t, want.t().contiguous())
    # a grad-free forward with unchanged weights: no launch (poison the copies, refresh, still poisoned) ...
    def fresh():
        torch.cuda.synchronize()
        return all(torch.equal(bank.get(c)[0], c.weight.detach().reshape(c.out_channels, c.in_channels).bfloat16()) for c in convs)
    bank.flat.fill_(float("nan"))
    with torch.no_grad():
        bank.refresh()
    torch.cuda.synchronize()
    assert torch.isnan(bank.get(convs[0])[0].float()).all()
    # ... a training forward (gradients enabled) always re-casts: the optimizer moves the masters between forwards
    bank.refresh()
    assert fresh()
    # an in-place update seen by the version counters re-casts in a grad-free forward too
    with torch.no_grad():
        convs[0].weight.mul_(0.5)
        bank.refresh()
    assert fresh()
    # a write the version counters cannot see (`.data`): stale until invalidate()
    convs[2].weight.data.mul_(2.0)
    with torch.no_grad():
        bank.refresh()
    assert not fresh()
    bank.invalidate()
    with torch.no_grad():
        bank.refresh()
    assert fresh()
    # once a refresh is part of a HIP graph, replays move the masters behind the counters' back (the captured optimizer step):
    # from then on every refresh re-casts, also grad-free ones (train by replay, then an eager validation forward)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        bank.refresh()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr), torch.no_grad():
        bank.refresh()
        for c in convs:
            c.weight.mul_(0.9)                           # stands for the captured optimizer step
    versions = [c.weight._version for c in convs]
    gr.replay()
    gr.replay()
    torch.cuda.synchronize()
    assert [c.weight._version for c in convs] == versions        # replays do not bump Python-side versions ...
    assert not fresh()                                            # ... and the bank holds the copies of the replay's START
    with torch.no_grad():
        bank.refresh()                                            # the eager validation forward after train-by-replay
    assert fresh()
    # a first build inside a capture is refused with a clear error (its table upload is not capturable)
    bank2 = Fm.WeightBank(convs)
    gr2 = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr2):
            with pytest.raises(L.MrlaHipError, match="eagerly before capturing"):
                bank2.refresh()
    except RuntimeError:
        pass                                                      # (an empty capture may itself be refused by the runtime)
    # fp32 weight gradient straight from the reduction kernel
    b, h, w, k, n = 3, 14, 14, 256, 1024
    m = b * h * w
    x, _ = _operands(b, h, w, k, n, salt=11)
    dy = bf16_round(detgen.normalish((m, n), 123))
    xt, dyt = torch.from_numpy(x).cuda().bfloat16().reshape(m, k), torch.from_numpy(dy).cuda().bfloat16()
    rows = L.load().mrla_conv1x1_wgrad_rows(m, k, n, L.BF16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    part = torch.empty((rows, n, k), dtype=torch.float32, device="cuda")
    d16 = torch.full((n, k), float("nan"), dtype=torch.bfloat16, device="cuda")
    d32 = torch.full((n, k), float("nan"), dtype=torch.float32, device="cuda")
    L.call("mrla_conv1x1_wgrad", P(dyt), P(xt), P(part), P(d16), m, k, n, L.BF16, L.BF16, st)
    L.call("mrla_conv1x1_wgrad", P(dyt), P(xt), P(part), P(d32), m, k, n, L.BF16, L.F32, st)
    torch.cuda.synchronize()
    assert torch.equal(d32.bfloat16(), d16)
    want = dy.astype(np.float64).T @ x.reshape(m, k).astype(np.float64)
    assert relmax(d32.cpu().numpy(), want) < 1e-5


@pytest.mark.parametrize("shape", [(8, 56, 56, 256, 64), (8, 28, 28, 128, 512), (8, 14, 14, 1024, 256)],
                         ids=lambda s: "x".join(map(str, s)))
def test_conv_bn_act_with_and_without_the_weight_bank(shape):
    """conv_bn_act under bf16 autocast with the banked working copies vs the per-call cast (what autocast does): the
    banked bf16 weights are the very values the cast produces, so outputs, input gradients and BatchNorm gradients are
    bit-identical; the weight gradient comes in fp32 from the reduction kernel and equals the bf16 path's after one
    rounding.  (Whole models cannot be compared bit for bit: MIOpen's strided convolutions are not run-to-run
    deterministic at small batches -- scripts/determinism_probe.py.)"""
    from mrla_amd import functional as Fm
    b, h, w, k, n = shape
    torch.manual_seed(k + n)
    conv = torch.nn.Conv2d(k, n, 1, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(n).cuda()
    x = torch.randn(b, k, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gup = torch.randn(b, n, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    gsc = torch.randn(b, k, h, w, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    res = []
    for bank in (Fm.WeightBank([conv]), None):
        conv.zero_grad(); bn.zero_grad(); bn.reset_running_stats()
        xt = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16), Fm.batched_bookkeeping(0, bank.refresh() if bank else None):
            out, through = Fm.conv_bn_act(xt, conv, bn, relu=True, passthrough=True)
        torch.autograd.backward([out, through], [gup, gsc])
        torch.cuda.synchronize()
        res.append((out.detach(), xt.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_var.clone(),
                    conv.weight.grad.clone()))
    assert res[0][5].dtype == torch.float32
    from mrla_amd import _lib as L
    own_fwd = L.load().mrla_conv1x1_rows(b * h * w, k, n, L.BF16) > 0
    own_dgrad = L.load().mrla_conv1x1_rows(b * h * w, n, k, L.BF16) > 0
    for i in range(5):
        if (own_fwd and own_dgrad) or (own_fwd and i != 1):
            assert torch.equal(res[0][i], res[1][i]), i
        else:       # a stock (MIOpen) direction in between: the same operator on the same values, not run-to-run bit-stable
            a, r = res[0][i].float(), res[1][i].float()
            assert ((a - r).abs() <= 2.0 ** -6 * r.abs() + 1e-2 * r.abs().max()).all(), i
    if own_fwd:
        assert torch.equal(res[0][5].bfloat16(), res[1][5].bfloat16())
    else:           # (the stock forward's output, hence the BatchNorm backward feeding dY, is not bit-stable between the runs)
        a, r = res[0][5].float(), res[1][5].float()
        assert ((a - r).norm() / r.norm()).item() < 2.0 ** -7


@pytest.mark.parametrize("ratio", [30.0, 1000.0])
@pytest.mark.parametrize("shape", [(16, 28, 28, 128, 512), (16, 56, 56, 256, 64)], ids=["wide", "narrow"])
def test_conv_bn_statistics_when_the_mean_dwarfs_sigma(shape, ratio):
    """The GEMM epilogue's BatchNorm moments and the BatchNorm backward on conv outputs with |mean| / sigma = 30 and
    "10^3" per channel -- as far as bf16 resolves it: a bf16 tensor whose values spread over two or three neighbouring
    grid points has |mean| / sigma of a few hundred at most (asserted >= 200 on the rounded tensor).  The epilogue takes its moments
    about a per-workgroup pivot (records merged in double), the backward its sums about the saved mean: batch statistics to
    1e-5 / 1e-4, outputs to the fp32 evaluation of the affine, parameter gradients to 1e-3 -- where raw one-pass fp32
    sums lose eps * ratio^2 of the variance (6 % at 10^3)."""
    from mrla_amd import functional as Fm
    b, h, w, k, n = shape
    m = b * h * w
    g = torch.Generator(device="cuda").manual_seed(int(ratio) + k)
    # y[m, c] = mean_c * (1 + noise / ratio) with |mean_c| ~ 64 .. 120: x = 1 + small noise, w[c, :] = mean_c / k
    noise = torch.randn((b, h, w, k), device="cuda", generator=g) * (k ** 0.5) / min(ratio, 400.0)
    x = (1.0 + noise).bfloat16()
    sign = 2 * (torch.rand((n,), device="cuda", generator=g) > 0.5).float() - 1
    means = (64 + 56 * torch.rand((n,), device="cuda", generator=g)) * sign
    wt = (means[:, None] / k).expand(n, k).contiguous().bfloat16()
    bn = torch.nn.BatchNorm2d(n).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.6, 1.4, generator=g)
        bn.bias.uniform_(-0.3, 0.3, generator=g)
    xt = x.permute(0, 3, 1, 2)
    assert xt.is_contiguous(memory_format=torch.channels_last)
    yt, part = Fm._Conv1x1Fn.apply(xt, wt, True)                    # the GEMM itself is pinned elsewhere: take ITS output
    yt = yt.detach().requires_grad_(True)
    out = Fm.bn_act(yt, bn, relu=False, pre_moments=part)
    gup = torch.randn((b, n, h, w), device="cuda", generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    out.backward(gup)
    torch.cuda.synchronize()
    y = yt.detach().permute(0, 2, 3, 1).reshape(m, n).double()
    mean, var = y.mean(0), y.var(0, unbiased=False)
    real_ratio = (mean.abs() / var.sqrt()).median().item()
    assert real_ratio >= min(ratio, 200.0) * 0.9, real_ratio
    assert (var > 1e3 * bn.eps).all()                              # the variance matters next to eps
    assert ((bn.running_mean.double() - 0.1 * mean).abs() / (0.1 * mean.abs())).max().item() < 1e-5
    assert ((bn.running_var.double() - (0.9 + 0.1 * var * m / (m - 1))).abs() / (0.1 * var)).max().item() < 1e-4
    inv = 1.0 / torch.sqrt(var + bn.eps)
    yhat = (y - mean) * inv
    want = yhat * bn.weight.double() + bn.bias.double()
    got = out.detach().permute(0, 2, 3, 1).reshape(m, n).double()
    # the affine y -> sc*y + sh is evaluated in fp32 on y ~ ratio * sigma: eps * ratio of a unit-variance output, + bf16
    assert ((got - want).abs() <= 2.0 ** -7 * want.abs() + 4e-7 * real_ratio + 1e-2 * 2.0 ** -7).all()
    gu = gup.permute(0, 2, 3, 1).reshape(m, n).double()
    dgamma, dbeta = (gu * yhat).sum(0), gu.sum(0)
    assert ((bn.weight.grad.double() - dgamma).abs().max() / dgamma.abs().max()).item() < 1e-3
    assert ((bn.bias.grad.double() - dbeta).abs().max() / dbeta.abs().max()).item() < 1e-5
    # the gradient wrt the conv output: gamma/sigma * (g - mean(g) - yhat * mean(g * yhat)), stored in bf16
    dy = (bn.weight.double() * inv) * (gu - dbeta / m - yhat * dgamma / m)
    dgot = yt.grad.permute(0, 2, 3, 1).reshape(m, n).double()
    assert ((dgot - dy).abs() <= 2.0 ** -7 * dy.abs() + 1e-3 * dy.abs().max()).all()


@pytest.mark.parametrize("shape", [(4, 28, 28, 256, 512, 2), (3, 14, 14, 512, 1024, 2), (2, 15, 13, 128, 256, 2), (2, 14, 14, 1024, 2048, 2)],
                         ids=lambda s: "x".join(map(str, s)))
def test_strided_downsample_convolution_on_the_gemm_matches_stock_modules(shape):
    """The downsample branch of a stage's first block (resnet_mrla_light.py:196-199: 1x1 convolution with stride 2 +
    BatchNorm): conv_bn_act runs it as the stride-1 GEMM on the subsampled input (forward with the BatchNorm statistics in
    its epilogue, input gradient scattered back into a zero-filled channels_last tensor, weight gradient) -- vs the stock
    strided nn.Conv2d -> nn.BatchNorm2d in fp32 on the same bf16 operands: outputs, running statistics, every gradient,
    and zeros exactly at the pixels the stride skips (odd map sizes included)."""
    from mrla_amd import functional as Fm
    b, h, w, k, n, st = shape
    x, wt = _operands(b, h, w, k, n, salt=7)
    conv = torch.nn.Conv2d(k, n, 1, stride=st, bias=False).cuda().to(memory_format=torch.channels_last)
    bn = torch.nn.BatchNorm2d(n).cuda()
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(wt).view(n, k, 1, 1))
        bn.weight.copy_(torch.from_numpy(1 + 0.2 * detgen.uniform((n,), 5)))
        bn.bias.copy_(torch.from_numpy(0.1 * detgen.uniform((n,), 6)))
    xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
    used = []
    orig = Fm._Conv1x1Fn.apply
    try:
        Fm._Conv1x1Fn.apply = staticmethod(lambda *a: (used.append(a[0].shape), orig(*a))[1])
        out = Fm.conv_bn_act(xt, conv, bn, relu=False)
    finally:
        Fm._Conv1x1Fn.apply = orig
    ho, wo = (h + st - 1) // st, (w + st - 1) // st
    assert used == [torch.Size((b, k, ho, wo))], used                 # the GEMM path, on the subsampled input
    assert out.shape == (b, n, ho, wo) and out.is_contiguous(memory_format=torch.channels_last)
    gup = torch.from_numpy(bf16_round(detgen.normalish((b, n, ho, wo), 9))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)
    out.backward(gup)
    conv_r = torch.nn.Conv2d(k, n, 1, stride=st, bias=False).cuda()
    bn_r = torch.nn.BatchNorm2d(n).cuda()
    conv_r.load_state_dict(conv.state_dict()); bn_r.load_state_dict(bn.state_dict())
    bn_r.running_mean.zero_(); bn_r.running_var.fill_(1.0); bn_r.num_batches_tracked.zero_()
    xr = xt.detach().float().requires_grad_(True)
    zr = bn_r(conv_r(xr).bfloat16().float())
    zr.backward(gup.float())
    a, r = out.detach().float(), zr.detach()
    bad = (a - r).abs() > 2.0 ** -7 * (r.abs() + 0.05 * r.abs().max())
    assert bad.float().mean().item() < 1e-4
    # (reductions over 256 - 1024 channels: the two products differ in the last bf16 bit of a few outputs)
    assert torch.allclose(bn.running_mean, bn_r.running_mean, rtol=1e-3, atol=2e-5)
    assert torch.allclose(bn.running_var, bn_r.running_var, rtol=1e-3, atol=1e-6)
    assert xt.grad.is_contiguous(memory_format=torch.channels_last)
    mask = torch.zeros(h, w, dtype=torch.bool, device="cuda")
    mask[::st, ::st] = True
    assert float(xt.grad[:, :, ~mask].abs().max()) == 0.0             # exact zeros where the stride skips
    for got, want, tol in ((bn.weight.grad, bn_r.weight.grad, 2e-2), (bn.bias.grad, bn_r.bias.grad, 2e-2),
                           (conv.weight.grad, conv_r.weight.grad, 3e-2), (xt.grad.float(), xr.grad, 3e-2)):
        assert ((got.float() - want).norm() / want.norm()).item() < tol


@pytest.mark.parametrize("dtype,fmt", [(torch.float32, "nhwc"), (torch.float32, "nchw"), (torch.float16, "nhwc")])
def test_strided_downsample_in_other_dtypes_is_the_stride_1_convolution_on_the_subsampled_input(dtype, fmt):
    """resnet/train.py trains in fp32 (:397-409; no autocast) and deit's recipe is fp16: the strided 1x1 downsample convolution
    (resnet_mrla_light.py:196-199) must not reach MIOpen as a STRIDED convolution in any dtype -- its input gradient is right
    when launched eagerly and garbage from the second replay of a HIP graph on (profiles/r05_notes.md section 2).  conv_bn_act
    subsamples and runs the stride-1 convolution: same outputs, running statistics and gradients as the stock strided modules
    (fp32: to accumulation noise), exact zeros at the skipped pixels, the input's memory format kept -- and the input gradient
    of three replays of a captured forward + backward equals the eagerly launched one."""
    from mrla_amd import functional as Fm
    b, h, w, k, n, st = 8, 14, 14, 64, 128, 2
    cl = fmt == "nhwc"
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(k, n, 1, stride=st, bias=False).cuda().to(dtype)
    bn = torch.nn.BatchNorm2d(n).cuda()
    if cl:
        conv = conv.to(memory_format=torch.channels_last)
    x0 = torch.randn(b, k, h, w, device="cuda").to(dtype)
    x0 = x0.contiguous(memory_format=torch.channels_last) if cl else x0
    gup = torch.randn(b, n, h // st, w // st, device="cuda").to(dtype)
    gup = gup.contiguous(memory_format=torch.channels_last) if cl else gup
    calls = []
    orig = torch.nn.functional.conv2d
    xt = x0.clone().requires_grad_(True)
    try:
        torch.nn.functional.conv2d = lambda inp, wt, *a, **kw: (calls.append((tuple(inp.shape), a, kw)), orig(inp, wt, *a, **kw))[1]
        out = Fm.conv_bn_act(xt, conv, bn, relu=False)
    finally:
        torch.nn.functional.conv2d = orig
    assert calls == [((b, k, h // st, w // st), (), {})], calls           # stride 1, on the subsampled input
    out.backward(gup)
    conv_r = torch.nn.Conv2d(k, n, 1, stride=st, bias=False).cuda().to(dtype)
    bn_r = torch.nn.BatchNorm2d(n).cuda()
    conv_r.load_state_dict(conv.state_dict())
    xr = x0.clone().requires_grad_(True)
    zr = bn_r(conv_r(xr))
    zr.backward(gup)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert ((out.float() - zr.float()).norm() / zr.float().norm()).item() < tol
    assert torch.allclose(bn.running_mean, bn_r.running_mean, rtol=1e-3, atol=1e-5)
    assert torch.allclose(bn.running_var, bn_r.running_var, rtol=1e-3, atol=1e-5)
    assert xt.grad.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
    mask = torch.zeros(h, w, dtype=torch.bool, device="cuda")
    mask[::st, ::st] = True
    assert float(xt.grad[:, :, ~mask].abs().max()) == 0.0
    for got, want in ((xt.grad, xr.grad), (conv.weight.grad, conv_r.weight.grad), (bn.weight.grad, bn_r.weight.grad)):
        assert ((got.float() - want.float()).norm() / want.float().norm()).item() < 5 * tol
    # under replay: the pool poisoned with NaN between replays (a solver that relies on memory it zeroed at capture time shows)
    xs = x0.clone().requires_grad_(True)
    bn.train()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            (gx,) = torch.autograd.grad(Fm.conv_bn_act(xs, conv, bn, relu=False), xs, gup)
    torch.cuda.current_stream().wait_stream(side)
    want = gx.detach().clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        (gx_static,) = torch.autograd.grad(Fm.conv_bn_act(xs, conv, bn, relu=False), xs, gup)
    for _ in range(3):
        gx_static.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(gx_static).all()
        assert ((gx_static.float() - want.float()).norm() / want.float().norm()).item() < 5 * tol


@pytest.mark.parametrize("shape", [(3, 28, 28, 512, 128), (3, 28, 28, 128, 512), (4, 56, 56, 64, 256), (5, 14, 14, 1024, 256),
                                   (6, 7, 7, 512, 2048)], ids=lambda s: "x".join(map(str, s)))
def test_the_gemms_are_bit_reproducible_run_to_run(shape):
    """No atomics, fixed summation orders: forward (with its BatchNorm moment records), input gradient and weight gradient of this
    build's 1x1 GEMMs return the same bits every time (tests/test_lean_gpu.py found a bottleneck whose outputs differ between two
    runs of one path at b = 3: that is the stock 3x3 convolution, not these)."""
    from mrla_amd import functional as Fm
    b, h, w, k, n = shape
    x, wt = _operands(b, h, w, k, n, salt=3)
    gup = torch.from_numpy(bf16_round(detgen.normalish((b, n, h, w), 17))).cuda().bfloat16().contiguous(memory_format=torch.channels_last)

    def once():
        xt = torch.from_numpy(x).cuda().bfloat16().permute(0, 3, 1, 2).requires_grad_(True)
        wtt = torch.from_numpy(wt).cuda().bfloat16().requires_grad_(True)
        y, part = Fm._Conv1x1Fn.apply(xt, wtt, True)
        y.backward(gup)
        torch.cuda.synchronize()
        return y.detach().clone(), part.clone(), xt.grad.clone(), wtt.grad.clone()
    first = once()
    for _ in range(3):
        for a, r in zip(once(), first):
            assert torch.equal(a, r)
