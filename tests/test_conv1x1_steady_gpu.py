"""The 1x1-convolution MFMA GEMMs IN STEADY STATE, at the sizes bench.py runs them (b = 64 and b = 256 of every ResNet-50
shape class; resnet_mrla_light.py:93-102 conv1 / bn1, conv3 / bn3 and their backward).

tests/test_conv1x1_gpu.py covers the tile classes and ragged edges at small batches, where a workgroup sees one or two
units of work -- less than the depth of the LDS-DMA rings, so ring wrap-around, slot reuse, the constant `vmcnt(N)` waits,
the "issue past the range with an out-of-bounds offset" trick and the per-lane running BatchNorm sums over many units are
never reached there.  Here every case asserts FROM THE PLANNER (mrla_conv1x1_plan / mrla_conv1x1_wgrad_plan) that a
workgroup walks more units than its pipeline is deep, compares with a float64 product of the same bf16 operands rounded
once to bf16 (<= 1 bf16 ulp; moment partials vs float64 sums of the rounded outputs), poisons outputs and workspaces, and
runs everything twice requiring bit-equality (a stale-slot race shows up as a run-to-run difference before it shows up
as an error).  Inputs come from a seeded torch generator on the GPU; the float64 products run on the GPU as well."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

# (h, w, k, n) of mrla_conv1x1_fwd: conv3 / conv1 of stages 1-3 (wide: n % 256 == 0, narrow: n = 64)
FWD = [(56, 56, 64, 256), (56, 56, 256, 64), (28, 28, 128, 512), (14, 14, 256, 1024)]
# mrla_conv1x1_fwd_add = the input gradient of conv1 with the shortcut's gradient in the epilogue: x = dY[m, planes]
ADD = [(56, 56, 64, 256), (28, 28, 128, 512), (14, 14, 256, 1024)]
WGRAD = FWD + [(7, 7, 512, 2048), (14, 14, 1024, 256), (28, 28, 512, 128), (7, 7, 2048, 512)]


def _P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _operands(m, k, n, seed):
    """bf16 activations with per-channel means and scales (BatchNorm inputs are not centred) and a kaiming-scaled weight."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randn((m, k), device="cuda", generator=g)
    x = x * (0.5 + torch.rand((k,), device="cuda", generator=g)) + torch.randn((k,), device="cuda", generator=g)
    w = torch.randn((n, k), device="cuda", generator=g) * (2.0 / k) ** 0.5
    return x.bfloat16(), w.bfloat16()


def _assert_bf16_close(got, want64, what):
    """|got - bf16(want)| <= 1 bf16 ulp of the value (2^-7 relative) + a floor of 1e-3 of the tensor's largest ulp."""
    want = want64.float().bfloat16().float()
    tol = want.abs() * 2.0 ** -7 + 1e-3 * want64.abs().max().item() * 2.0 ** -7 + 1e-30
    bad = (got.float() - want).abs() > tol
    n = int(bad.sum())
    assert n == 0, f"{what}: {n} of {bad.numel()} beyond 1 bf16 ulp; worst {(got.float() - want).abs().max().item()}"


def _ragged_m(m0, k, n, addend):
    """The largest m < m0 with m % 32 != 0 that the planner takes with >= 2 ring depths of units per workgroup."""
    from mrla_amd import _lib as L
    for m in range(m0 - 1, m0 - 4000, -1):
        if m % 32 == 0:
            continue
        plan = L.conv1x1_plan(m, k, n, addend)
        if plan is not None and plan[0] >= 2 * plan[1]:
            return m
    raise AssertionError("no ragged pixel count with a deep pipeline near " + str(m0))


def _run_fwd(x, w, m, k, n, rows, moments):
    from mrla_amd import _lib as L
    y = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((rows, n, L.GEMM_MOMENTS), float("nan"), dtype=torch.float32, device="cuda") if moments else None
    L.call("mrla_conv1x1_fwd", _P(x), _P(w), _P(y), _P(part), m, k, n, L.BF16, _stream())
    return y, part


@pytest.mark.parametrize("batch", [64, 256, "ragged"])
@pytest.mark.parametrize("shape", FWD, ids=lambda s: "x".join(map(str, s)))
def test_forward_gemm_and_moment_partials_in_steady_state(shape, batch):
    from mrla_amd import _lib as L
    h, w_, k, n = shape
    wide = n % 256 == 0
    if batch == "ragged":
        m = _ragged_m(256 * h * w_, k, n, False)
        assert m % 32
    else:
        m = batch * h * w_
    upw, depth, wgs, rows = L.conv1x1_plan(m, k, n)
    # the point of this file: the software pipeline wraps around (wide form: LDS ring of `depth` blocks; narrow form: a
    # register double buffer per pixel-wave, which only b = 256 fills more than once over)
    assert upw > depth or (not wide and batch == 64 and upw >= depth), (upw, depth)
    assert rows == L.load().mrla_conv1x1_rows(m, k, n, L.BF16) and m % rows == 0
    x, w = _operands(m, k, n, seed=1000 + k + n)
    y, part = _run_fwd(x, w, m, k, n, rows, True)
    y2, part2 = _run_fwd(x, w, m, k, n, rows, True)
    y3, _ = _run_fwd(x, w, m, k, n, rows, False)
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(part, part2), "two runs of the same launch differ"
    assert torch.equal(y, y3), "the kernel without the moments epilogue stores different outputs"
    want = x.double() @ w.double().t()
    _assert_bf16_close(y, want, "y")
    del want
    # statistics of the stored (rounded) tensor, as the stand-alone moments pass would read them back
    from tests.test_conv1x1_gpu import raw_sums
    g = y.double()
    s = raw_sums(part)                                           # records (S1, S2, pivot, count) per row -> raw sums
    assert part[:, :, 3].double().sum(0).eq(m).all()
    s1, s2 = g.sum(0), (g * g).sum(0)
    assert ((s[:, 0] - s1).abs().max() / s1.abs().max()).item() < 1e-5
    assert ((s[:, 1] - s2).abs().max() / s2.abs().max()).item() < 1e-5
    # ... and the variance the per-channel kernel takes from the records (merged about one pivot, as it does)
    mean, var = s1 / m, g.var(dim=0, unbiased=False)
    r = part.double()
    P = r[0, :, 2]
    d = r[..., 2] - P
    S1 = (r[..., 0] + r[..., 3] * d).sum(0)
    S2 = (r[..., 1] + 2 * d * r[..., 0] + r[..., 3] * d * d).sum(0)
    mean_k, var_k = S1 / m + P, S2 / m - (S1 / m) ** 2
    assert ((mean_k - mean).abs() / var.sqrt()).max().item() < 1e-5
    assert ((var_k - var).abs() / var).max().item() < 1e-5


@pytest.mark.parametrize("batch", [64, 256, "ragged"])
@pytest.mark.parametrize("shape", ADD, ids=lambda s: "x".join(map(str, s)))
def test_gemm_with_addend_in_steady_state(shape, batch):
    from mrla_amd import _lib as L
    h, w_, k, n = shape
    m = _ragged_m(256 * h * w_, k, n, True) if batch == "ragged" else batch * h * w_
    upw, depth, _, _ = L.conv1x1_plan(m, k, n, True)
    assert upw > depth, (upw, depth)
    x, w = _operands(m, k, n, seed=2000 + k + n)
    g = torch.Generator(device="cuda").manual_seed(77)
    add = torch.randn((m, n), device="cuda", generator=g).bfloat16()

    def run(dst, addend):
        L.call("mrla_conv1x1_fwd_add", _P(x), _P(w), _P(addend), _P(dst), m, k, n, L.BF16, _stream())
        return dst
    y = run(torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda"), add)
    y2 = run(torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda"), add)
    inplace = add.clone()
    run(inplace, inplace)
    torch.cuda.synchronize()
    assert torch.equal(y, y2), "two runs of the same launch differ"
    assert torch.equal(y, inplace), "in place (addend aliasing y) differs"
    want = x.double() @ w.double().t()
    want += add.double()
    _assert_bf16_close(y, want, "y")


@pytest.mark.parametrize("batch", [64, 256, "ragged"])
@pytest.mark.parametrize("shape", WGRAD, ids=lambda s: "x".join(map(str, s)))
def test_weight_gradient_gemm_in_steady_state(shape, batch):
    from mrla_amd import _lib as L
    h, w_, k, n = shape
    m = 256 * h * w_ - 37 if batch == "ragged" else batch * h * w_
    plan = L.conv1x1_wgrad_plan(m, k, n)
    assert plan is not None
    chunks, stages, tn, tk, splits, tiles = plan
    assert chunks > stages, (chunks, stages)           # every LDS stage is refilled at least once
    assert splits == L.load().mrla_conv1x1_wgrad_rows(m, k, n, L.BF16)
    x, _ = _operands(m, k, n, seed=3000 + k + n)
    g = torch.Generator(device="cuda").manual_seed(78)
    dy = (torch.randn((m, n), device="cuda", generator=g) * (0.5 + torch.rand((n,), device="cuda", generator=g))).bfloat16()

    def run():
        part = torch.full((splits, n, k), float("nan"), dtype=torch.float32, device="cuda")
        dw = torch.full((n, k), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.call("mrla_conv1x1_wgrad", _P(dy), _P(x), _P(part), _P(dw), m, k, n, L.BF16, L.BF16, _stream())
        return part, dw
    part, dw = run()
    part2, dw2 = run()
    torch.cuda.synchronize()
    assert torch.equal(dw, dw2) and torch.equal(part, part2), "two runs of the same launch differ"
    want = dy.double().t() @ x.double()
    assert torch.isfinite(dw.float()).all()
    _assert_bf16_close(dw, want, "dw")
    # the partial tiles themselves: their float64 sum is the product to fp32 accuracy
    psum = part.double().sum(0)
    assert ((psum - want).abs().max() / want.abs().max()).item() < 1e-5


# the K-streaming kernel (k >= 512): conv1 of stages 2-4, conv3 of stage 4 and the input gradients with a wide reduction
KSTREAM = [(28, 28, 512, 128), (28, 28, 512, 256), (14, 14, 1024, 256), (14, 14, 1024, 512), (7, 7, 2048, 512),
           (7, 7, 512, 2048), (14, 14, 512, 1024)]


@pytest.mark.parametrize("batch", [64, 256, "ragged"])
@pytest.mark.parametrize("shape", KSTREAM, ids=lambda s: "x".join(map(str, s)))
def test_kstream_gemm_at_bench_sizes(shape, batch):
    """mrla_conv1x1_fwd on the wide reductions (both operands streamed through a three- or four-stage LDS ring, 16 - 64
    chunks per tile: the ring wraps 4 - 21 times) vs a float64 product rounded once; two launches bit-equal; poisoned
    output; with and without the moment-record epilogue (records vs float64 statistics of the stored tensor).  The batches
    cover both kernels of the planner: 256 x 256 tiles where they fill the chip, the smaller tiles elsewhere (7 x 7 maps,
    n = 128)."""
    from mrla_amd import _lib as L
    h, w_, k, n = shape
    m = 256 * h * w_ - 37 if batch == "ragged" else batch * h * w_
    chunks, stages, _, rows = L.conv1x1_plan(m, k, n)
    assert chunks >= 4 * stages and rows > 0 and rows == L.load().mrla_conv1x1_rows(m, k, n, L.BF16)
    x, w = _operands(m, k, n, seed=4000 + k + n)
    y, part = _run_fwd(x, w, m, k, n, rows, True)
    y2, part2 = _run_fwd(x, w, m, k, n, rows, True)
    y3, _ = _run_fwd(x, w, m, k, n, rows, False)
    torch.cuda.synchronize()
    assert torch.equal(y, y2) and torch.equal(part, part2), "two runs of the same launch differ"
    assert torch.equal(y, y3), "the kernel without the moments epilogue stores different outputs"
    want = x.double() @ w.double().t()
    _assert_bf16_close(y, want, "y")
    del want
    # the moment records (one row per pixel tile): counts, and mean / variance of the stored tensor merged as the
    # per-channel kernel merges them
    g = y.double()
    r = part.double()
    assert r[:, :, 3].sum(0).eq(m).all() and (r[:, :, 3] > 0).all()
    tm = int(r[0, 0, 3].item())
    assert torch.equal(part[:, :, 2], y[::tm].float()[:rows]), "pivots are the first pixel of every tile"
    mean, var = g.mean(0), g.var(dim=0, unbiased=False)
    P = r[0, :, 2]
    d = r[..., 2] - P
    S1 = (r[..., 0] + r[..., 3] * d).sum(0)
    S2 = (r[..., 1] + 2 * d * r[..., 0] + r[..., 3] * d * d).sum(0)
    mean_k, var_k = S1 / m + P, S2 / m - (S1 / m) ** 2
    assert ((mean_k - mean).abs() / var.sqrt()).max().item() < 1e-5
    assert ((var_k - var).abs() / var).max().item() < 1e-5
