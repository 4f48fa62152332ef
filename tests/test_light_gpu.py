"""GPU parity: HIP MRLA-light path (through the C ABI) vs the numpy oracle and the reference's goldens.

Protocol (SURVEY.md section 7 "bf16 tolerance"):
  fp32  : kernel vs fp64 oracle on the same fp32 inputs, error relative to the tensor's max-abs; bound 1e-6 for
          activations / input gradients (north_star), 5e-6 for parameter gradients, 3e-5 for the cancelling dWq / dWk
          sums (tests/cases.py: the bounds with their measured maxima and reasons).
  bf16  : inputs/weights pre-rounded to bf16; kernel output (bf16) vs the fp64 oracle rounded once to bf16 must
          agree within 1 bf16 ulp (2^-7 relative), plus an absolute floor for values near zero.
"""
import numpy as np
import pytest
import torch

from oracle import mrla_numpy as mn
from tests import cases

pytestmark = pytest.mark.gpu

ACT_TOL, PAR_TOL, QK_TOL, GOLD_TOL, TINY_BN_TOL = cases.ACT_TOL, cases.PAR_TOL, cases.QK_TOL, cases.GOLD_TOL, cases.TINY_BN_TOL


def par_tol(name):
    """Bound for a parameter gradient by (reference or oracle) name: the cancelling Wq / Wk sums get QK_TOL."""
    return QK_TOL if ("Wq" in name or "Wk" in name or name in ("dwq", "dwk")) else PAR_TOL


relmax = cases.relmax          # error relative to the tensor's max-abs; every call is recorded (tests/cases.py)


def to_dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def run_light(x, o, P, d, mode, dp_mask, p_drop, gup, dtype=torch.float32, rm=None, rv=None, cl=False):
    from mrla_amd.functional import mrla_light
    fmt = torch.channels_last if cl else torch.contiguous_format
    xt = to_dev(x, dtype).contiguous(memory_format=fmt).requires_grad_(True)
    ot = to_dev(o, dtype).contiguous(memory_format=fmt).requires_grad_(True)
    prm = {k: to_dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
    rmt = to_dev(P["bn_mrla.running_mean"] if rm is None else rm)
    rvt = to_dev(P["bn_mrla.running_var"] if rv is None else rv)
    dp = None
    if dp_mask is not None:
        dp = to_dev(dp_mask / (1.0 - p_drop))
    out = mrla_light(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d,
                     o_prev=ot, lam=prm["mrla.lambda_t"],
                     bn=dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"], running_mean=rmt, running_var=rvt,
                             training=(mode != "eval"), momentum=0.1, eps=1e-5), dp=dp, res=True)
    if cl:
        assert out.is_contiguous(memory_format=torch.channels_last)
    out.backward(to_dev(gup, dtype).contiguous(memory_format=fmt))
    torch.cuda.synchronize()
    g = dict(out=out.detach().float().cpu().numpy(), dx=xt.grad.float().cpu().numpy(), do=ot.grad.float().cpu().numpy(),
             rm=rmt.cpu().numpy(), rv=rvt.cpu().numpy())
    for k, v in prm.items():
        g["grad/" + k] = v.grad.cpu().numpy()
    return g


def oracle_light(x, o, P, d, mode, dp_mask, p_drop, gup):
    P = {k: np.asarray(v, np.float64) for k, v in P.items()}
    dp = None if dp_mask is None else mn.drop_path_scale(dp_mask, p_drop)
    out, cache = mn.light_tail_fwd(np.asarray(x, np.float64), np.asarray(o, np.float64),
                                   P["mrla.mrla.Wq.weight"].ravel(), P["mrla.mrla.Wk.weight"].ravel(),
                                   P["mrla.mrla.Wv.weight"][:, 0], P["mrla.lambda_t"].ravel(), P["bn_mrla.weight"],
                                   P["bn_mrla.bias"], P["bn_mrla.running_mean"], P["bn_mrla.running_var"], d,
                                   training=(mode != "eval"), dp=dp)
    g = mn.light_tail_bwd(np.asarray(gup, np.float64), cache)
    return out, cache, g


@pytest.mark.parametrize("case", cases.LIGHT_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["train", "eval", "traindp"])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_light_tail_fp32_vs_oracle_and_reference(case, mode, cl):
    name, b, c, h, w, d = case
    G = cases.golden("light_blocks")
    x, o, gup = cases.light_inputs(name, b, c, h, w)
    P = cases.block_params(c, 1)
    key = f"{name}/{mode}/"
    mask = G[key + "dp_mask"] if mode == "traindp" else None
    got = run_light(x, o, P, d, mode, mask, 0.25, gup, cl=cl)
    out, cache, g = oracle_light(x, o, P, d, mode, mask, 0.25, gup)
    assert relmax(got["out"], out) < ACT_TOL
    assert relmax(got["dx"], g["dx"]) < ACT_TOL
    assert relmax(got["do"], g["do_prev"]) < ACT_TOL
    assert relmax(got["rm"], cache["bn"]["new_rm"]) < ACT_TOL
    assert relmax(got["rv"], cache["bn"]["new_rv"]) < ACT_TOL
    for ours, theirs in (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
                         ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta")):
        assert relmax(got["grad/" + ours].ravel(), np.asarray(g[theirs]).ravel()) < par_tol(ours), ours
    # and directly against what the reference itself produced (fp32, so its own rounding is in the budget)
    sub = (lambda a: a[:, ::8]) if name == "s2048" else (lambda a: a)
    assert relmax(sub(got["out"]), G[key + "out"]) < GOLD_TOL
    assert relmax(sub(got["dx"]), G[key + "dx"]) < GOLD_TOL
    assert relmax(sub(got["do"]), G[key + "do"]) < GOLD_TOL
    assert relmax(got["grad/mrla.mrla.Wv.weight"], G[key + "grad/mrla.mrla.Wv.weight"]) < GOLD_TOL
    assert relmax(got["grad/mrla.lambda_t"], G[key + "grad/mrla.lambda_t"]) < GOLD_TOL
    # ... and against the reference run in float64 (light_blocks_f64.npz): every parameter gradient, the Wq / Wk sums included
    # (torch.rand draws another stochastic-depth mask in float64: those cases are covered where the masks coincide)
    G64 = cases.golden("light_blocks_f64")
    if mask is not None and not np.array_equal(G64[key + "dp_mask"], mask):
        return
    for ours in ("mrla.mrla.Wq.weight", "mrla.mrla.Wk.weight", "mrla.mrla.Wv.weight", "mrla.lambda_t", "bn_mrla.weight",
                 "bn_mrla.bias"):
        assert relmax(got["grad/" + ours].ravel(), G64[key + "grad/" + ours].ravel()) < par_tol(ours), ours


def bf16_round(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def assert_bf16_close(got, want64, what):
    want = bf16_round(want64)
    tol = np.abs(want) * 2.0 ** -7 + 1e-3 * np.abs(want64).max() * 2.0 ** -7 + 1e-30
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} beyond 1 bf16 ulp; worst {np.abs(got - want).max()}"


# (17, 256, 56, 56): 4352 slabs -> workgroups loop over 2 images each, ragged last group;
# (3, 2048, 7, 7): 2048 channels do not divide into 72-plane slabs -> ragged last slab.
# (2, 64, 1, 1) / (3, 64, 1, 3) / (2, 128, 3, 1): degenerate maps -- every tap but the centre (or a row / column of
# taps) falls into the padding; (1, 256, 9, 9): a single image (BatchNorm over h*w only), 2 strips with a ragged one.
# (128, 2048, 7, 7) / (128, 256, 3, 56) / (256, 512, 4, 28): enough waves for the NHWC backward apply pass to walk 2 / 2 / 4
# images per workgroup (BG > 1: the dWv partial rows and the per-image coefficients change inside a workgroup), with
# 1 / 8 / 4 strip-waves -- the oracle covers dx, do and every parameter gradient there.
STAGE_SHAPES = [(4, 256, 56, 56, 32), (4, 512, 28, 28, 32), (4, 1024, 14, 14, 32), (3, 2048, 7, 7, 32),
                (17, 256, 56, 56, 32), (2, 64, 1, 1, 32), (3, 64, 1, 3, 32), (2, 128, 3, 1, 32), (1, 256, 9, 9, 32),
                (128, 2048, 7, 7, 32), (128, 256, 3, 56, 32), (256, 512, 4, 28, 32)]


@pytest.mark.parametrize("shape", STAGE_SHAPES, ids=lambda s: "x".join(map(str, s[:4])))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_light_tail_resnet50_stage_shapes(shape, dtype, cl):
    b, c, h, w, d = shape
    from oracle import detgen
    s = detgen.seed_of(f"stage/{c}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.1 * detgen.normalish((b, c, h, w), s + 1)
    o = detgen.normalish((b, c, h, w), s + 2)
    gup = detgen.normalish((b, c, h, w), s + 3)
    P = cases.block_params(c, 7)
    mask = np.array(([1, 0, 1, 1] * ((b + 3) // 4))[:b], dtype=np.float32)
    if b >= 128:
        from mrla_amd import _lib as L
        rows = L.load().mrla_light_wgrad_rows(b, c, h, w, L.BF16 if dtype == torch.bfloat16 else L.F32, L.NHWC)
        assert 0 < rows < b, "this shape is meant to put several images into one workgroup of the NHWC apply_bwd pass"
    if dtype == torch.bfloat16:
        x, o, gup = bf16_round(x), bf16_round(o), bf16_round(gup)
    got = run_light(x, o, P, d, "train", mask, 0.2, gup, dtype, cl=cl)
    out, cache, g = oracle_light(x, o, P, d, "train", mask, 0.2, gup)
    # BatchNorm over a handful of values (b*h*w <= 9) is ill-conditioned: 1/sigma amplifies the fp32 input rounding
    tiny = b * h * w <= 9
    tol = TINY_BN_TOL if tiny else ACT_TOL
    if dtype == torch.float32:
        assert relmax(got["out"], out) < tol
        assert relmax(got["dx"], g["dx"]) < tol
        assert relmax(got["do"], g["do_prev"]) < tol
    elif b * h * w <= 9:
        for k, want in (("out", out), ("dx", g["dx"]), ("do", g["do_prev"])):
            assert relmax(got[k], want) < 2.0 ** -6, k
    else:
        assert_bf16_close(got["out"], out, "out")
        assert_bf16_close(got["dx"], g["dx"], "dx")
        assert_bf16_close(got["do"], g["do_prev"], "do")
    for ours, theirs in (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
                         ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta")):
        # bf16 I/O too: the oracle sees the same bf16-rounded inputs and the parameter gradients are fp32 sums, so the
        # kernel's fp32 accumulation is what is measured and the fp32 bounds hold (measured <= 3.5e-6; 8.2e-6 tiny-BN)
        ptol = TINY_BN_TOL if tiny else par_tol(theirs)
        assert relmax(got["grad/" + ours].ravel(), np.asarray(g[theirs]).ravel()) < ptol, ours


def test_fused_relu_add_producer_fp32_and_bf16():
    """pre-activation form: x_t = relu(pre + identity) formed inside the statistics kernel; the backward returns the
    gradient wrt `pre` and the total gradient wrt the identity (resnet_mrla_light.py:113-116 as one op)."""
    from mrla_amd.functional import mrla_light
    from oracle import detgen
    for dtype, (b, c, h, w, d), cl in ((torch.float32, (2, 256, 7, 5, 32), False), (torch.float32, (3, 256, 56, 56, 32), False),
                                       (torch.bfloat16, (3, 512, 28, 28, 32), False), (torch.bfloat16, (2, 2048, 7, 7, 32), False),
                                       (torch.float32, (2, 256, 7, 5, 32), True), (torch.float32, (3, 256, 56, 56, 32), True),
                                       (torch.bfloat16, (3, 512, 28, 28, 32), True), (torch.bfloat16, (2, 2048, 7, 7, 32), True),
                                       (torch.float32, (2, 64, 9, 17, 32), True)):
        fmt = torch.channels_last if cl else torch.contiguous_format
        s = detgen.seed_of(f"fuse/{c}/{h}")
        pre, o, gup = (detgen.normalish((b, c, h, w), s + i) for i in range(3))
        if dtype == torch.bfloat16:
            pre, o, gup = bf16_round(pre), bf16_round(o), bf16_round(gup)
        P = cases.block_params(c, 5)
        pt = to_dev(pre, dtype).contiguous(memory_format=fmt).requires_grad_(True)
        ot = to_dev(o, dtype).contiguous(memory_format=fmt).requires_grad_(True)
        prm = {k: to_dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
        rm, rv = to_dev(P["bn_mrla.running_mean"]), to_dev(P["bn_mrla.running_var"])
        out = mrla_light(pt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d, o_prev=ot,
                         lam=prm["mrla.lambda_t"], bn=dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"],
                                                           running_mean=rm, running_var=rv, training=True), res=True,
                         pre_activation=True)
        out.backward(to_dev(gup, dtype).contiguous(memory_format=fmt))
        x = pre.astype(np.float64) + o
        if dtype == torch.bfloat16:
            x = bf16_round(x).astype(np.float64)
        x = np.maximum(x, 0)
        want, cache, g = oracle_light(x, o, P, d, "train", None, 0.0, gup)
        dpre = g["dx"] * (x > 0)
        dot = g["do_prev"] + dpre
        got = [t.detach().float().cpu().numpy() for t in (out, pt.grad, ot.grad)]
        if dtype == torch.float32:
            assert relmax(got[0], want) < ACT_TOL and relmax(got[1], dpre) < ACT_TOL and relmax(got[2], dot) < ACT_TOL
        else:
            assert_bf16_close(got[0], want, "out"); assert_bf16_close(got[1], dpre, "dpre"); assert_bf16_close(got[2], dot, "do")
        assert relmax(prm["mrla.mrla.Wv.weight"].grad.cpu().numpy()[:, 0], g["dwv"]) < PAR_TOL


def test_layer_only_and_module_forms():
    """mrla_light_layer (a1) and light mrla_module (a2) through the same kernels, fp32."""
    from mrla_amd.functional import mrla_light
    name, b, c, h, w, d = cases.LIGHT_CASES[1]
    G = cases.golden("light_blocks")
    x, o, gup = cases.light_inputs(name, b, c, h, w)
    P = cases.block_params(c, 1)
    xt, ot = to_dev(x).requires_grad_(True), to_dev(o).requires_grad_(True)
    wq, wk, wv, lam = (to_dev(P[k]).requires_grad_(True) for k in
                       ("mrla.mrla.Wq.weight", "mrla.mrla.Wk.weight", "mrla.mrla.Wv.weight", "mrla.lambda_t"))
    layer = mrla_light(xt, wq, wk, wv, d)
    assert relmax(layer.detach().cpu().numpy(), G[f"{name}/train/layer_out"]) < GOLD_TOL
    m = mrla_light(xt, wq, wk, wv, d, o_prev=ot, lam=lam)
    assert relmax(m.detach().cpu().numpy(), G[f"{name}/train/m"]) < GOLD_TOL
    lo, cache = mn.light_layer_fwd(x.astype(np.float64), P["mrla.mrla.Wq.weight"].ravel().astype(np.float64),
                                   P["mrla.mrla.Wk.weight"].ravel().astype(np.float64),
                                   P["mrla.mrla.Wv.weight"][:, 0].astype(np.float64), d)
    gl = mn.light_layer_bwd(gup.astype(np.float64), cache)
    layer.backward(to_dev(gup))
    assert relmax(xt.grad.cpu().numpy(), gl["dx"]) < ACT_TOL
    assert relmax(wv.grad.cpu().numpy()[:, 0], gl["dwv"]) < PAR_TOL
    assert relmax(wq.grad.cpu().numpy().ravel(), gl["dwq"]) < QK_TOL


def test_cpu_tensor_is_rejected_loudly():
    from mrla_amd import _lib
    from mrla_amd.functional import mrla_light
    with pytest.raises(_lib.MrlaHipError):
        mrla_light(torch.zeros(1, 32, 4, 4), torch.zeros(1, 1, 3), torch.zeros(1, 1, 3), torch.zeros(32, 1, 3, 3), 32)


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("shape", [(3, 256, 14, 14, 32), (2, 192, 14, 14, 16)], ids=["resnet-stage", "deit-width"])
def test_fp16_storage_matches_fp32_arithmetic_on_the_same_values(shape, cl):
    """float16 activations (the reference's DeiT recipe trains under fp16 autocast, deit/engine.py:37): the kernels
    accumulate in fp32, so the fp16 path must equal the fp32 path run on the same fp16-representable inputs up to the
    rounding of the stored results (1 fp16 ulp = 2^-10)."""
    from mrla_amd.functional import mrla_light
    b, c, h, w, d = shape
    torch.manual_seed(2)
    fmt = torch.channels_last if cl else torch.contiguous_format
    mk = lambda *s: torch.randn(*s, device="cuda")
    x16, o16, g16 = (mk(b, c, h, w).half().contiguous(memory_format=fmt) for _ in range(3))
    k = 5
    wq, wk, wv, lam = mk(1, 1, k) * 0.5, mk(1, 1, k) * 0.5, mk(c, 1, 3, 3) * 0.3, mk(c, 1, 1)
    res = []
    for dt in (torch.float16, torch.float32):
        bn = torch.nn.BatchNorm2d(c).cuda()
        xt, ot = x16.detach().to(dt).clone().requires_grad_(True), o16.detach().to(dt).clone().requires_grad_(True)
        prm = [p.clone().requires_grad_(True) for p in (wq, wk, wv, lam)]
        out = mrla_light(xt, prm[0], prm[1], prm[2], d, o_prev=ot, lam=prm[3],
                         bn=dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var,
                                 training=True, momentum=0.1, eps=1e-5), res=True)
        out.backward(g16.to(dt))
        res.append([t.detach().float() for t in (out, xt.grad, ot.grad, prm[2].grad, prm[3].grad, bn.weight.grad)])
    for i, (a, r) in enumerate(zip(*res)):
        tol = 2.0 ** -9 if i < 3 else 2e-3          # stored fp16 tensors: 2 ulps; fp32 parameter gradients: input rounding only
        bad = (a - r).abs() > tol * (r.abs() + 0.05 * r.abs().max())
        assert bad.float().mean().item() < 1e-4, (i, (a - r).abs().max().item())


@pytest.mark.parametrize("shape", [(2, 64, 5, 100, 32), (1, 32, 3, 70, 16)], ids=["w100", "w70-c32"])
def test_wide_nchw_maps_run_through_one_internal_channels_last_conversion(shape):
    """The reference takes NCHW maps of any size (mmdet: 800x1333 images); the NCHW slab kernels need a plane row to fit
    one wave (W <= 64).  Wider NCHW inputs go through the NHWC kernels and come back NCHW-contiguous: same numbers, same
    memory contract (mrla_light_module.py:62-64,72 use .view on the result)."""
    from mrla_amd import _lib as L
    b, c, h, w, d = shape
    assert L.load().mrla_light_wgrad_rows(b, c, h, w, L.F32, L.NCHW) == L.EUNSUPPORTED
    from oracle import detgen
    s = detgen.seed_of(f"wide/{w}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.1 * detgen.normalish((b, c, h, w), s + 1)
    o, gup = detgen.normalish((b, c, h, w), s + 2), detgen.normalish((b, c, h, w), s + 3)
    P = cases.block_params(c, 7)
    got = run_light(x, o, P, d, "train", None, 0.0, gup)
    out, cache, g = oracle_light(x, o, P, d, "train", None, 0.0, gup)
    assert relmax(got["out"], out) < ACT_TOL and relmax(got["dx"], g["dx"]) < ACT_TOL
    assert relmax(got["do"], g["do_prev"]) < ACT_TOL
    assert relmax(got["grad/mrla.mrla.Wv.weight"][:, 0], g["dwv"]) < PAR_TOL
    from mrla_amd.functional import mrla_light
    y = mrla_light(to_dev(x), to_dev(P["mrla.mrla.Wq.weight"]), to_dev(P["mrla.mrla.Wk.weight"]), to_dev(P["mrla.mrla.Wv.weight"]), d)
    assert y.is_contiguous() and y.view(b, c // d, d, h, w).shape[1] == c // d


@pytest.mark.parametrize("cl,ratio", [(False, 30.0), (False, 1000.0), (True, 30.0), (True, 1000.0)],
                         ids=["nchw-30", "nchw-1000", "nhwc-30", "nhwc-1000"])
def test_closed_form_bn_mrla_statistics_with_offset_inputs(cl, ratio):
    """bn_mrla's batch statistics come in closed form from per-(image, channel) moments of V and o_{t-1}.  Both kernel
    families -- the NHWC row pipeline and (since round 4) the NCHW slab kernels that serve the reference's own layout
    contract (mrla_light_module.py:62-64) -- take those moments about per-plane pivots, so |mean| / sigma ~ 1e3 in o and,
    through the 3x3 taps, in V costs nothing (raw fp32 sums: error ~ eps * ratio^2, 10 % parameter gradients at 1e3).  The
    elementwise passes evaluate affine forms of V and o in fp32, which bounds every output at ~ eps * ratio."""
    from oracle import detgen
    b, c, h, w, d = 4, 64, 14, 14, 32
    s = detgen.seed_of("cond")
    x = np.maximum(detgen.normalish((b, c, h, w), s) + ratio, 0)
    o = detgen.normalish((b, c, h, w), s + 1) - ratio
    gup = detgen.normalish((b, c, h, w), s + 2)
    P = cases.block_params(c, 11)
    got = run_light(x, o, P, d, "train", None, 0.0, gup, cl=cl)
    out, cache, g = oracle_light(x, o, P, d, "train", None, 0.0, gup)
    tol = max(2e-4, 1.5e-6 * ratio)
    assert relmax(got["rv"], cache["bn"]["new_rv"]) < tol
    assert relmax(got["out"] - x, out - x) < tol          # the normalised branch (x itself is ~ratio)
    assert relmax(got["dx"], g["dx"]) < 3 * tol
    assert relmax(got["do"], g["do_prev"]) < 3 * tol
    # parameter gradients: the backward statistics pass takes sum dOut*V and sum dOut*o about the forward pivots
    # (mrla_light_stats_bwd's `mom`), so they hold to 1e-3 at ratio 1e3 as well (raw fp32 sums: 10 % there)
    ptol = 1e-3
    for ours, theirs in (("mrla.mrla.Wv.weight", "dwv"), ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"),
                         ("bn_mrla.bias", "dbeta")):
        assert relmax(got["grad/" + ours].ravel(), np.asarray(g[theirs]).ravel()) < ptol, ours
