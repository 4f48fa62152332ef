"""GPU: the row pipeline with an image's ROWS cut into ranges (light_nhwc_wide.h: RowCut; the geometry a detection batch gets --
few, very large images would otherwise leave most CUs without a workgroup) computes what the uncut kernels compute.

`mrla_tuning_row_ranges(2)` cuts wherever an image has >= 16 rows, so that small shapes -- the ones the fp64 oracle finishes in
seconds -- run through the cut instances of every pass:
  * pass by pass through the C ABI, same coefficient tensors in both modes: what a pass WRITES per pixel (x_t, out, dx, do)
    must be BIT-IDENTICAL -- a range only re-fetches its halo rows and re-computes one row of dU --; what it SUMS (moment
    records, dWv / bn3 partial rows) changes its order of summation only: relative 2e-6 of the largest entry (fp32 sums of
    <= 10^5 terms);
  * the whole tail (mrla_amd.functional.mrla_light, forward + backward) against the fp64 oracle at the bounds of the uncut
    tests (tests/test_light_gpu.py), on whole and ragged strips, row counts that do not divide, fp32 / bf16;
  * the bottleneck with the deferred bn3 (resnet_mrla_light.py:100-116) against itself uncut.
The detection-size test (tests/test_det_backbone_gpu.py) runs the default mode, where the real shapes are cut."""
import ctypes

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture
def row_ranges():
    """Yields a setter for mrla_tuning_row_ranges; the default mode comes back afterwards."""
    from mrla_amd import _lib as L
    lib = L.load()
    try:
        yield lib.mrla_tuning_row_ranges
    finally:
        lib.mrla_tuning_row_ranges(0)


P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731


def _run_passes(lib, L, mode, T, coef=None):
    """Every streaming pass of the tail once, through the C ABI, with the row-range mode `mode`.  `coef`: the per-(image, channel)
    coefficient tensors of another run (gate, BatchNorm scale / shift, backward coefficients, dyx) -- with them the apply
    passes see exactly the same inputs in both modes."""
    b, c, h, w, d, dt = T["shape"]
    lib.mrla_tuning_row_ranges(mode)
    lay, st = L.NHWC, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    q = lambda f: f(b, c, h, w, dt, lay)                                      # noqa: E731
    msplits, bsplits, rows = q(lib.mrla_light_mom_splits), q(lib.mrla_light_bmom_splits), q(lib.mrla_light_wgrad_rows)
    assert min(msplits, bsplits, rows) >= 1
    f32 = dict(device="cuda", dtype=torch.float32)
    x, o, g, y3 = T["x"], T["o"], T["g"], T["y3"]
    ks = T["wq"].numel()
    R = dict(counts=(msplits, bsplits, rows))
    ok = lambda rc, what: L.check(rc, what)                                   # noqa: E731
    # forward statistics: x given, and the fused producer (x_t = relu(psc*y3 + psh + o) formed and stored)
    mom = torch.full((msplits, b, c, L.FWD_MOMENTS), float("nan"), **f32)
    ok(lib.mrla_light_stats_fwd(P(x), P(o), P(T["wv"]), P(mom), b, c, h, w, dt, lay, 0, st), "stats_fwd")
    R["mom"] = mom[0].clone()
    momf = torch.full((msplits, b, c, L.FWD_MOMENTS), float("nan"), **f32)
    xt = torch.empty_like(x)
    ok(lib.mrla_light_stats_fwd_fused(P(y3), P(T["psc"]), P(T["psh"]), P(o), P(T["wv"]), P(momf), P(xt), b, c, h, w, dt, lay, st),
       "stats_fwd_fused")
    R["momf"], R["xt"] = momf[0].clone(), xt
    if coef is None:
        gate = torch.empty(b, c // d, **f32)
        bn = torch.empty(4, c, **f32)
        rm, rv = torch.zeros(c, **f32), torch.ones(c, **f32)
        ok(lib.mrla_light_gate_fwd(P(mom), P(T["wq"]), P(T["wk"]), ks, P(gate), b, c, h * w, d, st), "gate_fwd")
        ok(lib.mrla_light_bn_fwd(P(mom), P(gate), P(T["lam"]), P(T["gamma"]), P(T["beta"]), P(rm), P(rv), 1, 0.1, 1e-5, P(bn[0]),
                                 P(bn[1]), P(bn[2]), P(bn[3]), b, c, h * w, d, st), "bn_fwd")
    else:
        gate, bn = coef["gate"], coef["bn"]
    out, outf = torch.empty_like(x), torch.empty_like(x)
    ok(lib.mrla_light_apply_fwd(P(x), P(o), P(T["wv"]), P(gate), P(bn[0]), P(bn[1]), P(T["lam"]), P(T["dp"]), P(out), b, c, h, w, d, 1,
                                dt, lay, 0, st), "apply_fwd")
    ok(lib.mrla_light_apply_fwd_fused(P(y3), P(T["psc"]), P(T["psh"]), P(o), P(T["wv"]), P(gate), P(bn[0]), P(bn[1]), P(T["lam"]),
                                      P(T["dp"]), P(outf), b, c, h, w, d, 1, dt, lay, st), "apply_fwd_fused")
    R["out"], R["outf"] = out, outf
    # backward statistics (about the pivots of `mom`: the first run's record in both modes)
    mom_b = mom if coef is None else coef["mom"]
    bmom = torch.full((bsplits, b, c, 3), float("nan"), **f32)
    ok(lib.mrla_light_stats_bwd(P(g), P(x), P(o), P(T["wv"]), P(mom_b), P(bmom), b, c, h, w, dt, lay, 0, st), "stats_bwd")
    R["bmom"] = bmom[0].clone()                   # (the pass folds its partial records into record 0)
    if coef is None:
        cb, small = torch.empty(c, 4, **f32), torch.empty(3, c, **f32)
        dyx, dwqk = torch.empty(b, c, **f32), torch.empty(b, 2 * ks, **f32)
        ok(lib.mrla_light_bn_bwd(P(mom), P(bmom), P(gate), P(T["lam"]), P(T["gamma"]), P(T["dp"]), P(bn[2]), P(bn[3]), 1, P(cb), None,
                                 P(small[0]), P(small[1]), P(small[2]), b, c, h * w, d, st), "bn_bwd")
        ok(lib.mrla_light_gate_bwd(P(mom), P(bmom), P(gate), P(cb), None, P(T["dp"]), P(T["wq"]), P(T["wk"]), ks, P(dyx), P(dwqk), b, c,
                                   h * w, d, st), "gate_bwd")
        R["coef"] = dict(gate=gate, bn=bn, cb=cb, dyx=dyx, mom=mom)
    else:
        cb, dyx = coef["cb"], coef["dyx"]
    # backward apply: plain (x given, no mask), and the fused producer's (relu mask; 16-bit types: bn3's sums folded in)
    for relu, pre in ((0, False), (1, False), (1, True)):
        if pre and dt == L.F32:
            continue
        dx, do = torch.empty_like(x), torch.empty_like(x)
        dwv = torch.full((rows, c * 9), float("nan"), **f32)
        tm = torch.full((rows, c, 2), float("nan"), **f32) if pre else None
        ok(lib.mrla_light_apply_bwd(P(g), P(xt if relu else x), P(o), P(T["wv"]), P(gate), P(cb), P(T["lam"]), P(T["dp"]), P(dyx), P(dx),
                                    P(do), P(dwv), P(y3) if pre else None, P(T["center"]) if pre else None, P(tm), b, c, h, w, d, 1,
                                    relu, dt, lay, 0, st), "apply_bwd")
        key = f"bwd{relu}{int(pre)}"
        R[key + "/dx"], R[key + "/do"], R[key + "/dwv"] = dx, do, dwv.double().sum(0)
        if pre:
            R[key + "/tmom"] = tm.double().sum(0)
    torch.cuda.synchronize()
    return R


# (the last one: workgroups of the backward apply pass that walk TWO images each -- mrla_light_wgrad_rows < b uncut)
SHAPES = [(2, 64, 37, 45, 32), (3, 128, 16, 9, 32), (1, 256, 50, 23, 32), (2, 512, 24, 28, 32), (2, 64, 33, 7, 16),
          (128, 512, 28, 28, 32)]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16], ids=["bf16", "fp32", "fp16"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s[:4])))
def test_every_pass_cut_against_uncut_through_the_c_abi(shape, dtype, row_ranges):
    from mrla_amd import _lib as L
    lib = L.load()
    b, c, h, w, d = shape
    dt = {torch.bfloat16: L.BF16, torch.float32: L.F32, torch.float16: L.F16}[dtype]
    g = torch.Generator(device="cuda").manual_seed(b * 1000 + c + h)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)             # noqa: E731
    act = lambda: rnd(b, c, h, w).to(dtype).contiguous(memory_format=torch.channels_last)      # noqa: E731
    T = dict(shape=(b, c, h, w, d, dt), x=act(), o=act(), g=act(), y3=act(), wv=rnd(c, 9) * 0.3, wq=rnd(5), wk=rnd(5), lam=rnd(c),
             gamma=rnd(c).abs() + 0.5, beta=rnd(c) * 0.1, psc=rnd(c).abs() + 0.3, psh=rnd(c) * 0.2, center=rnd(c) * 0.1,
             dp=torch.tensor(([1.25, 0.0, 1.25] * b)[:b], device="cuda"))
    if b >= 128 and dtype != torch.bfloat16:
        pytest.skip("the large shape runs once")
    A = _run_passes(lib, L, 1, T)
    if b >= 128:
        assert A["counts"][2] < b, "this shape is meant to put several images into one workgroup of apply_bwd"
    Bc = _run_passes(lib, L, 2, T, coef=A["coef"])
    # the rows really were cut -- more partial records / rows than the uncut launch has
    assert all(n2 > n1 for n1, n2 in zip(A["counts"], Bc["counts"])), (A["counts"], Bc["counts"])
    for k in A:
        if k in ("counts", "coef"):
            continue
        a, bb = A[k], Bc[k]
        assert torch.isfinite(bb.float()).all(), k
        if k in ("xt", "out", "outf") or k.endswith("/dx") or k.endswith("/do"):
            assert torch.equal(a, bb), (k, float((a.float() - bb.float()).abs().max()))
        else:           # sums: another order of summation
            err = float((a.double() - bb.double()).abs().max() / a.double().abs().max().clamp_min(1e-30))
            assert err < 2e-6, (k, err)


@pytest.mark.parametrize("act", [0, 1], ids=["linear", "gelu"])
@pytest.mark.parametrize("shape", [(2, 128, 37, 30, 32), (1, 64, 24, 9, 16)], ids=lambda s: "x".join(map(str, s[:4])))
def test_layer_form_without_o_prev_cut_against_uncut(shape, act, row_ranges):
    """The passes as mrla_light_layer uses them (mrla_light_module.py:52-74: no o_prev, no BatchNorm; act = 1: the GELU on V of
    the DeiT form on an image) -- the other template instances of the cut kernels."""
    from mrla_amd import _lib as L
    lib = L.load()
    b, c, h, w, d = shape
    dt, lay, st = L.F32, L.NHWC, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    gen = torch.Generator(device="cuda").manual_seed(c + h + act)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=gen)           # noqa: E731
    x, g = (rnd(b, c, h, w).contiguous(memory_format=torch.channels_last) for _ in range(2))
    wv, wq, wk = rnd(c, 9) * 0.3, rnd(3), rnd(3)
    f32 = dict(device="cuda", dtype=torch.float32)

    def run(mode, coef=None):
        lib.mrla_tuning_row_ranges(mode)
        q = lambda f: f(b, c, h, w, dt, lay)                                  # noqa: E731
        ms, bs, rows = q(lib.mrla_light_mom_splits), q(lib.mrla_light_bmom_splits), q(lib.mrla_light_wgrad_rows)
        mom = torch.full((ms, b, c, L.FWD_MOMENTS), float("nan"), **f32)
        L.check(lib.mrla_light_stats_fwd(P(x), None, P(wv), P(mom), b, c, h, w, dt, lay, act, st), "stats_fwd")
        if coef is None:
            gate = torch.empty(b, c // d, **f32)
            L.check(lib.mrla_light_gate_fwd(P(mom), P(wq), P(wk), 3, P(gate), b, c, h * w, d, st), "gate_fwd")
        else:
            gate = coef["gate"]
        out = torch.empty_like(x)
        L.check(lib.mrla_light_apply_fwd(P(x), None, P(wv), P(gate), None, None, None, None, P(out), b, c, h, w, d, 0, dt, lay, act, st),
                "apply_fwd")
        bmom = torch.full((bs, b, c, 3), float("nan"), **f32)
        L.check(lib.mrla_light_stats_bwd(P(g), P(x), None, P(wv), P(mom if coef is None else coef["mom"]), P(bmom), b, c, h, w, dt, lay,
                                         act, st), "stats_bwd")
        if coef is None:
            dyx, dwqk = torch.empty(b, c, **f32), torch.empty(b, 6, **f32)
            L.check(lib.mrla_light_gate_bwd(P(mom), P(bmom), P(gate), None, None, None, P(wq), P(wk), 3, P(dyx), P(dwqk), b, c, h * w, d,
                                            st), "gate_bwd")
        else:
            dyx = coef["dyx"]
        dx = torch.empty_like(x)
        dwv = torch.full((rows, c * 9), float("nan"), **f32)
        L.check(lib.mrla_light_apply_bwd(P(g), P(x), None, P(wv), P(gate), None, None, None, P(dyx), P(dx), None, P(dwv), None, None,
                                         None, b, c, h, w, d, 0, 0, dt, lay, act, st), "apply_bwd")
        torch.cuda.synchronize()
        return dict(counts=(ms, bs, rows), mom=mom[0].clone(), out=out, bmom=bmom[0].clone(), dx=dx, dwv=dwv.double().sum(0),
                    coef=dict(gate=gate, dyx=dyx, mom=mom))
    A = run(1)
    Bc = run(2, coef=A["coef"])
    assert all(n2 > n1 for n1, n2 in zip(A["counts"], Bc["counts"])), (A["counts"], Bc["counts"])
    for k in ("out", "dx"):
        assert torch.equal(A[k], Bc[k]), (k, float((A[k] - Bc[k]).abs().max()))
    for k in ("mom", "bmom", "dwv"):
        err = float((A[k].double() - Bc[k].double()).abs().max() / A[k].double().abs().max().clamp_min(1e-30))
        assert err < 2e-6, (k, err)


CUT_CASES = [(4, 256, 56, 56, 32), (4, 512, 28, 28, 32), (17, 256, 56, 56, 32), (2, 64, 37, 45, 32), (3, 128, 16, 9, 16),
             (1, 192, 50, 23, 32)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", CUT_CASES, ids=lambda s: "x".join(map(str, s[:4])))
def test_light_tail_with_cut_rows_vs_oracle(shape, dtype, row_ranges):
    """tests/test_light_gpu.py's stage-shape test (train-mode BatchNorm, stochastic depth, channels_last, all parameter gradients,
    fp64 oracle, the same bounds) with every pass cut into row ranges."""
    from mrla_amd import _lib as L
    from tests import test_light_gpu as tl
    lib = L.load()
    b, c, h, w, d = shape
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    row_ranges(1)
    uncut = lib.mrla_light_wgrad_rows(b, c, h, w, dt, L.NHWC)
    row_ranges(2)
    assert lib.mrla_light_wgrad_rows(b, c, h, w, dt, L.NHWC) > uncut and lib.mrla_light_mom_splits(b, c, h, w, dt, L.NHWC) > 1
    if shape in tl.STAGE_SHAPES:
        tl.test_light_tail_resnet50_stage_shapes(shape, dtype, True)
        return
    from oracle import detgen
    s = detgen.seed_of(f"cut/{c}/{h}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.1 * detgen.normalish((b, c, h, w), s + 1)
    o, gup = detgen.normalish((b, c, h, w), s + 2), detgen.normalish((b, c, h, w), s + 3)
    prm = cases.block_params(c, 9)
    if dtype == torch.bfloat16:
        x, o, gup = tl.bf16_round(x), tl.bf16_round(o), tl.bf16_round(gup)
    got = tl.run_light(x, o, prm, d, "train", None, 0.0, gup, dtype, cl=True)
    out, cache, gr = tl.oracle_light(x, o, prm, d, "train", None, 0.0, gup)
    for k, want in (("out", out), ("dx", gr["dx"]), ("do", gr["do_prev"])):
        if dtype == torch.float32:
            assert tl.relmax(got[k], want) < tl.ACT_TOL, k
        else:
            tl.assert_bf16_close(got[k], want, k)
    for ours, theirs in (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
                         ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta")):
        assert tl.relmax(got["grad/" + ours].ravel(), np.asarray(gr[theirs]).ravel()) < tl.par_tol(theirs), ours


def test_fused_producer_with_cut_rows_vs_oracle(row_ranges):
    """x_t = relu(pre + identity) formed inside the cut statistics pass; dpre and the total identity gradient (fp32, bf16)."""
    from tests import test_light_gpu as tl
    row_ranges(2)
    tl.test_fused_relu_add_producer_fp32_and_bf16()


@pytest.mark.parametrize("shape", [(3, 64, 56, 56), (2, 128, 28, 28), (2, 64, 40, 23)], ids=lambda s: "x".join(map(str, s)))
def test_bottleneck_with_deferred_bn3_cut_against_uncut(shape, row_ranges):
    """One MRLA_Bottleneck (deferred bn3 -> fused tail, bn3's backward sums folded into the cut apply pass; sequence entry points),
    bf16 autocast, train mode: forward + backward with cut rows against the same block uncut.  The stock convolutions around the
    tail see inputs that differ in last bits (the gate's sums are taken in another order), so: to bf16 resolution."""
    from mrla_amd import _lib as L, resnet
    lib = L.load()
    b, planes, h, w = shape
    c = planes * 4
    torch.manual_seed(21)
    blk = resnet.MRLA_Bottleneck(c, planes, drop_path=0.0).cuda().to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        blk.bn3.weight.uniform_(0.3, 1.2)
        blk.bn3.bias.uniform_(-0.2, 0.2)
        blk.mrla.lambda_t.normal_()
    state = {k: v.clone() for k, v in blk.state_dict().items()}
    gen = torch.Generator(device="cuda").manual_seed(4)
    x0 = torch.randn(b, c, h, w, device="cuda", generator=gen).bfloat16().contiguous(memory_format=torch.channels_last)
    gup = torch.randn(b, c, h, w, device="cuda", generator=gen).bfloat16().contiguous(memory_format=torch.channels_last)
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True

    def run(mode):
        row_ranges(mode)
        blk.load_state_dict(state)
        blk.zero_grad(set_to_none=True)
        xt = x0.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(xt)
        out.backward(gup)
        r = dict(out=out.detach().float(), dx=xt.grad.float())
        r.update({"grad:" + k: p.grad.float() for k, p in blk.named_parameters()})
        r.update({"buf:" + k: v.clone().float() for k, v in blk.named_buffers() if v.is_floating_point()})
        return r
    try:
        a = run(1)
        rows_uncut = lib.mrla_light_wgrad_rows(b, c, h, w, L.BF16, L.NHWC)
        s = run(2)
        assert lib.mrla_light_wgrad_rows(b, c, h, w, L.BF16, L.NHWC) > rows_uncut
    finally:
        torch.backends.cudnn.deterministic = det
    for k in a:
        err = float((a[k] - s[k]).norm() / a[k].norm().clamp_min(1e-20))
        assert err < (4e-3 if k in ("out", "dx") else 1e-2), (k, err)       # (bf16: 2^-8 per element)
    assert float(s["grad:bn3.weight"].abs().max()) > 0


@pytest.mark.parametrize("arch", ["resnet50_mrlal", "resnet50_mrlab"])
def test_whole_models_with_cut_rows_track_the_eager_restatement(arch, row_ranges):
    """tests/test_models_gpu.py's five-SGD-steps and bf16-autocast train-step tests with the rows cut wherever a map has >= 16
    rows (stages 1 and 2 at these batches): the light model end to end, and the MRLA-base model, whose value backward walks whole
    images and zeroes the partial rows of the row ranges it does not write."""
    from tests import test_models_gpu as tm
    row_ranges(2)
    tm.test_five_sgd_steps_track_the_eager_restatement(arch)
    tm.test_bf16_autocast_train_step_tracks_the_eager_restatement(arch)


def test_modes_and_counts(row_ranges):
    """0: cuts only where the launch would leave CUs idle AND the tensor is large (ranges of >= 12 rows); 1: never; 2: wherever >= 16 rows.  Off the row
    pipeline (C % 64 != 0, NCHW) nothing is ever cut; the x_t-free tail reports itself unavailable where rows are cut."""
    from mrla_amd import _lib as L
    lib = L.load()
    q = lambda f, s, lay=L.NHWC: f(*s, L.BF16, lay)                           # noqa: E731
    small, det = (4, 256, 56, 56), (2, 512, 100, 168)
    assert q(lib.mrla_light_mom_splits, small) == 1 and q(lib.mrla_light_lean_supported, small) == 1
    assert q(lib.mrla_light_mom_splits, det) == 24 and q(lib.mrla_light_lean_supported, det) == 0
    assert row_ranges(1) == 0
    assert q(lib.mrla_light_mom_splits, det) == 3 and q(lib.mrla_light_lean_supported, det) == 1
    assert row_ranges(2) == 1
    assert q(lib.mrla_light_mom_splits, small) == 7 and q(lib.mrla_light_lean_supported, small) == 0
    assert q(lib.mrla_light_mom_splits, (4, 96, 56, 56)) == 1 and q(lib.mrla_light_mom_splits, small, L.NCHW) == 1
    assert q(lib.mrla_light_mom_splits, (4, 256, 15, 56)) == 1                 # fewer than 16 rows
    assert row_ranges(5) == L.EINVAL and row_ranges(0) == 2
