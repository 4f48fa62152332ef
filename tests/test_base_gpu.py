"""GPU parity: HIP MRLA-base path (rings + softmax over depth, through the C ABI) vs the numpy oracle and the
reference's goldens (tests/golden/base_chains.npz).  Same tolerance protocol as tests/test_light_gpu.py."""
import numpy as np
import pytest
import torch

from oracle import detgen, mrla_numpy as mn
from tests import cases
from tests.test_light_gpu import ACT_TOL, GOLD_TOL, assert_bf16_close, bf16_round, par_tol, relmax, to_dev

pytestmark = pytest.mark.gpu

# whole models in fp32 (50+ layers: the per-op 1e-7 roundings add up).  vs the eager restatement on the same GPU (the very
# same MIOpen convolutions on both sides) and vs the reference's CPU logits (other convolution arithmetic on their side)
MODEL_TOL, MODEL_REF_TOL = 5e-6, 1e-5


def run_chain(xs, gups, params, d, training, dtype=torch.float32, hint=None, dp=None, cl=False):
    """Fused block tails of a whole stage on the GPU; returns per-layer outs / grads.  cl: channels_last activations
    and slot-major NHWC rings."""
    from mrla_amd import _lib as L, functional as Fm
    Tn = len(xs)
    b, c, h, w = xs[0].shape
    layout = L.NCHW
    if cl:
        probe = torch.empty((b, c, h, w), dtype=dtype, device="cuda").contiguous(memory_format=torch.channels_last)
        layout = Fm.BaseStage.layout_for(probe, d)
        if layout != L.NHWC:
            pytest.skip("shape not served by the NHWC MRLA-base kernels (the stage stays NCHW)")
    stage = Fm.BaseStage(b, c, h, w, d, dtype, torch.device("cuda"), hint or Tn, layout)
    xts, prms, rms, rvs, outs = [], [], [], [], []
    loss = 0.0
    for t in range(Tn):
        P = params[t]
        xt = to_dev(xs[t], dtype)
        if cl:
            xt = xt.contiguous(memory_format=torch.channels_last)
        xt.requires_grad_(True)
        prm = {k: to_dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
        rm, rv = to_dev(P["bn_mrla.running_mean"]), to_dev(P["bn_mrla.running_var"])
        out = Fm.mrla_base(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d, stage,
                           bn=dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"], running_mean=rm,
                                   running_var=rv, training=training, momentum=0.1, eps=1e-5),
                           dp=None if dp is None else to_dev(dp[t]))
        loss = loss + (out.float() * to_dev(gups[t])).sum()
        xts.append(xt); prms.append(prm); rms.append(rm); rvs.append(rv); outs.append(out)
    loss.backward()
    torch.cuda.synchronize()
    res = []
    for t in range(Tn):
        g = dict(out=outs[t].detach().float().cpu().numpy(), dx=xts[t].grad.float().cpu().numpy(),
                 rm=rms[t].cpu().numpy(), rv=rvs[t].cpu().numpy())
        for k, v in prms[t].items():
            g["grad/" + k] = v.grad.cpu().numpy()
        res.append(g)
    K, V = stage.views()
    return res, K.cpu().numpy(), V.float().cpu().numpy()


def oracle_chain(xs, gups, params, d, training, dp=None, rnd=None, rnd_dv=None):
    Tn = len(xs)
    K = V = None
    caches, outs = [], []
    for t in range(Tn):
        P = {k: np.asarray(v, np.float64) for k, v in params[t].items()}
        out, K, V, cache = mn.base_tail_fwd(
            np.asarray(xs[t], np.float64), P["mrla.mrla.Wq.weight"].ravel(), P["mrla.mrla.Wk.weight"].ravel(),
            P["mrla.mrla.Wv.weight"][:, 0], P["bn_mrla.weight"], P["bn_mrla.bias"], P["bn_mrla.running_mean"],
            P["bn_mrla.running_var"], d, K, V, training=training, dp=None if dp is None else dp[t],
            **({} if rnd is None else {"rnd": rnd}), **({} if rnd_dv is None else {"rnd_dv": rnd_dv}))
        caches.append(cache); outs.append(out)
    dK, dV = np.zeros_like(K), np.zeros_like(V)
    grads = [None] * Tn
    for t in reversed(range(Tn)):
        g = mn.base_tail_bwd(np.asarray(gups[t], np.float64), dK, dV, caches[t])
        grads[t] = g
        dK, dV = g["dK_prev"], g["dV_prev"]
    return outs, caches, grads, K, V


def assert_mostly_close(got, want, tol, max_bad_frac, what):
    want = np.asarray(want, np.float64)
    err = np.abs(np.asarray(got, np.float64) - want)
    bad = err > tol * (np.abs(want) + 0.05 * np.abs(want).max())
    assert bad.mean() <= max_bad_frac, f"{what}: {bad.sum()} of {bad.size} elements off by more than {tol:.2e} (rel)"
    l2 = np.linalg.norm(err) / np.linalg.norm(want)
    assert l2 < 2.0 ** -7, f"{what}: relative L2 error {l2:.3e}"


PAIRS = (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
         ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta"))


@pytest.mark.parametrize("case", cases.BASE_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_base_chain_fp32_vs_oracle_and_reference(case, mode, cl):
    name, b, c, h, w, d, Tn = case
    G = cases.golden("base_chains")
    xs, gups = zip(*[cases.base_inputs(name, t, b, c, h, w) for t in range(Tn)])
    params = [cases.block_params(c, 10 + t, light=False) for t in range(Tn)]
    got, K, V = run_chain(xs, gups, params, d, mode == "train", hint=2 if name == "chain5" else None, cl=cl)   # chain5: ring growth
    outs, caches, grads, Ko, Vo = oracle_chain(xs, gups, params, d, mode == "train")
    assert relmax(K, Ko) < ACT_TOL and relmax(V, Vo) < ACT_TOL
    assert relmax(K, G[f"{name}/{mode}/K"]) < GOLD_TOL and relmax(V, G[f"{name}/{mode}/V"]) < GOLD_TOL
    for t in range(Tn):
        key = f"{name}/{mode}/{t}/"
        assert relmax(got[t]["out"], outs[t]) < ACT_TOL, t
        assert relmax(got[t]["dx"], grads[t]["dx"]) < ACT_TOL, t
        assert relmax(got[t]["rv"], caches[t]["bn"]["new_rv"]) < ACT_TOL, t
        for ours, theirs in PAIRS:
            assert relmax(got[t]["grad/" + ours].ravel(), np.asarray(grads[t][theirs]).ravel()) < par_tol(ours), (t, ours)
        assert relmax(got[t]["out"], G[key + "out"]) < GOLD_TOL
        assert relmax(got[t]["dx"], G[key + "dx"]) < GOLD_TOL
        assert relmax(got[t]["grad/mrla.mrla.Wv.weight"], G[key + "grad/mrla.mrla.Wv.weight"]) < GOLD_TOL


@pytest.mark.parametrize("shape", [(3, 256, 56, 56, 16, 3), (2, 1024, 14, 14, 16, 6), (2, 2048, 7, 7, 16, 3),
                                   (2, 256, 7, 7, 1, 3),          # channel_wise_mrla (one head per channel)
                                   (2, 1024, 14, 14, 16, 23),     # resnet101_mrlab stage 3: the full 23-layer history
                                   (3, 512, 28, 28, 16, 4)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_base_chain_resnet_stage_shapes(shape, dtype, cl):
    b, c, h, w, d, Tn = shape
    xs, gups, dps = [], [], []
    for t in range(Tn):
        s = detgen.seed_of(f"bstage/{c}/{t}")
        x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.1 * detgen.normalish((b, c, h, w), s + 1)
        gu = detgen.normalish((b, c, h, w), s + 3)
        if dtype == torch.bfloat16:
            x, gu = bf16_round(x), bf16_round(gu)
        xs.append(x); gups.append(gu)
        dps.append(np.array(([1, 1, 0] * 2)[t % 2:t % 2 + b], dtype=np.float64) / 0.8)
    params = [cases.block_params(c, 20 + t, light=False) for t in range(Tn)]
    got, K, V = run_chain(xs, gups, params, d, True, dtype, dp=dps, cl=cl)
    if dtype == torch.float32:
        outs, caches, grads, Ko, Vo = oracle_chain(xs, gups, params, d, True, dp=dps)
        for t in range(Tn):
            assert relmax(got[t]["out"], outs[t]) < ACT_TOL, t
            assert relmax(got[t]["dx"], grads[t]["dx"]) < ACT_TOL, t
            for ours, theirs in PAIRS:
                assert relmax(got[t]["grad/" + ours].ravel(), np.asarray(grads[t][theirs]).ravel()) < par_tol(ours), (t, ours)
    else:
        # the bf16 path stores v_j, attn and dA_t in bf16 between kernels (as eager bf16 does): the oracle rounds at
        # the same three points (`rnd`), everything else stays fp64, so the ReLU masks agree and elementwise bounds hold
        rnd = lambda a: bf16_round(a).astype(np.float64)  # noqa: E731
        # the channels_last path also stores dV_t in bf16 between its two kernels (as autograd does for a bf16 V)
        outs, caches, grads, Ko, Vo = oracle_chain(xs, gups, params, d, True, dp=dps, rnd=rnd, rnd_dv=rnd if cl else None)
        for t in range(Tn):
            # a 1-ulp difference in a stored attn (fp32 vs fp64 accumulation, ~1e-5 of the elements) can still flip a
            # ReLU mask, so allow a 1e-4 fraction of outliers; everything else within 2 bf16 ulps of the tensor scale
            assert_mostly_close(got[t]["out"], outs[t], 2.0 ** -7, 1e-4, f"out[{t}]")
            assert_mostly_close(got[t]["dx"], grads[t]["dx"], 2.0 ** -6, 1e-4, f"dx[{t}]")
            for ours, theirs in PAIRS:
                # one flipped ReLU mask moves a channel's dWv / dgamma by one element's worth (b*h*w is only 392 here):
                # all but a 1e-3 fraction of the entries within 2 %, and the whole tensor within 2 % in L2
                a, r = got[t]["grad/" + ours].ravel().astype(np.float64), np.asarray(grads[t][theirs]).ravel()
                err = np.abs(a - r) / max(np.abs(r).max(), 1e-12)     # (dWq / dWk of the first layer are exactly zero)
                assert np.quantile(err, 0.999) < 2e-2 and err.max() < 0.1, (t, ours, err.max())
                assert np.linalg.norm(a - r) <= 2e-2 * np.linalg.norm(r) + 1e-12, (t, ours)


def test_bare_base_layer_api_matches_reference_attn():
    """mrla_base_layer.forward(x, prev_K, prev_V) -> (attn, K, V) as in mrla_base_module.py:54-89."""
    from mrla_amd.layers import mrla_base_layer
    name, b, c, h, w, d, Tn = cases.BASE_CASES[0]
    G = cases.golden("base_chains")
    K = V = None
    for t in range(3):
        lay = mrla_base_layer(c, dim_perhead=d, init_cell=(t == 0)).cuda()
        P = cases.block_params(c, 10 + t, light=False)
        lay.load_state_dict({k[len("mrla.mrla."):]: torch.from_numpy(v) for k, v in P.items() if k.startswith("mrla.mrla.")})
        x, _ = cases.base_inputs(name, t, b, c, h, w)
        attn, K, V = lay(to_dev(x), K, V)
        assert tuple(K.shape) == (b, t + 1, c) and tuple(V.shape) == (b, t + 1, c, h, w)
        assert relmax(attn.detach().cpu().numpy(), G[f"{name}/eval/{t}/attn"]) < GOLD_TOL


def test_resnet50_mrlab_logits_match_reference_and_eager():
    from mrla_amd import models
    from oracle import eager_models as em
    G = cases.golden("models")
    net = models.resnet50_mrlab().cuda()
    ref = em.eager_resnet50_mrlab().cuda()
    vals = detgen.fill_state_dict(net.state_dict())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    ref.load_state_dict(net.state_dict())
    net.eval(); ref.eval()
    x = torch.from_numpy(cases.image_batch(4)).cuda()
    with torch.no_grad():
        y, yr = net(x), ref(x)
    assert relmax(y.cpu().numpy(), yr.cpu().numpy()) < MODEL_TOL          # (measured 3.9e-7)
    assert relmax(y.cpu().numpy(), G["resnet50_mrlab/eval4/logits"]) < MODEL_REF_TOL   # (measured 6.3e-7)
    net.train(); ref.train()
    xb = torch.from_numpy(cases.image_batch(4, "img-train")).cuda()
    tgt = (torch.arange(4) * 37 % 1000).cuda()
    y, yr = net(xb), ref(xb)
    assert relmax(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 4 * MODEL_TOL   # train mode, 53+16 BatchNorms at batch 4 (measured 1.6e-6)
    torch.nn.functional.cross_entropy(y, tgt).backward()
    torch.nn.functional.cross_entropy(yr, tgt).backward()
    gp, gr = dict(net.named_parameters()), dict(ref.named_parameters())
    worst = (0.0, "")
    dots = np.zeros(3)
    for k in gp:
        a, b_ = gp[k].grad.cpu().numpy().ravel().astype(np.float64), gr[k].grad.cpu().numpy().ravel().astype(np.float64)
        if np.abs(b_).sum() < 1e-4:
            continue
        worst = max(worst, (np.abs(a - b_).sum() / np.abs(b_).sum(), k))
        dots += np.array([a @ b_, a @ a, b_ @ b_])
    # fp32, batch 4, 16 train-mode BNs + ReLU masks: per-parameter sums of tiny Wq/Wk gradients are noise-limited
    # (the chain tests above pin every gradient to 5e-5); the whole gradient must still point the same way
    assert worst[0] < 0.5, worst           # run-to-run noise of the tiny Wq/Wk sums alone reaches 0.25 (MIOpen atomics)
    assert dots[0] / np.sqrt(dots[1] * dots[2]) > 0.9999


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_repeated_backward_over_one_stage_restarts_the_gradient_rings(cl):
    """retain_graph=True / two losses over shared activations: every backward pass over a stage must start from zeroed dK
    and from the layers that take part in THIS pass (BaseStage.begin_layer_backward), so a second pass reproduces the first
    bit for bit and a pass that skips the deepest layer does not read its stale dA slot."""
    from mrla_amd import _lib as L, functional as Fm
    b, c, h, w, d, Tn = 2, 64, 6, 5, 16, 4
    fmt = torch.channels_last if cl else torch.contiguous_format
    stage = Fm.BaseStage(b, c, h, w, d, torch.float32, torch.device("cuda"), Tn, L.NHWC if cl else L.NCHW)
    xs, outs, prms = [], [], []
    for t in range(Tn):
        x, _ = cases.base_inputs("chain5", t, b, c, h, w)
        P = cases.block_params(c, 10 + t, light=False)
        xt = to_dev(x).contiguous(memory_format=fmt).requires_grad_(True)
        prm = {k: to_dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
        out = Fm.mrla_base(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d, stage,
                           bn=dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"],
                                   running_mean=to_dev(P["bn_mrla.running_mean"]), running_var=to_dev(P["bn_mrla.running_var"]),
                                   training=True, momentum=0.1, eps=1e-5))
        xs.append(xt); outs.append(out); prms.append(prm)
    gups = [to_dev(cases.base_inputs("chain5", t, b, c, h, w)[1]) for t in range(Tn)]
    leaves = xs + [v for p in prms for v in p.values()]

    def grads_of(loss, retain):
        for v in leaves:
            v.grad = None
        loss.backward(retain_graph=retain)
        return [v.grad.clone() if v.grad is not None else None for v in leaves]
    full = sum((o * g).sum() for o, g in zip(outs, gups))
    g1 = grads_of(full, True)
    g2 = grads_of(full, True)
    for a, r in zip(g1, g2):
        assert torch.equal(a, r)
    # a loss without the deepest layer: equals the same chain built with one layer less (oracle-checked elsewhere)
    part = sum((o * g).sum() for o, g in zip(outs[:-1], gups[:-1]))
    g3 = grads_of(part, True)
    g4 = grads_of(part, True)
    for a, r in zip(g3, g4):
        assert (a is None and r is None) or torch.equal(a, r)
    xs64 = [cases.base_inputs("chain5", t, b, c, h, w)[0] for t in range(Tn - 1)]
    params = [cases.block_params(c, 10 + t, light=False) for t in range(Tn - 1)]
    _, _, og, _, _ = oracle_chain(xs64, [g.cpu().numpy() for g in gups[:-1]], params, d, True)
    for t in range(Tn - 1):
        assert relmax(g3[t].cpu().numpy(), og[t]["dx"]) < ACT_TOL, t
    # a pass whose deepest layer is SHALLOWER than the previous pass's last layer (t decreases across the pass boundary:
    # the ordering heuristic alone would take it for a continuation): passes are told apart by the autograd graph-task id
    shallow = sum((o * g).sum() for o, g in zip(outs[:2], gups[:2]))
    grads_of(part, True)                                   # ends at layer 1 ...
    g5 = grads_of(shallow, True)                           # ... and this one starts at layer 2 of 4
    _, _, og2, _, _ = oracle_chain(xs64[:2], [g.cpu().numpy() for g in gups[:2]], params[:2], d, True)
    for t in range(2):
        assert relmax(g5[t].cpu().numpy(), og2[t]["dx"]) < ACT_TOL, t
    # and directly after a partial pass that stopped at layer 3 (torch.autograd.grad down to x_3 only)
    torch.autograd.grad(full, [xs[2]], retain_graph=True)
    g6 = grads_of(shallow, True)
    for a, r in zip(g5, g6):
        assert (a is None and r is None) or torch.equal(a, r)


def test_wide_nchw_base_stage_goes_through_nhwc_rings():
    """NCHW maps wider than a wave (W > 64) on MRLA-base: the stage picks the slot-major NHWC rings, converts the input
    once per layer, and hands NCHW-contiguous results back (resnet_mrla_base.py accepts any map size)."""
    b, c, h, w, d, Tn = 1, 64, 3, 70, 16, 3
    xs, gups = zip(*[cases.base_inputs("wide", t, b, c, h, w) for t in range(Tn)])
    params = [cases.block_params(c, 30 + t, light=False) for t in range(Tn)]
    from mrla_amd import _lib as L, functional as Fm
    probe = torch.empty((b, c, h, w), device="cuda")
    assert Fm.BaseStage.layout_for(probe, d) == L.NHWC
    stage = Fm.BaseStage(b, c, h, w, d, torch.float32, torch.device("cuda"), Tn, L.NHWC)
    outs, xts = [], []
    for t in range(Tn):
        P = params[t]
        xt = to_dev(xs[t]).requires_grad_(True)
        out = Fm.mrla_base(xt, to_dev(P["mrla.mrla.Wq.weight"]), to_dev(P["mrla.mrla.Wk.weight"]), to_dev(P["mrla.mrla.Wv.weight"]),
                           d, stage, bn=dict(weight=to_dev(P["bn_mrla.weight"]), bias=to_dev(P["bn_mrla.bias"]),
                                             running_mean=to_dev(P["bn_mrla.running_mean"]),
                                             running_var=to_dev(P["bn_mrla.running_var"]), training=True))
        assert out.is_contiguous()
        outs.append(out); xts.append(xt)
    sum((o * to_dev(g)).sum() for o, g in zip(outs, gups)).backward()
    want, _, grads, _, _ = oracle_chain(xs, gups, params, d, True)
    for t in range(Tn):
        assert relmax(outs[t].detach().cpu().numpy(), want[t]) < ACT_TOL, t
        assert relmax(xts[t].grad.cpu().numpy(), grads[t]["dx"]) < ACT_TOL, t


@pytest.mark.parametrize("shape", [(3, 64, 6, 9), (4, 256, 14, 14), (2, 1024, 14, 14), (8, 128, 28, 28)],
                         ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_deferred_bn3_backward_sums_ride_in_the_base_value_backward(shape, dtype):
    """resnet_mrla_base.py:103-104,120-127 on an NHWC stage: with bn3's affine deferred into the MRLA-base pooling pass, its
    BACKWARD statistics (sum dpre, sum dpre * (y3 - mean)) are taken inside mrla_base_value_bwd_dv -- the kernel that forms
    dpre -- instead of by a 2N pass of mrla_bn_plane_dmoments (33 launches fewer per resnet101_mrlab step).  Two layers of a
    stage, both with their own bn3; compared with the same chain where the hand-over is refused (the box emptied between
    the kernels, so bn3's backward runs its own statistics pass): outputs and every other gradient bit-identical, bn3's
    parameter gradients and the constants of its input gradient equal up to fp32 summation order."""
    from mrla_amd import _lib as L, functional as Fm
    b, c, h, w = shape
    d = 16
    torch.manual_seed(9)
    mk = lambda *s: torch.randn(*s, device="cuda")                                             # noqa: E731
    cl = lambda t: t.contiguous(memory_format=torch.channels_last)                              # noqa: E731
    probe = cl(torch.empty((b, c, h, w), dtype=dtype, device="cuda"))
    assert Fm.BaseStage.layout_for(probe, d) == L.NHWC
    dt = L.BF16 if dtype == torch.bfloat16 else L.F32
    assert L.load().mrla_base_value_bwd_pre_sums(b, c, h, w, dt, L.NHWC) == 1
    k = 3 if c == 64 else 5
    convs = [cl(mk(b, c, h, w).to(dtype)) for _ in range(2)]
    idn, gs = cl(mk(b, c, h, w).to(dtype)), [cl(mk(b, c, h, w).to(dtype)) for _ in range(2)]
    wts = [(mk(1, 1, k) * 0.5, mk(1, 1, k) * 0.5, mk(c, 1, 3, 3) * 0.3) for _ in range(2)]
    results, launches = [], []
    for handed in (False, True):
        stage = Fm.BaseStage(b, c, h, w, d, dtype, torch.device("cuda"), 2, L.NHWC)
        bn3s = [torch.nn.BatchNorm2d(c).cuda() for _ in range(2)]
        bnms = [torch.nn.BatchNorm2d(c).cuda() for _ in range(2)]
        with torch.no_grad():
            for bn in bn3s + bnms:
                bn.weight.copy_(torch.linspace(0.5, 1.5, c)); bn.bias.copy_(torch.linspace(-0.3, 0.3, c))
        xins = [t.clone().requires_grad_(True) for t in convs]
        oin = idn.clone().requires_grad_(True)
        prms = [[p.clone().requires_grad_(True) for p in wt] for wt in wts]
        loss, prev, outs = 0.0, oin, []
        for t in range(2):
            pre = Fm.bn_act(xins[t], bn3s[t], relu=False, defer=True)
            assert getattr(pre, "_mrla_bn_box", None) is not None
            if not handed:
                pre._mrla_bn_box = None              # no box: the value backward takes no sums, bn3 runs its own pass
            out = Fm.mrla_base(pre, prms[t][0], prms[t][1], prms[t][2], d, stage,
                               bn=dict(weight=bnms[t].weight, bias=bnms[t].bias, running_mean=bnms[t].running_mean,
                                       running_var=bnms[t].running_var, training=True, momentum=0.1, eps=1e-5),
                               identity=prev)
            outs.append(out)
            loss = loss + (out.float() * gs[t].float()).sum()
            prev = out
        timer = Fm.KernelTimer(["mrla_bn_plane_dmoments", "mrla_base_value_bwd_dv"])
        Fm.TIMER = timer
        try:
            loss.backward()
        finally:
            Fm.TIMER = None
        torch.cuda.synchronize()
        launches.append(timer.summary())
        results.append([o.detach() for o in outs] + [x.grad for x in xins] + [oin.grad]
                       + [g for bn in bn3s for g in (bn.weight.grad, bn.bias.grad)]
                       + [bn.weight.grad for bn in bnms] + [p.grad for pr in prms for p in pr])
    assert launches[0]["mrla_bn_plane_dmoments"]["launches"] == 2 and "mrla_bn_plane_dmoments" not in launches[1]
    assert launches[1]["mrla_base_value_bwd_dv"]["launches"] == 2
    for i, (a, bb) in enumerate(zip(*results)):
        if i in (5, 6, 7, 8):                    # bn3.weight.grad / bn3.bias.grad of the two layers
            assert ((a - bb).abs().max() / bb.abs().max()).item() < 1e-3, i          # VERDICT bound; measured ~1e-6
            cases.record(((a - bb).abs().max() / bb.abs().max()).item(), depth=1)
        elif i in (2, 3):                        # gradient wrt conv3's output: e*dpre + f*y3 + h with those constants
            af, bf_ = a.float(), bb.float()
            unit = (2.0 ** -7 if dtype == torch.bfloat16 else 1e-5) * (bf_.abs() + 0.05 * bf_.abs().max())
            assert ((af - bf_).abs() <= unit).all(), i
            assert (af != bf_).float().mean().item() < (1e-3 if dtype == torch.bfloat16 else 1.0), i
        else:
            assert torch.equal(a, bb), i
