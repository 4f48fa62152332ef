"""GPU parity: HIP token (DeiT) MRLA-light path vs the numpy oracle and the reference's goldens."""
import numpy as np
import pytest
import torch

from oracle import detgen, eager_models as em, mrla_numpy as mn
from tests import cases
from tests.test_light_gpu import ACT_TOL, GOLD_TOL, PAR_TOL, QK_TOL, assert_bf16_close, bf16_round, par_tol, relmax, to_dev

pytestmark = pytest.mark.gpu

NAMES = ("normx.weight", "normx.bias", "normo.weight", "normo.bias", "mrla.Wq.weight", "mrla.Wk.weight", "mrla.Wv.weight",
         "lambda_t")
ORACLE = ("dlnx_w", "dlnx_b", "dlno_w", "dlno_b", "dwq", "dwk", "dwv", "dlam")


def run(x, o, P, d, gup, dtype, res):
    from mrla_amd.functional import mrla_token_light
    xt, ot = to_dev(x, dtype).requires_grad_(True), to_dev(o, dtype).requires_grad_(True)
    prm = [to_dev(P[k]).requires_grad_(True) for k in NAMES]
    out = mrla_token_light(xt, ot, *prm, d, eps=1e-6, res=res)
    out.backward(to_dev(gup, dtype))
    torch.cuda.synchronize()
    return (out.detach().float().cpu().numpy(), xt.grad.float().cpu().numpy(), ot.grad.float().cpu().numpy(),
            [p.grad.cpu().numpy() for p in prm])


def oracle(x, o, P, d, gup, res):
    P = {k: np.asarray(v, np.float64) for k, v in P.items()}
    out, cache = mn.token_light_fwd(np.asarray(x, np.float64), np.asarray(o, np.float64), P["normx.weight"],
                                    P["normx.bias"], P["normo.weight"], P["normo.bias"], P["mrla.Wq.weight"].ravel(),
                                    P["mrla.Wk.weight"].ravel(), P["mrla.Wv.weight"][:, 0], P["lambda_t"], d)
    g = mn.token_light_bwd(np.asarray(gup, np.float64), cache)
    if res:
        out = out + x
        g["dxt"] = g["dxt"] + gup
    return out, g


@pytest.mark.parametrize("case", cases.TOKEN_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("res", [False, True], ids=["module", "block"])
def test_token_module_fp32(case, res):
    name, b, n, c, d = case
    G = cases.golden("token_modules")
    x, o, gup = cases.token_inputs(name, b, n, c)
    P = cases.token_params(c)
    out, dx, do, pg = run(x, o, P, d, gup, torch.float32, res)
    want, g = oracle(x, o, P, d, gup, res)
    assert relmax(out, want) < ACT_TOL
    assert relmax(dx, g["dxt"]) < ACT_TOL
    assert relmax(do, g["dot"]) < ACT_TOL
    for got, key in zip(pg, ORACLE):
        assert relmax(got.ravel(), np.asarray(g[key]).ravel()) < par_tol(key), key
    sub = (lambda a: a[:, ::4]) if n == 197 else (lambda a: a)
    if res:      # the golden gradients were taken through the block residual x + module(x, o)
        assert relmax(sub(dx), G[name + "/dx"]) < GOLD_TOL
        assert relmax(sub(do), G[name + "/do"]) < GOLD_TOL
        assert relmax(pg[6], G[name + "/grad/mrla.Wv.weight"]) < GOLD_TOL
    else:
        assert relmax(sub(out), G[name + "/module_out"]) < GOLD_TOL


@pytest.mark.parametrize("c", [192, 384, 768], ids=["tiny", "small", "base"])
def test_token_block_bf16_and_batch(c):
    b, n, d = 5 if c == 192 else 3, 197, 16
    s = detgen.seed_of("tokbig" if c == 192 else f"tokbig{c}")
    x = bf16_round(detgen.normalish((b, n, c), s) * 1.3 + 0.2)
    o = bf16_round(detgen.normalish((b, n, c), s + 1))
    gup = bf16_round(detgen.normalish((b, n, c), s + 2))
    P = cases.token_params(c, salt=9)
    out, dx, do, pg = run(x, o, P, d, gup, torch.bfloat16, True)
    want, g = oracle(x, o, P, d, gup, True)
    assert_bf16_close(out, want, "out")
    assert_bf16_close(dx, g["dxt"], "dx")
    assert_bf16_close(do, g["dot"], "do")
    for got, key in zip(pg, ORACLE):
        assert relmax(got.ravel(), np.asarray(g[key]).ravel()) < par_tol(key), key


@pytest.mark.parametrize("case", cases.GELU_LAYER_CASES, ids=lambda c: c[0])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_gelu_map_layer_module(case, cl, dtype):
    """layers.mrlal_layer (MRLA_ACT_GELU branch of the light map kernels, deit_mrla_light.py:157-180) in both layouts vs
    the oracle and the reference's golden."""
    from mrla_amd.layers import mrlal_layer
    name, b, c, h, w, d = case
    G = cases.golden("token_modules")
    x, gup = cases.gelu_layer_inputs(name, b, c, h, w)
    if dtype == torch.bfloat16:
        x, gup = bf16_round(x), bf16_round(gup)
    P = cases.gelu_layer_params(c)
    lay = mrlal_layer(c, dim_perhead=d).cuda()
    lay.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    fmt = torch.channels_last if cl else torch.contiguous_format
    xt = to_dev(x, dtype).contiguous(memory_format=fmt).requires_grad_(True)
    y = lay(xt)
    y.backward(to_dev(gup, dtype).contiguous(memory_format=fmt))
    P64 = {k: v.astype(np.float64) for k, v in P.items()}
    want, cache = mn.light_layer_fwd(x.astype(np.float64), P64["Wq.weight"].ravel(), P64["Wk.weight"].ravel(),
                                     P64["Wv.weight"][:, 0], d, act_gelu=True)
    g = mn.light_layer_bwd(gup.astype(np.float64), cache)
    out, dx = y.detach().float().cpu().numpy(), xt.grad.float().cpu().numpy()
    if dtype == torch.float32:
        assert relmax(out, want) < ACT_TOL and relmax(dx, g["dx"]) < ACT_TOL
        sub = (lambda a: a[:, ::4]) if c > 64 else (lambda a: a)
        assert relmax(sub(out), G[name + "/out"]) < GOLD_TOL and relmax(sub(dx), G[name + "/dx"]) < GOLD_TOL
        assert relmax(lay.Wv.weight.grad.cpu().numpy(), G[name + "/grad/Wv.weight"]) < GOLD_TOL
    else:
        assert_bf16_close(out, want, "out")
        assert_bf16_close(dx, g["dx"], "dx")
    assert relmax(lay.Wv.weight.grad.cpu().numpy()[:, 0], g["dwv"]) < PAR_TOL
    assert relmax(lay.Wq.weight.grad.cpu().numpy().ravel(), g["dwq"]) < QK_TOL
    assert relmax(lay.Wk.weight.grad.cpu().numpy().ravel(), g["dwk"]) < QK_TOL


def test_deit_mrlal_tiny_logits_match_reference_and_eager():
    from mrla_amd import vit
    G = cases.golden("models")
    net = vit.deit_mrlal_tiny_patch16_224().cuda()
    ref = em.eager_deit_mrlal_tiny_patch16_224().cuda()
    vals = detgen.fill_state_dict(ref.state_dict())
    ref.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    net.load_state_dict(ref.state_dict())                  # same keys as the reference / the eager restatement
    net.eval(); ref.eval()
    x = torch.from_numpy(cases.image_batch(4)).cuda()
    with torch.no_grad():
        y, yr = net(x), ref(x)
    assert relmax(y.cpu().numpy(), yr.cpu().numpy()) < 5e-6          # same GPU GEMMs on both sides (measured 9.4e-7)
    assert relmax(y.cpu().numpy(), G["deit_mrlal_tiny/eval4/logits"]) < 1e-5   # the reference's CPU logits (measured 8.9e-7)
    net.train(); ref.train()
    xb = torch.from_numpy(cases.image_batch(4, "img-train")).cuda()
    tgt = (torch.arange(4) * 37 % 1000).cuda()
    y, yr = net(xb), ref(xb)
    torch.nn.functional.cross_entropy(y, tgt).backward()
    torch.nn.functional.cross_entropy(yr, tgt).backward()
    gp, gr = dict(net.named_parameters()), dict(ref.named_parameters())
    for k in gp:
        a, b_ = gp[k].grad.cpu().numpy().ravel().astype(np.float64), gr[k].grad.cpu().numpy().ravel().astype(np.float64)
        if np.abs(b_).sum() < 1e-6:
            continue
        assert np.abs(a - b_).sum() / np.abs(b_).sum() < 2e-3, k
