"""GPU parity on SEEDED RANDOM shapes: the HIP paths (through the C ABI) vs the fp64 numpy oracle on shapes nobody chose
by hand -- odd batches, non-square maps, widths that are no multiple of the 7-column strips, channel counts on and off the
64-channel fast paths, short and long MRLA-base histories, token grids from 2x2 to 9x9.  The fixed cases of the other test
files pin the reference's own shapes; this file looks for the ragged edges between them (strip remainders, slab
remainders, several images per workgroup, rings that wrap) with the same fp32 bounds (tests/cases.py).

The shapes are drawn from a fixed seed (the parametrisation is identical on every machine); inputs come from
oracle/detgen.py like everywhere else."""
import numpy as np
import pytest
import torch

from oracle import detgen, mrla_numpy as mn
from tests import cases
from tests.test_light_gpu import ACT_TOL, TINY_BN_TOL, assert_bf16_close, bf16_round, oracle_light, par_tol, relmax, run_light

pytestmark = pytest.mark.gpu


def _shapes(seed, n, draw):
    rng = np.random.RandomState(seed)
    out, seen = [], set()
    while len(out) < n:
        s = draw(rng)
        if s not in seen:
            seen.add(s)
            out.append(s)
    return out


def _light_shape(rng):
    d = int(rng.choice([8, 16, 32]))
    c = int(d * rng.choice([2, 3, 4, 8, 12, 16]))             # 16 .. 512: on and off the c % 64 == 0 row pipeline
    return (int(rng.randint(1, 7)), c, int(rng.randint(1, 24)), int(rng.randint(1, 24)), d)


LIGHT_SHAPES = _shapes(20240401, 14, _light_shape)
# The row pipeline's workgroups hold neighbouring channel groups side by side once there are >= 8 of them (wide_shape() in
# light_nhwc_wide.hip): these shapes put 4 groups x 2 strips in a workgroup with the strips walked unevenly (5 strips on 2
# strip slots), ragged last strips, a group count 4 does not divide (9 -> one group per workgroup), 12 and 16 groups, and a
# map wider than 8 strips.
LIGHT_SHAPES += [(3, 512, 5, 23, 32), (2, 768, 6, 30, 32), (2, 576, 4, 9, 32), (2, 1024, 9, 16, 16), (1, 512, 3, 61, 32)]


@pytest.mark.parametrize("shape", LIGHT_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_light_tail_random_shapes_fp32(shape, cl):
    b, c, h, w, d = shape
    s = detgen.seed_of(f"rand-light/{shape}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.2 * detgen.normalish((b, c, h, w), s + 1)
    o = detgen.normalish((b, c, h, w), s + 2)
    gup = detgen.normalish((b, c, h, w), s + 3)
    P = cases.block_params(c, 21)
    mask = np.array(([1, 1, 0, 1, 0, 1] * 2)[:b], dtype=np.float32)
    mode = "traindp" if b > 1 else "train"
    got = run_light(x, o, P, d, mode, mask if b > 1 else None, 0.25, gup, cl=cl)
    out, cache, g = oracle_light(x, o, P, d, mode, mask if b > 1 else None, 0.25, gup)
    tiny = b * h * w <= 9                   # BatchNorm over a handful of values: 1/sigma amplifies input rounding
    tol = TINY_BN_TOL if tiny else ACT_TOL
    if b * h * w == 1:
        pytest.skip("a single value per channel: train-mode BatchNorm is degenerate (variance 0)")
    assert relmax(got["out"], out) < tol
    assert relmax(got["dx"], g["dx"]) < tol
    assert relmax(got["do"], g["do_prev"]) < tol
    assert relmax(got["rv"], cache["bn"]["new_rv"]) < tol
    for ours, theirs in (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
                         ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta")):
        want = np.asarray(g[theirs]).ravel()
        assert relmax(got["grad/" + ours].ravel(), want) < (TINY_BN_TOL if tiny else par_tol(theirs)), ours


@pytest.mark.parametrize("shape", LIGHT_SHAPES[:8] + LIGHT_SHAPES[-5:], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_light_tail_random_shapes_bf16(shape, cl):
    """bf16 I/O protocol: inputs pre-rounded to bf16, the kernel's bf16 outputs vs the fp64 oracle on
    the same inputs rounded once to bf16 -- within 1 bf16 ulp elementwise; the parameter gradients (fp32 sums) to the fp32
    bounds."""
    b, c, h, w, d = shape
    if b * h * w <= 9:
        pytest.skip("BatchNorm over a handful of values: covered in fp32 with its own bound")
    s = detgen.seed_of(f"rand-light16/{shape}")
    x = bf16_round(np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.2 * detgen.normalish((b, c, h, w), s + 1))
    o = bf16_round(detgen.normalish((b, c, h, w), s + 2))
    gup = bf16_round(detgen.normalish((b, c, h, w), s + 3))
    P = cases.block_params(c, 23)
    got = run_light(x, o, P, d, "train", None, 0.0, gup, torch.bfloat16, cl=cl)
    out, cache, g = oracle_light(x, o, P, d, "train", None, 0.0, gup)
    assert_bf16_close(got["out"], out, "out")
    assert_bf16_close(got["dx"], g["dx"], "dx")
    assert_bf16_close(got["do"], g["do_prev"], "do")
    for ours, theirs in (("mrla.mrla.Wq.weight", "dwq"), ("mrla.mrla.Wk.weight", "dwk"), ("mrla.mrla.Wv.weight", "dwv"),
                         ("mrla.lambda_t", "dlam"), ("bn_mrla.weight", "dgamma"), ("bn_mrla.bias", "dbeta")):
        assert relmax(got["grad/" + ours].ravel(), np.asarray(g[theirs]).ravel()) < par_tol(theirs), ours


def _base_shape(rng):
    d = int(rng.choice([1, 8, 16]))
    c = int(max(d, 16) * rng.choice([1, 2, 4, 8]))            # 16 .. 128 (64 / 128: the slot-major NHWC rings)
    return (int(rng.randint(1, 5)), c, int(rng.randint(2, 12)), int(rng.randint(2, 12)), d, int(rng.randint(1, 8)))


BASE_SHAPES = _shapes(20240402, 10, _base_shape)
BASE_SHAPES += [(2, 512, 5, 23, 16, 3), (2, 768, 4, 30, 16, 2)]      # (the value backward's workgroups of neighbouring channel groups)


@pytest.mark.parametrize("shape", BASE_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_base_chain_random_shapes_fp32(shape, cl):
    from tests.test_base_gpu import PAIRS, oracle_chain, run_chain
    b, c, h, w, d, Tn = shape
    s = detgen.seed_of(f"rand-base/{shape}")
    xs = [np.maximum(detgen.normalish((b, c, h, w), s + 10 * t), 0) + 0.2 * detgen.normalish((b, c, h, w), s + 10 * t + 1)
          for t in range(Tn)]
    gups = [detgen.normalish((b, c, h, w), s + 10 * t + 2) for t in range(Tn)]
    params = [cases.block_params(c, 40 + t, light=False) for t in range(Tn)]
    got, K, V = run_chain(xs, gups, params, d, True, cl=cl)          # (skips when the shape keeps NCHW rings under cl)
    outs, caches, grads, Ko, Vo = oracle_chain(xs, gups, params, d, True)
    assert relmax(K, Ko) < ACT_TOL and relmax(V, Vo) < ACT_TOL
    for t in range(Tn):
        assert relmax(got[t]["out"], outs[t]) < ACT_TOL, t
        assert relmax(got[t]["dx"], grads[t]["dx"]) < ACT_TOL, t
        for ours, theirs in PAIRS:
            want = np.asarray(grads[t][theirs]).ravel()
            if np.abs(want).max() < 1e-12:
                continue                                # (d = 1, t = 1: softmax over one key -- exactly zero gradients)
            assert relmax(got[t]["grad/" + ours].ravel(), want) < par_tol(theirs), (t, ours)


def _token_shape(rng):
    side = int(rng.randint(2, 10))
    return (int(rng.randint(1, 5)), 1 + side * side, int(16 * rng.choice([2, 4, 8, 12])), 16)     # c = 32 .. 192


TOKEN_SHAPES = _shapes(20240403, 8, _token_shape)


@pytest.mark.parametrize("shape", TOKEN_SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("res", [False, True], ids=["module", "block"])
def test_token_module_random_shapes_fp32(shape, res):
    from tests.test_tokens_gpu import ORACLE, oracle, run
    b, n, c, d = shape
    s = detgen.seed_of(f"rand-tok/{shape}")
    x = detgen.normalish((b, n, c), s) * 1.4 + 0.25
    o = detgen.normalish((b, n, c), s + 1) * 0.8 - 0.1
    gup = detgen.normalish((b, n, c), s + 2)
    P = cases.token_params(c, salt=17)
    out, dx, do, pg = run(x, o, P, d, gup, torch.float32, res)
    want, g = oracle(x, o, P, d, gup, res)
    assert relmax(out, want) < ACT_TOL
    assert relmax(dx, g["dxt"]) < ACT_TOL
    assert relmax(do, g["dot"]) < ACT_TOL
    for got, key in zip(pg, ORACLE):
        assert relmax(got.ravel(), np.asarray(g[key]).ravel()) < par_tol(key), key
