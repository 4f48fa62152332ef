"""Shared case tables + deterministic inputs for the golden fixtures (mirrors oracle/make_goldens.py)."""
import os

import numpy as np

from oracle import detgen
from oracle.mrla_numpy import k_size_for

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

LIGHT_CASES = [("s64", 2, 64, 8, 8, 32), ("s256", 2, 256, 7, 5, 32), ("s2048", 1, 2048, 7, 7, 32),
               ("s128d16", 3, 128, 5, 6, 16)]
BASE_CASES = [("chain5", 2, 64, 6, 5, 16, 5), ("chain23", 1, 32, 2, 3, 16, 23), ("chain3cw", 2, 16, 4, 4, 1, 3),
              ("chain23n", 1, 64, 3, 4, 16, 23)]
TOKEN_CASES = [("t17", 2, 17, 32, 16), ("t197", 2, 197, 192, 16), ("t197s", 1, 197, 384, 16), ("t197b", 1, 197, 768, 16)]
GELU_LAYER_CASES = [("g64", 2, 64, 6, 5, 16), ("g192", 2, 192, 14, 14, 16)]

_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
    return _cache[name]


def light_inputs(name, b, c, h, w):
    s = detgen.seed_of("light/" + name)
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.25 * detgen.normalish((b, c, h, w), s + 5)
    return x.astype(np.float32), detgen.normalish((b, c, h, w), s + 1), detgen.normalish((b, c, h, w), s + 2)


def base_inputs(name, t, b, c, h, w):
    s = detgen.seed_of(f"base/{name}/{t}")
    x = np.maximum(detgen.normalish((b, c, h, w), s), 0) + 0.25 * detgen.normalish((b, c, h, w), s + 5)
    return x.astype(np.float32), detgen.normalish((b, c, h, w), s + 2)


def token_inputs(name, b, n, c):
    s = detgen.seed_of("tok/" + name)
    return (detgen.normalish((b, n, c), s) * 1.5 + 0.3, detgen.normalish((b, n, c), s + 1) * 0.7 - 0.2,
            detgen.normalish((b, n, c), s + 2))


def gelu_layer_inputs(name, b, c, h, w):
    s = detgen.seed_of("gelu/" + name)
    return detgen.normalish((b, c, h, w), s), detgen.normalish((b, c, h, w), s + 1)


def gelu_layer_params(c, salt=4):
    k = k_size_for(c)
    return detgen.fill_state_dict(_shapes({"Wq.weight": (1, 1, k), "Wk.weight": (1, 1, k), "Wv.weight": (c, 1, 3, 3)}), salt)


def _shapes(spec):
    return {k: np.empty(v, dtype=np.float32) for k, v in spec.items()}


def block_params(c, salt, light=True):
    """Deterministic MRLA + bn_mrla parameters of one ResNet block, under the reference's key names."""
    k = k_size_for(c)
    spec = {"mrla.mrla.Wq.weight": (1, 1, k), "mrla.mrla.Wk.weight": (1, 1, k), "mrla.mrla.Wv.weight": (c, 1, 3, 3),
            "bn_mrla.weight": (c,), "bn_mrla.bias": (c,), "bn_mrla.running_mean": (c,), "bn_mrla.running_var": (c,)}
    if light:
        spec["mrla.lambda_t"] = (c, 1, 1)
    return detgen.fill_state_dict(_shapes(spec), salt)


def token_params(c, salt=3):
    k = k_size_for(c)
    spec = {"mrla.Wq.weight": (1, 1, k), "mrla.Wk.weight": (1, 1, k), "mrla.Wv.weight": (c, 1, 3, 3),
            "lambda_t": (c,), "normx.weight": (c,), "normx.bias": (c,), "normo.weight": (c,), "normo.bias": (c,)}
    return detgen.fill_state_dict(_shapes(spec), salt)


def image_batch(b, tag="img"):
    return detgen.normalish((b, 3, 224, 224), detgen.seed_of(tag))
