"""TEST INFRASTRUCTURE ONLY -- torch-RNG-independent deterministic tensors.

A counter-based integer hash (splitmix64 finaliser) -> 24-bit uniforms -> float32, using only exact
integer / dyadic-rational arithmetic, so the container that generates the golden fixtures and the GPU
box that replays them produce bit-identical inputs and weights without shipping them.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def seed_of(name, salt=0):
    """Stable 32-bit seed from a string (e.g. a state_dict key)."""
    return (zlib.crc32(name.encode()) + 0x9E3779B1 * salt) & 0xFFFFFFFF


def uniform(shape, seed, lo=-1.0, hi=1.0):
    """float32 uniform in [lo, hi) on a 2^-24 grid."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + (np.uint64(seed) << np.uint64(32))
        bits = _mix(_mix(ctr)) >> np.uint64(40)                  # top 24 bits
    u = bits.astype(np.float64) / float(1 << 24)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(shape, seed, std=1.0):
    """Irwin-Hall(4) scaled to unit variance: bell-shaped, bounded (+-3.46 std), no transcendental calls."""
    n = int(np.prod(shape)) if len(shape) else 1
    acc = np.zeros(n, dtype=np.float64)
    for i in range(4):
        acc += uniform((n,), (seed * 4 + i) & 0xFFFFFFFF, 0.0, 1.0).astype(np.float64)
    z = (acc - 2.0) * 1.7320508075688772            # var of IH(4) = 4/12
    return (z * std).astype(np.float32).reshape(shape)


def fill_state_dict(sd, salt=0):
    """Deterministic, well-conditioned values for every tensor of a ResNet/DeiT MRLA state_dict.

    `sd`: mapping name -> object with .shape (torch tensors or numpy arrays).  Returns name -> float32 /
    int64 numpy arrays.  Scales are chosen so activations stay O(1) through 16..33 blocks with BN in
    either mode (this is for parity, not accuracy).
    """
    out = {}
    for name in sorted(sd.keys()):
        shape = tuple(sd[name].shape)
        s = seed_of(name, salt)
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, dtype=np.int64)
        elif leaf == "running_mean":
            out[name] = 0.1 * uniform(shape, s)
        elif leaf == "running_var":
            out[name] = 1.0 + 0.25 * uniform(shape, s)
        elif leaf == "lambda_t":
            out[name] = 0.5 * uniform(shape, s)
        elif parent in ("Wq", "Wk"):
            out[name] = 0.6 * uniform(shape, s)
        elif leaf == "bias":
            out[name] = 0.05 * uniform(shape, s)
        elif leaf == "weight" and len(shape) == 1:                      # norm scales
            base = 0.35 if parent == "bn3" else 1.0
            out[name] = base * (1.0 + 0.2 * uniform(shape, s))
        elif leaf == "weight" and len(shape) >= 2:
            fan_out = shape[0] * int(np.prod(shape[2:])) if len(shape) > 2 else shape[0]
            fan_in = int(np.prod(shape[1:]))
            std = np.sqrt(2.0 / max(fan_out if len(shape) > 2 else fan_in, 1))
            out[name] = normalish(shape, s, std)
        else:                                                           # cls_token, pos_embed, ...
            out[name] = 0.02 * normalish(shape, s)
    return out
