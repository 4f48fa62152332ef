"""GPU micro-benchmark of the fused BatchNorm(+ReLU) passes through the C ABI on every BatchNorm shape of
ResNet-50 at b=256 (bf16, channels_last by default).
Usage: [KBENCH_LIB=scripts/variants/libmrla_hip_<name>.so] python scripts/bnbench.py [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

if os.environ.get("KBENCH_LIB"):          # an experiment build (scripts/build_variant.sh) instead of the product library
    L.LIB_PATH = os.path.abspath(os.environ["KBENCH_LIB"])
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(os.environ.get("B", 256))
LAY = L.NCHW if os.environ.get("LAYOUT", "nhwc") == "nchw" else L.NHWC
FMT = torch.channels_last if LAY == L.NHWC else torch.contiguous_format
# (channels, side, count per step) of resnet50: stem, then per stage bn1/bn2/bn3/downsample
SHAPES = [(64, 112, 1), (64, 56, 6), (256, 56, 4), (128, 56, 1), (128, 28, 7), (512, 28, 5), (256, 28, 1), (256, 14, 11),
          (1024, 14, 7), (512, 14, 1), (512, 7, 5), (2048, 7, 4)]
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


tot = {}
print(f"{'shape':>14} {'MB':>7} | " + " | ".join(f"{k:>16}" for k in ("moments", "act_fwd", "dmoments", "act_bwd")))
for c, hw, cnt in SHAPES:
    x = torch.randn(B, c, hw, hw, device="cuda").bfloat16().contiguous(memory_format=FMT)
    g = torch.randn(B, c, hw, hw, device="cuda").bfloat16().contiguous(memory_format=FMT)
    y = torch.empty_like(x)
    rows = lib.mrla_bn_moment_rows(B, c, hw, hw, LAY)
    mom = torch.empty(rows, c, 2, device="cuda")
    sc, sh = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    cb = torch.randn(c, 3, device="cuda")
    nbytes = x.numel() * 2
    K = {"moments": (1, lambda: lib.mrla_bn_plane_moments(P(x), P(mom), None, B, c, hw, hw, L.BF16, LAY, st)),
         "act_fwd": (2, lambda: lib.mrla_bn_act_fwd(P(x), P(sc), P(sh), 1, P(y), B, c, hw, hw, L.BF16, LAY, st)),
         "dmoments": (2, lambda: lib.mrla_bn_plane_dmoments(P(g), P(x), P(sc), P(sh), None, 1, P(mom), B, c, hw, hw, L.BF16, LAY, st)),
         "act_bwd": (3, lambda: lib.mrla_bn_act_bwd(P(g), P(x), P(sc), P(sh), P(cb), 1, P(y), B, c, hw, hw, L.BF16, LAY, st))}
    cells = []
    for name, (passes, fn) in K.items():
        rc = fn()
        assert rc == 0, (name, rc)
        t = timeit(fn)
        tot[name] = tot.get(name, 0.0) + t * cnt
        cells.append(f"{t * 1e6:7.1f}us {passes * nbytes / t / 1e12:5.2f}TB")
    print(f"{c:5d}x{hw:3d}^2 x{cnt:2d} {nbytes / 1e6:7.1f} | " + " | ".join(cells))
print("per training step (ms):", {k: round(v * 1e3, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()) * 1e3, 3))
