"""GPU micro-benchmark of the MRLA streaming kernels through the C ABI, per ResNet-50 stage shape (b=256, bf16).
Usage: [KBENCH_LIB=scripts/variants/libmrla_hip_<name>.so] [LAYOUT=nhwc] [STAGE=0..3] python scripts/kbench.py [reps] [kernel-substring]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

if os.environ.get("KBENCH_LIB"):          # an experiment build (scripts/build_variant.sh) instead of the product library
    L.LIB_PATH = os.path.abspath(os.environ["KBENCH_LIB"])

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
only = sys.argv[2] if len(sys.argv) > 2 else ""
B = int(os.environ.get("B", 256))
RELU = int(os.environ.get("RELU", 1))
LAY = L.NHWC if os.environ.get("LAYOUT", "nchw") == "nhwc" else L.NCHW
FMT = torch.channels_last if LAY == L.NHWC else torch.contiguous_format
STAGES = [(256, 56), (512, 28), (1024, 14), (2048, 7)]
if os.environ.get("STAGE"):               # one stage only (counter passes: the summaries average over equal grid sizes)
    STAGES = [STAGES[int(os.environ["STAGE"])]]
dt = torch.bfloat16
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
for c, hw in STAGES:
    d, ks = 32, (7 if c == 2048 else 5)
    n = B * c * hw * hw
    x = torch.randn(B, c, hw, hw, device="cuda").to(dt).contiguous(memory_format=FMT)
    o = torch.randn(B, c, hw, hw, device="cuda").to(dt).contiguous(memory_format=FMT)
    g = torch.randn(B, c, hw, hw, device="cuda").to(dt).contiguous(memory_format=FMT)
    out, dx, do = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    wv = torch.randn(c, 9, device="cuda") * 0.3
    wq, wk = torch.randn(ks, device="cuda"), torch.randn(ks, device="cuda")
    lam, gamma, beta = torch.randn(c, device="cuda"), torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    dp = torch.ones(B, device="cuda")
    mom = torch.empty(B, c, L.FWD_MOMENTS, device="cuda")
    bmom = torch.empty(B, c, 3, device="cuda")
    gate = torch.empty(B, c // d, device="cuda")
    bn = torch.empty(4, c, device="cuda")
    cb = torch.empty(c, 4, device="cuda")
    small = torch.empty(3, c, device="cuda")
    dyx = torch.empty(B, c, device="cuda")
    dwqk = torch.empty(B, 2 * ks, device="cuda")
    rows = lib.mrla_light_wgrad_rows(B, c, hw, hw, L.BF16, LAY)
    dwv = torch.empty(rows, c * 9, device="cuda")
    pre_tmom = torch.empty(rows, c, 2, device="cuda")
    K = {
        "stats_fwd": (2, lambda: lib.mrla_light_stats_fwd(P(x), P(o), P(wv), P(mom), B, c, hw, hw, L.BF16, LAY, 0, st)),
        "gate_fwd": (0, lambda: lib.mrla_light_gate_fwd(P(mom), P(wq), P(wk), ks, P(gate), B, c, hw * hw, d, st)),
        "bn_fwd": (0, lambda: lib.mrla_light_bn_fwd(P(mom), P(gate), P(lam), P(gamma), P(beta), P(rm), P(rv), 1, 0.1, 1e-5,
                                                     P(bn[0]), P(bn[1]), P(bn[2]), P(bn[3]), B, c, hw * hw, d, st)),
        "apply_fwd": (3, lambda: lib.mrla_light_apply_fwd(P(x), P(o), P(wv), P(gate), P(bn[0]), P(bn[1]), P(lam), P(dp), P(out),
                                                          B, c, hw, hw, d, 1, L.BF16, LAY, 0, st)),
        "stats_bwd": (3, lambda: lib.mrla_light_stats_bwd(P(g), P(x), P(o), P(wv), P(mom), P(bmom), B, c, hw, hw, L.BF16, LAY, 0, st)),
        "bn_bwd": (0, lambda: lib.mrla_light_bn_bwd(P(mom), P(bmom), P(gate), P(lam), P(gamma), P(dp), P(bn[2]), P(bn[3]), 1,
                                                     P(cb), None, P(small[0]), P(small[1]), P(small[2]), B, c, hw * hw, d, st)),
        "gate_bwd": (0, lambda: lib.mrla_light_gate_bwd(P(mom), P(bmom), P(gate), P(cb), None, P(dp), P(wq), P(wk), ks, P(dyx), P(dwqk),
                                                        B, c, hw * hw, d, st)),
        "apply_bwd": (5, lambda: lib.mrla_light_apply_bwd(P(g), P(x), P(o), P(wv), P(gate), P(cb), P(lam), P(dp), P(dyx), P(dx),
                                                          P(do), P(dwv), None, None, None, B, c, hw, hw, d, 1, RELU, L.BF16, LAY, 0, st)),
        "apply_bwd+bn3sums": (6, lambda: lib.mrla_light_apply_bwd(P(g), P(x), P(o), P(wv), P(gate), P(cb), P(lam), P(dp), P(dyx), P(dx),
                                                                  P(do), P(dwv), P(out), P(bn[2]), P(pre_tmom), B, c, hw, hw, d, 1, 1,
                                                                  L.BF16, LAY, 0, st)),
        "stats_fused": (3, lambda: lib.mrla_light_stats_fwd_fused(P(g), P(bn[0]), P(bn[1]), P(o), P(wv), P(mom), P(out), B, c, hw, hw, L.BF16, LAY, st)),
    }
    if LAY == L.NHWC and lib.mrla_light_lean_supported(B, c, hw, hw, L.BF16, LAY) == 1:
        # the passes of the tail without a stored x_t (ABI 5): x_t re-formed from y3 (= g here), the affine (bn[0], bn[1]) and o
        K.update({
            "lean stats_fwd": (2, lambda: lib.mrla_light_stats_fwd_fused(P(g), P(bn[0]), P(bn[1]), P(o), P(wv), P(mom), None, B, c, hw, hw, L.BF16, LAY, st)),
            "lean apply_fwd": (3, lambda: lib.mrla_light_apply_fwd_fused(P(g), P(bn[0]), P(bn[1]), P(o), P(wv), P(gate), P(bn[0]), P(bn[1]),
                                                                         P(lam), P(dp), P(out), B, c, hw, hw, d, 1, L.BF16, LAY, st)),
            "lean stats_bwd": (3, lambda: lib.mrla_light_stats_bwd_fused(P(x), P(g), P(bn[0]), P(bn[1]), P(o), P(wv), P(mom), P(bmom), B, c, hw, hw,
                                                                         L.BF16, LAY, st)),
            "lean apply_bwd": (5, lambda: lib.mrla_light_apply_bwd_fused(P(x), P(g), P(bn[0]), P(bn[1]), P(o), P(wv), P(gate), P(cb), P(lam), P(dp),
                                                                         P(dyx), P(dx), P(do), P(dwv), P(bn[2]), P(pre_tmom), B, c, hw, hw, d, 1,
                                                                         L.BF16, LAY, st)),
        })
    for name, (passes, fn) in K.items():
        if only and only not in name:
            continue
        rc = fn()
        assert rc == 0, (name, rc)
        t = timeit(fn)
        tot[name] = tot.get(name, 0.0) + t
        bw = f"{passes * n * 2 / t / 1e12:5.2f} TB/s" if passes else "          "
        print(f"c={c:5d} {hw:2d}x{hw:<2d} {name:10s} {t * 1e6:8.1f} us  {bw}", flush=True)
    # reference point: a plain bf16 copy of the same tensor (2 passes)
    if not only:
        t = timeit(lambda: out.copy_(x))
        print(f"c={c:5d} {hw:2d}x{hw:<2d} {'torch copy':10s} {t * 1e6:8.1f} us  {2 * n * 2 / t / 1e12:5.2f} TB/s", flush=True)
blocks = [3, 4, 6, 3]
print("per-kernel sum over one block of each stage (us):", {k: round(v * 1e6, 1) for k, v in tot.items()})
