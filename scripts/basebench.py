"""GPU micro-benchmark of the channels_last MRLA-base kernels through the C ABI (bf16, b=128, ResNet-101 stage shapes)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = int(os.environ.get("B", 128))


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for c, hw, T, ts in ((256, 56, 3, (1, 3)), (512, 28, 4, (1, 4)), (1024, 14, 23, (1, 4, 12, 23)), (2048, 7, 3, (1, 3))):
    d, n = 16, B * c * hw * hw
    ring = torch.randn(T, B, hw, hw, c, device="cuda").bfloat16()
    dA = torch.randn(T, B, hw, hw, c, device="cuda").bfloat16()
    Pall = torch.softmax(torch.randn(B, c // d, T, T, device="cuda"), -1)
    x = torch.randn(B, hw, hw, c, device="cuda").bfloat16()
    g = torch.randn_like(x)
    attn, out, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    dv = torch.empty(B, hw, hw, c, device="cuda").bfloat16()
    rows = lib.mrla_base_tile_rows(B, c, hw, hw, L.BF16, L.NHWC)
    amom = torch.empty(rows, c, 2, device="cuda")
    sc, sh, cb = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.randn(c, 3, device="cuda")
    wv = torch.randn(c, 9, device="cuda")
    dyx = torch.randn(B, c, device="cuda")
    wrows = lib.mrla_light_wgrad_rows(B, c, hw, hw, L.BF16, L.NHWC)
    dwv = torch.empty(wrows, c * 9, device="cuda")
    mom = torch.empty(B, c, 6, device="cuda")
    print(f"--- c={c} {hw}x{hw} T={T} tile rows/image={rows // B}  tensor {n * 2 / 1e6:.0f} MB")
    t0 = timeit(lambda: lib.mrla_base_pool_value_fwd(P(x), None, None, None, P(wv), P(mom), None, P(ring[0]), B, c, hw, hw, L.BF16, L.NHWC, st))
    print(f"   pool_value          {t0 * 1e6:8.1f} us  {2 * n * 2 / t0 / 1e12:5.2f} TB/s")
    t0 = timeit(lambda: lib.mrla_base_tail_fwd(P(x), P(attn), P(sc), P(sh), None, P(out), B, c, hw, hw, L.BF16, L.NHWC, st))
    print(f"   tail_fwd            {t0 * 1e6:8.1f} us  {3 * n * 2 / t0 / 1e12:5.2f} TB/s")
    t0 = timeit(lambda: lib.mrla_base_value_bwd_dv(P(g), P(x), P(wv), P(dv), P(dyx), P(dx), P(dwv), B, c, hw, hw, 3, L.BF16, L.NHWC, st))
    print(f"   value_bwd_dv        {t0 * 1e6:8.1f} us  {n * 4 * 2 / t0 / 1e12:5.2f} TB/s")
    for t in ts:
        part = torch.empty(lib.mrla_base_pmom_rows(B, c, hw, hw, L.BF16, L.NHWC), t, c, device="cuda")
        ta = timeit(lambda: lib.mrla_base_attend_fwd(None, P(wv), P(ring), P(Pall), P(attn), P(amom), B, c, hw, hw, d, T, t, L.BF16, L.NHWC, st))
        tb = timeit(lambda: lib.mrla_base_attend_bwd(P(g), P(attn), P(sc), P(sh), None, P(cb), P(ring), P(dA), P(part), B, c, hw, hw, T, t, L.BF16, L.NHWC, st))
        tc = timeit(lambda: lib.mrla_base_dv_combine(P(dA), P(Pall), P(dv), B, c, hw, hw, d, T, T - t + 1, T, L.BF16, L.NHWC, st))
        print(f"   t={t:2d} attend_fwd {ta * 1e6:8.1f} us {(t + 1) * n * 2 / ta / 1e12:5.2f} TB/s | attend_bwd {tb * 1e6:8.1f} us "
              f"{(t + 3) * n * 2 / tb / 1e12:5.2f} TB/s | dv_combine({t} slots) {tc * 1e6:8.1f} us {n * (t + 1) * 2 / tc / 1e12:5.2f} TB/s")
