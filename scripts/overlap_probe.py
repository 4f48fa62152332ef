"""GPU probe: does an MFMA-bound kernel (weight-gradient / K-streaming GEMM, stock 3x3 weight gradient) overlap with an
HBM-bound pass (a BatchNorm-apply-sized elementwise op) when the two are launched on different HIP streams?
Prints serial vs two-stream wall time per pair.  Usage: python scripts/overlap_probe.py [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = 256
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def S(s):
    return ctypes.c_void_p(s.cuda_stream)


def wall(fn_main, fn_side, n_main, n_side, two_streams):
    def once():
        if two_streams:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(n_side):
                    fn_side(side)
            for _ in range(n_main):
                fn_main(main)
            main.wait_stream(side)
        else:
            for _ in range(n_side):
                fn_side(main)
            for _ in range(n_main):
                fn_main(main)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def hbm_pass(c, hw):
    x = torch.randn(B, hw, hw, c, device="cuda").bfloat16()
    y = torch.empty_like(x)
    return lambda s: torch.add(x, 1.0, out=y)      # (torch launches on the current stream: only used as fn_main)


def wgrad(cin, cout, hw):
    m = B * hw * hw
    x = torch.randn(m, cin, device="cuda").bfloat16()
    dy = torch.randn(m, cout, device="cuda").bfloat16()
    rows = L.load().mrla_conv1x1_wgrad_rows(m, cin, cout, L.BF16)
    part = torch.empty(rows, cout, cin, device="cuda")
    dw = torch.empty(cout, cin, device="cuda")
    return lambda s: L.call("mrla_conv1x1_wgrad", P(dy), P(x), P(part), P(dw), m, cin, cout, L.BF16, L.F32, S(s))


def kstream(cin, cout, hw):
    m = B * hw * hw
    x = torch.randn(m, cin, device="cuda").bfloat16()
    w = torch.randn(cout, cin, device="cuda").bfloat16()
    y = torch.empty(m, cout, device="cuda", dtype=torch.bfloat16)
    return lambda s: L.call("mrla_conv1x1_fwd", P(x), P(w), P(y), None, m, cin, cout, L.BF16, S(s))


def stock_wrw3(c, hw):
    x = torch.randn(B, c, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    w = torch.randn(c, c, 3, 3, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)

    def run(s):
        with torch.cuda.stream(s):
            torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, [False, True, False])
    return run


torch.backends.cudnn.benchmark = True
pairs = [
    ("wgrad 1024->256 @14 x4 | add 256ch @56", wgrad(1024, 256, 14), 4, hbm_pass(256, 56), 1),
    ("wgrad 256->1024 @14 x4 | add 64ch @56 x4", wgrad(256, 1024, 14), 4, hbm_pass(64, 56), 4),
    ("wgrad 64->256 @56 x2   | add 256ch @56", wgrad(64, 256, 56), 2, hbm_pass(256, 56), 1),
    ("kstream 1024->256 @14 x4 | add 256ch @56", kstream(1024, 256, 14), 4, hbm_pass(256, 56), 1),
    ("stock 3x3 wrw 256 @14 x2 | add 256ch @56", stock_wrw3(256, 14), 2, hbm_pass(256, 56), 1),
    ("stock 3x3 wrw 64 @56 x1 | add 256ch @56", stock_wrw3(64, 56), 1, hbm_pass(256, 56), 1),
]
for name, f_side, n_side, f_main, n_main in pairs:
    t_side = wall(lambda s: None, f_side, 0, n_side, False)
    t_main = wall(f_main, lambda s: None, n_main, 0, False)
    t_ser = wall(f_main, f_side, n_main, n_side, False)
    t_two = wall(f_main, f_side, n_main, n_side, True)
    print(f"{name}: side {t_side:7.1f} us  main {t_main:7.1f} us  serial {t_ser:7.1f} us  two streams {t_two:7.1f} us "
          f"({100 * (1 - t_two / t_ser):+.1f} % saved)", flush=True)
