"""GPU micro-benchmark: weight gradient of the ResNet-50 1x1 stride-1 convolutions (b=256, bf16, channels_last): the stock
backward (aten.convolution_backward, weight only: MIOpen's memset + atomics + cast) vs mrla_conv1x1_wgrad.
Usage: python scripts/wgradbench.py [reps]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(os.environ.get("B", 256))


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


P = lambda t: ctypes.c_void_p(t.data_ptr())
tot = [0.0, 0.0]
# (c_in, c_out, hw, count per step)
for (cin, cout, hw, cnt) in [(64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1), (128, 512, 28, 4),
                             (512, 128, 28, 3), (512, 256, 28, 1), (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1),
                             (512, 2048, 7, 3), (2048, 512, 7, 2)]:
    x = torch.randn(B, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    m = B * hw * hw
    rows = L.load().mrla_conv1x1_wgrad_rows(m, cin, cout, L.BF16)
    part = torch.empty(rows, cout, cin, device="cuda")
    dw = torch.empty(cout, cin, device="cuda", dtype=torch.bfloat16)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    t_ref = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                               [False, True, False]))
    t_own = timeit(lambda: L.call("mrla_conv1x1_wgrad", P(dy), P(x), P(part), P(dw), m, cin, cout, L.BF16, L.BF16, st))
    ref = torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, [False, True, False])[1]
    err = ((dw.float() - ref.view(cout, cin).float()).norm() / ref.float().norm()).item()
    gb = (x.numel() + dy.numel()) * 2 / 1e9
    tot[0] += t_ref * cnt
    tot[1] += t_own * cnt
    print(f"wgrad {cin:4d}->{cout:4d} @{hw:2d} x{cnt}: stock {t_ref*1e6:7.1f} us ({gb/t_ref/1e3:4.2f} TB/s)   own {t_own*1e6:7.1f} us "
          f"({gb/t_own/1e3:4.2f} TB/s)   splits {rows}   rel.diff {err:.1e}", flush=True)
print("network sums (ms): stock %.3f  own %.3f" % tuple(1e3 * t for t in tot))
