"""GPU micro-benchmark: the 1x1 convolutions with a wide reduction (K >= 512; b=256, bf16, channels_last) -- forward and
input gradient: MIOpen (F.conv2d / convolution_backward) vs the K-streaming GEMM (mrla_conv1x1_fwd).
Usage: [NOSTOCK=1] python scripts/ksbench.py [reps]"""
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(os.environ.get("B", 256))
torch.backends.cudnn.benchmark = bool(int(os.environ.get("BENCHMARK", "1")))


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


P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = [0.0, 0.0]
# (K, N, hw, count, what)
for (k, n, hw, cnt, what) in [(512, 128, 28, 3, "fwd conv1 s2"), (512, 256, 28, 1, "fwd conv1 s3.0"), (1024, 256, 14, 5, "fwd conv1 s3"),
                              (1024, 512, 14, 1, "fwd conv1 s4.0"), (2048, 512, 7, 2, "fwd conv1 s4"), (512, 2048, 7, 3, "fwd conv3 s4"),
                              (512, 128, 28, 4, "dgrad conv3 s2"), (1024, 256, 14, 6, "dgrad conv3 s3"), (2048, 512, 7, 3, "dgrad conv3 s4"),
                              (512, 2048, 7, 2, "dgrad conv1 s4"), (512, 1024, 14, 1, "dgrad conv1 s4.0")]:
    m = B * hw * hw
    x = torch.randn(B, k, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(n, k, 1, 1, device="cuda") * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    w2 = w.view(n, k).contiguous()
    y = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    with torch.no_grad():
        if os.environ.get("NOSTOCK"):
            t_stock = 0.0
        elif what.startswith("fwd"):
            t_stock = timeit(lambda: F.conv2d(x, w))
        else:       # the input gradient of a convolution n -> k channels: dX[m, n_out = n] = dY[m, k] * W[k, n]... as a stock call
            wt = (torch.randn(k, n, 1, 1, device="cuda") * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
            xin = torch.empty(B, n, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
            t_stock = timeit(lambda: torch.ops.aten.convolution_backward(x, xin, wt, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                                           [True, False, False]))
        t_own = timeit(lambda: L.call("mrla_conv1x1_fwd", P(x), P(w2), P(y), None, m, k, n, L.BF16, st))
    gb = (x.numel() + y.numel()) * 2 / 1e9
    tf = 2.0 * m * k * n / 1e12
    tot[0] += t_stock * cnt
    tot[1] += t_own * cnt
    print(f"{what:16s} K {k:4d} -> N {n:4d} @{hw:2d} x{cnt}: stock {t_stock*1e6:7.1f} us   own {t_own*1e6:7.1f} us ({gb/t_own/1e3:4.2f} TB/s, "
          f"{tf/t_own:5.0f} TFLOP/s)   HBM bound {gb/5.0*1e3:5.1f} us", flush=True)
print("network sums (ms): stock %.3f  own %.3f" % tuple(1e3 * t for t in tot))
