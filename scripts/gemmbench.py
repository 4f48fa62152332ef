"""GPU micro-benchmark: forward of the ResNet-50 1x1 stride-1 convolutions (b=256, bf16, channels_last): MIOpen (F.conv2d)
vs the MFMA GEMM with the BatchNorm-moments epilogue (mrla_conv1x1_fwd), and the stand-alone moments pass it replaces.
Usage: python scripts/gemmbench.py [reps]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L, functional as Fm  # noqa: E402

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


tot = [0.0, 0.0, 0.0, 0.0]
for (cin, cout, hw, n) in [(64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1), (128, 512, 28, 4),
                           (256, 1024, 14, 6)]:
    x = torch.randn(B, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, device="cuda") * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    w2 = w.view(cout, cin)
    m = B * hw * hw
    rows = L.load().mrla_conv1x1_rows(m, cin, cout, L.BF16)
    y = torch.empty(B, cout, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    mrows = L.load().mrla_bn_moment_rows(B, cout, hw, hw, L.NHWC)
    amom = torch.empty(mrows, cout, 2, device="cuda")
    st = Fm._stream()
    with torch.no_grad():
        t_conv = timeit(lambda: F.conv2d(x, w))
        t_gemm = timeit(lambda: Fm._Conv1x1Fn.apply(x, w2, False))
        t_gemm_m = timeit(lambda: Fm._Conv1x1Fn.apply(x, w2, True))
        t_mom = timeit(lambda: L.call("mrla_bn_plane_moments", Fm._ptr(y), Fm._ptr(amom), None, B, cout, hw, hw, L.BF16, L.NHWC, st))
    gb = (x.numel() + y.numel()) * 2 / 1e9
    for i, t in enumerate((t_conv, t_gemm, t_gemm_m, t_mom)):
        tot[i] += t * n
    print(f"1x1 {cin:4d}->{cout:4d} @{hw:2d} x{n}: conv2d {t_conv*1e6:7.1f} us ({gb/t_conv/1e3:4.2f} TB/s)   gemm {t_gemm*1e6:7.1f} us "
          f"({gb/t_gemm/1e3:4.2f})   gemm+moments {t_gemm_m*1e6:7.1f} us ({gb/t_gemm_m/1e3:4.2f})   moments pass {t_mom*1e6:6.1f} us   rows {rows}", flush=True)
print("network sums (ms): conv2d %.3f  gemm %.3f  gemm+moments %.3f  moments passes %.3f" % tuple(1e3 * t for t in tot))
