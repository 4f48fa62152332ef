"""GPU probe: which kernels one channels_last bf16 convolution launches (looking for the output memset MIOpen adds)."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

bench = int(os.environ.get("BENCHMARK", 0))
torch.backends.cudnn.benchmark = bool(bench)
cases = [(256, 64, 1, 56, 1), (64, 64, 3, 56, 1), (64, 256, 1, 56, 1), (512, 128, 1, 28, 1), (128, 128, 3, 28, 1), (256, 512, 1, 56, 2)]
for cin, cout, k, hw, stride in cases:
    conv = torch.nn.Conv2d(cin, cout, k, stride, k // 2, bias=False).cuda().bfloat16().to(memory_format=torch.channels_last)
    x = torch.randn(256, cin, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    for _ in range(3):
        y = conv(x)
        y.backward(y)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            y = conv(x)
            y.backward(y)
        torch.cuda.synchronize()
    print(f"--- conv {cin}->{cout} k{k} s{stride} @{hw}  out {y.numel() * 2 / 1e6:.0f} MB  in {x.numel() * 2 / 1e6:.0f} MB")
    for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total):
        if e.device_time_total > 0:
            print(f"   {e.device_time_total / 5:8.1f} us  {e.count // 5}x  {e.key[:100]}")
