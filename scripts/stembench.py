"""GPU micro-benchmark of the stem tail (b=256, 64x112x112, bf16 channels_last): stock bn_act + max_pool2d vs the fused
mrla_bn_relu_pool_* passes, forward and backward.  Usage: python scripts/stembench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import functional as Fm  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(os.environ.get("B", 256))
bn = torch.nn.BatchNorm2d(64).cuda()
pool = torch.nn.MaxPool2d(3, 2, 1)
x = torch.randn(B, 64, 112, 112, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
g = torch.randn(B, 64, 56, 56, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)


def run(fn):
    for _ in range(3):
        y = fn(); y.backward(g)
    torch.cuda.synchronize()
    names = ["mrla_bn_plane_moments", "mrla_bn_act_fwd", "mrla_bn_plane_dmoments", "mrla_bn_act_bwd", "mrla_bn_relu_pool_fwd",
             "mrla_bn_relu_pool_dmoments", "mrla_bn_relu_pool_bwd"]
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(reps):
        e[0].record(); y = fn(); e[1].record(); y.backward(g); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    Fm.TIMER = t = Fm.KernelTimer(names)
    y = fn(); y.backward(g); torch.cuda.synchronize()
    Fm.TIMER = None
    ks = {k: round(v["ms"] * 1e3, 1) for k, v in t.summary().items()}
    return tf / reps * 1e3, tb / reps * 1e3, ks


for label, fn in (("stock bn_act + max_pool2d", lambda: pool(Fm.bn_act(x, bn, relu=True))), ("fused", lambda: Fm.bn_relu_maxpool(x, bn, pool))):
    f, b_, ks = run(fn)
    print(f"{label:28s} forward {f:7.1f} us   backward {b_:7.1f} us   kernels (us): {ks}", flush=True)
