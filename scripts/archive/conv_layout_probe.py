"""GPU probe: 3x3 convolutions of the ResNet-50 stages, bf16, fwd+bwd: NCHW tensors vs channels_last tensors."""
import time
import torch
import torch.nn.functional as F

def run(cl, cin, cout, hw, stride, b=256, reps=10):
    x = torch.randn(b, cin, hw, hw, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device="cuda", dtype=torch.bfloat16) * 0.05
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
        w = w.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True); w.requires_grad_(True)
    y = F.conv2d(x, w, stride=stride, padding=1)
    g = torch.randn_like(y)
    if cl:
        g = g.contiguous(memory_format=torch.channels_last)
    def it():
        y = F.conv2d(x, w, stride=stride, padding=1)
        y.backward(g)
        x.grad = None; w.grad = None
        return y
    for _ in range(3):
        y = it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        it()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, y.is_contiguous(memory_format=torch.channels_last), y.is_contiguous()

for (cin, cout, hw, s) in [(64, 64, 56, 1), (128, 128, 56, 2), (128, 128, 28, 1), (256, 256, 28, 2), (256, 256, 14, 1),
                           (512, 512, 14, 2), (512, 512, 7, 1)]:
    a = run(False, cin, cout, hw, s)
    c = run(True, cin, cout, hw, s)
    print(f"conv3x3 {cin}->{cout} @{hw} s{s}: NCHW {a[0]*1e3:.3f} ms   channels_last {c[0]*1e3:.3f} ms  (out cl={c[1]})", flush=True)
