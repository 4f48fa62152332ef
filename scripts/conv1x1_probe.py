"""GPU probe: 1x1 stride-1 convolutions of ResNet-50 in channels_last, bf16, fwd+bwd: MIOpen conv2d vs a plain GEMM (F.linear)."""
import time
import torch
import torch.nn.functional as F

def bench(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

tot = [0.0, 0.0]
for (cin, cout, hw, n) in [(64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1), (128, 512, 28, 4), (512, 128, 28, 3),
                           (512, 256, 28, 1), (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1), (512, 2048, 7, 3), (2048, 512, 7, 2)]:
    b = 256
    x = torch.randn(b, cin, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = (torch.randn(cout, cin, 1, 1, device="cuda", dtype=torch.bfloat16) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(b, cout, hw, hw, device="cuda", dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    def conv():
        y = F.conv2d(x, w); y.backward(g); x.grad = None; w.grad = None
    def gemm():
        y = F.linear(x.permute(0, 2, 3, 1), w.view(cout, cin)).permute(0, 3, 1, 2); y.backward(g); x.grad = None; w.grad = None
    a, c = bench(conv), bench(gemm)
    tot[0] += a * n; tot[1] += c * n
    print(f"1x1 {cin:4d}->{cout:4d} @{hw:2d} x{n}: conv2d {a*1e3:.3f} ms   linear {c*1e3:.3f} ms", flush=True)
print(f"sum over the network's 1x1 stride-1 convs: conv2d {tot[0]*1e3:.2f} ms   linear {tot[1]*1e3:.2f} ms")
