"""GPU probe (not product): kernel-time table of the product's inference pass (eval, no_grad, bf16 autocast)."""
import contextlib
import io
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import models, vit  # noqa: E402

b = int(os.environ.get("B", 256))
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50_mrlal"
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = (getattr(vit, arch) if arch.startswith("deit") else getattr(models, arch))()
net = net.cuda().eval()
x = torch.randn(b, 3, 224, 224, device="cuda")
iters = 5
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3):
        net(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            net(x)
        torch.cuda.synchronize()
rows = sorted(((e.key, e.device_time_total / iters / 1e3, e.count // iters) for e in prof.key_averages()
               if e.device_time_total > 0), key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print(f"total kernel ms / pass: {tot:.3f}")
for k, ms, n in rows[:25]:
    print(f"{ms:8.3f} ms  {n:4d}x  {k[:110]}")
