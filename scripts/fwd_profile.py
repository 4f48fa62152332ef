"""GPU probe (not product): kernel-time table of the product's inference pass (eval, no_grad, bf16 autocast), or of
its training step with TRAIN=1."""
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
train = os.environ.get("TRAIN") == "1"
net = net.cuda().train(train)
x = torch.randn(b, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (b,), device="cuda")
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
iters = 5


def step():
    if not train:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            net(x)
        return
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(net(x).float(), y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
rows = sorted(((e.key, e.device_time_total / iters / 1e3, e.count // iters) for e in prof.key_averages()
               if e.device_time_total > 0), key=lambda r: -r[1])
tot = sum(r[1] for r in rows)
print(f"total kernel ms / pass: {tot:.3f}")
for k, ms, n in rows[:25]:
    print(f"{ms:8.3f} ms  {n:4d}x  {k[:110]}")
