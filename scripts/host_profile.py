"""GPU-box probe (not product): where the HOST time of an eager-launched training step goes (cProfile over N steps of
bench.py's step, no synchronisation inside the profiled region).  Usage: python scripts/host_profile.py [arch] [steps]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mrla_amd import models, vit  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50_mrlal"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(os.environ.get("B", 256))
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = getattr(vit, arch)(drop_path_rate=0.2) if arch.startswith("deit") else getattr(models, arch)(drop_path=0.2)
net = net.cuda().train()
x = torch.randn(batch, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (batch,), device="cuda")
step = bench.make_step(net, bench.sgd(net.parameters()), x, y)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_host = (time.perf_counter() - t0) / steps
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / steps
print(f"host issue time {t_host*1e3:.2f} ms/step; with the GPU drained {t_all*1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
