"""Per-stage time of the MRLA path kernels INSIDE the training step (resnet50_mrlal b=256, bf16 autocast, SGD -- bench.py's
step, launched kernel by kernel with a HIP-event pair around every C-ABI launch): the in-situ A/B instrument.
Usage: [KBENCH_LIB=scripts/variants/libmrla_hip_<name>.so] python scripts/instep_kernels.py [steps] [name-substring] [arch] [batch]
Prints one line per (kernel, bytes-per-launch) = per stage: launches per step, average us per launch, GB/s."""
import contextlib
import io
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import _lib as L  # noqa: E402

if os.environ.get("KBENCH_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["KBENCH_LIB"])
import bench  # noqa: E402
from mrla_amd import functional as Fm, models, vit  # noqa: E402

if os.environ.get("MRLA_LEAN") is not None:       # A/B of the tail without a stored x_t (functional.LEAN) against the storing passes
    Fm.LEAN = os.environ["MRLA_LEAN"] == "1"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2] if len(sys.argv) > 2 else ""
arch = sys.argv[3] if len(sys.argv) > 3 else "resnet50_mrlal"
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 256

torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = (getattr(vit, arch)(drop_path_rate=0.2) if arch.startswith("deit") else getattr(models, arch)(drop_path=0.2))
net = net.cuda().train()
if getattr(net, "channels_last", False):
    net.to(memory_format=torch.channels_last)
opt = bench.sgd(net.parameters())
gx = torch.Generator(device="cuda").manual_seed(0)
gy = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(batch, 3, 224, 224, device="cuda", generator=gx)
y = torch.randint(0, 1000, (batch,), device="cuda", generator=gy)
step = bench.make_step(net, opt, x, y)
for _ in range(6):
    step()
torch.cuda.synchronize()
dt_plain = bench.timed(step, steps, 0)
timer = Fm.KernelTimer()
Fm.TIMER = timer
dt = bench.timed(step, steps, 0)
Fm.TIMER = None
rows = {}
for name, nbytes, e0, e1, alg, path in timer.records:
    if only and only not in name:
        continue
    d = rows.setdefault((name, nbytes), [0, 0.0])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
out = {"lib": os.path.basename(L.LIB_PATH), "lean": Fm.LEAN, "arch": arch, "batch": batch, "steps": steps,
       "eager_ms_per_step": round(1e3 * dt_plain / steps, 3), "events_ms_per_step": round(1e3 * dt / steps, 3), "kernels": []}
tot = {}
for (name, nbytes), (n, ms) in sorted(rows.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    us = 1e3 * ms / n
    out["kernels"].append({"kernel": name, "bytes": nbytes, "launches_per_step": n / steps, "avg_us": round(us, 2),
                           "GBps": round(nbytes / us / 1e3, 1) if nbytes else None})
    tot[name] = tot.get(name, 0.0) + ms / steps
out["ms_per_step_by_kernel"] = {k: round(v, 4) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}
print(json.dumps(out))
