"""One-off: where does the HOST spend the eagerly launched detection step (2 x 3 x 800 x 1344; 11 - 13 ms eager against 8.8 ms of
GPU work under graph replay)?  cProfile over 20 eager steps, sorted by own time.  Usage: python scripts/archive/r06_host_profile.py"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from benchkit import common, core
sys.argv = ["bench.py", "--arch", "det_resnet50_mrlal", "--shape", "2x3x800x1344", "--no-baselines"]
args = common.parse()
common.AUTOCAST_DTYPE = common.AUTOCAST["bf16"]
net, what = core.build_model(args)
net = net.cuda().train()
if getattr(net, "channels_last", False):
    net.to(memory_format=torch.channels_last)
x, _ = core.synthetic_batch(args)
opt = common.sgd((p for p in net.parameters() if p.requires_grad), lr=common.DET_LR)
step = core.make_det_step(net, opt, x)
for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"20 eager steps: host issue {1e3 * t_issue / 20:.2f} ms per step, with the final synchronize {1e3 * t_all / 20:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
