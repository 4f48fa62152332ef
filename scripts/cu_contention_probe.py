"""GPU-box probe (not product): what does losing CUs to a collective's channel kernels cost the eager-launched training step?

Under torch.distributed the gradient all-reduce of resnet50_mrlal (103 MB per step, resnet/train.py:174) runs on RCCL's
channel kernels -- a few persistent workgroups that occupy CUs and move the buckets -- while the backward pass runs.
Many kernels here are sized "one workgroup per CU" (the GEMM planners, the stencil passes' image groups).  This probe times
bench.py's eager-launched step with a stand-in on a side stream: scripts/micro/occupy.hip on 0 / 8 / 16 / 32 workgroups,
each copying a slice of a 206 MB buffer (read + write = the traffic of a 103 MB all-reduce's local part) for the whole
duration of the step.  Usage: python scripts/cu_contention_probe.py [steps]"""
import contextlib
import ctypes
import io
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrla_amd import models  # noqa: E402

so = "/tmp/liboccupy.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC",
                       os.path.join(ROOT, "scripts", "micro", "occupy.hip"), "-o", so])
lib = ctypes.CDLL(so)
lib.occupy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = models.resnet50_mrlal(drop_path=0.2).cuda().train()
x = torch.randn(256, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (256,), device="cuda")
step = bench.make_step(net, bench.sgd(net.parameters()), x, y)
for _ in range(8):
    step()
torch.cuda.synchronize()
src = torch.empty(103 * 2 ** 20, dtype=torch.uint8, device="cuda")
dst = torch.empty_like(src)
side = torch.cuda.Stream()
base = None
for n_wg in (0, 8, 16, 32, 0):
    # rounds chosen so that the side kernel outlives the timed steps (it is stopped by finishing, not by a flag):
    # calibrate on one launch first
    rounds = 1
    if n_wg:
        per = (src.numel() // n_wg) // 16 * 16
        t0 = time.perf_counter()
        lib.occupy(src.data_ptr(), dst.data_ptr(), per, n_wg, 4, side.cuda_stream)
        side.synchronize()
        one = (time.perf_counter() - t0) / 4
        rounds = max(1, int(steps * 0.040 / one * 1.3))
        moved = 2 * per * n_wg / one / 1e9
    torch.cuda.synchronize()
    if n_wg:
        lib.occupy(src.data_ptr(), dst.data_ptr(), per, n_wg, rounds, side.cuda_stream)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / steps
    still_running = not side.query()
    side.synchronize()
    if base is None:
        base = dt
    extra = f"  side kernel: {moved:.0f} GB/s alone, still running at the end: {still_running}" if n_wg else ""
    print(f"{n_wg:3d} workgroups beside the WHOLE step: {dt * 1e3:7.3f} ms/step ({100 * (dt / base - 1):+5.1f} %){extra}", flush=True)

# the realistic duty cycle: ONE burst per step that moves what a 103 MB all-reduce moves locally (read + write 103 MB),
# launched beside the step (the real buckets go out during the backward pass)
for n_wg in (8, 16, 32):
    per = (src.numel() // n_wg) // 16 * 16
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.occupy(src.data_ptr(), dst.data_ptr(), per, n_wg, 1, side.cuda_stream)
    side.synchronize()
    burst = time.perf_counter() - t0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        lib.occupy(src.data_ptr(), dst.data_ptr(), per, n_wg, 1, side.cuda_stream)
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{n_wg:3d} workgroups, one {burst * 1e3:.2f} ms burst (2 x 103 MB) per step: {dt * 1e3:7.3f} ms/step "
          f"({100 * (dt / base - 1):+5.1f} %)", flush=True)
