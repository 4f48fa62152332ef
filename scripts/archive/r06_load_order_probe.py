"""One-off: does it matter whether libmrla_hip.so is loaded before the process's first HIP call?  (It did in
scripts/r06_small_batch.sh's first form: 'HIP runtime error at kernel launch'.)  Usage: python scripts/archive/r06_load_order_probe.py early|late"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
order = sys.argv[1]
import torch
from mrla_amd import _lib as L
if order == "early":
    lib = L.load()                       # before anything initialised HIP
x = torch.randn(4, 1000, device="cuda")
if order == "late":
    lib = L.load()
out = torch.empty(1000, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = lib.mrla_reduce_rows(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), 4, 1000, st)
torch.cuda.synchronize()
print(order, "rc", rc, "max err", float((out - x.sum(0)).abs().max()) if rc == 0 else None)
import subprocess
maps = open(f"/proc/{os.getpid()}/maps").read()
print(sorted({ln.split()[-1] for ln in maps.splitlines() if "amdhip" in ln or "libhsa" in ln}))
