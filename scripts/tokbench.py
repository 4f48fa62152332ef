"""GPU micro-benchmark + self-check of the DeiT token backward kernels through the C ABI at BASELINE config 4's shape
(b = 256, n = 197, c = 192, fp32 residual stream).  With KBENCH_LIB=<variant .so> the same calls run on an experiment
build (scripts/build_variant.sh); `--check` compares every output of the variant with the product library bit for bit.
Usage: [KBENCH_LIB=scripts/variants/libmrla_hip_<name>.so] python scripts/tokbench.py [reps] [--check]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L  # noqa: E402

PRODUCT = L.LIB_PATH
reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
B, N, C, D = int(os.environ.get("B", 256)), 197, int(os.environ.get("C", 192)), 16
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def load(path):
    lib = ctypes.CDLL(path)
    for name, argtypes in L.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
    return lib


def run(lib, tag, time_it=True):
    torch.manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = "cuda"
    x, o, g = (torch.randn(B, N, C, device=dev) for _ in range(3))
    wxw, wxb, wow, wob = (torch.randn(C, device=dev) * 0.3 + 1 for _ in range(4))
    wv, lam = torch.randn(C, 9, device=dev) * 0.3, torch.randn(C, device=dev)
    gate = torch.rand(B, C // D, device=dev)
    stats = torch.empty(B, N, 4, device=dev)
    mom = torch.empty(B, C, L.FWD_MOMENTS, device=dev)
    assert lib.mrla_token_norm_pool(P(x), P(o), P(wxw), P(wxb), 1e-6, P(stats), P(mom), B, N, C, L.F32, st) == 0
    dxn = torch.empty(B, N, C, device=dev)
    prow = lib.mrla_token_part_rows(B, N, C, L.F32)
    part = torch.empty(prow, C * L.TOKEN_PARTIALS, device=dev)
    bmom = torch.empty(B, C, L.BWD_MOMENTS, device=dev)
    side = int(round((N - 1) ** 0.5))
    dv = torch.randn(B, side, side, C, device=dev)
    dxn2, part2 = torch.empty_like(dxn), torch.empty_like(part)
    K = {
        "token_apply_bwd": (x.numel() * 4 * 4, lambda: lib.mrla_token_apply_bwd(
            P(g), P(x), P(o), P(stats), P(wxw), P(wxb), P(wow), P(wob), P(wv), P(gate), P(lam), P(dxn), P(part), P(bmom), B, N, C, D,
            L.F32, st)),
        "token_base_value_bwd": (x.numel() * 4 * 3, lambda: lib.mrla_token_base_value_bwd(
            P(g), P(x), P(stats), P(wxw), P(wxb), P(wv), P(dv), P(dxn2), P(part2), B, N, C, L.F32, st)),
    }
    res = {}
    for name, (nbytes, fn) in K.items():
        assert fn() == 0, name
        torch.cuda.synchronize()
        if time_it:
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / reps * 1e-3
            print(f"{tag:10s} {name:22s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:5.2f} TB/s", flush=True)
    res = dict(dxn=dxn.clone(), part=part.clone(), bmom=bmom[:, :, 1].clone(), dxn2=dxn2.clone(), part2=part2.clone())
    return res


variant = os.environ.get("KBENCH_LIB")
a = run(load(PRODUCT), "product")
if variant:
    b = run(load(os.path.abspath(variant)), os.path.basename(variant)[len("libmrla_hip_"):-3][:10])
    if "--check" in sys.argv:
        for k in a:
            same = torch.equal(a[k], b[k])
            nan = (torch.isnan(a[k]).sum().item(), torch.isnan(b[k]).sum().item())
            print(f"check {k:6s} {'bit-identical' if same else 'DIFFERS: max abs ' + str((a[k] - b[k]).abs().nan_to_num(1e30).max().item())} (NaNs: product {nan[0]}, variant {nan[1]})")
        assert all(torch.equal(a[k], b[k]) for k in a)
