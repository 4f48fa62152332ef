"""GPU probe: launch-size scaling of the NHWC BatchNorm affine passes."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mrla_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for c, hw in ((512, 7), (256, 14), (64, 56)):
    for B in (1, 8, 32, 64, 128, 256):
        x = torch.randn(B, c, hw, hw, device="cuda").bfloat16().contiguous(memory_format=torch.channels_last)
        g = torch.randn_like(x); y = torch.empty_like(x)
        sc, sh, cb = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda"), torch.randn(c, 3, device="cuda")
        tf1 = timeit(lambda: lib.mrla_bn_act_fwd(P(x), P(sc), P(sh), 1, P(y), B, c, hw, hw, L.BF16, L.NHWC, st))
        tf0 = timeit(lambda: lib.mrla_bn_act_fwd(P(x), P(sc), P(sh), 0, P(y), B, c, hw, hw, L.BF16, L.NHWC, st))
        tb = timeit(lambda: lib.mrla_bn_act_bwd(P(g), P(x), P(sc), P(sh), P(cb), 1, P(y), B, c, hw, hw, L.BF16, L.NHWC, st))
        tr = timeit(lambda: torch.relu(x))
        print(f"c={c} hw={hw} B={B:3d} {x.numel()*2/1e6:7.2f} MB  fwd relu {tf1:6.1f}us  fwd norelu {tf0:6.1f}us  bwd {tb:6.1f}us  torch.relu {tr:6.1f}us")
