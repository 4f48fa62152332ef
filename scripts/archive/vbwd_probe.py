import os, sys, ctypes, torch
sys.path.insert(0, ".")
from mrla_amd import _lib as L
if os.environ.get("KBENCH_LIB"): L.LIB_PATH = os.path.abspath(os.environ["KBENCH_LIB"])
lib = L.load()
P = lambda t: t.data_ptr() if t is not None else None
st = torch.cuda.current_stream().cuda_stream
for B, c, hw in ((128, 512, 28), (256, 512, 28)):
    x = torch.randn(B, hw, hw, c, device="cuda").bfloat16(); g = torch.randn_like(x); dv = torch.randn_like(x); pre = torch.randn_like(x)
    dx = torch.empty_like(x); wv = torch.randn(c, 9, device="cuda"); dyx = torch.randn(B, c, device="cuda"); cen = torch.randn(c, device="cuda")
    rows = lib.mrla_light_wgrad_rows(B, c, hw, hw, L.BF16, L.NHWC)
    dwv = torch.empty(rows, c * 9, device="cuda"); tm = torch.empty(rows, c, 2, device="cuda")
    fn = lambda: lib.mrla_base_value_bwd_dv(P(g), P(x), P(wv), P(dv), P(dyx), P(dx), P(dwv), P(pre), P(cen), P(tm), B, c, hw, hw, 3, L.BF16, L.NHWC, st)
    assert fn() == 0
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    print(os.environ.get("KBENCH_LIB", "product (wc 2)"), f"base_value_bwd_dv+bn3sums b={B} {c}x{hw}^2 rows {rows}:", [round(t, 1) for t in ts], "us")
