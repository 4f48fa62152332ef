"""Are a stock convolution's input gradient (dX) and weight gradient (dW) -- MIOpen, bf16 channels_last -- the same from a
replayed HIP graph as from eager launches?  (scripts/miopen_wrw_graph_probe.py looked at dW of the stride-1 3x3 shapes only.)
The captured region poisons the pool with NaN first.
usage: miopen_bwd_graph_probe.py [benchmark 0|1] [batch] [deterministic 0|1] [dtype bf16|fp16|fp32]
Round 6: a dtype argument (fp32 = resnet/train.py's own recipe, no autocast), and for every strided 1x1 shape a second line for
the route the product takes in every dtype -- the stride-1 convolution on the subsampled input (functional.conv_bn_act)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

bm = (sys.argv[1] if len(sys.argv) > 1 else "0") == "1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
torch.backends.cudnn.benchmark = bm
torch.backends.cudnn.deterministic = (sys.argv[3] if len(sys.argv) > 3 else "0") == "1"
DT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[sys.argv[4] if len(sys.argv) > 4 else "bf16"]
torch.manual_seed(0)
# (cin, cout, kernel, stride, input map) of ResNet-50's stock convolutions
shapes = [(512, 512, 3, 2, 14), (1024, 2048, 1, 2, 14), (256, 256, 3, 2, 28), (512, 1024, 1, 2, 28), (128, 128, 3, 2, 56),
          (256, 512, 1, 2, 56), (512, 512, 3, 1, 7), (256, 256, 3, 1, 14), (128, 128, 3, 1, 28), (64, 64, 3, 1, 56),
          (3, 64, 7, 2, 224)]
shapes = [(s, False) for s in shapes] + [(s, True) for s in shapes if s[2] == 1 and s[3] > 1]
for (ci, co, k, stride, hw), subsampled in shapes:
    conv = torch.nn.Conv2d(ci, co, k, padding=k // 2, stride=stride, bias=False).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(B, ci, hw, hw, device="cuda").to(memory_format=torch.channels_last).requires_grad_(True)
    ho = (hw + 2 * (k // 2) - k) // stride + 1
    gy = torch.randn(B, co, ho, ho, device="cuda").to(memory_format=torch.channels_last)

    def run(poison):
        if poison:
            junk = torch.full((256 << 20,), float("nan"), device="cuda")     # 1 GB of NaN, freed right away
            del junk
        conv.weight.grad = None
        x.grad = None
        with torch.autocast("cuda", dtype=DT or torch.bfloat16, enabled=DT is not None):
            if subsampled:
                from mrla_amd import functional as Fm
                out = torch.nn.functional.conv2d(Fm._SubsampleFn.apply(x, stride, stride), conv.weight)
            else:
                out = conv(x)
        out.backward(gy.to(out.dtype))
        return conv.weight.grad, x.grad

    rw, rx = (t.clone() for t in run(False))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run(True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dw, dx = run(True)
    res = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        def err(a, r):
            if not torch.isfinite(a).all():
                return float("nan")
            return round(float((a.float() - r.float()).abs().max() / r.float().abs().max()), 4)
        res.append((err(dw, rw), err(dx, rx)))
    bad = any(not (e[0] < 0.05 and e[1] < 0.05) for e in res)
    print(f"benchmark={bm} det={torch.backends.cudnn.deterministic} {sys.argv[4] if len(sys.argv) > 4 else 'bf16'} "
          f"{'SUBSAMPLE + stride-1 ' if subsampled else ''}conv {ci}->{co} {k}x{k}/s{stride} [{B},{ci},{hw},{hw}]: (dW, dX) max rel err vs eager per "
          f"replay: {res}{'   <-- REPLAY DIFFERS' if bad else ''}", flush=True)
