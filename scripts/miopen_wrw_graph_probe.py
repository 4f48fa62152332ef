"""Is a stock 3x3 convolution's weight gradient (MIOpen, bf16 channels_last) the same from a replayed HIP graph as from eager
launches?  The captured region first fills a large scratch tensor with NaN and frees it, so that whatever workspace the
convolution's backward takes from the graph's pool starts out poisoned on every replay -- a solver that relies on memory it
zeroed only once shows up as non-finite / different dW.
Usage: python scripts/miopen_wrw_graph_probe.py [benchmark 0|1] [batch,batch,...]     (default batches: 8 and 256)"""
import sys

import torch

bm = (sys.argv[1] if len(sys.argv) > 1 else "0") == "1"
torch.backends.cudnn.benchmark = bm
torch.manual_seed(0)
batches = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8, 256]
# the 3x3 convolutions of the ResNet-50 / -101 bottlenecks: (channels, input map, stride)
shapes = [(512, 7, 1), (256, 14, 1), (128, 28, 1), (64, 56, 1), (512, 14, 2), (256, 28, 2), (128, 56, 2)]
if len(sys.argv) <= 2:
    cases = [(8, c, hw, st) for c, hw, st in shapes[:4]] + [(256, 512, 7, 1)]
else:
    cases = [(b, c, hw, st) for b in batches for c, hw, st in shapes]
for (b, c, hw, stride) in cases:
    conv = torch.nn.Conv2d(c, c, 3, padding=1, stride=stride, bias=False).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(b, c, hw, hw, device="cuda").to(memory_format=torch.channels_last).requires_grad_(True)
    ho = (hw - 1) // stride + 1
    gy = torch.randn(b, c, ho, ho, device="cuda").to(memory_format=torch.channels_last)

    def run(poison):
        if poison:
            junk = torch.full((64 << 20,), float("nan"), device="cuda")     # 256 MB of NaN, freed right away
            del junk
        conv.weight.grad = None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = conv(x)
        out.backward(gy.to(out.dtype))
        return conv.weight.grad

    ref = run(False).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            run(True)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        dw = run(True)
    res = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        nf = int((~torch.isfinite(dw)).sum())
        err = float((dw - ref).abs().max() / ref.abs().max()) if nf == 0 else float("nan")
        res.append((nf, round(err, 5)))
    print(f"benchmark={bm} conv {c}->{c} 3x3/s{stride} [{b},{c},{hw},{hw}]: (non-finite dW entries, max rel err vs eager) per replay: {res}", flush=True)
