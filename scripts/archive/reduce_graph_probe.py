"""Is `dy.sum(0)` (the bias gradient of nn.Linear: a [rows, n] -> [n] column sum, torch's multi-block reduce kernel) the same
from a replayed HIP graph as from an eager launch?  The captured region first fills a large scratch tensor with NaN and frees
it, so that whatever workspace / semaphores the reduction takes from the graph's pool start out poisoned on every replay.
Background: deit_mrlal_tiny's replayed training step returned NaN in a random handful of Linear bias gradients from the second
replay on (profiles/r05_notes.md)."""
import sys
import torch

torch.manual_seed(0)
for dtype in (torch.bfloat16, torch.float32):
    for (m, n) in [(32 * 197, 576), (32 * 197, 192), (32 * 197, 768), (256 * 197, 576), (256 * 197, 768), (6304, 1000)]:
        x = torch.randn(m, n, device="cuda").to(dtype)

        def run(poison):
            if poison:
                junk = torch.full((64 << 20,), float("nan"), device="cuda")
                del junk
            return x.sum(0)
        ref = run(False).float().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                run(True)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = run(True)
        res = []
        for _ in range(4):
            g.replay()
            torch.cuda.synchronize()
            nf = int((~torch.isfinite(out)).sum())
            err = float((out.float() - ref).abs().max() / ref.abs().max()) if nf == 0 else float("nan")
            res.append((nf, round(err, 5)))
        print(f"{dtype} [{m}, {n}].sum(0): (non-finite entries, max rel err vs eager) per replay: {res}", flush=True)
