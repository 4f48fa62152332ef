"""Bisecting the non-finite Linear bias gradients of deit_mrlal_tiny under HIP-graph replay with stochastic depth on.
usage: deit_replay_debug2.py variant     (cache_off | block | linear_mul | mrla_off)"""
import contextlib, io, sys, torch
sys.path.insert(0, ".")
from mrla_amd import vit, graphs, layers
torch.manual_seed(0)
variant = sys.argv[1]
B = 32
out = {}

def check(named, tag):
    def go(fn):
        g = graphs.capture_step(fn, warmup=2)
        for i in range(3):
            g.replay(); torch.cuda.synchronize()
            bad = [k for k, p in named() if p.grad is not None and not torch.isfinite(p.grad).all()]
            print(f"{tag} replay {i}: non-finite grads: {len(bad)} {bad[:5]}", flush=True)
    return go

if variant in ("cache_off", "cache_on"):
    with contextlib.redirect_stdout(io.StringIO()):
        net = vit.deit_mrlal_tiny_patch16_224(drop_path_rate=0.1).cuda().train()
    x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
    def fwdbwd():
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=(variant == "cache_on")):
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        net.zero_grad(set_to_none=True)
        loss.backward()
    check(net.named_parameters, variant)(fwdbwd)
elif variant in ("block", "block_noattn", "block_nomrla"):
    blk = vit.Block(dim=192, num_heads=3, dim_mrla=16, qkv_bias=True, drop_path=0.1).cuda().train()
    blks = torch.nn.ModuleList([blk] + [vit.Block(dim=192, num_heads=3, dim_mrla=16, qkv_bias=True, drop_path=0.1).cuda().train() for _ in range(3)])
    x = torch.randn(B, 197, 192, device="cuda", requires_grad=True)
    def fwdbwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = x
            for b_ in blks:
                h = b_(h)
            loss = h.float().square().mean()
        blks.zero_grad(set_to_none=True)
        loss.backward()
    check(blks.named_parameters, variant)(fwdbwd)
elif variant == "linear_mul":
    lins = torch.nn.ModuleList([torch.nn.Linear(192, 192).cuda() for _ in range(24)])
    x = torch.randn(B, 197, 192, device="cuda", requires_grad=True)
    def fwdbwd():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            h = x
            for l in lins:
                s = torch.floor(0.9 + torch.rand((B,), device="cuda")) / 0.9
                h = h + l(h) * s.to(h.dtype).view(-1, 1, 1)
            loss = h.float().square().mean()
        lins.zero_grad(set_to_none=True)
        loss.backward()
    check(lins.named_parameters, variant)(fwdbwd)
