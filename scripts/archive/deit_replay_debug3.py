"""Bisecting, part 2.  usage: deit_replay_debug3.py variant   (fp32 | nomrla | mmbias | nomlp | noattn)"""
import sys, torch
sys.path.insert(0, ".")
from mrla_amd import vit, graphs, layers
torch.manual_seed(0)
variant = sys.argv[1]
B = 32

class _Lin(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return torch.nn.functional.linear(x, w, b)
    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        ones = torch.ones((1, dy2.shape[0]), dtype=dy2.dtype, device=dy2.device)
        return (dy2 @ w).view_as(x), dy2.t() @ x2, (ones @ dy2).view(-1)

class MMLinear(torch.nn.Linear):
    def forward(self, x):
        w, b = self.weight, self.bias
        if torch.is_autocast_enabled("cuda"):
            w, b = w.to(torch.bfloat16), b.to(torch.bfloat16)
            x = x.to(torch.bfloat16)
        return _Lin.apply(x, w, b)

def make_block():
    blk = vit.Block(dim=192, num_heads=3, dim_mrla=16, qkv_bias=True, drop_path=0.1)
    if variant == "mmbias":
        for mod in (blk.attn, blk.mlp):
            for name, m in list(mod.named_children()):
                if isinstance(m, torch.nn.Linear):
                    nm = MMLinear(m.in_features, m.out_features)
                    setattr(mod, name, nm)
    return blk.cuda().train()

blks = torch.nn.ModuleList([make_block() for _ in range(4)])
if variant == "nomrla":
    for b_ in blks:
        b_.mrla = None
x = torch.randn(B, 197, 192, device="cuda", requires_grad=True)

def block_fwd(b_, h):
    if variant == "nomrla":
        h = h + b_.drop_path(b_.attn(b_.norm1(h)))
        return h + b_.drop_path(b_.mlp(b_.norm2(h)))
    if variant == "noattn":
        ot = h
        h = h + b_.drop_path(b_.mlp(b_.norm2(h)))
        return b_.mrla(h, ot, fused_residual=True) if hasattr(b_.mrla, "forward") else h
    if variant == "nomlp":
        ot = h
        h = h + b_.drop_path(b_.attn(b_.norm1(h)))
        return b_.mrla(h, ot, fused_residual=True)
    return b_(h)

def fwdbwd():
    ctx = torch.autocast("cuda", dtype=torch.bfloat16, enabled=(variant != "fp32"))
    with ctx:
        h = x
        for b_ in blks:
            h = block_fwd(b_, h)
        loss = h.float().square().mean()
    blks.zero_grad(set_to_none=True)
    loss.backward()
g = graphs.capture_step(fwdbwd, warmup=2)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    bad = [k for k, p in blks.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(f"{variant} replay {i}: non-finite grads: {len(bad)} {bad[:6]}", flush=True)
