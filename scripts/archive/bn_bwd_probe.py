"""Diagnostic: is the fused BatchNorm backward (mrla_bn_plane_dmoments -> mrla_bn_stats_bwd -> mrla_bn_act_bwd) the float64
formula rounded once?  bf16 channels_last [64, 256, 56, 56]; prints the fraction of elements that are not bit-equal."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import functional as Fm  # noqa: E402

torch.manual_seed(0)
b, c, h, w = 64, 256, 56, 56
for relu in (False, True):
    bn = torch.nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.6, 1.4)
        bn.bias.uniform_(-0.3, 0.3)
    x = (torch.randn(b, c, h, w, device="cuda") * 0.7 + 0.3 * torch.randn(1, c, 1, 1, device="cuda")).bfloat16()
    x = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = (torch.randn(b, c, h, w, device="cuda") * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    y = Fm.bn_act(x, bn, relu)
    y.backward(g)
    torch.cuda.synchronize()
    xd, gd = x.detach().double(), g.double()
    mu, var = xd.mean(dim=(0, 2, 3), keepdim=True), xd.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    inv = 1.0 / torch.sqrt(var + bn.eps)
    yh = (xd - mu) * inv
    gam, bet = bn.weight.double()[None, :, None, None], bn.bias.double()[None, :, None, None]
    yv = yh * gam + bet
    dz = gd * (yv > 0) if relu else gd
    want = gam * inv * (dz - dz.mean(dim=(0, 2, 3), keepdim=True) - yh * (dz * yh).mean(dim=(0, 2, 3), keepdim=True))
    wr = want.float().bfloat16()
    got = x.grad
    neq = (got != wr)
    ratio = (got.double() / want)[want.abs() > 0.01]
    print(f"relu={relu}: forward not bit-equal {(y.detach() != (torch.relu(yv) if relu else yv).float().bfloat16()).float().mean().item():.3e}; "
          f"backward not bit-equal {neq.float().mean().item():.3e}; median got/want - 1 = {(ratio.median() - 1).item():.2e}; "
          f"L2 {((got.double() - want).norm() / want.norm()).item():.2e}")
    # the same through the C ABI pieces with float64 constants, to separate the constants from the elementwise pass
    e = (gam * inv).float()
    c1 = dz.mean(dim=(0, 2, 3), keepdim=True)
    c2 = (dz * yh).mean(dim=(0, 2, 3), keepdim=True)
    f_ = (-gam * inv * inv * c2).float()
    h_ = (gam * inv * (-c1 + inv * mu * c2)).float()
    emu = torch.addcmul(torch.addcmul(h_.expand_as(xd).contiguous(), f_.expand_as(xd), xd.float()), e.expand_as(xd), dz.float()).bfloat16()
    print(f"           fp32 emulation of e*dz + (f*x + h) with float64-derived constants: not bit-equal to float64 {(emu != wr).float().mean().item():.3e}, "
          f"to the kernel {(emu != got).float().mean().item():.3e}")
