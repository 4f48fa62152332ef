"""Diagnostic: where does the bf16 bottleneck diverge from the staged float64 reference?  Replays MRLA_Bottleneck's forward
piecewise through the product's functional API, keeps every intermediate and its gradient, and compares with the staged
reference of tests/test_block_bf16_gpu.py.  python scripts/block_probe.py [stage1|stage3]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import functional as Fm, layers, resnet  # noqa: E402
from oracle import eager_models as em  # noqa: E402
from tests.test_block_bf16_gpu import _RefConv, _bn, _stock_directions, rnd  # noqa: E402

STOCK = os.environ.get("PROBE_FP64_CONVS") is None      # default: stock operators as black boxes, as the test does

which = sys.argv[1] if len(sys.argv) > 1 else "stage1"
b, inplanes, planes, hw = {"stage1": (64, 256, 64, 56), "stage3": (64, 1024, 256, 14)}[which]
torch.manual_seed(1234)
blk = resnet.MRLA_Bottleneck(inplanes, planes, drop_path=0.2)
for mod in blk.modules():
    if isinstance(mod, torch.nn.Conv2d) and mod.groups == 1:
        torch.nn.init.kaiming_normal_(mod.weight, mode="fan_out", nonlinearity="relu")
    elif isinstance(mod, torch.nn.BatchNorm2d):
        torch.nn.init.uniform_(mod.weight, 0.6, 1.4)
        torch.nn.init.uniform_(mod.bias, -0.3, 0.3)
blk = blk.cuda().to(memory_format=torch.channels_last).train()
ref = em.EagerLightBottleneck(inplanes, planes, drop_path=0.2).cuda().double().train()
ref.load_state_dict({k: v.double() for k, v in blk.state_dict().items()})
g = torch.Generator(device="cuda").manual_seed(5)
x = torch.relu(torch.randn((b, inplanes, hw, hw), device="cuda", generator=g) + 0.3).bfloat16().contiguous(memory_format=torch.channels_last)
gup = (torch.randn((b, inplanes, hw, hw), device="cuda", generator=g) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
keep = (torch.rand((b,), device="cuda", generator=g) >= 0.2).float()
dp = keep / 0.8
layers.drop_path_scale = lambda batch, p, training, device: dp

# ---- product, piecewise (what _BottleneckTrunk.trunk_pre + light_block_tail do) ----
P, PG = {}, {}


def keep_(name, t):
    P[name] = t.detach()
    if t.requires_grad:
        t.register_hook(lambda gr, n=name: PG.__setitem__(n, gr.detach()))
    return t


xp = x.clone().requires_grad_(True)
with torch.autocast("cuda", dtype=torch.bfloat16):
    y1, part1 = Fm._Conv1x1Fn.apply(xp, blk.conv1.weight.to(torch.bfloat16), True) if Fm.conv1x1_applies(blk.conv1, xp) else (blk.conv1(xp), None)
    keep_("y1", y1)
    z1 = keep_("z1", Fm.bn_act(y1, blk.bn1, True, pre_moments=part1 if part1 is not None and part1.numel() else None))
    y2 = keep_("y2", blk.conv2(z1))
    z2 = keep_("z2", Fm.bn_act(y2, blk.bn2, True))
    y3, part3 = Fm._Conv1x1Fn.apply(z2, blk.conv3.weight.to(torch.bfloat16), True)
    keep_("y3", y3)
    pre = Fm.bn_act(y3, blk.bn3, False, defer=True, pre_moments=part3 if part3.numel() else None)
    pre.register_hook(lambda gr: PG.__setitem__("pre", gr.detach()))
    xid = xp * 1.0 if os.environ.get("PROBE_SPLIT_IDENT") else xp          # (a separate node: its gradient is `do` alone)
    if xid is not xp:
        xid.register_hook(lambda gr: PG.__setitem__("ident", gr.detach()))
    out = layers.light_block_tail(pre, xid, blk.mrla, blk.bn_mrla, blk.drop_path, pre_activation=True)
out.backward(gup)

# ---- staged reference with retained grads ----
R = {}
wc = lambda w: w.float().bfloat16().double()  # noqa: E731
xr = x.double().requires_grad_(True)
ident = rnd(xr, "bwd")
M = b * hw * hw
sd = (lambda c: _stock_directions(c, M)) if STOCK else (lambda c: (False, False))
R["y1"] = rnd(_RefConv.apply(xr, wc(ref.conv1.weight), 0, *sd(blk.conv1)))
z, *_ = _bn(R["y1"], ref.bn1)
R["z1"] = rnd(torch.relu(z))
R["y2"] = rnd(_RefConv.apply(R["z1"], wc(ref.conv2.weight), 1, STOCK, STOCK))
z, *_ = _bn(R["y2"], ref.bn2)
R["z2"] = rnd(torch.relu(z))
R["y3"] = rnd(_RefConv.apply(R["z2"], wc(ref.conv3.weight), 0, *sd(blk.conv3)))
pre_r, *_ = _bn(R["y3"], ref.bn3)
pre_r = rnd(pre_r)
R["pre"] = pre_r
xt = rnd(torch.relu(pre_r + ident), "fwd")
m = ref.mrla(xt, ident)
z, *_ = _bn(m, ref.bn_mrla)
out_r = rnd(xt + dp.double()[:, None, None, None] * z, "fwd")
for t in R.values():
    t.retain_grad()
out_r.backward(gup.double())
torch.cuda.synchronize()


def cmp(name, got, want):
    want_r = want.float().bfloat16().float()
    d = (got.float() - want_r).abs()
    unit = 2.0 ** -7 * (want_r.abs() + 0.05 * want.abs().max().item())
    l2 = ((got.double() - want).norm() / want.norm()).item()
    print(f"{name:8s} L2 {l2:.2e}   >1ulp {((d / unit) > 1).float().mean().item():.2e}   >2ulp {((d / unit) > 2).float().mean().item():.2e}"
          f"   exact {(d == 0).float().mean().item():.3f}   max|want| {want.abs().max().item():.3g}")


for k in ("y1", "z1", "y2", "z2", "y3"):
    cmp(k, P[k], R[k].detach())
cmp("out", out.detach(), out_r.detach())
for k in ("pre", "y3", "z2", "y2", "z1", "y1"):
    cmp("d" + k, PG[k], R[k].grad)
# internal consistency of the product's bn3 backward: dy3 from ITS dpre and y3, in float64
dz, y3p = PG["pre"].double(), P["y3"].double()
mu, var = y3p.mean(dim=(0, 2, 3), keepdim=True), y3p.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
inv = 1.0 / torch.sqrt(var + blk.bn3.eps)
yh = (y3p - mu) * inv
gam = blk.bn3.weight.double()[None, :, None, None]
chk = gam * inv * (dz - dz.mean(dim=(0, 2, 3), keepdim=True) - yh * (dz * yh).mean(dim=(0, 2, 3), keepdim=True))
cmp("dy3|own", PG["y3"], chk)
# the same consistency check for bn2 / bn1 (with their ReLU): float64 BatchNorm backward of the PRODUCT's own inputs
for bnm, yk, zk in ((blk.bn2, "y2", "z2"), (blk.bn1, "y1", "z1")):
    yv, gz = P[yk].double(), PG[zk].double()
    mu_, var_ = yv.mean(dim=(0, 2, 3), keepdim=True), yv.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    inv_ = 1.0 / torch.sqrt(var_ + bnm.eps)
    yh_ = (yv - mu_) * inv_
    g_, b_ = bnm.weight.double()[None, :, None, None], bnm.bias.double()[None, :, None, None]
    dzz = gz * ((yh_ * g_ + b_) > 0)
    chk_ = g_ * inv_ * (dzz - dzz.mean(dim=(0, 2, 3), keepdim=True) - yh_ * (dzz * yh_).mean(dim=(0, 2, 3), keepdim=True))
    cmp(f"d{yk}|own", PG[yk], chk_)
    same_in = (PG[zk].float() == R[zk].grad.float().bfloat16().float())
    same_out = (PG[yk].float() == R[yk].grad.float().bfloat16().float())
    print(f"   d{zk} equal to the reference's on {same_in.float().mean().item():.4f}; where it is, d{yk} equal on "
          f"{same_out[same_in].float().mean().item():.4f}; channels with any differing d{zk}: "
          f"{(~same_in).any(dim=0).any(dim=1).any(dim=1).float().mean().item():.3f}")
# which constants did the product's bn3 backward apply?  per channel: dy3 - e*dz = f*y3 + h, least squares over the pixels
e_x = gam * inv
res = (PG["y3"].double() - e_x * dz)
xm, rm_ = y3p.mean(dim=(0, 2, 3), keepdim=True), res.mean(dim=(0, 2, 3), keepdim=True)
f_eff = ((y3p - xm) * (res - rm_)).sum(dim=(0, 2, 3)) / ((y3p - xm) ** 2).sum(dim=(0, 2, 3))
h_eff = rm_.flatten() - f_eff * xm.flatten()
c1 = dz.mean(dim=(0, 2, 3))
c2 = (dz * yh).mean(dim=(0, 2, 3))
f_x = (-gam * inv * inv).flatten() * c2
h_x = (gam * inv).flatten() * (-c1 + (inv * mu).flatten() * c2)
print("bn3 backward constants applied vs float64 of the same inputs (worst channel, relative to the largest):",
      f"f {((f_eff - f_x).abs().max() / f_x.abs().max()).item():.2e}  h {((h_eff - h_x).abs().max() / h_x.abs().max()).item():.2e}",
      f"  |f|max {f_x.abs().max().item():.2e} |h|max {h_x.abs().max().item():.2e} e~{e_x.mean().item():.2f}")
sums = Fm  # noqa
cmp("dx", xp.grad, xr.grad)
pref = dict(ref.named_parameters())
for name, p in blk.named_parameters():
    e = ((p.grad.double() - pref[name].grad).norm() / pref[name].grad.norm()).item()
    print(f"grad {name:28s} rel L2 {e:.2e}")
