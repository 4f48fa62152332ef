"""Diagnostic (it found that sums of the UNROUNDED dpre differ systematically from sums of the stored tensor: where the MRLA
branch adds less than half an ulp to dOut, rounding drops it): the deferred-bn3 sums taken inside mrla_light_apply_bwd (pre / pre_center / pre_tmom) vs float64 sums of the
dpre it wrote.  [b, 256, 56, 56] bf16 channels_last, drop-path mask with dropped images."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import functional as Fm  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
c, h, w = 256, 56, 56
torch.manual_seed(0)
cl = lambda t: t.bfloat16().contiguous(memory_format=torch.channels_last)  # noqa: E731
y3 = cl(torch.randn(b, c, h, w, device="cuda") * 0.5 + 0.4 * torch.randn(1, c, 1, 1, device="cuda")).requires_grad_(True)
idn = cl(torch.relu(torch.randn(b, c, h, w, device="cuda")))
g = cl(torch.randn(b, c, h, w, device="cuda") * 0.1)
bn3, bnm = torch.nn.BatchNorm2d(c).cuda(), torch.nn.BatchNorm2d(c).cuda()
with torch.no_grad():
    for m in (bn3, bnm):
        m.weight.uniform_(0.6, 1.4); m.bias.uniform_(-0.3, 0.3)
wq, wk, wv, lam = (torch.randn(1, 1, 5, device="cuda") * 0.5, torch.randn(1, 1, 5, device="cuda") * 0.5,
                   torch.randn(c, 1, 3, 3, device="cuda") * 0.3, torch.randn(c, 1, 1, device="cuda"))
dp = ((torch.rand(b, device="cuda") >= 0.2).float() / 0.8)
caught = {}
orig_put = Fm._DeferredBnBox.put


def put(self, dpre, tmom, rows):
    caught.update(dpre=dpre.detach().clone(), tmom=tmom.detach().clone(), rows=rows, center=self.center.detach().clone())
    orig_put(self, dpre, tmom, rows)


Fm._DeferredBnBox.put = put
pre = Fm.bn_act(y3, bn3, relu=False, defer=True)
out = Fm.mrla_light(pre, wq, wk, wv, 32, o_prev=idn, lam=lam,
                    bn=dict(weight=bnm.weight, bias=bnm.bias, running_mean=bnm.running_mean, running_var=bnm.running_var,
                            training=True, momentum=0.1, eps=1e-5), dp=dp, res=True, pre_activation=True)
out.backward(g)
torch.cuda.synchronize()
assert caught, "the fused sums were not produced"
dpre, y = caught["dpre"].double(), y3.detach().double()
s = caught["tmom"].double().sum(0)                        # [c, 2]
cen = caught["center"].double()
e1 = dpre.sum(dim=(0, 2, 3))
e2 = (dpre * (y - cen[None, :, None, None])).sum(dim=(0, 2, 3))
print("rows", caught["rows"], "center vs batch mean", (cen - y.mean(dim=(0, 2, 3))).abs().max().item())
print("sum dpre:            max |fused - float64| / max|float64| =", ((s[:, 0] - e1).abs().max() / e1.abs().max()).item())
print("sum dpre*(y3 - mean): max |fused - float64| / max|float64| =", ((s[:, 1] - e2).abs().max() / e2.abs().max()).item(),
      " worst channel", int((s[:, 1] - e2).abs().argmax()))
print("absolute: max|fused - f64| sum1", (s[:, 0] - e1).abs().max().item(), "sum2", (s[:, 1] - e2).abs().max().item(),
      " max|f64| sum1", e1.abs().max().item(), "sum2", e2.abs().max().item(),
      " expected std of the rounding noise of dpre in sum1:", (8.5e-4 * dpre.pow(2).mean().sqrt() * (dpre[:, 0].numel()) ** 0.5).item())
# rounding bias: bf16 RNE of a quantity produced in fp32 is unbiased, but dpre = 0 wherever x_t <= 0: count the mask
print("fraction of dpre == 0:", (dpre == 0).float().mean().item())
# per image-group row (NOTE: the apply pass walks the image groups in reverse order: row r = group rows-1-r)
per = (dpre * (y - cen[None, :, None, None])).sum(dim=(2, 3))           # [b, c]
rows = caught["rows"]
per = per.view(rows, b // rows, c).sum(1).flip(0)
d = (caught["tmom"][:, :, 1].double() - per).abs()
print("per row: worst", d.max().item(), "at row", int(d.max(dim=1)[0].argmax()), " typical |row sum|", per.abs().mean().item())
