"""Diagnostic: which tensor of a stage-2-first bottleneck (stride 2, downsample) differs between two runs on identical
inputs?  Piecewise forward through the product's functional API, twice; bitwise comparison of every intermediate."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import functional as Fm, layers, resnet  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
down = torch.nn.Sequential(torch.nn.Conv2d(256, 512, 1, stride=2, bias=False), torch.nn.BatchNorm2d(512))
blk = resnet.MRLA_Bottleneck(256, 128, stride=2, downsample=down, drop_path=0.0)
with torch.no_grad():
    blk.bn3.weight.fill_(0.5)
blk = blk.cuda().to(memory_format=torch.channels_last).train()
x = torch.relu(torch.randn(b, 256, 56, 56, device="cuda")).bfloat16().contiguous(memory_format=torch.channels_last)


def run():
    T = {}
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        w1 = blk.conv1.weight.to(torch.bfloat16)
        y1, p1 = Fm._Conv1x1Fn.apply(x, w1, True)
        T["y1"], T["part1"] = y1, p1
        blk.bn1.train()
        z1 = Fm.bn_act(y1, blk.bn1, True, pre_moments=p1 if p1.numel() else None)
        T["z1"] = z1
        y2 = blk.conv2(z1)
        T["y2"] = y2
        z2 = Fm.bn_act(y2, blk.bn2, True)
        T["z2"] = z2
        y3, p3 = Fm._Conv1x1Fn.apply(z2, blk.conv3.weight.to(torch.bfloat16), True)
        T["y3"], T["part3"] = y3, p3
        pre = Fm.bn_act(y3, blk.bn3, False, defer=True, pre_moments=p3 if p3.numel() else None)
        T["bn3_sc"], T["bn3_sh"] = pre._mrla_affine
        yd = blk.downsample[0](x)
        T["ds_conv"] = yd
        idn = Fm.bn_act(yd, blk.downsample[1], False)
        T["ds_bn"] = idn
        out = layers.light_block_tail(pre, idn, blk.mrla, blk.bn_mrla, blk.drop_path, pre_activation=True)
        T["out"] = out
    torch.cuda.synchronize()
    return {k: v.detach().clone() for k, v in T.items()}


a = run()
for rep in range(3):
    c = run()
    diffs = [(k, (a[k].float() - c[k].float()).abs().max().item()) for k in a if not torch.equal(a[k], c[k])]
    print(f"run {rep + 1} vs run 0:", diffs if diffs else "bit-identical")
