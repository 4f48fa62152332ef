"""Diagnostic: where do resnet50_mrlal's activations differ between the weight-bank path and per-convolution casts?"""
import contextlib
import io
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mrla_amd import models, resnet  # noqa: E402

torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = models.resnet50_mrlal(drop_path=0.0).cuda().train()
with torch.no_grad():
    for mod in net.modules():
        if isinstance(mod, resnet._BottleneckTrunk):
            mod.bn3.weight.fill_(0.5)
x = torch.randn(4, 3, 224, 224, device="cuda")
outs = []
for rep, use_bank in enumerate((True, False, True, False)):
    acts = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, o.detach().float().clone()))
             for n, m in net.named_modules() if isinstance(m, resnet._BottleneckTrunk)]
    for mod in net.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.reset_running_stats()
    if use_bank:
        os.environ.pop("MRLA_NO_WEIGHT_BANK", None)
    else:
        os.environ["MRLA_NO_WEIGHT_BANK"] = "1"
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits = net(x)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    outs.append((logits.detach().float().clone(), acts))
for i in range(1, 4):
    print(f"run {i} vs run 0 (bank {'on' if i % 2 == 0 else 'off'}): logits max diff {(outs[i][0] - outs[0][0]).abs().max().item():.3e}")
    for n in outs[0][1]:
        d = (outs[i][1][n] - outs[0][1][n]).abs().max().item()
        if d > 0:
            print(f"   first differing block: {n} max diff {d:.3e} (max |act| {outs[0][1][n].abs().max().item():.3g})")
            break
