"""GPU probe (not product): a few eager fwd+bwd steps of resnet50_mrlal (NCHW, bf16 autocast) for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import eager_models as em  # noqa: E402

b = int(os.environ.get("B", 256))
arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50_mrlal"
plain = os.environ.get("NO_MRLA") == "1"
torch.manual_seed(0)
net = getattr(em, "eager_" + arch)().cuda().train()
if plain:   # ablation: same backbone without the MRLA tail
    for m in net.modules():
        if isinstance(m, em.EagerLightBottleneck):
            m.forward = (lambda blk: (lambda x: em._trunk_forward(blk, x)[0]))(m)
x = torch.randn(b, 3, 224, 224, device="cuda")
y = torch.randint(0, 1000, (b,), device="cuda")
opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
for i in range(int(os.environ.get("STEPS", 4))):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(net(x).float(), y)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(loss))
