"""GPU probe (not product): time the eager restatement of resnet50_mrlal, bf16 autocast, fwd and fwd+bwd,
NCHW vs channels_last, to decide which activation layout the HIP path should favour."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import eager_models as em  # noqa: E402


def bench(net, x, y, steps, train):
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
            if train:
                loss = torch.nn.functional.cross_entropy(out.float(), y)
        if train:
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    b = int(os.environ.get("B", 256))
    dev = "cuda"
    print(torch.cuda.get_device_name(0), "B", b, "NHWC env", os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"))
    for arch in sys.argv[1:] or ["resnet50_mrlal"]:
        for cl in (False, True):
            torch.manual_seed(0)
            net = getattr(em, "eager_" + arch)().to(dev)
            x = torch.randn(b, 3, 224, 224, device=dev)
            y = torch.randint(0, 1000, (b,), device=dev)
            if cl:
                net = net.to(memory_format=torch.channels_last)
                x = x.contiguous(memory_format=torch.channels_last)
            net.train()
            t_tr = bench(net, x, y, 8, True)
            net.eval()
            with torch.no_grad():
                t_fw = bench(net, x, y, 8, False)
            print(f"{arch} channels_last={cl}: fwd+bwd {t_tr*1e3:.1f} ms ({b/t_tr:.0f} img/s)  fwd(eval) {t_fw*1e3:.1f} ms ({b/t_fw:.0f} img/s)", flush=True)
            del net


if __name__ == "__main__":
    main()
