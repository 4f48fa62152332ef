"""Which gradients of resnet50_mrlal's bf16 training step differ between a replayed HIP graph and eager launches?
usage: replay_grad_diag.py [batch] [benchmark 0|1] [deterministic 0|1]"""
import contextlib, io, sys, torch
sys.path.insert(0, ".")
from mrla_amd import models, graphs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.backends.cudnn.benchmark = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
torch.backends.cudnn.deterministic = (sys.argv[3] if len(sys.argv) > 3 else "0") == "1"
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    net = models.resnet50_mrlal(drop_path=0.0).cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
def fwdbwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(net(x).float(), y)
    net.zero_grad(set_to_none=True)
    loss.backward()
for _ in range(3):
    fwdbwd()
torch.cuda.synchronize()
ref = {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}
g = graphs.capture_step(fwdbwd, warmup=2)
names = [k for k, _ in net.named_parameters()]
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    bad = []
    for k, p in net.named_parameters():
        e = float((p.grad.double() - ref[k]).norm() / ref[k].norm().clamp_min(1e-30))
        if not (e < 5e-2):
            bad.append((names.index(k), k, e))
    print(f"b={B} benchmark={torch.backends.cudnn.benchmark} deterministic={torch.backends.cudnn.deterministic} replay {i}: "
          f"{len(bad)} of {len(names)} gradients off by > 5 %; the ones closest to the loss: "
          + "; ".join(f"{k} ({e:.2e})" for _, k, e in bad[-4:]), flush=True)
