"""Where does deit_mrlal_tiny's backward go non-finite under HIP-graph replay with stochastic depth on?  (batch 32, bf16 autocast)
usage: deit_replay_debug.py [batch] [drop_path] [fused_attn 0|1] [mrla 0|1]"""
import contextlib, io, sys, torch
sys.path.insert(0, ".")
from mrla_amd import vit, graphs, layers
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
vit.Attention.fused_attn = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
variant = sys.argv[4] if len(sys.argv) > 4 else "std"
if variant == "nomask":      # stochastic depth drawn but all-keep: is it the zeros in the mask?
    orig = layers.drop_path_scale
    layers.drop_path_scale = lambda batch, p, tr, dev: None if (p == 0.0 or not tr) else (torch.rand((batch,), device=dev) * 0 + 1)
with contextlib.redirect_stdout(io.StringIO()):
    net = vit.deit_mrlal_tiny_patch16_224(drop_path_rate=dp).cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
out = {}
def fwdbwd():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = torch.nn.functional.cross_entropy(net(x).float(), y)
    net.zero_grad(set_to_none=True)
    loss.backward()
    out["loss"] = loss.detach()
g = graphs.capture_step(fwdbwd, warmup=2)
for i in range(3):
    g.replay(); torch.cuda.synchronize()
    bad = [k for k, p in net.named_parameters() if not torch.isfinite(p.grad).all()]
    print(f"fused_attn={vit.Attention.fused_attn} {variant} dp={dp} replay {i}: loss {float(out['loss']):.5f}; non-finite grads: {len(bad)}; last (closest to the input) 6: {bad[:6]}; first from the top: {bad[-4:]}", flush=True)
