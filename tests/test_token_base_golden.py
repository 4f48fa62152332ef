"""CPU: eager restatement of the DeiT + MRLA-base token module / network vs values the reference produced
(tests/golden/token_base.npz, made by oracle/make_goldens.py from deit/deit_mrla_base.py:204-277,280-450)."""
import numpy as np
import torch

from oracle import detgen, eager_models as em
from tests import cases

B, N, C, D, STEPS = 2, 17, 32, 16, 5


def chain_inputs(t):
    s = detgen.seed_of(f"tokbase/{t}")
    return (detgen.normalish((B, N, C), s) * 1.2 + 0.1).astype(np.float32), detgen.normalish((B, N, C), s + 1).astype(np.float32)


def load_det(mod, salt=0):
    vals = detgen.fill_state_dict(mod.state_dict(), salt=salt)
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


def run_chain(make_module, dev="cpu", dtype=torch.float32):
    mods, xs, outs, loss, K, V = [], [], [], 0.0, None, None
    for t in range(STEPS):
        m = make_module(t)
        load_det(m, salt=30 + t)
        m = m.to(dev)
        x, g = chain_inputs(t)
        x = torch.from_numpy(x).to(dev, dtype).requires_grad_(True)
        y, K, V = m(x, K, V)
        loss = loss + ((x + y).float() * torch.from_numpy(g).to(dev)).sum()
        mods.append(m); xs.append(x); outs.append(y)
    loss.backward()
    return mods, xs, outs


def rel(got, want):
    return cases.relmax(got, want, floor=1e-9)          # recorded (tests/cases.py)


def check_chain(mods, xs, outs, tol_act=cases.GOLD_TOL, tol_par=cases.GOLD_TOL, tol_qk=cases.QK_TOL):
    """vs what the reference computed in fp32 (tests/cases.py: the bounds; the cancelling Wq / Wk sums by name)."""
    G = cases.golden("token_base")
    for t in range(STEPS):
        assert rel(outs[t].detach().float().cpu().numpy(), G[f"{t}/module_out"]) < tol_act, t
        assert rel(xs[t].grad.float().cpu().numpy(), G[f"{t}/dx"]) < tol_act, t
        for pn, pv in mods[t].named_parameters():
            tol = tol_qk if ("Wq" in pn or "Wk" in pn) else tol_par
            assert rel(pv.grad.float().cpu().numpy(), G[f"{t}/grad/{pn}"]) < tol, (t, pn)


def test_eager_token_base_chain_vs_reference():
    # the eager restatement issues the very ATen calls of the reference: measured 0.0 on this container
    check_chain(*run_chain(lambda t: em.EagerTokenBaseModule(C, D, init_cell=(t % 4 == 0))), 1e-6, 1e-6, 1e-6)


def test_eager_deit_mrlab_logits_vs_reference():
    G = cases.golden("token_base")
    net = em.eager_deit_mrlab_tiny_patch16_224()
    vals = detgen.fill_state_dict(net.state_dict())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    net.eval()
    with torch.no_grad():
        logits = net(torch.from_numpy(cases.image_batch(2)))
    assert rel(logits.numpy(), G["deit_mrlab_tiny/eval2/logits"]) < 1e-5
