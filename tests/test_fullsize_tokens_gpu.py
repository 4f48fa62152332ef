"""GPU: the DeiT token modules at BASELINE size (config 4 of BASELINE.json: deit_mrlal_tiny_patch16_224, b = 256, n = 197,
c = 192, fp32 residual stream as under autocast; and the MRLA-base token variant of 8f rank 3 at the same size) tied to the
small cases that tests/test_tokens_gpu.py / tests/test_token_base_gpu.py compare with the oracle and the reference,
through properties that do not depend on the batch size:

  * every image is independent on this path (LayerNorm per token, pooling / gate / history per image): the outputs and
    the input gradients of images [i, j, k] inside a 256-image launch equal the same three images run as a batch of 3 --
    bit for bit (all per-image sums are taken inside one workgroup in a batch-independent order);
  * parameter gradients are sums over images: the full batch's equal the sum over eight 32-image sub-batches (fp32
    re-association only)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, N, C, D = 256, 197, 192, 32
PICK = [0, 101, 255]


def _inputs(seed, b=B):
    g = torch.Generator(device="cuda").manual_seed(seed)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x = r(b, N, C) * (0.5 + torch.rand(C, device="cuda", generator=g)) + 0.3 * r(C)
    return x, r(b, N, C), r(b, N, C)            # x, o_prev, upstream gradient


def _light_module(seed):
    from mrla_amd import layers
    torch.manual_seed(seed)
    m = layers.mrlal_module(C, D).cuda()
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.3 + (1.0 if p.dim() == 1 and p.numel() == C else 0.0))
    return m


def _run_light(m, x, o, g):
    x, o = x.clone().requires_grad_(True), o.clone().requires_grad_(True)
    m.zero_grad(set_to_none=True)
    out = m(x, o, fused_residual=True)
    out.backward(g)
    return out.detach(), x.grad, o.grad, [p.grad.clone() for p in m.parameters()]


def test_token_light_module_full_batch_slices_and_parameter_gradients():
    m = _light_module(5)
    x, o, g = _inputs(50)
    out, dx, do, gp = _run_light(m, x, o, g)
    assert torch.isfinite(out).all() and torch.isfinite(dx).all()
    o3, dx3, do3, _ = _run_light(m, x[PICK], o[PICK], g[PICK])
    assert torch.equal(out[PICK], o3), "outputs of images inside the full launch differ from the batch of 3"
    assert torch.equal(dx[PICK], dx3) and torch.equal(do[PICK], do3), "input gradients differ from the batch of 3"
    acc = None
    for s in range(0, B, 32):
        _, _, _, gs = _run_light(m, x[s:s + 32], o[s:s + 32], g[s:s + 32])
        acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
    for (name, _), a, b in zip(m.named_parameters(), gp, acc):
        err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        assert err < 2e-5, (name, err)


def _base_chain(mods, xs, gs):
    xs = [x.clone().requires_grad_(True) for x in xs]
    for m in mods:
        m.zero_grad(set_to_none=True)
    K = V = None
    loss = 0.0
    outs = []
    for m, x, g in zip(mods, xs, gs):
        y, K, V = m(x, K, V)
        loss = loss + (y * g).sum()
        outs.append(y.detach())
    loss.backward()
    return outs, [x.grad for x in xs], [[p.grad.clone() for p in m.parameters()] for m in mods]


def test_token_base_module_chain_full_batch_slices_and_parameter_gradients():
    """Four layers of one stage (init_cell on the first, as ViT_mrlab resets the history every 4 blocks), the fused token
    path (LayerNorm on load, V_t into the ring, rows written in place)."""
    from mrla_amd import functional as F_, layers
    mods = []
    for t in range(4):
        torch.manual_seed(20 + t)
        m = layers.mrlab_module(C, D, init_cell=(t == 0)).cuda()
        m.mrla.history_hint = 4
        with torch.no_grad():
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if p.dim() == 1 and p.numel() == C else 0.0))
        mods.append(m)
    data = [_inputs(70 + t) for t in range(4)]
    xs, gs = [d[0] for d in data], [d[2] for d in data]
    assert F_.token_base_supported(xs[0], D)
    outs, dxs, gps = _base_chain(mods, xs, gs)
    o3, dx3, _ = _base_chain(mods, [x[PICK] for x in xs], [g[PICK] for g in gs])
    for t in range(4):
        assert torch.isfinite(outs[t]).all()
        assert torch.equal(outs[t][PICK], o3[t]), t
        assert torch.equal(dxs[t][PICK], dx3[t]), t
    acc = None
    for s in range(0, B, 32):
        _, _, g8 = _base_chain(mods, [x[s:s + 32] for x in xs], [g[s:s + 32] for g in gs])
        acc = g8 if acc is None else [[a + b for a, b in zip(la, lb)] for la, lb in zip(acc, g8)]
    for t, m in enumerate(mods):
        for (name, _), a, b in zip(m.named_parameters(), gps[t], acc[t]):
            err = (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
            assert err < 2e-5, (t, name, err)
