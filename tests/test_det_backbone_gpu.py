"""GPU parity: the detection backbone (mmdetection/mmdet/models/backbones/resnet_mrlal.py) on the HIP path --
non-square inputs, BatchNorm frozen to its running statistics during training, frozen first stage."""
import pytest
import torch

from oracle import eager_models as em
from tests.test_det_backbone_golden import check_against_golden, load_det

pytestmark = pytest.mark.gpu


def test_det_backbone_fp32_vs_reference_golden():
    from mrla_amd import mmdet_backbone as mb
    net = mb.ResNet_mrlal(frozen_stages=1, norm_eval=True)
    load_det(net)
    check_against_golden(net.cuda(), "cuda", 5e-6, 5e-3)          # maps measured 8.3e-7


def test_det_backbone_wide_image_fp32_and_bf16_autocast_vs_eager():
    """A 3 x 192 x 320 image pair: stage-1 rows are 80 pixels wide (wider than one wave), widths are not multiples of the
    7-column strips.  fp32: product == eager restatement to rounding.  bf16 autocast: the product's distance to the fp32
    result is no worse than the eager bf16 pipeline's."""
    from mrla_amd import mmdet_backbone as mb
    net, ref = mb.ResNet_mrlal(frozen_stages=1, norm_eval=True), em.EagerDetBackbone(frozen_stages=1, norm_eval=True)
    load_det(net); load_det(ref)
    net, ref = net.cuda().train(), ref.cuda().train()
    x = torch.randn(2, 3, 192, 320, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))

    def run(model, amp):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            maps = model(x)
            loss = sum(m.float().square().mean() for m in maps)
        loss.backward()
        return [m.detach().float() for m in maps], {k: p.grad.double().ravel().clone() for k, p in model.named_parameters()
                                                    if p.grad is not None}

    r_maps, r_grads = run(ref, False)
    a_maps, a_grads = run(net, False)
    dist = lambda u, v: ((u - v).norm() / v.norm()).item()
    for u, v in zip(a_maps, r_maps):
        assert u.shape == v.shape and dist(u, v) < 1e-5
    assert set(a_grads) == set(r_grads) and len(a_grads) > 100
    for k, g in r_grads.items():
        if g.norm() > 1e-9:
            assert dist(a_grads[k], g) < 2e-3, k
    b_maps, b_grads = run(ref, True)
    c_maps, c_grads = run(net, True)
    for u, e, v in zip(c_maps, b_maps, r_maps):
        assert dist(u, v) <= 1.5 * dist(e, v) + 1e-3, (dist(u, v), dist(e, v))
    for k, g in r_grads.items():
        if g.norm() > 1e-9:
            assert dist(c_grads[k], g) <= 2.0 * dist(b_grads[k], g) + 2e-2, k


@pytest.mark.parametrize("shape", [(2, 256, 200, 336), (2, 512, 100, 168)], ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("mode", ["norm_eval", "train"])
def test_light_tail_at_detection_size_vs_eager(shape, mode):
    """The block tail where the detection backbone runs it (mmdetection/mmdet/models/backbones/resnet_mrlal.py:283-293 at
    2 x 3 x 800 x 1344: stage-1 maps 200 x 336, stage-2 100 x 168): two large images instead of 256 small ones, 48 / 24 column
    strips per image -- the passes spread them over strip RANGES and cut the rows into ranges as well (gridDim.z;
    mrla_light_wgrad_rows / mrla_light_bmom_splits / mrla_light_mom_splits count them: 8 row ranges of 25 / 13 rows here), each
    leaving partial rows / records that are folded into the ones the small kernels read.  Product (bf16,
    channels_last, x_t = relu(pre + identity) formed inside) vs the eager restatement in fp32 on the same bf16 inputs:
    out and both input gradients to the last bf16 bits, every parameter gradient to fp32-accumulation accuracy.
    `norm_eval`: bn_mrla is a fixed affine (what the backbone trains with); `train`: batch statistics."""
    from mrla_amd import _lib as L
    from mrla_amd.functional import mrla_light
    b, c, h, w = shape
    lib = L.load()
    assert lib.mrla_light_wgrad_rows(b, c, h, w, L.BF16, L.NHWC) > b and lib.mrla_light_bmom_splits(b, c, h, w, L.BF16, L.NHWC) > 1
    assert lib.mrla_light_mom_splits(b, c, h, w, L.BF16, L.NHWC) >= 24               # (3 or 6 strip ranges x 8 row ranges; more under MRLA_TEST_ROW_RANGES=2)
    g = torch.Generator(device="cuda").manual_seed(c + h)
    mk = lambda s=1.0: (s * torch.randn(b, c, h, w, device="cuda", generator=g)).bfloat16().contiguous(memory_format=torch.channels_last)
    pre, idn, gup = mk(), mk(), mk(0.1)
    ref = em.EagerLightModule(c).cuda()
    bn = torch.nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        ref.mrla.Wv.weight.mul_(0.5)
        bn.weight.uniform_(0.5, 1.5, generator=g); bn.bias.uniform_(-0.2, 0.2, generator=g)
        bn.running_mean.uniform_(-0.1, 0.1, generator=g); bn.running_var.uniform_(0.5, 1.5, generator=g)
    bn.train(mode == "train")
    bn0 = {k: v.clone() for k, v in bn.state_dict().items()}
    params = dict(wq=ref.mrla.Wq.weight, wk=ref.mrla.Wk.weight, wv=ref.mrla.Wv.weight, lam=ref.lambda_t, gamma=bn.weight, beta=bn.bias)

    def grads():
        out = {k: p.grad.detach().double().clone() for k, p in params.items()}
        for p in params.values():
            p.grad = None
        return out
    # product
    pg, ig = pre.clone().requires_grad_(True), idn.clone().requires_grad_(True)
    out = mrla_light(pg, params["wq"], params["wk"], params["wv"], 32, o_prev=ig, lam=params["lam"],
                     bn=dict(weight=bn.weight, bias=bn.bias, running_mean=bn.running_mean, running_var=bn.running_var,
                             training=(mode == "train"), momentum=0.1, eps=1e-5), res=True, pre_activation=True)
    out.backward(gup)
    got = dict(out=out.detach().float(), dpre=pg.grad.float(), didn=ig.grad.float(), **grads())
    stats = (bn.running_mean.clone(), bn.running_var.clone())
    bn.load_state_dict(bn0)
    # eager restatement, fp32 arithmetic on the same values (resnet_mrlal.py:108-112)
    pr, ir = pre.float().requires_grad_(True), idn.float().requires_grad_(True)
    xt = torch.relu(pr + ir)
    xt = xt + (xt.detach().bfloat16().float() - xt.detach())              # x_t as the product forms it: rounded once to bf16
    o = xt + bn(ref(xt, ir))
    o.backward(gup.float())
    want = dict(out=o.detach(), dpre=pr.grad, didn=ir.grad, **grads())
    rel = lambda a, r: ((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-30)).item()
    for k in ("out", "dpre", "didn"):
        a, r = got[k], want[k]
        bad = (a - r).abs() > 2.0 ** -6 * (r.abs() + 0.02 * r.abs().max())
        assert bad.float().mean().item() < 1e-4, (k, bad.float().mean().item())
        assert rel(a, r) < 4e-3, (k, rel(a, r))
    for k in params:
        assert rel(got[k], want[k]) < 2e-2, (k, rel(got[k], want[k]))
    if mode == "train":
        assert torch.allclose(stats[0], bn.running_mean, rtol=1e-3, atol=1e-5) and torch.allclose(stats[1], bn.running_var, rtol=1e-3, atol=1e-5)
