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
