"""GPU parity of the fused BatchNorm2d(+ReLU) passes vs torch's own BatchNorm2d + relu (fp64 reference)."""
import numpy as np
import pytest
import torch

from tests.test_light_gpu import relmax

pytestmark = pytest.mark.gpu

SHAPES = [(3, 64, 112, 112), (4, 64, 56, 56), (2, 128, 28, 28), (3, 256, 14, 14), (5, 512, 7, 7), (2, 24, 5, 3)]


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("cl", [False, True], ids=["nchw", "nhwc"])
def test_bn_act_matches_torch(shape, relu, training, cl):
    from mrla_amd.functional import bn_act
    torch.manual_seed(0)
    b, c, h, w = shape
    fmt = torch.channels_last if cl else torch.contiguous_format
    x = (torch.randn(shape, device="cuda") * 1.7 + 0.4).contiguous(memory_format=fmt)
    g = torch.randn(shape, device="cuda").contiguous(memory_format=fmt)
    bn = torch.nn.BatchNorm2d(c).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.3, 0.3)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.8, 1.2)
    ref = torch.nn.BatchNorm2d(c).cuda().double()
    ref.load_state_dict(bn.state_dict())
    bn.train(training); ref.train(training)
    xa = x.clone().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    y = bn_act(xa, bn, relu)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    y.backward(g); yr.backward(g.double())
    assert relmax(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 5e-6
    assert relmax(xa.grad.cpu().numpy(), xr.grad.cpu().numpy()) < 2e-5
    assert relmax(bn.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy()) < 5e-5
    assert relmax(bn.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy()) < 5e-5
    assert relmax(bn.running_mean.cpu().numpy(), ref.running_mean.cpu().numpy()) < 5e-6
    assert relmax(bn.running_var.cpu().numpy(), ref.running_var.cpu().numpy()) < 5e-6
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


def test_bn_act_bf16():
    from mrla_amd.functional import bn_act
    torch.manual_seed(1)
    x = torch.randn(4, 256, 56, 56, device="cuda").bfloat16()
    g = torch.randn_like(x)
    bn = torch.nn.BatchNorm2d(256).cuda()
    ref = torch.nn.BatchNorm2d(256).cuda().double()
    xa = x.clone().requires_grad_(True)
    xr = x.double().requires_grad_(True)
    y = bn_act(xa, bn, True)
    yr = torch.relu(ref(xr))
    y.backward(g); yr.backward(g.double())
    want = yr.detach().float().bfloat16().float().cpu().numpy()
    got = y.detach().float().cpu().numpy()
    assert (np.abs(got - want) <= np.abs(want) * 2.0 ** -7 + 1e-6).all()
    gw = xr.grad.float().cpu().numpy()
    gg = xa.grad.float().cpu().numpy()
    bad = np.abs(gg - gw) > 2.0 ** -7 * (np.abs(gw) + 0.05 * np.abs(gw).max())
    assert bad.mean() < 1e-4        # a y that rounds across zero flips its ReLU mask
