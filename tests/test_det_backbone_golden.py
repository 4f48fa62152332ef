"""CPU: eager restatement of the mmdetection backbone vs values produced by the reference
(tests/golden/det_backbone.npz <- mmdetection/mmdet/models/backbones/resnet_mrlal.py, see oracle/make_goldens.py gen_det),
and the host-side surface of the product class (constructor, state_dict keys, train() freezing rules)."""
import numpy as np
import torch

from oracle import detgen, eager_models as em
from tests import cases

DET_SHAPE = (2, 3, 96, 160)


def det_inputs():
    x = detgen.normalish(DET_SHAPE, detgen.seed_of("det/img")).astype(np.float32)
    gs = [detgen.normalish((2, c, 96 // s, 160 // s), detgen.seed_of(f"det/g{c}")).astype(np.float32)
          for c, s in ((256, 4), (512, 8), (1024, 16), (2048, 32))]
    return x, gs


def load_det(net):
    vals = detgen.fill_state_dict(net.state_dict())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


def rel(got, want):
    return cases.relmax(got, want, floor=1e-9)          # recorded (tests/cases.py)


def check_against_golden(net, dev, tol_map, tol_grad, amp=False):
    G = cases.golden("det_backbone")
    x_np, gs = det_inputs()
    x = torch.from_numpy(x_np).to(dev)
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        maps = net(x)
    assert len(maps) == 4
    for i, m in enumerate(maps):
        assert tuple(m.shape) == (2, 256 << i, 96 >> (i + 2), 160 >> (i + 2))
        assert rel(m.float().cpu().numpy()[:, ::8], G[f"eval/map{i}"]) < tol_map, i
    net.train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        maps = net(x)
        loss = sum((m.float() * torch.from_numpy(g).to(dev)).mean() for m, g in zip(maps, gs))
    loss.backward()
    assert abs(loss.item() - float(G["train/loss"][0])) < tol_map * max(1.0, abs(float(G["train/loss"][0])))
    frozen = sorted(k for k, p in net.named_parameters() if p.grad is None)
    assert frozen == list(G["train/frozen"])
    n = 0
    for k, p in net.named_parameters():
        if p.grad is None:
            continue
        want = G["train/gsum/" + k]
        if want[1] < 1e-7:
            continue
        assert abs(float(p.grad.double().abs().sum()) - want[1]) <= tol_grad * want[1] + 1e-6, k   # fp32 noise floor on the tiny Wq/Wk sums
        n += 1
    assert n > 100
    # norm_eval: no running statistic may have moved, no BatchNorm is in training mode
    assert not any(m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))


def test_eager_det_backbone_vs_reference():
    net = em.EagerDetBackbone(frozen_stages=1, norm_eval=True)
    load_det(net)
    check_against_golden(net, "cpu", 5e-6, 2e-3)          # maps measured 9.5e-7


def test_product_det_backbone_surface():
    from mrla_amd import mmdet_backbone as mb
    G = cases.golden("det_backbone")
    net = mb.ResNet_mrlal(frozen_stages=1, norm_eval=True, style="pytorch", drop_path=0.1,
                          init_cfg=None)
    assert sorted(net.state_dict().keys()) == list(G["state_keys"])
    assert all(m.bn3.weight.detach().abs().sum().item() == 0.0 for m in net.modules() if isinstance(m, mb.MRLA_Bottleneck))
    assert not any(isinstance(m.drop_path, type(net.layer1[0].mrla)) for m in net.layer1)      # no DropPath module in use
    assert all(isinstance(m.drop_path, torch.nn.Identity) for m in net.modules() if isinstance(m, mb.MRLA_Bottleneck))
    net.train()
    frozen = sorted(k for k, p in net.named_parameters() if not p.requires_grad)
    assert frozen == list(G["train/frozen"])
    assert not net.layer1.training and net.layer2.training
    assert not any(m.training for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d))
    net2 = mb.ResNet_mrlal(norm_eval=False, frozen_stages=-1).train()
    assert all(m.training for m in net2.modules() if isinstance(m, torch.nn.BatchNorm2d))
    assert all(p.requires_grad for p in net2.parameters())


def test_product_det_backbone_loads_a_classification_checkpoint(tmp_path):
    """init_cfg=dict(type='Pretrained', checkpoint=...) as in the reference's mmdet configs: a classification checkpoint
    written by resnet/train.py (`{'state_dict': {'module.<key>': ...}}`, with fc.*) initialises the backbone."""
    import io
    from contextlib import redirect_stdout

    from mrla_amd import mmdet_backbone as mb, models
    with redirect_stdout(io.StringIO()):
        cls = models.resnet50_mrlal()
    with torch.no_grad():
        for p in cls.parameters():
            p.add_(0.01)
    path = tmp_path / "checkpoint.pth.tar"
    torch.save({"epoch": 1, "arch": "resnet50_mrlal", "state_dict": {"module." + k: v for k, v in cls.state_dict().items()}},
               path)
    net = mb.ResNet_mrlal(init_cfg=dict(type="Pretrained", checkpoint=str(path)))
    sd, ref = net.state_dict(), cls.state_dict()
    assert "fc.weight" not in sd
    for k in sd:
        assert torch.equal(sd[k], ref[k]), k
