"""CPU: the eager torch restatement of the full models vs logits / gradient sums the reference produced
(tests/golden/models.npz; BASELINE config 1 = resnet50_mrlal forward on 8x3x224x224)."""
import numpy as np
import pytest
import torch

from oracle import detgen, eager_models as em
from tests import cases


def load_det(net):
    vals = detgen.fill_state_dict(net.state_dict())
    missing = net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return missing


def rel(got, want):
    return cases.relmax(got, want, floor=1e-9)          # recorded (tests/cases.py)


@pytest.mark.parametrize("arch,factory,nb", [("resnet50_mrlal", em.eager_resnet50_mrlal, 8),
                                             ("resnet50_mrlab", em.eager_resnet50_mrlab, 4),
                                             ("deit_mrlal_tiny", em.eager_deit_mrlal_tiny_patch16_224, 4)])
def test_full_model_logits_and_grads(arch, factory, nb):
    G = cases.golden("models")
    torch.manual_seed(0)
    net = factory()
    load_det(net)
    net.eval()
    with torch.no_grad():
        logits = net(torch.from_numpy(cases.image_batch(nb)))
    assert rel(logits.numpy(), G[f"{arch}/eval{nb}/logits"]) < 5e-6                   # (measured 5.4e-7)
    net.train()
    xb = torch.from_numpy(cases.image_batch(4, "img-train"))
    logits = net(xb)
    assert rel(logits.detach().numpy(), G[f"{arch}/train4/logits"]) < 1e-5          # (measured 5.8e-7)
    loss = torch.nn.functional.cross_entropy(logits, torch.arange(4) * 37 % 1000)
    loss.backward()
    grads = dict(net.named_parameters())
    n = 0
    for k in G.files:
        pre = f"{arch}/train4/gsum/"
        if k.startswith(pre):
            g = grads[k[len(pre):]].grad.numpy().ravel().astype(np.float64)
            want = G[k]
            if want[1] < 1e-4:      # mathematically-zero gradients (a shift in front of a train-mode BN): pure noise
                continue
            # fp32 end-to-end gradients through 16 train-mode BNs at batch 4 are noise-limited (~1e-3)
            assert abs(np.abs(g).sum() - want[1]) <= 2e-2 * want[1] + 1e-7, k
            n += 1
    assert n > 10


def test_state_dict_keys_match_reference_layout():
    sd = em.eager_resnet50_mrlal().state_dict()
    for k, shape in {"layer1.0.mrla.lambda_t": (256, 1, 1), "layer1.0.mrla.mrla.Wq.weight": (1, 1, 5),
                     "layer4.2.mrla.mrla.Wk.weight": (1, 1, 7), "layer3.5.mrla.mrla.Wv.weight": (1024, 1, 3, 3),
                     "layer2.3.bn_mrla.running_var": (512,), "layer1.0.downsample.1.weight": (256,)}.items():
        assert tuple(sd[k].shape) == shape
    assert sum(p.numel() for p in em.eager_resnet50_mrlal().parameters()) == 25_738_452
    assert sum(p.numel() for p in em.eager_resnet50_mrlab().parameters()) == 25_742_580
    assert sum(p.numel() for p in em.eager_deit_mrlal_tiny_patch16_224().parameters()) == 5_749_792
    sdb = em.eager_resnet50_mrlab().state_dict()
    assert "stages.2.5.mrla.mrla.Wv.weight" in sdb and "conv1.6.weight" in sdb
