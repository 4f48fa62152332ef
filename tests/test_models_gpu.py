"""GPU parity of the full models: product (HIP MRLA tails) vs the eager restatement on the same GPU with the
same deterministic weights, and vs logits the reference itself produced on CPU (tests/golden/models.npz)."""
import numpy as np
import pytest
import torch

from oracle import detgen, eager_models as em
from tests import cases

pytestmark = pytest.mark.gpu


def load_det(net):
    vals = detgen.fill_state_dict(net.state_dict())
    net.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})


def rel(got, want):
    return cases.relmax(got, want, floor=1e-9)          # recorded (tests/cases.py)


def test_resnet50_mrlal_eval_logits_match_reference_and_eager():
    from mrla_amd import models
    torch.backends.cudnn.allow_tf32 = False
    G = cases.golden("models")
    net = models.resnet50_mrlal().cuda()
    ref = em.eager_resnet50_mrlal().cuda()
    load_det(net)
    ref.load_state_dict(net.state_dict())           # same keys -> strict load
    net.eval(); ref.eval()
    x = torch.from_numpy(cases.image_batch(8)).cuda()
    with torch.no_grad():
        y, yr = net(x), ref(x)
    # MIOpen fp32 convolutions differ from the CPU's by ~1e-5; the product and eager share them exactly
    assert rel(y.cpu().numpy(), yr.cpu().numpy()) < 5e-6                            # (measured 5.4e-7)
    assert rel(y.cpu().numpy(), G["resnet50_mrlal/eval8/logits"]) < 1e-5           # (measured 7.0e-7)


def test_resnet50_mrlal_train_step_matches_eager():
    from mrla_amd import models
    G = cases.golden("models")
    net = models.resnet50_mrlal().cuda()
    ref = em.eager_resnet50_mrlal().cuda()
    load_det(net)
    ref.load_state_dict(net.state_dict())
    net.train(); ref.train()
    x = torch.from_numpy(cases.image_batch(4, "img-train")).cuda()
    tgt = (torch.arange(4) * 37 % 1000).cuda()
    y, yr = net(x), ref(x)
    # train mode at batch 4: 69 BatchNorms over 4 x h x w samples each (measured 1.5e-6 / 1.1e-6)
    assert rel(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 2e-5
    assert rel(y.detach().cpu().numpy(), G["resnet50_mrlal/train4/logits"]) < 2e-5
    torch.nn.functional.cross_entropy(y, tgt).backward()
    torch.nn.functional.cross_entropy(yr, tgt).backward()
    gp, gr = dict(net.named_parameters()), dict(ref.named_parameters())
    worst, dots = (0.0, ""), np.zeros(3)
    tight = (0.0, "")
    for k in gp:
        a, b = gp[k].grad.cpu().numpy().ravel().astype(np.float64), gr[k].grad.cpu().numpy().ravel().astype(np.float64)
        if np.abs(b).sum() < 1e-4:
            continue                       # mathematically-zero gradients: noise
        e = np.abs(a - b).sum() / np.abs(b).sum()
        worst = max(worst, (e, k))
        # the MRLA / BatchNorm parameters whose gradients are plain sums well above the noise: this is where the deferred
        # bn3 affine and the fused producer are wired end to end, so these get their own, tight bound
        if k.endswith(("mrla.mrla.Wv.weight", "mrla.lambda_t", "bn_mrla.weight", "bn_mrla.bias", "bn3.weight", "bn3.bias")):
            tight = max(tight, (e, k))
        dots += np.array([a @ b, a @ a, b @ b])
    print("model-level gradient check: worst", worst, "worst of the MRLA / bn3 parameters", tight)
    assert tight[0] < 2e-2, tight
    # fp32 noise through 16 train-mode BNs + ReLU masks at batch 4 (and MIOpen may pick different conv algorithms for
    # the two models): per-parameter sums of the tiny Wq/Wk gradients are noise-limited, the block-level tests pin every
    # gradient to 5e-5; the whole gradient must still point the same way
    assert worst[0] < 0.2, worst
    assert dots[0] / np.sqrt(dots[1] * dots[2]) > 0.9999
    sp, sr = net.state_dict(), ref.state_dict()
    for k in ("layer1.0.bn_mrla.running_mean", "layer4.2.bn_mrla.running_var", "layer2.1.bn_mrla.num_batches_tracked"):
        assert rel(sp[k].float().cpu().numpy(), sr[k].float().cpu().numpy()) < 1e-5, k       # (measured 4.1e-7)


@pytest.mark.parametrize("arch", ["resnet50_mrlal", "resnet50_mrlab"])
def test_nchw_and_channels_last_paths_agree(arch):
    """The class default runs NHWC inside (MRLA-base: slot-major NHWC history rings); switching it off runs the NCHW
    kernels: same logits, same gradients."""
    from mrla_amd import models
    a, b = getattr(models, arch)().cuda(), getattr(models, arch)().cuda()
    load_det(a)
    b.load_state_dict(a.state_dict())
    b.channels_last = False
    b.to(memory_format=torch.contiguous_format)
    wa, wb = next(m.weight for m in a.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3)), \
        next(m.weight for m in b.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3))
    assert wa.is_contiguous(memory_format=torch.channels_last) and wb.is_contiguous()
    a.train(); b.train()
    x = torch.from_numpy(cases.image_batch(4, "img-train")).cuda()
    ya, yb = a(x), b(x)
    assert rel(ya.detach().cpu().numpy(), yb.detach().cpu().numpy()) < 2e-5               # (measured 1.7e-6)
    ya.square().mean().backward(); yb.square().mean().backward()
    dots = np.zeros(3)
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        ga, gb = pa.grad.cpu().numpy().ravel().astype(np.float64), pb.grad.cpu().numpy().ravel().astype(np.float64)
        dots += np.array([ga @ gb, ga @ ga, gb @ gb])
    assert dots[0] / np.sqrt(dots[1] * dots[2]) > 0.9999


def test_state_dict_roundtrip_and_api_surface():
    from mrla_amd import models
    net = models.resnet50_mrlal(drop_rate=0.1, drop_path=0.2, num_classes=10)
    sd = em.eager_resnet50_mrlal(num_classes=10).state_dict()
    assert set(net.state_dict().keys()) == set(sd.keys())
    assert sum(p.numel() for p in models.resnet50_mrlal().parameters()) == 25_738_452


@pytest.mark.parametrize("arch", ["resnet50_mrlal", "resnet50_mrlab"])
def test_five_sgd_steps_track_the_eager_restatement(arch):
    """Same initial weights, same batches, SGD(momentum, weight decay) as resnet/train.py:199-201, fp32, no stochastic depth:
    the loss trajectory and the BatchNorm running statistics of the product follow the eager restatement step by step."""
    from mrla_amd import models
    net, ref = getattr(models, arch)().cuda(), getattr(em, "eager_" + arch)().cuda()
    load_det(net)
    ref.load_state_dict(net.state_dict())
    net.train(); ref.train()
    opt_a = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    opt_b = torch.optim.SGD(ref.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator(device="cuda").manual_seed(0)
    la, lb = [], []
    for step in range(5):
        x = torch.randn(8, 3, 224, 224, device="cuda", generator=g)
        y = torch.randint(0, 1000, (8,), device="cuda", generator=g)
        for model, opt, losses in ((net, opt_a, la), (ref, opt_b, lb)):
            loss = torch.nn.functional.cross_entropy(model(x), y)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
    for a, b in zip(la, lb):
        assert abs(a - b) <= 5e-3 * abs(b) + 1e-4, (la, lb)     # small lr: the two trajectories stay within rounding drift
    sa, sb = net.state_dict(), ref.state_dict()
    for k in sa:
        if k.endswith("running_var") or k.endswith("running_mean"):
            assert rel(sa[k].float().cpu().numpy(), sb[k].float().cpu().numpy()) < 2e-2, k     # 5 steps of rounding drift
    assert int(sa["bn1.num_batches_tracked"]) == 5


def test_half_precision_model_runs_and_keeps_its_statistics_dtype():
    """model.half() (the reference tolerates it): BatchNorm running statistics become 2-byte buffers; the kernels read and
    update float32, so the wrappers must go through float32 copies and write the update back -- never hand a 2-byte
    buffer to the C ABI as float*."""
    from mrla_amd import models
    net = models.resnet50_mrlal().cuda()
    load_det(net)
    x = torch.from_numpy(cases.image_batch(4)).cuda()
    net.eval()
    with torch.no_grad():
        want = net(x)
    h = models.resnet50_mrlal().cuda()
    h.load_state_dict(net.state_dict())
    h.half().eval()
    assert h.layer1[0].bn_mrla.running_var.dtype == torch.float16
    with torch.no_grad():
        got = h(x.half())
    assert torch.isfinite(got).all()
    assert rel(got.float().cpu().numpy(), want.cpu().numpy()) < 3e-2          # fp16 storage of every activation
    h.train()
    rv0 = h.layer2[1].bn_mrla.running_var.clone()
    h(x.half()).float().square().mean().backward()
    rv1 = h.layer2[1].bn_mrla.running_var
    assert rv1.dtype == torch.float16 and torch.isfinite(rv1).all() and not torch.equal(rv0, rv1)
    assert h.layer2[1].mrla.lambda_t.grad is not None and torch.isfinite(h.layer2[1].mrla.lambda_t.grad).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_kernels_launch_on_the_tensor_device_not_the_current_one():
    from mrla_amd.functional import mrla_light
    assert torch.cuda.current_device() == 0
    dev = torch.device("cuda", 1)
    c = 64
    x = torch.randn(2, c, 8, 8, device=dev).requires_grad_(True)
    args = (torch.randn(1, 1, 3, device=dev), torch.randn(1, 1, 3, device=dev), torch.randn(c, 1, 3, 3, device=dev))
    y1 = mrla_light(x, *args, 32)
    y1.sum().backward()
    with torch.cuda.device(1):
        x2 = x.detach().clone().requires_grad_(True)
        y2 = mrla_light(x2, *args, 32)
        y2.sum().backward()
    torch.cuda.synchronize(1)
    assert torch.equal(y1, y2) and torch.equal(x.grad, x2.grad)


@pytest.mark.parametrize("arch", ["resnet50_mrlal", "resnet50_mrlab"])
def test_bf16_autocast_train_step_tracks_the_eager_restatement(arch):
    """The configuration bench.py times (bf16 autocast, channels_last): the 1x1 convolutions then run on the MFMA GEMM with
    the BatchNorm statistics in its epilogue.  One training step vs the eager restatement under the same autocast: logits
    and loss agree to bf16 accuracy, every gradient is finite and the whole gradient points the same way."""
    from mrla_amd import functional as Fm, models
    net, ref = getattr(models, arch)().cuda(), getattr(em, "eager_" + arch)().cuda()
    load_det(net)
    ref.load_state_dict(net.state_dict())
    net.train(); ref.train()
    x = torch.from_numpy(cases.image_batch(8, "img-train")).cuda()
    tgt = (torch.arange(8) * 37 % 1000).cuda()
    used = []
    orig = Fm._Conv1x1Fn.apply
    try:
        Fm._Conv1x1Fn.apply = staticmethod(lambda *a: (used.append(1), orig(*a))[1])
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = net(x)
    finally:
        Fm._Conv1x1Fn.apply = orig
    assert len(used) >= 18, "the MFMA GEMM path was not taken under autocast"   # 13 conv3 + 4 conv1 + 1 downsample (k <= 256)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yr = ref(x)
    la = torch.nn.functional.cross_entropy(y.float(), tgt)
    lb = torch.nn.functional.cross_entropy(yr.float(), tgt)
    assert abs(la.item() - lb.item()) < 3e-2 * abs(lb.item())
    assert rel(y.detach().float().cpu().numpy(), yr.detach().float().cpu().numpy()) < 6e-2
    la.backward(); lb.backward()

    def cosine(pa_, pb_):
        dots = np.zeros(3)
        for (k, pa), (_, pb) in zip(pa_.named_parameters(), pb_.named_parameters()):
            assert torch.isfinite(pa.grad).all(), k
            a, b = pa.grad.double().flatten(), pb.grad.double().flatten()
            dots += np.array([float(a @ b), float(a @ a), float(b @ b)])
        return dots[0] / np.sqrt(dots[1] * dots[2])
    # bf16 storage of every activation through ~50 train-mode BatchNorms at batch 8: two equivalent implementations
    # differ in the last bit of many activations, and the gradient direction only agrees to ~0.9; the fp32 tests pin
    # the arithmetic
    c_eager = cosine(net, ref)
    # the same product network with the stock convolution in place of the GEMM: only the summation order inside the
    # 1x1 convolutions differs
    net2 = getattr(models, arch)().cuda()
    load_det(net2)
    net2.train()
    applies = Fm.conv1x1_applies
    try:
        Fm.conv1x1_applies = lambda conv, x_, strided=False: False
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y2 = net2(x)
    finally:
        Fm.conv1x1_applies = applies
    torch.nn.functional.cross_entropy(y2.float(), tgt).backward()
    c_stock = cosine(net, net2)
    print(f"{arch}: gradient cosine vs eager {c_eager:.4f}, GEMM vs stock 1x1 convolutions {c_stock:.4f}; "
          f"logits GEMM vs stock {rel(y.detach().float().cpu().numpy(), y2.detach().float().cpu().numpy()):.3e}")
    assert c_eager > 0.8 and c_stock > 0.8


def test_per_forward_bookkeeping_is_batched_but_equivalent():
    """One training forward bumps every BatchNorm's num_batches_tracked exactly once (collected and applied by one
    _foreach_add_ at the end of forward_features) and serves every block its own stochastic-depth row from one table."""
    from mrla_amd import functional as Fm, models
    net = models.resnet50_mrlal(drop_path=0.5).cuda().train()
    x = torch.from_numpy(cases.image_batch(8, "img-train")).cuda()
    seen = []
    orig = Fm._Bookkeeping.drop_path_row

    def spy(self, batch, p, device):
        row = orig(self, batch, p, device)
        seen.append(row)
        return row
    try:
        Fm._Bookkeeping.drop_path_row = spy
        with torch.autocast("cuda", dtype=torch.bfloat16):
            net(x).float().sum().backward()
    finally:
        Fm._Bookkeeping.drop_path_row = orig
    counters = [m.num_batches_tracked for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert len(counters) == 53 + 16 and all(int(c) == 1 for c in counters)
    assert len(seen) == 16 and all(r.shape == (8,) for r in seen)
    rows = torch.stack(seen)
    assert set(rows.unique().tolist()) <= {0.0, 2.0} and rows.float().mean().item() not in (0.0, 2.0)
    assert len({tuple(r.tolist()) for r in rows}) > 8            # independent rows, not one mask repeated
    assert Fm.current_bookkeeping() is None
    # outside a model forward nothing is deferred
    bn = torch.nn.BatchNorm2d(64).cuda()
    Fm.bn_act(torch.randn(2, 64, 8, 8, device="cuda"), bn, relu=True)
    assert int(bn.num_batches_tracked) == 1


def test_resnet18_mrlal_extension_matches_its_eager_restatement():
    """BASELINE.json's config 1 by its literal name.  `resnet18_mrlal` is a BUILD-SIDE EXTENSION (the reference defines no
    BasicBlock network: SURVEY.md section 8(a)-note) -- torchvision's BasicBlock + the reference's light tail -- so there is
    no reference output to compare with: the product is compared with the eager restatement of the same definition
    (oracle/eager_models.py::EagerLightBasicBlock) on the same deterministic weights -- eval logits at 8 x 3 x 224 x 224
    (config 1's input), a train step with every gradient, NCHW against channels_last, bf16 autocast -- and the MRLA operators
    at its channel counts (64 / 128 / 256 / 512) are pinned against the oracle and the reference at module level
    (tests/test_light_gpu.py)."""
    from mrla_amd import models
    torch.backends.cudnn.allow_tf32 = False
    net, ref = models.resnet18_mrlal().cuda(), em.eager_resnet18_mrlal().cuda()
    load_det(net)
    ref.load_state_dict(net.state_dict())
    net.eval(); ref.eval()
    x = torch.from_numpy(cases.image_batch(8)).cuda()
    with torch.no_grad():
        y, yr = net(x), ref(x)
    assert rel(y.cpu().numpy(), yr.cpu().numpy()) < 5e-6
    net.train(); ref.train()
    xt = torch.from_numpy(cases.image_batch(4, "img-train")).cuda()
    tgt = (torch.arange(4) * 37 % 1000).cuda()
    y, yr = net(xt), ref(xt)
    assert rel(y.detach().cpu().numpy(), yr.detach().cpu().numpy()) < 2e-5
    torch.nn.functional.cross_entropy(y, tgt).backward()
    torch.nn.functional.cross_entropy(yr, tgt).backward()
    gp, gr = dict(net.named_parameters()), dict(ref.named_parameters())
    tight, dots = (0.0, ""), np.zeros(3)
    for k in gp:
        a, b = gp[k].grad.cpu().numpy().ravel().astype(np.float64), gr[k].grad.cpu().numpy().ravel().astype(np.float64)
        if np.abs(b).sum() < 1e-4:
            continue
        if k.endswith(("mrla.mrla.Wv.weight", "mrla.lambda_t", "bn_mrla.weight", "bn_mrla.bias", "bn2.weight", "bn2.bias")):
            tight = max(tight, (np.abs(a - b).sum() / np.abs(b).sum(), k))
        dots += np.array([a @ b, a @ a, b @ b])
    assert tight[0] < 2e-2, tight                          # where the deferred bn2 affine and the fused producer are wired
    assert dots[0] / np.sqrt(dots[1] * dots[2]) > 0.9999
    for k in ("layer1.0.bn_mrla.running_mean", "layer4.1.bn_mrla.running_var", "layer2.0.bn2.running_var"):
        assert rel(net.state_dict()[k].float().cpu().numpy(), ref.state_dict()[k].float().cpu().numpy()) < 1e-5, k
    # the NCHW kernels (the reference's layout contract) against the channels_last ones, and bf16 autocast
    b = models.resnet18_mrlal().cuda()
    b.load_state_dict(net.state_dict())
    b.channels_last = False
    b.to(memory_format=torch.contiguous_format)
    b.train()
    net.zero_grad(set_to_none=True)
    ya, yb = net(xt), b(xt)
    assert rel(ya.detach().cpu().numpy(), yb.detach().cpu().numpy()) < 2e-5
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y16 = net(xt)
    torch.nn.functional.cross_entropy(y16.float(), tgt).backward()
    assert torch.isfinite(y16).all() and all(torch.isfinite(p.grad).all() for p in net.parameters())
    assert rel(y16.detach().float().cpu().numpy(), ya.detach().cpu().numpy()) < 0.1
