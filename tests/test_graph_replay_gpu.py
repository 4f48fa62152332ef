"""GPU: the HIP-graph-replayed training step of the PRODUCT models computes what the eagerly launched step computes.

bench.py's `value` is measured on a replayed graph, the reference launches its step eagerly (resnet/train.py:387-409), and a
library kernel was once found to misbehave under replay (MIOpen's split-K 3x3 weight gradient at small batches, finite but
wrong: profiles/r04_notes.md section 10).  So: same weights, momentum buffers, BatchNorm statistics, inputs and generator state
(both legs drop the same images), K steps launched eagerly vs K replays -- through `mrla_amd.graphed_step`, the recipe a
training loop gets (its WeightBank refresh under capture, the deferred-bn3 hand-over, the stochastic-depth table, the K/V rings
of MRLA-base all take part)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(arch, drop_path):
    import contextlib
    import io
    from mrla_amd import models, vit
    torch.manual_seed(0)
    with contextlib.redirect_stdout(io.StringIO()):
        if arch.startswith("deit"):
            return getattr(vit, arch)(drop_path_rate=drop_path).cuda().train()
        return getattr(models, arch)(drop_path=drop_path).cuda().train()


def _data(batch):
    g = torch.Generator(device="cuda").manual_seed(5)
    return (torch.randn(batch, 3, 224, 224, device="cuda", generator=g),
            torch.randint(0, 1000, (batch,), device="cuda", generator=g))


@pytest.mark.parametrize("arch", ["resnet50_mrlal", "resnet101_mrlab", "deit_mrlal_tiny_patch16_224"])
def test_replayed_bf16_step_equals_eager_steps(arch):
    """bf16 autocast, batch 32, stochastic depth on, SGD with momentum and weight decay (resnet/train.py:199-201): four
    consecutive replays, each against an eager step from the same state.  MIOpen accumulates its weight gradients with
    atomics, so eager vs eager is not bit-equal either: that run-to-run difference is measured in the same call and the replay
    must stay within a small multiple of it (and the whole update within 2 % in absolute terms)."""
    import mrla_amd
    from mrla_amd import graphs
    was = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True            # MIOpen's find mode, as resnet/train.py:247 runs (see the module docstring)
    try:
        net = _build(arch, 0.1)
        opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        x, y = _data(32)
        step = mrla_amd.graphed_step(net, opt, torch.nn.functional.cross_entropy, (x, y), verify=0)
        assert step.graph is not None
        rep = graphs.replay_matches_eager(step.eager, step.graph.replay, net, opt, steps=4, replay_loss=step._static[0])
    finally:
        torch.backends.cudnn.benchmark = was
    print({k: v for k, v in rep.items() if k != "what"})
    assert "nonfinite" not in rep and "counter_mismatch" not in rep
    assert rep["ok"], rep
    assert rep["weights_rel_l2"] < 1e-3                                   # all parameters as one vector, relative to the weights
    assert rep["update_rel_l2"] < max(2e-2, 3 * rep["noise_update_rel_l2"])   # relative to what a step changed
    assert rep["buffers_rel_l2"] < max(1e-2, 3 * rep["noise_buffers_rel_l2"])   # BatchNorm running statistics
    assert rep["optim_rel_l2"] < max(2e-2, 3 * rep["noise_optim_rel_l2"])       # momentum buffers
    assert all(abs(a - b) < 1e-2 * abs(a) + 1e-3 for a, b in zip(rep["loss_eager"], rep["loss_replay"])), rep


def test_replayed_fp32_step_equals_eager_steps():
    """fp32 (resnet/train.py trains without AMP, :397-409) at batch 32 with MIOpen's find mode (:247): the recipe INTEGRATION.md
    2b recommends for train.py, `graphed_step(..., autocast=None)`.  The strided 1x1 downsample convolutions take the
    subsample + stride-1 route in every dtype (functional.conv_bn_act), so nothing of MIOpen's strided input gradient -- wrong
    from the second replay on -- is in the graph.  Where the eager step is bit-reproducible run to run, the replay must be
    bit-equal to it; otherwise within fp32 accumulation noise."""
    import mrla_amd
    from mrla_amd import graphs
    was = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        net = _build("resnet50_mrlal", 0.1)
        opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9, weight_decay=1e-4)
        x, y = _data(32)
        step = mrla_amd.graphed_step(net, opt, torch.nn.functional.cross_entropy, (x, y), autocast=None, verify=0)
        rep = graphs.replay_matches_eager(step.eager, step.graph.replay, net, opt, steps=4, replay_loss=step._static[0])
    finally:
        torch.backends.cudnn.benchmark = was
    print({k: v for k, v in rep.items() if k != "what"})
    assert rep["ok"], rep
    if rep["noise_weights_rel_l2"] == 0.0 and rep["noise_buffers_rel_l2"] == 0.0 and rep["noise_optim_rel_l2"] == 0.0:
        assert rep["weights_rel_l2"] == 0.0 and rep["buffers_rel_l2"] == 0.0 and rep["optim_rel_l2"] == 0.0
    else:
        # (MIOpen's atomically accumulated fp32 weight gradients: the summation order is not the same run to run, and not the
        # same between a replayed graph and eager launches either)
        assert rep["weights_rel_l2"] < 1e-5 and rep["update_rel_l2"] < max(5e-3, 10 * rep["noise_update_rel_l2"])


def test_graphed_step_is_a_drop_in_training_step():
    """`mrla_amd.graphed_step` as a loop uses it: built with its self-check on (verify=2), fed fresh batches, it keeps
    training -- the loss of a repeated batch goes down, the logits are exposed; building it leaves the training state as it was
    (resume-from-checkpoint runs start where resnet/train.py's would); a smaller last batch is stepped eagerly on its own
    tensors; another image size is refused."""
    import mrla_amd
    net = _build("resnet50_mrlal", 0.0)
    opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
    x, y = _data(32)               # (32: the smallest batch MIOpen's immediate-mode 3x3 weight gradient replays correctly at)
    entry = {k: v.detach().clone() for k, v in net.state_dict().items()}
    rng = torch.cuda.get_rng_state()
    step = mrla_amd.graphed_step(net, opt, torch.nn.functional.cross_entropy, (x, y))
    assert step.report is not None and step.report["ok"] and step.graph is not None
    # the warm-up, capture and self-check steps were real optimizer steps on the example batch -- and left no trace: weights,
    # BatchNorm statistics and counters, the generator are as on entry; the momentum buffers they created are zero (= none)
    now = net.state_dict()
    assert all(torch.equal(v, now[k]) for k, v in entry.items()), [k for k, v in entry.items() if not torch.equal(v, now[k])][:3]
    assert int(net.bn1.num_batches_tracked) == 0 and torch.equal(torch.cuda.get_rng_state(), rng)
    assert len(opt.state) > 0 and all(float(st["momentum_buffer"].abs().max()) == 0.0 for st in opt.state.values())
    losses = []
    for _ in range(6):
        losses.append(float(step(x, y)))
    assert step.last_launch == "graph" and int(net.bn1.num_batches_tracked) == 6
    assert step.output.shape == (32, 1000) and all(v == v for v in losses)
    assert losses[-1] < losses[0]
    x2, y2 = _data(32)
    step(x2 * 0.5, y2)                                   # another batch of the captured shape
    assert torch.isfinite(step.loss)
    # the tail batch of an epoch: stepped eagerly ON THE TENSORS GIVEN (the static buffers keep the previous batch)
    held = step.static[0].clone()
    w = net.fc.weight.detach().clone()
    loss8 = step(x[:8], y[:8])
    assert step.last_launch == "eager (other shape)" and step.output.shape == (8, 1000) and torch.isfinite(loss8)
    assert torch.equal(step.static[0], held) and not torch.equal(net.fc.weight.detach(), w)
    assert int(net.bn1.num_batches_tracked) == 8
    step(x, y)
    assert step.last_launch == "graph" and step.output.shape == (32, 1000)      # and the graph is still good afterwards
    with pytest.raises(mrla_amd._lib.MrlaHipError):
        step(x[:8, :, :128, :128], y[:8])
    # a mismatch is reported, not trained on: a "replay" that skips the optimizer is caught by the same check
    from mrla_amd import graphs
    rep = graphs.replay_matches_eager(step.eager, lambda: None, net, opt, steps=2)
    assert not rep["ok"] and rep["update_rel_l2"] > 0.5


def test_deit_fp16_autocast_with_loss_scaler_tracks_the_eager_restatement():
    """The recipe the reference trains DeiT with (deit/engine.py:37,51: `torch.cuda.amp.autocast()` = fp16, timm's
    NativeScaler = GradScaler): one scaled step of deit_mrlal_tiny through the product vs the eager restatement from the same
    weights -- same loss, the scaler does not skip the step (no inf / NaN in any unscaled gradient), and the gradients point
    the same way."""
    import numpy as np
    from mrla_amd import vit
    from oracle import eager_models as em
    torch.manual_seed(0)
    net = vit.deit_mrlal_tiny_patch16_224().cuda().train()
    ref = em.eager_deit_mrlal_tiny_patch16_224().cuda().train()
    ref.load_state_dict(net.state_dict())
    x, y = _data(16)
    out = {}
    for name, model in (("product", net), ("eager", ref)):
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        before = [p.detach().clone() for p in model.parameters()]
        with torch.autocast("cuda", dtype=torch.float16):
            logits = model(x)
            loss = torch.nn.functional.cross_entropy(logits.float(), y)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.unscale_(opt)
        grads = [p.grad.detach().double().flatten() for p in model.parameters()]
        assert all(torch.isfinite(g).all() for g in grads), name
        scaler.step(opt)
        scaler.update()
        moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(model.parameters(), before))
        out[name] = dict(loss=float(loss), grads=torch.cat(grads), scale=scaler.get_scale(), moved=moved,
                         logits=logits.detach().float())
    a, b = out["product"], out["eager"]
    assert a["scale"] == 1024.0 and b["scale"] == 1024.0          # update() halves the scale after a skipped step
    assert a["moved"] > 0 and b["moved"] > 0
    assert abs(a["loss"] - b["loss"]) < 2e-2 * abs(b["loss"])
    rel = float((a["logits"] - b["logits"]).abs().max() / b["logits"].abs().max())
    cos = float(a["grads"] @ b["grads"] / np.sqrt(float(a["grads"] @ a["grads"]) * float(b["grads"] @ b["grads"])))
    print(f"deit_mrlal_tiny fp16 autocast + GradScaler: loss {a['loss']:.4f} vs {b['loss']:.4f}, logits rel {rel:.3e}, gradient cosine {cos:.4f}")
    # (the bound of test_models_gpu.py's bf16 twin: 16-bit storage through 12 blocks at batch 16 -- two equivalent
    # implementations agree on the gradient's direction to ~0.9; the fp32 tests pin the arithmetic)
    assert rel < 5e-2 and cos > 0.8


def test_deit_fp16_loss_scaler_step_replays_from_one_graph():
    """deit/engine.py:37,51 inside the graph: fp16 autocast, GradScaler (timm's NativeScaler), fused AdamW (deit/main.py's
    default optimizer is adamw) -- scale, backward, unscale + inf check + update and the scale's own update are all device
    work, so `graphed_step(..., scaler=...)` captures them.  (a) the self-check passes (replays == eager steps from the same
    state, the scaler's tensors included); (b) replays track an eager loop of the same recipe; (c) an inf injected into the
    loss is handled as the eager scaler handles it: the step is skipped (weights stand still) and the scale halves."""
    import mrla_amd
    was = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = True
    try:
        x, y = _data(32)

        def build():
            net = _build("deit_mrlal_tiny_patch16_224", 0.0)
            # (fused: the step takes the scaler's grad_scale / found_inf on the device; capturable: Adam's step counter too)
            opt = torch.optim.AdamW(net.parameters(), lr=5e-4, weight_decay=0.05, fused=True, capturable=True)
            return net, opt, torch.amp.GradScaler("cuda", init_scale=4096.0, growth_interval=4)
        poison = torch.ones((), device="cuda")

        def loss_fn(logits, target, factor):
            return torch.nn.functional.cross_entropy(logits, target) * factor      # factor = 1, or inf: every gradient inf / NaN

        net, opt, scaler = build()
        step = mrla_amd.graphed_step(net, opt, loss_fn, (x, y, poison), autocast=torch.float16, scaler=scaler, verify=2)
        assert step.graph is not None and step.report["ok"], step.report
        assert scaler.get_scale() == 4096.0 and len(opt.state) > 0                 # the state restore covers the scaler too
        ref, ropt, rscaler = build()
        ref.load_state_dict(net.state_dict())

        def eager_ref(extra):
            with torch.autocast("cuda", dtype=torch.float16):
                loss = loss_fn(ref(x).float(), y, extra)
            ropt.zero_grad(set_to_none=True)
            rscaler.scale(loss).backward()
            rscaler.step(ropt)
            rscaler.update()
            return float(loss)
        got, want = [], []
        for k in range(6):
            extra = torch.full((), float("inf"), device="cuda") if k == 2 else poison
            before = net.head.weight.detach().clone()
            got.append(float(step(x, y, extra)))
            want.append(eager_ref(extra))
            assert step.last_launch == "graph"
            moved = not torch.equal(net.head.weight.detach(), before)
            assert moved == (k != 2), (k, moved)                                       # the poisoned step is skipped on the device
            assert scaler.get_scale() == rscaler.get_scale(), (k, scaler.get_scale(), rscaler.get_scale())
        assert scaler.get_scale() == 2048.0 or scaler.get_scale() == 4096.0           # halved at step 2 (grown back after 4 good ones)
        assert all(abs(a - b) <= 2e-2 * abs(b) + 1e-3 for a, b in zip(got, want) if b == b and abs(b) != float("inf")), (got, want)
        assert got[2] == float("inf") and want[2] == float("inf")
        assert got[-1] < got[0]
    finally:
        torch.backends.cudnn.benchmark = was


def test_a_library_kernel_that_misbehaves_under_replay_is_caught_before_training_on_it():
    """MIOpen's immediate-mode split-K 3x3 weight gradient at batch 16 (the 7x7 stage) is right eagerly and on the first
    replay, garbage from the second replay on (scripts/miopen_wrw_graph_probe.py; profiles/r05_notes.md).  graphed_step's
    self-check must refuse such a graph (raise) or -- on_mismatch="eager" -- hand out the eagerly launched step; if a future
    MIOpen is fixed the check simply passes."""
    import warnings
    import mrla_amd
    from mrla_amd import graphs
    was = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    try:
        net = _build("resnet50_mrlal", 0.0)
        opt = torch.optim.SGD(net.parameters(), lr=0.02, momentum=0.9)
        x, y = _data(16)
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            step = mrla_amd.graphed_step(net, opt, torch.nn.functional.cross_entropy, (x, y), verify=3, on_mismatch="eager")
        if step.report["ok"]:
            assert step.graph is not None
        else:
            assert step.graph is None and any("does not reproduce" in str(m.message) for m in w)
            print("caught:", {k: step.report[k] for k in ("weights_rel_l2", "update_rel_l2", "worst_parameter")})
            with pytest.raises(graphs.GraphReplayMismatch):
                net2 = _build("resnet50_mrlal", 0.0)
                opt2 = torch.optim.SGD(net2.parameters(), lr=0.02, momentum=0.9)
                mrla_amd.graphed_step(net2, opt2, torch.nn.functional.cross_entropy, (x, y), verify=3)
        l0 = float(step(x, y))
        for _ in range(4):
            l1 = float(step(x, y))
        assert l1 == l1 and l1 < l0                      # whichever way it launches, it trains
    finally:
        torch.backends.cudnn.benchmark = was
