"""CPU: the arithmetic of mrla_amd.graphs.replay_matches_eager (the check every bench line and every graphed_step carries) on
a toy model, with plain callables standing in for the eager step and the graph replay: an identical "replay" measures exactly
zero, a replay that skips the optimizer / scales the gradients / corrupts one parameter / forgets a BatchNorm counter is
reported with the right measure, and the training state afterwards is where the eager steps left it."""
import pytest
import torch


@pytest.fixture
def toy(monkeypatch):
    # (the checker snapshots / restores the CUDA generator: on this CPU-only box the CPU generator stands in)
    monkeypatch.setattr(torch.cuda, "get_rng_state", torch.get_rng_state)
    monkeypatch.setattr(torch.cuda, "set_rng_state", torch.set_rng_state)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda: None)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.BatchNorm1d(16), torch.nn.ReLU(), torch.nn.Dropout(0.2),
                              torch.nn.Linear(16, 4))
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    x, y = torch.randn(32, 8), torch.randint(0, 4, (32,))

    def step(scale=1.0, skip_opt=False):
        loss = torch.nn.functional.cross_entropy(net(x), y) * scale
        opt.zero_grad()
        loss.backward()
        if not skip_opt:
            opt.step()
        return loss.detach() / scale
    step()                                             # (momentum buffers exist)
    return net, opt, step


def test_an_identical_replay_measures_zero_and_training_continues(toy):
    from mrla_amd import graphs
    net, opt, step = toy
    before = [p.detach().clone() for p in net.parameters()]
    rep = graphs.replay_matches_eager(step, step, net, opt, steps=3)
    assert rep["ok"] and all(rep[k] == 0.0 for k in graphs._MEASURES) and all(rep["noise_" + k] == 0.0 for k in graphs._MEASURES)
    assert rep["loss_eager"] == rep["loss_replay"] and len(rep["loss_eager"]) == 3        # same dropout masks on every leg
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))  # three real steps were taken
    assert int(net[1].num_batches_tracked) == 1 + 3                                       # ... exactly three


def test_wrong_replays_are_named(toy):
    from mrla_amd import graphs
    net, opt, step = toy
    rep = graphs.replay_matches_eager(step, lambda: None, net, opt, steps=2)              # nothing happens on replay
    assert not rep["ok"] and rep["update_rel_l2"] == pytest.approx(1.0) and rep.get("counter_mismatch")
    rep = graphs.replay_matches_eager(step, lambda: step(scale=1.5), net, opt, steps=2)   # gradients 1.5 x
    assert not rep["ok"] and 0.05 < rep["update_rel_l2"] < 0.6 and rep["weights_rel_l2"] < rep["update_rel_l2"]   # (momentum dilutes the 0.5 g)
    rep = graphs.replay_matches_eager(step, lambda: step(skip_opt=True), net, opt, steps=2)
    assert not rep["ok"] and rep["buffers_rel_l2"] == 0.0 and "counter_mismatch" not in rep   # forward ran, weights stood still

    def poisoned():
        step()
        with torch.no_grad():
            net[4].bias[1] = float("nan")
    rep = graphs.replay_matches_eager(step, poisoned, net, opt, steps=1)
    assert not rep["ok"] and rep["nonfinite"] == "param:4.bias"

    def one_parameter_off():
        step()
        with torch.no_grad():
            net[0].bias.add_(1e-2)
    rep = graphs.replay_matches_eager(step, one_parameter_off, net, opt, steps=1)
    assert not rep["ok"] and rep["worst_parameter"] == "0.bias" and rep["update_rel_l2_worst_parameter"] > 1.0
    assert all(torch.isfinite(p).all() for p in net.parameters())                         # the state continues from the EAGER leg


def test_integer_state_is_compared_exactly_and_named(toy):
    """TrainingState.now() widens everything to float64; which entries were integer is recorded BEFORE that, so a step
    counter (or any integer buffer) that differs is a counter mismatch, not a rounding-level float difference."""
    from mrla_amd import graphs
    net, opt, step = toy
    net.register_buffer("seen", torch.zeros((), dtype=torch.int64))
    st = graphs.TrainingState(net, opt)
    assert "buffer:seen" in st.exact_keys and "buffer:1.num_batches_tracked" in st.exact_keys
    assert "buffer:1.running_mean" not in st.exact_keys and not any(k.startswith("param:") for k in st.exact_keys)
    a, start = st.now(), st.initial()
    net.seen += 1
    b = st.now()
    assert graphs.compare_states(b, a, start, st.exact_keys)["counter_mismatch"] == "buffer:seen"
    assert "counter_mismatch" not in graphs.compare_states(a, a, start, st.exact_keys)


def test_restore_puts_back_every_tensor_in_place_and_zeroes_state_created_since(toy, monkeypatch):
    """What graphed_step does on exit: weights, BatchNorm buffers and counters, momentum buffers and the generator are as on
    entry (same tensors, same addresses); optimizer state that did not exist on entry is zeroed (= a first step)."""
    from mrla_amd import graphs
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)           # no state yet
    x = torch.randn(8, 4)
    entry = graphs.TrainingState(net, opt)
    want = {k: v.clone() for k, v in net.state_dict().items()}
    ptrs = [p.data_ptr() for p in net.parameters()]
    rng = torch.get_rng_state()
    for _ in range(3):
        opt.zero_grad()
        net(x).square().mean().backward()
        opt.step()
        torch.rand(3)
    assert int(net[1].num_batches_tracked) == 3 and len(opt.state) == 4
    entry.restore(zero_new_state=True)
    assert all(torch.equal(v, net.state_dict()[k]) for k, v in want.items())
    assert [p.data_ptr() for p in net.parameters()] == ptrs
    assert all(float(st["momentum_buffer"].abs().max()) == 0.0 for st in opt.state.values())
    assert torch.equal(torch.get_rng_state(), rng)
    # a first step from zeroed momentum buffers == a first step without any (dampening 0)
    ref = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    ref.load_state_dict(want)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    for m, o in ((net, opt), (ref, ropt)):
        o.zero_grad()
        m(x).square().mean().backward()
        o.step()
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), ref.parameters()))


def _bare_step(net, opt, static):
    """A GraphedStep without its constructor (no GPU here): the launch logic of __call__ / eager only."""
    from mrla_amd import graphs
    st = object.__new__(graphs.GraphedStep)
    st.model, st.optimizer, st.loss_fn, st.exchange, st.scaler = net, opt, torch.nn.functional.cross_entropy, None, None
    st.autocast, st.static, st.graph, st.loss, st.output, st.last_launch = None, static, None, None, None, None
    return st


def test_a_batch_of_another_size_is_stepped_eagerly_on_the_tensors_given():
    """The tail batch of an epoch: `step(x, y)` with fewer images reaches the MODEL (not the static buffers, which still hold
    the previous batch), takes exactly one optimizer step and leaves the static buffers alone; a different image shape is
    refused."""
    from mrla_amd import _lib
    torch.manual_seed(0)
    seen = []
    net = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 3))
    net.register_forward_pre_hook(lambda m, inp: seen.append(tuple(inp[0].shape)))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    x, y = torch.randn(8, 3, 2, 2), torch.randint(0, 3, (8,))
    st = _bare_step(net, opt, [x.clone(), y.clone()])
    st(x, y)
    assert seen[-1] == (8, 3, 2, 2) and st.last_launch == "eager"
    w = net[1].weight.detach().clone()
    xs, ys = torch.randn(5, 3, 2, 2), torch.randint(0, 3, (5,))
    loss = st(xs, ys)
    assert seen[-1] == (5, 3, 2, 2) and st.last_launch == "eager (other shape)" and st.output.shape == (5, 3)
    assert torch.equal(st.static[0], x) and torch.equal(st.static[1], y)            # untouched
    want = torch.nn.functional.cross_entropy(torch.nn.functional.linear(xs.flatten(1), w, net[1].bias.detach()
                                                                        + 0.1 * net[1].bias.grad), ys)
    assert not torch.equal(net[1].weight.detach(), w) and loss == loss and want == want
    assert st.eager(xs, ys).shape == () and seen[-1] == (5, 3, 2, 2)                 # the documented explicit form
    with pytest.raises(_lib.MrlaHipError):
        st(torch.randn(5, 3, 4, 4), ys)
    with pytest.raises(_lib.MrlaHipError):
        st(xs)


def test_a_loss_scaler_needs_an_optimizer_that_takes_found_inf_on_the_device():
    from mrla_amd import _lib, graphs
    net = torch.nn.Linear(4, 2)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)                                   # foreach form: GradScaler would .item()

    class _Scaler:
        def is_enabled(self):
            return True
    with pytest.raises(_lib.MrlaHipError, match="fused=True"):
        graphs.GraphedStep(net, opt, torch.nn.functional.cross_entropy, (torch.zeros(1).to("meta"),), scaler=_Scaler())


def _verdict_worker(rank, world, port, out):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from mrla_amd import graphs
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = (graphs._all_ranks(True), graphs._all_ranks(rank == 0), graphs._all_ranks(False))
    dist.destroy_process_group()


def test_every_rank_reaches_the_same_verdict_gloo_world2():
    """graphed_step with a process group: the self-check's verdict is the AND over the ranks, so no rank raises / retries /
    falls back alone (which would hang the others at the next collective)."""
    import socket
    import torch.multiprocessing as mp
    from mrla_amd import graphs
    assert graphs._all_ranks(True) is True and graphs._all_ranks(False) is False      # no process group: its own verdict
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = mp.Manager().dict()
    mp.spawn(_verdict_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0] == out[1] == (True, False, False)


def test_restore_resets_a_loss_scaler_that_was_created_since_to_its_initial_scale(monkeypatch):
    """A GradScaler creates its scale / growth-tracker tensors at the first scale() call -- i.e. during graphed_step's warm-up:
    the restore puts them back to what a first call starts from (init_scale, 0), not to zero."""
    from mrla_amd import graphs
    monkeypatch.setattr(torch.cuda, "get_rng_state", torch.get_rng_state)
    monkeypatch.setattr(torch.cuda, "set_rng_state", torch.set_rng_state)

    class _Scaler:                        # the two attributes TrainingState looks at, created lazily like torch.amp.GradScaler's
        _init_scale = 1024.0
        _scale = _growth_tracker = None
    net, sc = torch.nn.Linear(2, 2), _Scaler()
    entry = graphs.TrainingState(net, None, sc)
    sc._scale, sc._growth_tracker = torch.full((), 256.0), torch.full((), 3, dtype=torch.int32)
    entry.restore(zero_new_state=True)
    assert float(sc._scale) == 1024.0 and int(sc._growth_tracker) == 0
    later = graphs.TrainingState(net, None, sc)          # once they exist they are part of the snapshot like everything else
    sc._scale.fill_(64.0)
    later.restore(zero_new_state=True)
    assert float(sc._scale) == 1024.0
