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
