"""N > 1 with mrla_amd.distributed.FlatGradientExchange (resnet/train.py:174's DistributedDataParallel, as ONE flat fp32 gradient
buffer whose all-reduce fits into the step's HIP graph): the tiers, in the order in which a failure of a later one can no
longer cost the number of an earlier one:
  0. plain DistributedDataParallel, launched kernel by kernel, for the full region -- ALWAYS first, its finished line kept: the
     reported line is never worse than what `train.py` unchanged would get (`config.ddp_eager_first`);
  1. the flat exchange launched eagerly for the full region, with the per-kernel events (roofline);
  2. schedule A (one all-reduce after backward) captured, checked against the eager step, timed for the full region;
  3. schedule B (buckets sent from backward's hooks) captured, probed for --ab-steps, timed fully only if faster.
Where the collective cannot be captured (gloo; a failed pre-flight) the step replays from two graphs around an eagerly launched
all-reduce."""
import sys

import torch

from . import common
from .common import sgd, timed
from .core import GRAPH_NOT_REPRODUCED, prefer_ddp_line, print_line_with
from .ranks import all_ranks_ok, capture, leave_without_the_communicator, measure_exchange_schedules
from .report import report


def ddp_eager_first(run):
    """Tier 0.  Wrap the model in torch's DistributedDataParallel, time `steps` eagerly launched steps as the contract says,
    build that region's finished line, then take the wrapper away again (its reducer's hooks go with it) -- the flat tiers
    continue from the weights these steps left, identical on every rank."""
    args, D, R = run.args, run.D, run.R
    ddp = D.wrap_data_parallel(run.net, device_ids=[run.local], force=True)
    opt = sgd(run.net.parameters())
    st = run.new_step(ddp, opt)
    run.warm_up(st, max(2, args.warmup))
    fb = run.measured_eagerly_first(st)          # the contract's region, then once more with the per-kernel events (roofline)
    dt, rms = fb["dt"], fb["rank_ms"]
    in_sync, finite = run.states_after()
    rec = {"ms_per_step": round(1e3 * dt / args.steps, 3), "images_per_sec": round(run.world * args.batch * args.steps / dt, 1),
           "steps": args.steps, "rank_ms_per_step": rms, "replicas_in_sync": in_sync,
           "what": "torch DistributedDataParallel (32 MB buckets, overlapped with backward), launched kernel by kernel: "
                   "resnet/train.py:174 unchanged; timed before any other tier"}
    line = None
    if run.rank == 0:
        line = report(dict(R, dt=dt, dt_eager=dt, dt_events=fb["dt_events"], timer=fb["timer"], use_graph=False, legs=False,
                           rank_ms=rms, dp="ddp", in_sync=in_sync, finite=finite, net=None, ddp_first=rec,
                           launch="kernel by kernel (PyTorch eager launches, DistributedDataParallel): the tier that ran first; "
                                  "no later tier finished faster"), emit=False)
    del st, opt, ddp, fb                 # (the wrapper's reducer takes its autograd hooks with it)
    import gc
    gc.collect()
    run.net.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    R["ddp_first"] = rec
    return dt, line


def flat_split(run):
    """The collective cannot be captured: two graphs around an eagerly launched all-reduce (after backward)."""
    args, D, R, rank, world = run.args, run.D, run.R, run.rank, run.world
    net = run.net = run.net.cuda().train()
    run.find_first(net)
    ddp = ddp_eager_first(run) if args.ddp_first else None
    opt = run.opt = sgd(net.parameters())
    ex = D.FlatGradientExchange(net.parameters(), overlap=False)
    run.eager_step = run.step = run.new_step(net, opt, ex)
    R.update(exchange=ex, schedule="after_backward")
    run.warm_up(run.step, args.warmup)
    x, y = run.x, run.y

    def part1():
        with common.autocast():
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        ex.gather()
        part1.loss = loss.detach()

    def part3():
        ex.adopt()
        opt.step()
    err = None
    try:                                             # (no collective inside either capture: a failure here is local)
        g1 = capture(part1, run.dist_on, 2)        # (thread-local capture mode: the watchdog may query eager works meanwhile)
        ex.allreduce_flat()
        g3 = capture(part3, run.dist_on, 0)
    except Exception as e:                           # noqa: BLE001
        err = e
    if all_ranks_ok(err is None, "capture/split", rank, world):
        def replay2():
            g1.replay()
            ex.allreduce_flat()
            g3.replay()
        R["replay"], ok = run.check(run.eager_step, replay2, part1.loss, net, "split")
        if ok:
            run.split_graphs, run.step = (g1, g3), replay2
            run.launch = ("two HIP graphs per step (fwd+loss+bwd+gradient gather | SGD) around an eagerly launched all-reduce ("
                          + run.split_why + ")")
        else:
            run.launch = GRAPH_NOT_REPRODUCED
    else:
        print(f"warning: HIP graph capture failed ({type(err).__name__ if err else 'on another rank'}: {err}); timing eager "
              "launches", file=sys.stderr)
        ex.adopt()
    return ddp


def flat_schedules(run):
    """The flat exchange; with --exchange ab both schedules are measured on this hardware."""
    args, D, R, rank = run.args, run.D, run.R, run.rank
    net = run.net = run.net.cuda().train()
    run.find_first(net)
    ddp = ddp_eager_first(run) if args.ddp_first else None
    opt = run.opt = sgd(net.parameters())
    names = {"after": ["after_backward"], "overlap": ["bucketed_overlap"], "ab": ["after_backward", "bucketed_overlap"]}[args.exchange]
    if not run.use_graph:
        cands = {}
        for i, name in enumerate(names):
            ex = D.FlatGradientExchange(net.parameters(), overlap=(name == "bucketed_overlap"), broadcast=(i == 0))
            st = run.new_step(net, opt, ex)
            run.warm_up(st, args.warmup if i == 0 else 2)
            t = timed(st, args.ab_steps, 1) / args.ab_steps if len(names) > 1 else None
            cands[name] = dict(exchange=ex, step=st, ms=None if t is None else round(1e3 * t, 3))
        schedule = min(cands, key=lambda k: cands[k]["ms"]) if len(names) > 1 else names[0]
        for k, v in cands.items():
            if k != schedule:
                v["exchange"].remove_hooks()
        R.update(exchange=cands[schedule]["exchange"], schedule=schedule,
                 ab_ms={k: v["ms"] for k, v in cands.items()} if len(names) > 1 else None)
        run.eager_step = run.step = cands[schedule]["step"]
        return ddp
    state = {}

    def prepare(name, first):        # after_backward first: it registers no hooks that the other would trigger
        ex = D.FlatGradientExchange(net.parameters(), overlap=(name == "bucketed_overlap"), broadcast=first)
        st = run.new_step(net, opt, ex)
        run.warm_up(st, args.warmup if first else 2)
        if first:
            R.update(exchange=ex, schedule=name)
            fbk = state["eager"] = run.measured_eagerly_first(st)
            if rank == 0:          # the eager region's finished line, should the very first capture break
                state["eager_line"] = report(dict(R, dt=fbk["dt"], dt_eager=fbk["dt"], dt_events=fbk["dt_events"],
                                                  timer=fbk["timer"], use_graph=False, legs=False, rank_ms=fbk["rank_ms"],
                                                  launch="kernel by kernel"), emit=False)
        return dict(exchange=ex, step=st, name=name)

    def capture_graph(h, name):
        h["graph"] = run.capture_voted(h["step"], name, 3)
        h["static_loss"] = h["step"].loss
        return h["graph"].replay

    def verify(h, replay, name):
        return run.check(h["step"], replay, h["static_loss"], net, name)

    def after_region(rec):           # kept with that region: per-rank times, replicas in sync, weights finite
        rec["rank_ms"] = dict(common.RANK_MS)
        rec["in_sync"], rec["finite"] = run.states_after()
        # ... and the finished LINE of that region, built NOW: if a later, optional capture breaks, rank 0 prints it and
        # every rank leaves at once -- RCCL's watchdog thread aborts the process within moments of a broken capture
        # that had collectives in it, so nothing may be left to compute then
        if rank == 0:
            fbk = state["eager"]
            rec["line"] = report(dict(R, dt=rec["dt"], dt_eager=fbk["dt"], dt_events=fbk["dt_events"], timer=fbk["timer"],
                                      use_graph=True, legs=False, exchange=rec["handle"]["exchange"],
                                      schedule=rec["handle"]["name"],
                                      ab_ms={rec["handle"]["name"]: round(1e3 * rec["dt"] / args.steps, 3)},
                                      replay=rec["check"], rank_ms=rec["rank_ms"], in_sync=rec["in_sync"],
                                      finite=rec["finite"], launch=run.graph_launch), emit=False)

    for attempt in (0, 1):
        recs, chosen, failure = measure_exchange_schedules(
            names, prepare, capture_graph, lambda replay, n: timed(replay, n, 1 if n != args.steps else 0), args.steps,
            args.ab_steps, verify, after_region)
        if chosen is not None or failure is not None or attempt == 1:
            break
        # no schedule's replay reproduced the eager step (the ranks agree: every check was voted): once more with
        # MIOpen's deterministic solvers, on fresh exchanges
        for v in recs.values():
            v["handle"]["exchange"].remove_hooks()
        if not run.deterministic_retry(recs[names[0]]["check"], recs[names[0]]["handle"]["step"], max(3, args.warmup)):
            break
    run.eager_record = state["eager"]
    ab_ms = {k: v["ab_ms"] for k, v in recs.items() if v["ab_ms"] is not None} if len(names) > 1 else None
    if failure is not None:
        fname, ferr = failure
        why = str(ferr)[:200]
        print(f"warning: HIP graph capture of the data-parallel step failed for schedule {fname} ({why})", file=sys.stderr, flush=True)
        if chosen is None:
            # the first capture broke: the eager region measured before it (or tier 0, if that was faster)
            print("reporting the eager steps measured before it", file=sys.stderr, flush=True)
            broke = ("kernel by kernel (PyTorch eager launches; the HIP graph capture of the step failed -- "
                     f"{why} -- so this is the eager region timed before the capture; the communicator was not used again)")
            if not prefer_ddp_line(run, state["eager"]["dt"], ddp, broke) and rank == 0:
                print_line_with(state["eager_line"], launch=broke)
            leave_without_the_communicator(0)
        # an optional later schedule broke: the finished graph-replayed region of the earlier one stands
        print(f"reporting the finished graph-replayed region of schedule {chosen}", file=sys.stderr, flush=True)
        suffix = (f" (schedule {chosen}; the capture of the optional schedule {fname} failed -- {why} -- after this region had "
                  "been timed; the communicator was not used again)")
        if not prefer_ddp_line(run, recs[chosen]["dt"], ddp, run.graph_launch + suffix) and rank == 0:
            print_line_with(recs[chosen]["line"], gradient_exchange_ab_ms=ab_ms, launch_suffix=suffix)
        leave_without_the_communicator(0)
    if chosen is None:                 # no replay reproduced the eager step: time the eager launches of schedule 0
        h = recs[names[0]]["handle"]
        for k, v in recs.items():
            if k != names[0]:
                v["handle"]["exchange"].remove_hooks()
        R.update(exchange=h["exchange"], schedule=names[0], ab_ms=ab_ms, replay=recs[names[0]]["check"])
        run.eager_step = run.step = h["step"]
        run.launch = GRAPH_NOT_REPRODUCED
    else:
        c = recs[chosen]
        for k, v in recs.items():
            if k != chosen:
                v["handle"]["exchange"].remove_hooks()      # the loser's hooks must not fire in the winner's eager steps
        R.update(exchange=c["handle"]["exchange"], schedule=chosen, ab_ms=ab_ms, replay=c["check"])
        run.eager_step, run.graph = c["handle"]["step"], c["handle"]["graph"]
        run.step, run.launch = run.graph.replay, run.graph_launch
        R["dt_done"] = c["dt"]         # the full region of the chosen schedule has been timed already
        R["rank_ms_done"] = c["rank_ms"]
    return ddp


__all__ = ["ddp_eager_first", "flat_schedules", "flat_split"]
