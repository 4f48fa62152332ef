"""One rank's measurement: build the model and the synthetic batch, warm up, capture, PROVE the replay, time the region, hand the
record to report().  The N > 1 flows (benchkit/dataparallel.py) drive the same object."""
import contextlib
import io
import json
import sys
import time

import torch

from . import common
from .common import make_step, sgd, timed
from .ranks import CaptureBroken, all_ranks_ok, capture, check_replay, leave_without_the_communicator, preflight_capture, rank0_first
from .report import report

GRAPH_NOT_REPRODUCED = ("kernel by kernel (PyTorch eager launches; the captured HIP graph did NOT reproduce the eagerly launched "
                        "step -- config.replay_check -- so the eager launches are what is timed)")


def build_model(args):
    """The timed network (random init under torch.manual_seed(0), as resnet/train.py:158 builds it) and what a step of it is."""
    torch.manual_seed(0)
    what = "fwd+bwd+SGD"
    with contextlib.redirect_stdout(io.StringIO()):
        if args.eager:
            from oracle import eager_models as em              # (--eager: the diagnostic that times the restatement instead)
            kw = {"drop_path_rate": args.drop_path} if args.arch.startswith("deit") else {"drop_path": args.drop_path}
            net = getattr(em, "eager_" + args.arch)(**kw)
        elif args.arch.startswith("det_"):
            # mmdetection/mmdet/models/backbones/resnet_mrlal.py:283-293 as configs/_base_/models/faster_rcnn_r50mrlal_fpn.py:4-14
            # builds it: norm_eval, frozen stem + stage 1, four output maps
            from mrla_amd import mmdet_backbone as mb
            net = mb.ResNet_mrlal(frozen_stages=1, norm_eval=True)
            what = "backbone fwd+bwd+SGD (norm_eval, frozen stem + stage 1; loss = sum of mean squares of the four maps)"
        else:
            from mrla_amd import models, vit
            if args.arch.startswith("deit"):
                net = getattr(vit, args.arch)(drop_path_rate=args.drop_path)
            else:
                net = getattr(models, args.arch)(drop_path=args.drop_path)
    if args.channels_last >= 0 and hasattr(net, "channels_last"):
        net.channels_last = bool(args.channels_last)
        net.to(memory_format=torch.channels_last if args.channels_last else torch.contiguous_format)
    return net, what


def synthetic_batch(args):
    gx = torch.Generator(device="cuda").manual_seed(0)
    gy = torch.Generator(device="cuda").manual_seed(1)
    if args.arch.startswith("det_"):
        b, c, h, w = (int(v) for v in (args.shape or "2x3x800x1344").split("x"))
        args.batch = b
        return torch.randn(b, c, h, w, device="cuda", generator=gx), None
    x = torch.randn(args.batch, 3, 224, 224, device="cuda", generator=gx)
    return x, torch.randint(0, 1000, (args.batch,), device="cuda", generator=gy)


def make_det_step(net, opt, x):
    """A detection backbone has no head here: forward -> (C2 .. C5), a loss that reaches every map, backward, SGD."""
    def step():
        with common.autocast():
            loss = sum(m.float().square().mean() for m in net(x))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        step.loss = loss.detach()
        return step.loss
    step.loss = None
    return step


class Run:
    """The state one rank's measurement shares between its phases."""

    def __init__(self, args, rank, local, world, dist_on, seen):
        from mrla_amd import distributed as D, functional as Fm
        self.D, self.Fm = D, Fm
        self.args, self.rank, self.local, self.world, self.dist_on, self.seen = args, rank, local, world, dist_on, seen
        self.net, what = build_model(args)
        self.layout = "channels_last" if getattr(self.net, "channels_last", False) else "NCHW"
        # N > 1: the exchange that fits into the graph (flat) unless the caller asks for eager launches or for DDP
        self.dp = "none" if not dist_on else (args.dp if args.dp != "auto" else ("ddp" if args.graph == 0 else "flat"))
        # (a gloo exchange stages through the host: not capturable)
        self.use_graph = args.graph == 1 or (args.graph < 0 and self.dp != "ddp" and (not dist_on or args.backend == "nccl"))
        self.launch_note, self.split_why = "", None
        self._preflight()
        self.x, self.y = synthetic_batch(args)
        self.R = dict(args=args, rank=rank, world=world, seen=seen, dist_on=dist_on, dp=self.dp, layout=self.layout, x=self.x,
                      net=None, exchange=None, schedule=None, ab_ms=None, legs=True, replay=None, rank_ms=None, what=what,
                      ddp_first=None)
        self.opt = self.step = self.eager_step = self.graph = self.split_graphs = None
        self.eager_record = None   # a finished eager region with the per-kernel events (N > 1 flat: taken before the captures)
        self.launch = "kernel by kernel (PyTorch eager launches)" + self.launch_note
        self.graph_launch = ("one HIP graph per step (captured fwd+loss+bwd" + ("+gradient all-reduce" if dist_on else "")
                             + "+SGD), replayed")

    # ---- can the collective itself go into a HIP graph?  (a gloo exchange stages through the host; RCCL: ask the pre-flight)
    def _preflight(self):
        a = self.args
        if self.dist_on and self.dp == "flat" and a.graph != 0:
            if a.backend != "nccl":
                self.split_why = f"a {a.backend} all-reduce stages through the host and cannot be captured"
            elif a.split_graph:
                self.split_why = "--split-graph"
            elif self.use_graph:
                try:
                    preflight_capture(self.seen)
                except Exception as e:               # noqa: BLE001
                    print(f"warning: pre-flight capture of a 4-element all-reduce failed ({type(e).__name__}: {e}); the step is "
                          "replayed from two graphs around an eagerly launched all-reduce", file=sys.stderr)
                    self.split_why = f"the pre-flight capture of a small all-reduce failed ({type(e).__name__})"
            if self.split_why:
                self.use_graph = False
        elif self.use_graph and self.dist_on and a.backend == "nccl":
            try:
                preflight_capture(self.seen)
            except Exception as e:                   # noqa: BLE001
                print(f"warning: pre-flight capture of a 4-element all-reduce failed ({type(e).__name__}: {e}); launching the "
                      "step eagerly", file=sys.stderr)
                self.use_graph, self.launch_note = False, " [pre-flight capture of a small all-reduce failed: no graph]"

    # ---- small pieces every flow uses -------------------------------------------------------------------------------------
    def new_step(self, net, opt, exchange=None):
        if self.y is None:
            return make_det_step(net, opt, self.x)
        return make_step(net, opt, self.x, self.y, exchange)

    @staticmethod
    def warm_up(st, n):
        for _ in range(n):
            st()

    def check(self, eager_step, replay, static_loss, net, tag):
        return check_replay(eager_step, replay, static_loss, net, self.opt, self.rank, self.world, tag)

    def deterministic_retry(self, first_check, st, n_warm):
        """The replay did not reproduce the eager step.  With --deterministic -1: switch torch.backends.cudnn.deterministic on --
        MIOpen then leaves out its atomically accumulating (split-K) solvers, which are right when launched eagerly and garbage
        from the second replay of a graph on -- re-run the warm-up (MIOpen searches again among the others) and tell the caller
        to capture and check once more.  Returns True when it did."""
        args, R = self.args, self.R
        if args.deterministic != -1 or torch.backends.cudnn.deterministic:
            return False

        def ms_per_step(n=2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                st()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n
        before = ms_per_step()
        torch.backends.cudnn.deterministic = True
        if self.rank == 0:
            print("note: torch.backends.cudnn.deterministic = True from here on (MIOpen's atomically accumulating solvers do not "
                  "survive graph replay); warming up, capturing and checking once more", file=sys.stderr, flush=True)
        self.warm_up(st, n_warm)
        after = ms_per_step()
        # MIOpen's deterministic solver list can be catastrophically slow (resnet50_mrlal b = 256 on MI355X: 7.3 s per step
        # against 30 ms -- bit-reproducible, and useless): a retry that costs more than it can win is abandoned, on every rank
        worth_it = all_ranks_ok(after <= 1.5 * before, "deterministic-worth-it", self.rank, self.world)
        R["replay_first_attempt"] = first_check
        if not worth_it:
            torch.backends.cudnn.deterministic = False
            R["miopen_deterministic_why"] = ("tried after the first captured graph did not reproduce the eager step, and switched "
                                             f"off again: the eager step took {after:.0f} ms with MIOpen's deterministic solvers "
                                             f"against {before:.0f} ms without")
            if self.rank == 0:
                print(f"note: deterministic solvers run the step in {after:.0f} ms against {before:.0f} ms: switched off again",
                      file=sys.stderr, flush=True)
            self.warm_up(st, 1)
            torch.cuda.synchronize()
            return False
        R["miopen_deterministic_why"] = ("switched on after the first captured graph did not reproduce the eager step "
                                         f"(update_rel_l2 {(first_check or {}).get('update_rel_l2')}, noise "
                                         f"{(first_check or {}).get('noise_update_rel_l2')}, worst parameter "
                                         f"{(first_check or {}).get('worst_parameter')})")
        torch.cuda.synchronize()
        return True

    def find_first(self, module):
        """MIOpen's solver search (torch.backends.cudnn.benchmark), rank 0 alone first: forward + loss + backward of the bare
        module, twice -- every convolution's forward, input-gradient and weight-gradient problem of the step -- with NO
        optimizer step and NO collective (the other ranks are waiting, and the replicas must stay identical)."""
        if self.world == 1 or not self.args.benchmark:
            return
        x, y = self.x, self.y

        def go():
            for _ in range(2):
                with common.autocast():
                    loss = torch.nn.functional.cross_entropy(module(x).float(), y)
                loss.backward()
            module.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        rank0_first(go, self.rank, self.world, "find")
        self.R["find_s"] = round(time.perf_counter() - t0, 1)

    def measured_eagerly_first(self, st):
        """A complete measurement -- `steps` steps, bracketed as the contract says, per-kernel events on -- taken BEFORE any
        collective goes into a stream capture.  If the FIRST capture then fails on any rank, THIS is what the line reports (the
        ranks agree on that through the TCP store and leave without touching the communicator): an N > 1 run never ends
        without its number.  (A later, optional capture that fails costs nothing: measure_exchange_schedules.)"""
        d = timed(st, self.args.steps, 0)                # as a training loop launches it: one C call per tail, no events
        rms = dict(common.RANK_MS)
        t = self.Fm.KernelTimer()                        # ... and once more with an event pair around every kernel (roofline)
        self.Fm.TIMER = t
        de = timed(st, self.args.steps, 0)
        self.Fm.TIMER = None
        return dict(dt=d, dt_events=de, timer=t, rank_ms=rms)

    def capture_voted(self, st, tag, warm):
        """capture(st) with the ranks' vote; raises CaptureBroken when it failed on any rank."""
        args, err = self.args, None

        def poisoned():
            st()
            raise RuntimeError(f"injected failure inside the capture (--inject-capture-failure {args.inject_capture_failure})")
        inject = args.inject_capture_failure in ("all", tag) or (args.inject_capture_failure == "first" and tag in
                                                                 ("after_backward", "ddp"))
        try:
            g = capture(poisoned if inject else st, True, 0 if inject else warm)
        except Exception as e:                 # noqa: BLE001 -- whatever the runtime throws out of a broken capture
            g, err = None, e
        if all_ranks_ok(err is None, "capture/" + tag, self.rank, self.world):
            return g
        raise CaptureBroken(f"{type(err).__name__}: {err}" if err is not None else "failed on another rank")

    def report_and_leave(self, rec):
        if self.rank == 0:
            report(dict(self.R, **rec))
        leave_without_the_communicator(0)

    def states_after(self):
        # after warm-up, A/B, timed and event-timed steps: do all ranks still hold the same weights?  (they do if and only if
        # every step's exchange -- captured or not -- handed every rank the same averaged gradients) ... and are they numbers?
        in_sync = self.D.replicas_in_sync(list(self.net.parameters())) if self.dist_on else None
        finite = bool(torch.isfinite(torch.stack([p.detach().float().abs().max() for p in self.net.parameters()])).all())
        return in_sync, finite

    # ---- N = 1 (and --dp ddp): one process group-less model, or torch's DistributedDataParallel ------------------------------
    def single_or_ddp(self):
        args, D = self.args, self.D
        self.net = self.net.cuda().train()
        self.find_first(self.net)
        if self.dist_on and self.use_graph:
            # capturing a DDP step (PyTorch's whole-network-capture recipe): the wrapper is built in a side-stream context and
            # at least 11 DDP iterations run eagerly on a side stream before the capture
            side0 = torch.cuda.Stream()
            side0.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side0):
                self.net = D.wrap_data_parallel(self.net, device_ids=[self.local], force=args.ddp_probe)
            torch.cuda.current_stream().wait_stream(side0)
        else:
            self.net = D.wrap_data_parallel(self.net, device_ids=[self.local], force=args.ddp_probe)
        self.opt = sgd((p for p in self.net.parameters() if p.requires_grad), lr=common.DET_LR if self.y is None else 0.1)
        self.eager_step = self.step = self.new_step(self.net, self.opt)
        self.warm_up(self.step, args.warmup)                      # warm-up without the timer
        if self.use_graph and self.dist_on:
            fb = self.measured_eagerly_first(self.eager_step)
            try:
                self.graph = self.capture_voted(self.eager_step, "ddp", 11)
            except CaptureBroken as e:
                print(f"warning: HIP graph capture of the data-parallel step failed ({e}); reporting the eager steps measured "
                      "before it", file=sys.stderr, flush=True)
                self.report_and_leave(dict(dt=fb["dt"], dt_eager=fb["dt"], dt_events=fb["dt_events"], timer=fb["timer"],
                                           use_graph=False, legs=False, rank_ms=fb["rank_ms"], net=None, in_sync=None, finite=None,
                                           launch="kernel by kernel (PyTorch eager launches; the HIP graph capture of the step "
                                                  f"failed -- {str(e)[:200]} -- so this is the eager region timed before the "
                                                  "capture; the communicator was not used again)"))
            self.R["replay"], ok = self.check(self.eager_step, self.graph.replay, self.eager_step.loss, self.net, "ddp")
            if ok:
                self.step, self.launch = self.graph.replay, self.graph_launch
            else:
                self.graph, self.launch = None, GRAPH_NOT_REPRODUCED
        elif self.use_graph:
            # the whole training step is launch-order static (no host sync inside): capture it once into a HIP graph and
            # replay it -- the same kernels on the same buffers, minus the launch gaps -- after proving that the replay
            # computes what the eager launches compute (config.replay_matches_eager)
            try:
                for attempt in (0, 1):
                    self.graph = capture(self.eager_step, self.dist_on, 2)
                    self.R["replay"], ok = self.check(self.eager_step, self.graph.replay, self.eager_step.loss, self.net,
                                                      f"n1/{attempt}")
                    if ok:
                        self.step, self.launch = self.graph.replay, self.graph_launch
                        break
                    self.graph, self.launch = None, GRAPH_NOT_REPRODUCED
                    if attempt == 1 or not self.deterministic_retry(self.R["replay"], self.eager_step, max(3, args.warmup)):
                        break
            except Exception as e:                   # noqa: BLE001
                print(f"warning: HIP graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
                self.graph, self.step = None, self.eager_step

    # ---- the timed region: exactly `steps` steps between barrier + synchronize --------------------------------------------
    def timed_region_and_report(self, ddp=None):
        """ddp: tier 0's (seconds of its region, its finished line) when plain DistributedDataParallel was timed first."""
        args, R, Fm = self.args, self.R, self.Fm
        use_graph = self.graph is not None or self.split_graphs is not None
        timer = Fm.KernelTimer()                             # every C-ABI launch
        if use_graph:
            if R.get("dt_done") is not None:                 # (N > 1 flat: measure_exchange_schedules timed the chosen schedule's region)
                dt = R.pop("dt_done")
                R["rank_ms"] = R.pop("rank_ms_done")
            else:
                dt = timed(self.step, args.steps, 0)
                R["rank_ms"] = dict(common.RANK_MS) if self.dist_on else None
        else:
            dt = timed(self.step, args.steps, 0)             # eager launches as a training loop issues them (no events)
            R["rank_ms"] = dict(common.RANK_MS) if self.dist_on else None
        if self.eager_record is not None:                    # the eager regions were taken before the captures
            timer, dt_eager, dt_events = self.eager_record["timer"], self.eager_record["dt"], self.eager_record["dt_events"]
        else:
            # the same `steps` steps launched kernel by kernel: what resnet/train.py's loop gets unchanged ...
            dt_eager = timed(self.eager_step, args.steps, 0) if use_graph else dt
            # ... and once more with a HIP-event pair on the launch stream around EVERY kernel (events cannot be read out of a
            # replayed graph; this region feeds `roofline` / `mrla_kernels` only; the C ABI is then called pass by pass)
            Fm.TIMER = timer
            dt_events = timed(self.eager_step, args.steps, 0)
            Fm.TIMER = None
        R["dt_events"] = dt_events
        in_sync, finite = self.states_after()
        if prefer_ddp_line(self, dt, ddp, self.launch):
            pass                                             # (never worse than resnet/train.py:174 unchanged: tier 0's line stands)
        elif self.rank == 0:
            net = self.net
            self.graph = self.split_graphs = self.step = self.eager_step = None   # (report() may hand the GPU to child processes)
            self.net = None
            report(dict(R, dt=dt, dt_eager=dt_eager, timer=timer, use_graph=use_graph, launch=self.launch, net=net,
                        in_sync=in_sync, finite=finite))
        if self.dist_on:
            self.D.barrier()
            torch.distributed.destroy_process_group()


def prefer_ddp_line(run, dt, ddp, launch):
    """The tier that finished took `dt` for the region; tier 0 (plain DistributedDataParallel, eager: benchkit/dataparallel.py)
    took ddp[0] and left its finished line ddp[1].  If tier 0 was faster, rank 0 prints ITS line -- saying what it was
    preferred over -- and True comes back: the printed line is never slower than resnet/train.py:174 unchanged."""
    if ddp is None or dt <= ddp[0]:
        return False
    steps = run.args.steps
    if run.rank == 0:
        print(f"note: the tier that finished ({1e3 * dt / steps:.2f} ms per step) is slower than plain DistributedDataParallel "
              f"launched eagerly ({1e3 * ddp[0] / steps:.2f} ms): reporting the latter", file=sys.stderr, flush=True)
        print_line_with(ddp[1], slower_tier={"launch": launch, "ms_per_step": round(1e3 * dt / steps, 3)},
                        launch_suffix=f" [preferred over a slower later tier: {launch[:120]}]")
    return True


def print_line_with(rec0, **config_updates):
    """Rank 0: print a finished line (a JSON string) after editing its `config`."""
    rec = json.loads(rec0)
    for k, v in config_updates.items():
        if k == "launch_suffix":
            rec["config"]["launch"] += v
        else:
            rec["config"][k] = v
    print(json.dumps(rec), flush=True)
