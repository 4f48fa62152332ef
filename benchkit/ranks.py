"""N > 1: starting the ranks, the votes that never touch a communicator in unknown state, capture helpers and the schedule
measurement whose finished records survive a later failure.  (resnet/train.py:127-136,153,174.)"""
import json
import os
import subprocess
import sys

import torch

from .common import BENCH, STATUS_ENV, free_port


def launch_ranks(args):
    """`python bench.py --gpus N` with no torch.distributed environment: what resnet/train.py:127-133 does with mp.spawn,
    here as ONE child `python -m torch.distributed.run` (one process per GPU below it).  Returns the exit code."""
    import tempfile
    argv = [a for a in sys.argv[1:]]
    status = os.path.join(tempfile.gettempdir(), f"mrla_bench_status_{os.getpid()}")
    attempts = [[]] if args.graph == 0 else [[], ["--graph", "0"]]
    rc = 1
    for extra in attempts:
        if os.path.exists(status):
            os.remove(status)
        env = dict(os.environ, **{STATUS_ENV: status})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), BENCH] + argv + extra
        print("bench.py: starting " + " ".join(cmd[1:8]) + " ...", file=sys.stderr, flush=True)
        rc = subprocess.call(cmd, env=env, cwd=os.getcwd())
        if os.path.exists(status):
            os.remove(status)
        if rc == 0 or extra:
            break
        print(f"bench.py: the ranks ended with code {rc}; starting ONE fresh set of ranks with --graph 0 (no stream capture)",
              file=sys.stderr, flush=True)
    return rc


def all_ranks_ok(ok, tag, rank, world):
    """Do all ranks agree that `tag` succeeded?  Voted through the process group's TCP store -- NOT through the collective
    library: after a failed capture with a collective in it the communicator must not be touched again."""
    import datetime
    import torch.distributed as dist
    if not dist.is_initialized():
        return ok
    store = dist.distributed_c10d._get_default_store()
    store.set(f"mrla_bench/{tag}/{rank}", "1" if ok else "0")
    keys = [f"mrla_bench/{tag}/{r}" for r in range(world)]
    store.wait(keys, datetime.timedelta(seconds=300))
    return all(store.get(k) == b"1" for k in keys)


def leave_without_the_communicator(code=0):
    """End this rank without running any destructor that would talk to a communicator in unknown state."""
    sys.stdout.flush()
    sys.stderr.flush()
    path = os.environ.get(STATUS_ENV)
    if path and code != 0:
        try:
            with open(path, "a") as fh:
                fh.write("capture_broken\n")
        except OSError:
            pass
    os._exit(code)



def ranks_seen(world, backend):
    """An all-reduce of ones over the default group: the number of ranks the collective library actually connected."""
    import torch.distributed as dist
    if not dist.is_initialized():
        return 1
    t = torch.ones(1, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    return int(round(float(t.item())))


def preflight_capture(world):
    """Capture + replay a 4-element all-reduce: does this RCCL / driver pair hold a collective inside a HIP graph?"""
    import torch.distributed as dist
    t = torch.ones(4, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(t)                                      # communicator + its streams exist before the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        dist.all_reduce(t)
    t.fill_(1.0)
    g.replay()
    torch.cuda.synchronize()
    got = float(t[0].item())
    if abs(got - world) > 1e-3:
        raise RuntimeError(f"replayed all-reduce of ones gave {got}, expected {world}")
    del g


def capture(step, dist_on, warm):
    """PyTorch's whole-network-capture recipe (mrla_amd.graphs.capture_step -- the one `mrla_amd.graphed_step` hands to a
    training loop): `warm` eager steps on a side stream, then one captured step."""
    from mrla_amd import graphs
    return graphs.capture_step(step, warmup=warm, distributed=dist_on)


REPLAY_CHECK_STEPS = 3
REPLAY_TOL = 1e-2


def check_replay(eager_step, replay, static_loss, net, opt, rank, world, tag):
    """mrla_amd.graphs.replay_matches_eager on the step that is about to be timed: REPLAY_CHECK_STEPS replays against as many
    eagerly launched steps from the same weights / momentum / BatchNorm buffers / generator state (so both legs drop the same
    images), the eager leg twice for the noise floor.  Returns (record for the line, ok on ALL ranks)."""
    from mrla_amd import graphs
    rep = graphs.replay_matches_eager(eager_step, replay, net, opt, steps=REPLAY_CHECK_STEPS, replay_loss=static_loss,
                                      tol=REPLAY_TOL)
    if not rep["ok"]:
        print(f"warning: rank {rank}: the replayed graph ({tag}) does not reproduce the eager step: "
              + json.dumps({k: v for k, v in rep.items() if k != "what"}), file=sys.stderr, flush=True)
    ok = all_ranks_ok(rep["ok"], "replay/" + tag, rank, world)
    rec = {k: (float(f"{v:.3e}") if isinstance(v, float) else v) for k, v in rep.items()}
    rec["what"] = (f"{REPLAY_CHECK_STEPS} consecutive replays of the captured step, each against an eagerly launched step from the SAME "
                   "weights, momentum, BatchNorm buffers, inputs and generator state (mrla_amd.graphs.replay_matches_eager); "
                   "weights_rel_l2 = |w_replay - w_eager| / |w_eager| and update_rel_l2 = the same difference relative to what the "
                   "step changed, all parameters as one vector, maxima over the steps; noise_* = eager vs eager from the same state "
                   "(MIOpen accumulates its weight gradients with atomics); ok = each measure <= max(tol, 4 x its noise floor)")
    return rec, ok


class CaptureBroken(RuntimeError):
    """A stream capture with a collective in it failed (on this rank or on another): the communicator must not be used again."""


def measure_exchange_schedules(names, prepare, capture_graph, time_region, steps, ab_steps, verify=None, after_region=None):
    """Time the gradient-exchange schedules `names` (first = ONE all-reduce after backward) as replayed HIP graphs, such that a
    failure of a later, optional schedule never costs the finished measurement of an earlier one:
      * schedule 0: prepare -> capture -> [verify] -> the FULL timed region of `steps` steps.  That record is complete before
        anything else is tried;
      * every further schedule: prepare -> capture -> `ab_steps` steps; only if that is faster than the best full record does
        it run [verify and] its own full region.
    prepare(name, first) -> handle; capture_graph(handle, name) -> replay callable (raises CaptureBroken when the capture
    failed on any rank); time_region(run, n) -> seconds (max over ranks); verify(handle, run, name) -> (record, ok);
    after_region(record) is called after every full region (the caller attaches what it wants kept with that measurement).
    Returns (records, chosen, failure): records[name] = dict(handle, run, dt [full region, seconds] or None, ab_ms, check);
    chosen = the name with the fastest FULL region (None if schedule 0's capture failed); failure = (name, exception) of
    the schedule whose capture broke, else None -- the caller then reports `chosen`'s record and leaves without touching the
    communicator."""
    records, chosen, failure = {}, None, None
    for i, name in enumerate(names):
        h = prepare(name, i == 0)
        try:
            run = capture_graph(h, name)
        except CaptureBroken as e:
            failure = (name, e)
            break
        rec = records[name] = dict(handle=h, run=run, dt=None, ab_ms=None, check=None, ok=True)
        if i > 0:
            t = time_region(run, ab_steps) / ab_steps
            rec["ab_ms"] = round(1e3 * t, 3)
            if chosen is not None and t >= records[chosen]["dt"] / steps:
                continue                       # not faster than the best finished region: no full region for it
        if verify is not None:
            rec["check"], rec["ok"] = verify(h, run, name)
            if not rec["ok"]:
                continue                       # a replay that does not reproduce the eager step is never timed as `value`
        rec["dt"] = time_region(run, steps)
        if after_region is not None:
            after_region(rec)
        if rec["ab_ms"] is None:
            rec["ab_ms"] = round(1e3 * rec["dt"] / steps, 3)
        if chosen is None or rec["dt"] < records[chosen]["dt"]:
            chosen = name
    return records, chosen, failure



def rank0_first(fn, rank, world, tag):
    """Run fn() on rank 0 while the other ranks wait (TCP store, not a collective), then on the others together; fn must
    not contain a collective.  Used for MIOpen's solver search (torch.backends.cudnn.benchmark, resnet/train.py:247): eight
    searches writing one user find-db at once can leave the ranks with different solvers for the same convolution, and the
    contract's time is the slowest rank's.  Rank 0 searches alone; the others find its records (and its compiled kernels) on
    disk."""
    import datetime
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return fn()
    store = dist.distributed_c10d._get_default_store()
    key = f"mrla_bench/first/{tag}"
    if rank == 0:
        try:
            return fn()
        finally:
            store.set(key, "1")
    store.wait([key], datetime.timedelta(seconds=3600))
    return fn()

