#!/usr/bin/env python3
"""Headline benchmark: images/sec fwd+bwd of resnet50_mrlal, batch 256 per GPU, bf16 autocast, synthetic
ImageNet-shaped data (BASELINE.json metric / configs[1]; configs[2] for --gpus N > 1).

    python bench.py --gpus N --steps K --warmup W          # N > 1: starts its own N ranks (below)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # the driver's line: the ranks are already there

A "step" = forward + loss + backward + SGD update on one synthetic batch already resident in HBM.  On a single GPU the
timed steps replay the whole step from one HIP graph (`config.launch`; `--graph 0` times PyTorch's kernel-by-kernel
launches instead) -- AFTER the graph has proven itself: `config.replay_matches_eager` / `config.replay_check` = three
consecutive replays, each against an eagerly launched step from the same weights, momentum, BatchNorm buffers, inputs and
generator state (mrla_amd.graphs.replay_matches_eager; the eager step twice for the run-to-run noise floor).  A graph that
does not reproduce the eager step is never timed as `value`: the eager launches are timed instead and `config.launch` says
so (`--deterministic -1`: one more capture + check with torch.backends.cudnn.deterministic first; `config.miopen`).
`eager_launch_images_per_sec` / `eager_launch_ms_per_step`: the same steps launched eagerly, i.e. what resnet/train.py gets
UNCHANGED (:387-409); `eager_launch_with_kernel_events_ms_per_step`: once more with a HIP-event pair around every kernel (the
region `roofline` / `mrla_kernels` come from).  `config.weights_finite`: the weights are still numbers at the end.

N > 1 (resnet/train.py:127-133 spawns its own workers with mp.spawn; :153 init_process_group; :174 DDP).  Launched
plainly with `--gpus N`, this file starts `python -m torch.distributed.run ... bench.py <same arguments>` as a CHILD
process before anything has touched the GPU, lets rank 0's JSON line through and exits with the child's code.  Every
rank proves the process group (`config.ranks_seen` = an all-reduce of ones over `config.backend`) and, after the last step,
that the exchange did its job (`config.replicas_in_sync`: bit-identical weights on all ranks).  The step is captured
into one HIP graph with the gradient exchange inside it (mrla_amd/distributed.py: FlatGradientExchange) after a
pre-flight (capture + replay of a 4-element all-reduce).  If the collective cannot be captured -- the pre-flight fails, or
the backend is gloo -- the step still replays from graphs: forward + loss + backward + the gradient gather from one, the
optimizer step from a second, the all-reduce launched eagerly between them (`config.launch` says which tier ran).  Two exchange schedules are timed on the hardware -- ONE all-reduce after backward, and ~25 MB buckets sent from
backward's hooks while backward still runs (what DistributedDataParallel does) -- both reported
(`config.gradient_exchange_ab_ms`); the faster one runs the timed region (`--exchange` pins one).  An N > 1 run never
ends without its number: BEFORE any collective goes into a capture, `steps` eagerly launched steps are timed as the
contract says; if capturing the full step then fails on any rank (the ranks vote through the process group's TCP store,
not through the collective library), rank 0 prints the line from that eager region (`config.launch` says so) and every
rank leaves without touching the communicator again -- its state is unknown after a failed capture, and limping on with
it is a hang on N ranks, not a fallback.  If the capture of the OPTIONAL second schedule fails, the line is the first
schedule's finished graph-replayed region (built before the second was tried: measure_exchange_schedules).  MIOpen's solver
search runs on rank 0 alone first (`config.miopen_find_rank0_first_s`); `config.rank_ms_per_step`: the fastest / slowest
rank's own step time.  (Ranks that die instead: the self-launching parent starts ONE fresh set with
`--graph 0`.)  `--dp ddp` runs torch's DistributedDataParallel, launched kernel by kernel.  Rank 0 prints ONE JSON line.
Besides the contract keys it carries
  roofline      -- HBM roofline of the dominant MRLA kernel (mrla_light_apply_bwd), timed live with HIP events on the
                   launch stream over `steps` steps launched kernel by kernel (the timed region itself when it is not
                   graph-replayed, else the same steps run once more right after it: events cannot be read out of a
                   replayed graph; `eager_launch_ms_per_step` is that region's step time).  `achieved` / `frac`:
                   SURVEY.md section 8(d)'s algorithmic bytes, 5*N*sizeof(bf16) per launch (dOut, x_t, o_prev in; dx, do
                   out).  `achieved_fused` / `frac_fused`: the 6*N*s the launch is built to move (+ conv3's output y3:
                   bn3's backward sums are folded into this pass, which deleted a 2*N*s pass of its own).  `path_frac`:
                   section 8(d)'s compulsory bytes of the whole MRLA path per step (3N forward + 5N backward per block)
                   over the time of ALL kernels of the path (two passes per direction + the [b,c]-sized kernels).
                   `traffic`: HBM bytes per launch from THIS round's committed PMC passes (`traffic_source`), quoted only if
                   those passes were taken on the library loaded now (`traffic_tied_by`: its sha256, or that of its sources;
                   scripts/lib_identity.py), else null with `traffic_stale`: true;
  cpu_baseline  -- the eager CPU restatement (oracle/eager_models.py, kind "port") forward on the host cores, the best of a
                   thread-count sweep, bounded sample, rank 0 at N=1 only;
  eager_rocm    -- the same restatement run eager on this GPU (the north-star's ">=4x" denominator), N=1 only;
  other_configs -- BASELINE configs 4 and 5 (deit_mrlal_tiny_patch16_224 b=256, resnet101_mrlab b=128), each measured
                   by a child process of the default N=1 run after the headline (`--no-others` skips them).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROUND = "r05"             # profiles/<ROUND>_* are this build's measurements; older rounds are never substituted
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide)
# forward+backward flops per 224x224 image (3 x the hook-counted forward MACs*2 of SURVEY.md section 8d); these are
# almost entirely MIOpen convolution flops, not this build's kernels -- reported for the "fraction of compute roofline"
MODEL_GFLOP_PER_IMAGE = {"resnet50_mrlal": 24.8, "resnet101_mrlab": 48.7}
OTHER_CONFIGS = (("deit_mrlal_tiny_patch16_224", 256), ("resnet101_mrlab", 128))      # BASELINE.json configs 4 and 5
STATUS_ENV = "MRLA_BENCH_STATUS_FILE"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet50_mrlal")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--drop-path", type=float, default=0.2, help="resnet/train.py:67 default")
    ap.add_argument("--no-baselines", action="store_true", help="skip the cpu_baseline / eager_rocm / other_configs legs")
    ap.add_argument("--no-others", action="store_true", help="skip the other_configs leg (BASELINE configs 4 and 5)")
    ap.add_argument("--sgd-fused", type=int, default=1, help="1: torch.optim.SGD(fused=True) (one pass), 0: the foreach implementation")
    ap.add_argument("--no-forward-only", action="store_true", help="skip the inference-pass leg (counter passes over the training step)")
    ap.add_argument("--eager", action="store_true", help="time the eager restatement instead (diagnostic)")
    ap.add_argument("--channels-last", type=int, default=-1,
                    help="1 / 0: force torch.channels_last on / off; -1: the model class default (on for resnet*_mrlal)")
    ap.add_argument("--benchmark", type=int, default=1,
                    help="torch.backends.cudnn.benchmark for the timed model: 1 as resnet/train.py:247 sets it (MIOpen picks its "
                         "solvers by measuring them during the warm-up steps), 0 for MIOpen's immediate-mode choice")
    ap.add_argument("--deterministic", type=int, default=0,
                    help="torch.backends.cudnn.deterministic for the timed model (resnet/train.py:107-110 sets it with --seed): MIOpen "
                         "then leaves out its atomically accumulating (split-K) solvers -- the ones that are right when launched "
                         "eagerly and garbage from the second replay of a graph on.  1 / 0 (default): on / off; -1: off, and switched "
                         "ON for one more capture + check if the replayed graph does not reproduce the eager step (abandoned again if "
                         "the eager step then runs > 1.5 x slower).  Not the default: MIOpen's deterministic solver list runs "
                         "resnet50_mrlal b = 256 at 7.3 s per step on MI355X (bit-reproducible, 240 x slower; profiles/r05_notes.md)")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: the timed steps replay the whole step (fwd+bwd+SGD) from one HIP graph; 0: launched kernel by "
                         "kernel; -1 (default): 1, except with --dp ddp or a non-RCCL backend")
    ap.add_argument("--backend", default=os.environ.get("MRLA_DIST_BACKEND", "nccl"))
    ap.add_argument("--dp", choices=["auto", "flat", "ddp"], default="auto",
                    help="gradient exchange at N > 1.  flat: mrla_amd.distributed.FlatGradientExchange (one flat gradient "
                         "buffer; the whole step, exchange included, replays from one HIP graph like the N = 1 point); ddp: "
                         "torch DistributedDataParallel (bucketed, overlapped with backward, launched kernel by kernel); auto: "
                         "flat unless --graph 0")
    ap.add_argument("--exchange", choices=["ab", "after", "overlap"], default="ab",
                    help="schedule of the flat exchange.  after: ONE all-reduce after backward; overlap: ~25 MB buckets sent "
                         "from backward's hooks as they fill (DistributedDataParallel's schedule, resnet/train.py:174); ab "
                         "(default): time both on this hardware, report both, run the timed region with the faster")
    ap.add_argument("--ab-steps", type=int, default=6, help="steps per schedule of the --exchange ab comparison")
    ap.add_argument("--split-graph", action="store_true",
                    help="N > 1: replay the step from TWO HIP graphs (fwd+loss+bwd+gradient gather | SGD) around an eagerly "
                         "launched all-reduce -- what runs by itself when the collective cannot be captured (gloo; a failed "
                         "pre-flight); this flag forces it (diagnostic)")
    ap.add_argument("--inject-capture-failure", nargs="?", const="first", default="",
                    choices=["", "first", "after_backward", "bucketed_overlap", "ddp"],
                    help="diagnostic: raise inside the stream capture of the data-parallel step, after its collective has been "
                         "enqueued.  first (the default value): the first capture -- exercises 'report the eager region, leave "
                         "without the communicator'; bucketed_overlap: the optional second schedule of --exchange ab -- exercises "
                         "'report the first schedule's finished graph-replayed region'")
    ap.add_argument("--ddp-probe", action="store_true",
                    help="diagnostic on one GPU: a ONE-rank process group around the model, so that the N > 1 path -- the "
                         "exchange schedules, their hooks and the RCCL all-reduce launches, captured with --graph 1 -- runs "
                         "without a second GPU (nobody to exchange with: it measures the overhead side only)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# N > 1 launched plainly: start the ranks (nothing in this function may touch the GPU)
# ------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


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
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv + extra
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


# ------------------------------------------------------------------------------------------------------------------
def make_step(net, opt, x, y, exchange=None):
    """resnet/train.py:397-409 (forward, criterion, zero_grad, backward, optimizer step) under bf16 autocast.  `step.loss` is
    the latest call's loss tensor (right after a capture: the graph's static loss, which every replay overwrites)."""
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if exchange is not None:
            exchange.reduce()              # the N > 1 gradient average (capturable)
        opt.step()
        # (a detached alias: holding the loss itself would keep the step's autograd graph and the parameters' AccumulateGrad
        # nodes -- with the stream they were created on -- alive into the next step, and a later capture segfaults in capture_end)
        step.loss = loss.detach()
        return step.loss
    step.loss = None
    return step


SGD_FUSED = True          # (--sgd-fused 0: the foreach implementation -- four multi-tensor passes instead of one)


def sgd(params):
    """resnet/train.py:199-201: torch.optim.SGD(lr 0.1, momentum 0.9, weight decay 1e-4).  `fused=True` is the same optimizer
    in its single-pass implementation (gradient, weight and momentum buffer read once, weight and buffer written once: 5
    tensor-passes per step instead of foreach's 11); product run and eager baseline both use it."""
    params = list(params)
    if SGD_FUSED:
        try:
            return torch.optim.SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4, fused=True)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4)


RANK_MS = {}              # per-rank step time of the latest timed() region: {"min": ..., "max": ...} (ms; N > 1 only)


def timed(step, steps, warmup):
    """The contract's timing rule: `steps` steps between barrier + synchronize on both sides, the MAX over ranks.  Each rank's
    own time (up to its synchronize, before the closing barrier) is gathered too: RANK_MS shows a slow rank."""
    from mrla_amd import distributed as D
    for _ in range(warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    own = time.perf_counter() - t0
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0)
    lo, hi = D.min_max_over_ranks(own)
    RANK_MS.clear()
    RANK_MS.update(min=round(1e3 * lo / max(1, steps), 3), max=round(1e3 * hi / max(1, steps), 3))
    return dt


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arch, budget_s=75.0):
    """Eager CPU restatement, forward only, b=32 fp32.  The host's fair number: the thread count is
    swept ({16, 32, 64, 128} and the 8 of the survey container, capped at the logical CPUs; one timed iteration each after
    a common warm-up), channels_last is tried at the best count, and >= 3 iterations are timed with the winner."""
    from oracle import eager_models as em
    cores = os.cpu_count() or 1
    cand = sorted({min(t, cores) for t in (8, 16, 32, 64, 128)})
    net = getattr(em, "eager_" + arch)().eval()
    xb = torch.randn(32, 3, 224, 224)
    t_start = time.perf_counter()
    tried = {}

    def one(x, n=1):
        t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        return (time.perf_counter() - t0) / n

    with torch.no_grad():
        torch.set_num_threads(cand[len(cand) // 2])
        net(xb)                                                          # warm-up (allocator, oneDNN primitives)
        for t in cand:
            if time.perf_counter() - t_start > budget_s * 0.5 and tried:
                break
            torch.set_num_threads(t)
            tried[str(t)] = round(32 / one(xb), 2)
        best_t = int(max(tried, key=tried.get))
        torch.set_num_threads(best_t)
        fmt, x_best = "contiguous (NCHW)", xb
        if time.perf_counter() - t_start < budget_s * 0.6:
            net_cl, x_cl = net.to(memory_format=torch.channels_last), xb.contiguous(memory_format=torch.channels_last)
            net_cl(x_cl)
            cl = round(32 / one(x_cl), 2)
            tried[f"{best_t}+channels_last"] = cl
            if cl > tried[str(best_t)]:
                fmt, x_best = "channels_last", x_cl
            else:
                net.to(memory_format=torch.contiguous_format)
        n, t0 = 0, time.perf_counter()
        while n < 3 or (time.perf_counter() - t_start < budget_s * 0.8 and n < 10):
            net(x_best)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(32 * n / dt, 2), "unit": "images/sec", "cores": best_t, "kind": "port",
            "cpu": cpu_model_name(), "logical_cpus": cores, "threads_tried": tried, "memory_format": fmt,
            "sample": f"forward only (eval, no_grad), fp32, batch 32, {n} iterations with torch.set_num_threads({best_t}) "
                      f"(the best of threads_tried: one timed iteration each after a common warm-up), {fmt}"}


def eager_rocm(arch, batch, drop_path, steps=6):
    from oracle import eager_models as em
    torch.manual_seed(0)
    was_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = False      # the baseline always gets MIOpen's full solver list (it launches eagerly)
    kw = {"drop_path_rate": drop_path} if arch.startswith("deit") else {"drop_path": drop_path}
    net = getattr(em, "eager_" + arch)(**kw).cuda().train()
    x = torch.randn(batch, 3, 224, 224, device="cuda")
    y = torch.randint(0, 1000, (batch,), device="cuda")
    step = make_step(net, sgd(net.parameters()), x, y)
    dt = timed(step, steps, 3)
    net.eval()
    # the north-star's denominator is the eager FORWARD; resnet/train.py:247 runs with cudnn.benchmark = True (MIOpen's
    # exhaustive find), so the forward is timed under both settings, after the find has finished, and the FASTER one is
    # the denominator that is reported (the conservative ratio)
    was = torch.backends.cudnn.benchmark
    fwd = {}
    for bm in (False, True):
        torch.backends.cudnn.benchmark = bm
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            t_find = time.perf_counter()
            for _ in range(3):
                net(x)
            torch.cuda.synchronize()
            t_find = time.perf_counter() - t_find
            t0 = time.perf_counter()
            for _ in range(steps):
                net(x)
            torch.cuda.synchronize()
            fw = (time.perf_counter() - t0) / steps
        fwd[bm] = (round(batch / fw, 1), round(t_find, 1))
    torch.backends.cudnn.benchmark = was
    torch.backends.cudnn.deterministic = was_det
    return {"fwd_bwd_images_per_sec": round(batch * steps / dt, 1),
            "fwd_images_per_sec": max(fwd[False][0], fwd[True][0]),
            "fwd_images_per_sec_benchmark_false": fwd[False][0], "fwd_images_per_sec_benchmark_true": fwd[True][0],
            "warmup_s_benchmark_false": fwd[False][1], "warmup_s_benchmark_true": fwd[True][1],
            "what": "oracle/eager_models.py (stock ATen/MIOpen ops) on this GPU, same batch/dtype/optimizer; forward "
                    "timed with torch.backends.cudnn.benchmark False and True (resnet/train.py:247), the faster one is "
                    f"fwd_images_per_sec; fwd_bwd with the flag as the product run has it ({bool(was)})"}


def forward_only(net, x, steps=10, graph=True):
    """Inference pass (eval, no_grad, bf16 autocast) of the product network: the numerator of the north-star's
    ">= 4x the eager PyTorch-ROCm forward" target (eager_rocm.fwd_images_per_sec is its denominator).  Timed both as
    PyTorch launches it and (graph=True) replayed from one HIP graph; `fwd_images_per_sec` is the faster of the two."""
    was = net.training
    net.eval()
    res = {"mode": "eval, no_grad, bf16 autocast"}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
        torch.cuda.synchronize()
        fw = (time.perf_counter() - t0) / steps
        res.update(fwd_images_per_sec=round(x.shape[0] / fw, 1), ms=round(1e3 * fw, 3), launch="kernel by kernel")
        if graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    net(x)
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    net(x)
                g.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    g.replay()
                torch.cuda.synchronize()
                fg = (time.perf_counter() - t0) / steps
                res.update(eager_launch_fwd_images_per_sec=res["fwd_images_per_sec"], eager_launch_ms=res["ms"],
                           graph_fwd_images_per_sec=round(x.shape[0] / fg, 1), graph_ms=round(1e3 * fg, 3))
                if fg < fw:            # the headline forward figure is the faster way of launching the same kernels
                    res.update(fwd_images_per_sec=res["graph_fwd_images_per_sec"], ms=res["graph_ms"],
                               launch="one HIP graph, replayed")
            except Exception as e:
                print(f"warning: forward HIP graph capture failed ({type(e).__name__}: {e})", file=sys.stderr)
    net.train(was)
    return res


def library_identity():
    """{"lib_sha256": of the libmrla_hip.so this process loads, "src_sha256": of the sources it is built from} -- what the
    committed counter passes are tied to (scripts/lib_identity.py writes the same record into them)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lib_identity", os.path.join(ROOT, "scripts", "lib_identity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.identity(ROOT)


def counters_current(meta):
    """Was a committed counter pass taken on the library that is being timed now?  ("library sha256" | "source sha256" |
    None): the same binary, or a rebuild from byte-identical kernel sources."""
    if not isinstance(meta, dict):
        return None
    try:
        me = library_identity()
    except OSError:
        return None
    if meta.get("lib_sha256") and meta.get("lib_sha256") == me["lib_sha256"]:
        return "library sha256"
    if meta.get("src_sha256") and meta.get("src_sha256") == me["src_sha256"]:
        return "source sha256"
    return None


def pmc_traffic(args, kernel):
    """(HBM bytes per launch of `kernel`, source file, tie) from THIS round's committed rocprofv3 PMC passes of this exact
    workload (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md; scripts/pmc_bench.sh), else
    (None, None, None): a kernel may have changed since an older round's pass, so older files are never substituted.  The
    file carries the identity of the library it was measured on (`_meta`); if that is not the library loaded now the bytes
    are NOT reported (None, file, "stale")."""
    rel = os.path.join("profiles", f"{ROUND}_pmc_traffic_{args.arch}_b{args.batch}.json")
    # C-ABI entry point -> the device kernel's name in the profile where they differ (the token MRLA-base module runs the flat
    # history kernels of base_nhwc.hip; the value backward's kernels are called base_value_bwd_*)
    alias = {"token_base_attend_bwd": "base_attend_bwd", "base_value_bwd_dv": "base_value_bwd",
             "base_pool_value_fwd": "light_stats_fwd_fused", "token_base_value_fwd": "token_value_fwd"}
    try:
        table = json.load(open(os.path.join(ROOT, rel)))
        name = kernel.replace("mrla_", "")
        rec = table.get(name) or table.get(alias.get(name, ""))
        if rec:
            tie = counters_current(table.get("_meta"))
            if tie is None:
                return None, rel, "stale"
            return int(rec["hbm_bytes_per_launch"]), rel, tie
    except (OSError, ValueError, KeyError):
        pass
    return None, None, None


def mfma_counter(args):
    """Whole-step MFMA utilisation from the committed counter pass of this workload (scripts/pmc_mfma.sh), or None."""
    rel = os.path.join("profiles", f"{ROUND}_pmc_mfma_whole_step_{args.arch}_b{args.batch}.json")
    try:
        rec = json.load(open(os.path.join(ROOT, rel)))
        tie = counters_current(rec.get("_meta"))
        if tie is None:
            return {"mfma_util": None, "stale": True, "source": rel,
                    "what": "the committed counter pass was taken on another build of libmrla_hip.so: not reported"}
        return {"mfma_util": round(float(rec["mfma_busy_over_gpu_active_all_simds"]), 4), "tied_by": tie,
                "mfma_busy_over_cu_busy": round(float(rec["mfma_busy_over_cu_busy"]), 4), "source": rel,
                "what": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs "
                        "x 1024): the fraction of SIMD-cycles the matrix pipe was busy while the GPU was active, over every "
                        "kernel of the training step (rocprofv3 --pmc, a separate run of this command launched kernel by "
                        "kernel); the MFMA work is MIOpen's convolutions and this build's 1x1 GEMMs -- the MRLA kernels "
                        "issue none"}
    except (OSError, ValueError, KeyError):
        return None


def is_path_kernel(name):
    """The MRLA path proper -- not the BatchNorm / convolution kernels of the 8(f) rows."""
    return name.startswith(("mrla_light_", "mrla_base_", "mrla_token_", "mrla_reduce_rows2"))


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


def run_other_configs():
    """BASELINE configs 4 and 5 as child processes of the default N = 1 run (fresh processes: their own MIOpen state and
    HIP graph; this process is idle meanwhile)."""
    out = {}
    for arch, batch in OTHER_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--arch", arch, "--batch", str(batch), "--steps", "10",
               "--warmup", "3", "--no-baselines"]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not lines:
                out[arch] = {"error": f"rc {p.returncode}: " + p.stderr.decode(errors="replace")[-400:]}
                continue
            rec = json.loads(lines[-1])
            out[arch] = {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "batch": batch,
                         "steps": rec["steps"], "launch": rec["config"]["launch"], "roofline": rec["roofline"],
                         "replay_matches_eager": rec["config"].get("replay_matches_eager"), "miopen": rec["config"].get("miopen"),
                         "replay_check": rec["config"].get("replay_check"), "weights_finite": rec["config"].get("weights_finite"),
                         "eager_launch_ms_per_step": rec.get("eager_launch_ms_per_step"),
                         "workload": rec["config"]["workload"], "wall_s": round(time.perf_counter() - t0, 1)}
        except (subprocess.TimeoutExpired, ValueError, KeyError) as e:
            out[arch] = {"error": f"{type(e).__name__}: {e}"[:400]}
    return out


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


def main():
    args = parse()
    global SGD_FUSED
    SGD_FUSED = bool(args.sgd_fused)
    # dmabuf IPC: RCCL between processes (and CUDA-tensor sharing) fails with `hipIpcGetMemHandle: invalid argument` on this
    # image's driver without it.  Set for EVERY rank before the first GPU call -- the ranks of the driver's own
    # `python -m torch.distributed.run ... bench.py` line never pass through launch_ranks().
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ProcessGroupNCCL's watchdog thread must never query an event that was last recorded inside a stream capture
    # (hipErrorCapturedEvent -> std::terminate: seen in one run out of two of the one-rank RCCL tests once the captured steps
    # carried collectives on side streams).  Its event cache hands events of captured collectives to later eager ones, and
    # the flight recorder keeps events of captured collectives for the watchdog to retire: both off for this process.
    for k, v in (("TORCH_NCCL_CUDA_EVENT_CACHE", "0"), ("TORCH_FR_BUFFER_SIZE", "0"), ("TORCH_NCCL_RETHROW_CUDA_ERRORS", "0")):
        os.environ.setdefault(k, v)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.ddp_probe:
        sys.exit(launch_ranks(args))              # (nothing before this line has touched the GPU)

    from mrla_amd import distributed as D
    rank, local, world = D.env_world()
    dist_on = world > 1
    local = local % max(1, torch.cuda.device_count())     # (lets a 1-GPU box exercise the N>1 code path over gloo)
    torch.cuda.set_device(local)
    if args.ddp_probe and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        kw = {"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}
        torch.distributed.init_process_group(args.backend, rank=0, world_size=1, **kw)
        dist_on = True
    D.init_from_env(args.backend)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; measuring {world} rank(s)", file=sys.stderr)
    seen = ranks_seen(world, args.backend) if dist_on else 1
    if seen != world:
        print(f"error: the process group connected {seen} rank(s), WORLD_SIZE is {world}", file=sys.stderr)
        sys.exit(3)

    from mrla_amd import functional as Fm
    torch.backends.cudnn.benchmark = bool(args.benchmark)
    torch.backends.cudnn.deterministic = args.deterministic == 1
    if not args.benchmark and args.graph < 0:
        # MIOpen's immediate mode picks, for some small-batch 3x3 shapes, a weight-gradient solver that accumulates into memory
        # it zeroes only once: eager launches are right, a REPLAYED graph returns garbage dW from the second replay on
        # (scripts/miopen_wrw_graph_probe.py, profiles/r04_notes.md section 10).  Find mode (the default here, and what
        # resnet/train.py:247 runs with) is not affected.  So --benchmark 0 launches eagerly unless --graph 1 asks otherwise
        # (and then config.replay_matches_eager says whether the replay held).
        args.graph = 0
        if rank == 0:
            print("note: --benchmark 0 (MIOpen's immediate mode): launching the step kernel by kernel (--graph 0); pass --graph 1 "
                  "to replay it from a HIP graph anyway", file=sys.stderr)
    torch.manual_seed(0)
    if args.eager:
        from oracle import eager_models as em
        kw = {"drop_path_rate": args.drop_path} if args.arch.startswith("deit") else {"drop_path": args.drop_path}
        net = getattr(em, "eager_" + args.arch)(**kw)
    else:
        from mrla_amd import models, vit
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            if args.arch.startswith("deit"):
                net = getattr(vit, args.arch)(drop_path_rate=args.drop_path)
            else:
                net = getattr(models, args.arch)(drop_path=args.drop_path)
    if args.channels_last >= 0 and hasattr(net, "channels_last"):
        net.channels_last = bool(args.channels_last)
        net.to(memory_format=torch.channels_last if args.channels_last else torch.contiguous_format)
    layout = "channels_last" if getattr(net, "channels_last", False) else "NCHW"
    # N > 1: the exchange that fits into the graph (flat) unless the caller asks for eager launches or for DDP
    dp = "none" if not dist_on else (args.dp if args.dp != "auto" else ("ddp" if args.graph == 0 else "flat"))
    # (a gloo exchange stages through the host: not capturable)
    use_graph = args.graph == 1 or (args.graph < 0 and dp != "ddp" and (not dist_on or args.backend == "nccl"))
    launch_note = ""
    # can the collective itself go into a HIP graph?  (a gloo exchange stages through the host; RCCL: ask the pre-flight)
    split_why = None
    if dist_on and dp == "flat" and args.graph != 0:
        if args.backend != "nccl":
            split_why = f"a {args.backend} all-reduce stages through the host and cannot be captured"
        elif args.split_graph:
            split_why = "--split-graph"
        elif use_graph:
            try:
                preflight_capture(seen)
            except Exception as e:
                print(f"warning: pre-flight capture of a 4-element all-reduce failed ({type(e).__name__}: {e}); the step is "
                      "replayed from two graphs around an eagerly launched all-reduce", file=sys.stderr)
                split_why = f"the pre-flight capture of a small all-reduce failed ({type(e).__name__})"
        if split_why:
            use_graph = False
    elif use_graph and dist_on and args.backend == "nccl":
        try:
            preflight_capture(seen)
        except Exception as e:
            print(f"warning: pre-flight capture of a 4-element all-reduce failed ({type(e).__name__}: {e}); launching the "
                  "step eagerly", file=sys.stderr)
            use_graph, launch_note = False, " [pre-flight capture of a small all-reduce failed: no graph]"

    gx = torch.Generator(device="cuda").manual_seed(0)
    gy = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(args.batch, 3, 224, 224, device="cuda", generator=gx)
    y = torch.randint(0, 1000, (args.batch,), device="cuda", generator=gy)

    R = dict(args=args, rank=rank, world=world, seen=seen, dist_on=dist_on, dp=dp, layout=layout, x=x, net=None,
             exchange=None, schedule=None, ab_ms=None, legs=True, replay=None, rank_ms=None)
    graph = None
    launch = "kernel by kernel (PyTorch eager launches)" + launch_note
    graph_launch = "one HIP graph per step (captured fwd+loss+bwd" + ("+gradient all-reduce" if dist_on else "") + "+SGD), replayed"
    not_reproduced = ("kernel by kernel (PyTorch eager launches; the captured HIP graph did NOT reproduce the eagerly launched "
                      "step -- config.replay_check -- so the eager launches are what is timed)")

    def warm_up(st, n):
        for _ in range(n):
            st()

    def deterministic_retry(first_check, st, n_warm):
        """The replay did not reproduce the eager step.  With --deterministic -1 (the default): switch
        torch.backends.cudnn.deterministic on -- MIOpen then leaves out its atomically accumulating (split-K) solvers, which
        are right when launched eagerly and garbage from the second replay of a graph on -- re-run the warm-up (MIOpen
        searches again among the others) and tell the caller to capture and check once more.  Returns True when it did."""
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
        if rank == 0:
            print("note: torch.backends.cudnn.deterministic = True from here on (MIOpen's atomically accumulating solvers do not "
                  "survive graph replay); warming up, capturing and checking once more", file=sys.stderr, flush=True)
        warm_up(st, n_warm)
        after = ms_per_step()
        # MIOpen's deterministic solver list can be catastrophically slow (resnet50_mrlal b = 256 on MI355X: 7.3 s per step
        # against 30 ms -- bit-reproducible, and useless): a retry that costs more than it can win is abandoned, on every rank
        worth_it = all_ranks_ok(after <= 1.5 * before, "deterministic-worth-it", rank, world)
        R["replay_first_attempt"] = first_check
        if not worth_it:
            torch.backends.cudnn.deterministic = False
            R["miopen_deterministic_why"] = ("tried after the first captured graph did not reproduce the eager step, and switched "
                                             f"off again: the eager step took {after:.0f} ms with MIOpen's deterministic solvers "
                                             f"against {before:.0f} ms without")
            if rank == 0:
                print(f"note: deterministic solvers run the step in {after:.0f} ms against {before:.0f} ms: switched off again",
                      file=sys.stderr, flush=True)
            warm_up(st, 1)
            torch.cuda.synchronize()
            return False
        R["miopen_deterministic_why"] = ("switched on after the first captured graph did not reproduce the eager step "
                                         f"(update_rel_l2 {(first_check or {}).get('update_rel_l2')}, noise "
                                         f"{(first_check or {}).get('noise_update_rel_l2')}, worst parameter "
                                         f"{(first_check or {}).get('worst_parameter')})")
        torch.cuda.synchronize()
        return True

    def find_first(module):
        """MIOpen's solver search (torch.backends.cudnn.benchmark), rank 0 alone first: forward + loss + backward of the bare
        module, twice -- every convolution's forward, input-gradient and weight-gradient problem of the step -- with NO
        optimizer step and NO collective (the other ranks are waiting, and the replicas must stay identical)."""
        if world == 1 or not args.benchmark:
            return

        def go():
            for _ in range(2):
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = torch.nn.functional.cross_entropy(module(x).float(), y)
                loss.backward()
            module.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        rank0_first(go, rank, world, "find")
        R["find_s"] = round(time.perf_counter() - t0, 1)

    def measured_eagerly_first(st):
        """A complete measurement -- `steps` steps, bracketed as the contract says, per-kernel events on -- taken BEFORE any
        collective goes into a stream capture.  If the FIRST capture then fails on any rank, THIS is what the line reports (the
        ranks agree on that through the TCP store and leave without touching the communicator): an N > 1 run never ends
        without its number.  (A later, optional capture that fails costs nothing: measure_exchange_schedules.)"""
        d = timed(st, args.steps, 0)                     # as a training loop launches it: one C call per tail, no events
        rms = dict(RANK_MS)
        t = Fm.KernelTimer()                             # ... and once more with an event pair around every kernel (roofline)
        Fm.TIMER = t
        de = timed(st, args.steps, 0)
        Fm.TIMER = None
        return dict(dt=d, dt_events=de, timer=t, rank_ms=rms)

    def capture_voted(st, tag, warm):
        """capture(st) with the ranks' vote; raises CaptureBroken when it failed on any rank."""
        err = None

        def poisoned():
            st()
            raise RuntimeError(f"injected failure inside the capture (--inject-capture-failure {args.inject_capture_failure})")
        inject = args.inject_capture_failure in ("all", tag) or (args.inject_capture_failure == "first" and tag in
                                                                 ("after_backward", "ddp"))
        try:
            g = capture(poisoned if inject else st, True, 0 if inject else warm)
        except Exception as e:                 # noqa: BLE001 -- whatever the runtime throws out of a broken capture
            g, err = None, e
        if all_ranks_ok(err is None, "capture/" + tag, rank, world):
            return g
        raise CaptureBroken(f"{type(err).__name__}: {err}" if err is not None else "failed on another rank")

    def report_and_leave(rec):
        if rank == 0:
            report(dict(R, **rec))
        leave_without_the_communicator(0)

    def states_after():
        # after warm-up, A/B, timed and event-timed steps: do all ranks still hold the same weights?  (they do if and only if
        # every step's exchange -- captured or not -- handed every rank the same averaged gradients) ... and are they numbers?
        in_sync = D.replicas_in_sync(list(net.parameters())) if dist_on else None
        finite = bool(torch.isfinite(torch.stack([p.detach().float().abs().max() for p in net.parameters()])).all())
        return in_sync, finite

    split_graphs = None
    eager_record = None            # a finished eager region with the per-kernel events (N > 1 flat: taken before the captures)
    if dp == "flat" and split_why:
        # ---- the collective cannot be captured: two graphs around an eagerly launched all-reduce (after backward) ----
        net = net.cuda().train()
        find_first(net)
        opt = sgd(net.parameters())
        ex = D.FlatGradientExchange(net.parameters(), overlap=False)
        eager_step = step = make_step(net, opt, x, y, ex)
        R.update(exchange=ex, schedule="after_backward")
        warm_up(step, args.warmup)

        def part1():
            with torch.autocast("cuda", dtype=torch.bfloat16):
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
            g1 = capture(part1, dist_on, 2)        # (thread-local capture mode: the watchdog may query eager works meanwhile)
            ex.allreduce_flat()
            g3 = capture(part3, dist_on, 0)
        except Exception as e:                           # noqa: BLE001
            err = e
        if all_ranks_ok(err is None, "capture/split", rank, world):
            def replay2():
                g1.replay()
                ex.allreduce_flat()
                g3.replay()
            R["replay"], ok = check_replay(eager_step, replay2, part1.loss, net, opt, rank, world, "split")
            if ok:
                split_graphs, step = (g1, g3), replay2
                launch = ("two HIP graphs per step (fwd+loss+bwd+gradient gather | SGD) around an eagerly launched all-reduce ("
                          + split_why + ")")
            else:
                launch = not_reproduced
        else:
            print(f"warning: HIP graph capture failed ({type(err).__name__ if err else 'on another rank'}: {err}); timing eager "
                  "launches", file=sys.stderr)
            ex.adopt()
    elif dp == "flat":
        # ---- the flat exchange; with --exchange ab both schedules are measured on this hardware ----
        net = net.cuda().train()
        find_first(net)
        opt = sgd(net.parameters())
        names = {"after": ["after_backward"], "overlap": ["bucketed_overlap"], "ab": ["after_backward", "bucketed_overlap"]}[args.exchange]
        if use_graph:
            state = {}

            def prepare(name, first):        # after_backward first: it registers no hooks that the other would trigger
                ex = D.FlatGradientExchange(net.parameters(), overlap=(name == "bucketed_overlap"), broadcast=first)
                st = make_step(net, opt, x, y, ex)
                warm_up(st, args.warmup if first else 2)
                if first:
                    R.update(exchange=ex, schedule=name)
                    fbk = state["eager"] = measured_eagerly_first(st)
                    if rank == 0:          # the eager region's finished line, should the very first capture break (as above)
                        state["eager_line"] = report(dict(R, dt=fbk["dt"], dt_eager=fbk["dt"], dt_events=fbk["dt_events"],
                                                          timer=fbk["timer"], use_graph=False,
                                                          legs=False, rank_ms=fbk["rank_ms"], launch="kernel by kernel"), emit=False)
                return dict(exchange=ex, step=st, name=name)

            def capture_graph(h, name):
                h["graph"] = capture_voted(h["step"], name, 3)
                h["static_loss"] = h["step"].loss
                return h["graph"].replay

            def verify(h, run, name):
                return check_replay(h["step"], run, h["static_loss"], net, opt, rank, world, name)

            def after_region(rec):           # kept with that region: per-rank times, replicas in sync, weights finite
                rec["rank_ms"] = dict(RANK_MS)
                rec["in_sync"], rec["finite"] = states_after()
                # ... and the finished LINE of that region, built NOW: if a later, optional capture breaks, rank 0 prints it and
                # every rank leaves at once -- RCCL's watchdog thread aborts the process within moments of a broken capture
                # that had collectives in it, so nothing may be left to compute then
                if rank == 0:
                    fbk = state["eager"]
                    rec["line"] = report(dict(R, dt=rec["dt"], dt_eager=fbk["dt"], dt_events=fbk["dt_events"], timer=fbk["timer"],
                                              use_graph=True, legs=False,
                                              exchange=rec["handle"]["exchange"], schedule=rec["handle"]["name"],
                                              ab_ms={rec["handle"]["name"]: round(1e3 * rec["dt"] / args.steps, 3)},
                                              replay=rec["check"], rank_ms=rec["rank_ms"], in_sync=rec["in_sync"],
                                              finite=rec["finite"], launch=graph_launch), emit=False)

            for attempt in (0, 1):
                recs, chosen, failure = measure_exchange_schedules(
                    names, prepare, capture_graph, lambda run, n: timed(run, n, 1 if n != args.steps else 0), args.steps,
                    args.ab_steps, verify, after_region)
                if chosen is not None or failure is not None or attempt == 1:
                    break
                # no schedule's replay reproduced the eager step (the ranks agree: every check was voted): once more with
                # MIOpen's deterministic solvers, on fresh exchanges
                for v in recs.values():
                    v["handle"]["exchange"].remove_hooks()
                if not deterministic_retry(recs[names[0]]["check"], recs[names[0]]["handle"]["step"], max(3, args.warmup)):
                    break
            fb = state["eager"]
            eager_record = fb
            ab_ms = {k: v["ab_ms"] for k, v in recs.items() if v["ab_ms"] is not None} if len(names) > 1 else None
            if failure is not None:
                fname, ferr = failure
                why = str(ferr)[:200]
                print(f"warning: HIP graph capture of the data-parallel step failed for schedule {fname} ({why})", file=sys.stderr, flush=True)
                if chosen is None:
                    # the first capture broke: the eager region measured before it
                    print("reporting the eager steps measured before it", file=sys.stderr, flush=True)
                    if rank == 0:
                        rec0 = json.loads(state["eager_line"])
                        rec0["config"]["launch"] = ("kernel by kernel (PyTorch eager launches; the HIP graph capture of the step failed -- "
                                                    f"{why} -- so this is the eager region timed before the capture; the communicator was "
                                                    "not used again)")
                        print(json.dumps(rec0), flush=True)
                    leave_without_the_communicator(0)
                # an optional later schedule broke: the finished graph-replayed region of the earlier one stands
                print(f"reporting the finished graph-replayed region of schedule {chosen}", file=sys.stderr, flush=True)
                if rank == 0:
                    rec0 = json.loads(recs[chosen]["line"])
                    rec0["config"]["launch"] += (f" (schedule {chosen}; the capture of the optional schedule {fname} failed -- {why} -- "
                                                 "after this region had been timed; the communicator was not used again)")
                    rec0["config"]["gradient_exchange_ab_ms"] = ab_ms
                    print(json.dumps(rec0), flush=True)
                leave_without_the_communicator(0)
            if chosen is None:                 # no replay reproduced the eager step: time the eager launches of schedule 0
                h = recs[names[0]]["handle"]
                for k, v in recs.items():
                    if k != names[0]:
                        v["handle"]["exchange"].remove_hooks()
                R.update(exchange=h["exchange"], schedule=names[0], ab_ms=ab_ms, replay=recs[names[0]]["check"])
                eager_step = step = h["step"]
                launch = not_reproduced
            else:
                c = recs[chosen]
                for k, v in recs.items():
                    if k != chosen:
                        v["handle"]["exchange"].remove_hooks()      # the loser's hooks must not fire in the winner's eager steps
                R.update(exchange=c["handle"]["exchange"], schedule=chosen, ab_ms=ab_ms, replay=c["check"])
                eager_step, graph = c["handle"]["step"], c["handle"]["graph"]
                step, launch = graph.replay, graph_launch
                R["dt_done"] = c["dt"]         # the full region of the chosen schedule has been timed already
                R["rank_ms_done"] = c["rank_ms"]
            del recs
        else:
            cands = {}
            for i, name in enumerate(names):
                ex = D.FlatGradientExchange(net.parameters(), overlap=(name == "bucketed_overlap"), broadcast=(i == 0))
                st = make_step(net, opt, x, y, ex)
                warm_up(st, args.warmup if i == 0 else 2)
                t = timed(st, args.ab_steps, 1) / args.ab_steps if len(names) > 1 else None
                cands[name] = dict(exchange=ex, step=st, ms=None if t is None else round(1e3 * t, 3))
            schedule = min(cands, key=lambda k: cands[k]["ms"]) if len(names) > 1 else names[0]
            for k, v in cands.items():
                if k != schedule:
                    v["exchange"].remove_hooks()
            R.update(exchange=cands[schedule]["exchange"], schedule=schedule,
                     ab_ms={k: v["ms"] for k, v in cands.items()} if len(names) > 1 else None)
            eager_step = step = cands[schedule]["step"]
            del cands
    else:
        net = net.cuda().train()
        find_first(net)
        if dist_on and use_graph:
            # capturing a DDP step (PyTorch's whole-network-capture recipe): the wrapper is built in a side-stream context and
            # at least 11 DDP iterations run eagerly on a side stream before the capture
            side0 = torch.cuda.Stream()
            side0.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side0):
                net = D.wrap_data_parallel(net.cuda().train(), device_ids=[local], force=args.ddp_probe)
            torch.cuda.current_stream().wait_stream(side0)
        else:
            net = D.wrap_data_parallel(net.cuda().train(), device_ids=[local], force=args.ddp_probe)
        opt = sgd(net.parameters())
        eager_step = step = make_step(net, opt, x, y)
        warm_up(step, args.warmup)                           # warm-up without the timer
        if use_graph and dist_on:
            fb = measured_eagerly_first(eager_step)
            try:
                graph = capture_voted(eager_step, "ddp", 11)
            except CaptureBroken as e:
                print(f"warning: HIP graph capture of the data-parallel step failed ({e}); reporting the eager steps measured "
                      "before it", file=sys.stderr, flush=True)
                report_and_leave(dict(dt=fb["dt"], dt_eager=fb["dt"], dt_events=fb["dt_events"], timer=fb["timer"], use_graph=False,
                                      legs=False,
                                      rank_ms=fb["rank_ms"],
                                      launch="kernel by kernel (PyTorch eager launches; the HIP graph capture of the step failed -- "
                                             f"{str(e)[:200]} -- so this is the eager region timed before the capture; the "
                                             "communicator was not used again)"))
            R["replay"], ok = check_replay(eager_step, graph.replay, eager_step.loss, net, opt, rank, world, "ddp")
            if ok:
                step, launch = graph.replay, graph_launch
            else:
                graph, launch = None, not_reproduced
        elif use_graph:
            # the whole training step is launch-order static (no host sync inside): capture it once into a HIP graph and
            # replay it -- the same kernels on the same buffers, minus the launch gaps -- after proving that the replay
            # computes what the eager launches compute (config.replay_matches_eager)
            try:
                for attempt in (0, 1):
                    graph = capture(eager_step, dist_on, 2)
                    R["replay"], ok = check_replay(eager_step, graph.replay, eager_step.loss, net, opt, rank, world, f"n1/{attempt}")
                    if ok:
                        step, launch = graph.replay, graph_launch
                        break
                    graph, launch = None, not_reproduced
                    if attempt == 1 or not deterministic_retry(R["replay"], eager_step, max(3, args.warmup)):
                        break
            except Exception as e:
                print(f"warning: HIP graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
                graph, step = None, eager_step
    use_graph = graph is not None or split_graphs is not None

    timer = Fm.KernelTimer()                             # every C-ABI launch
    # the timed region: exactly `steps` steps between barrier + synchronize
    if use_graph:
        if R.get("dt_done") is not None:                 # (N > 1 flat: measure_exchange_schedules timed the chosen schedule's region)
            dt = R.pop("dt_done")
            R["rank_ms"] = R.pop("rank_ms_done")
        else:
            dt = timed(step, args.steps, 0)
            R["rank_ms"] = dict(RANK_MS) if dist_on else None
    else:
        dt = timed(step, args.steps, 0)                  # eager launches as a training loop issues them (no events)
        R["rank_ms"] = dict(RANK_MS) if dist_on else None
    if eager_record is not None:                         # the eager regions were taken before the captures
        timer, dt_eager, dt_events = eager_record["timer"], eager_record["dt"], eager_record["dt_events"]
    else:
        # the same `steps` steps launched kernel by kernel: what resnet/train.py's loop gets unchanged ...
        dt_eager = timed(eager_step, args.steps, 0) if use_graph else dt
        # ... and once more with a HIP-event pair on the launch stream around EVERY kernel (events cannot be read out of a
        # replayed graph; this region feeds `roofline` / `mrla_kernels` only; the C ABI is then called pass by pass)
        Fm.TIMER = timer
        dt_events = timed(eager_step, args.steps, 0)
        Fm.TIMER = None
    R["dt_events"] = dt_events
    in_sync, finite = states_after()
    if rank == 0:
        graph = split_graphs = step = eager_step = None  # (report() may hand the GPU to child processes)
        report(dict(R, dt=dt, dt_eager=dt_eager, timer=timer, use_graph=use_graph, launch=launch, net=net, in_sync=in_sync,
                    finite=finite))
    if dist_on:
        D.barrier()
        torch.distributed.destroy_process_group()


def report(R, emit=True):
    """Rank 0: build the ONE JSON line from a finished measurement R (see main()), print it (emit) and return it."""
    from mrla_amd import functional as Fm  # noqa: F401
    args, world, seen, dist_on, dp = R["args"], R["world"], R["seen"], R["dist_on"], R["dp"]
    dt, dt_eager, timer, launch, use_graph = R["dt"], R["dt_eager"], R["timer"], R["launch"], R["use_graph"]
    exchange, schedule, ab_ms, layout, net, x = R["exchange"], R["schedule"], R["ab_ms"], R["layout"], R["net"], R["x"]
    ips = world * args.batch * args.steps / dt
    ks = timer.summary()
    path_k = {k: v for k, v in ks.items() if is_path_kernel(k)}
    big = {k: v for k, v in path_k.items() if v["bytes"] > 0}
    dom_name = max(big, key=lambda k: big[k]["ms"]) if big else None        # the path's kernel with the most time
    dom = ks.get(dom_name)
    roofline = None
    if dom:
        sec = dom["ms"] * 1e-3
        ach, ach_f = dom["bytes_alg"] / sec / 1e9, dom["bytes"] / sec / 1e9
        path_ms = sum(v["ms"] for v in path_k.values())
        path_b = sum(v["bytes_path"] for v in path_k.values())
        # DeiT keeps its residual stream (and therefore the token MRLA kernels) in fp32 under autocast, as the reference does
        kdt = "fp32" if args.arch.startswith("deit") else "bf16"
        traffic, traffic_src, traffic_tie = pmc_traffic(args, dom_name)
        roofline = {"bound": "hbm", "kernel": f"{dom_name}<{kdt}>", "achieved": round(ach, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_tie == "stale",
                    "traffic_tied_by": None if traffic_tie == "stale" else traffic_tie,
                    "launches": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                    "algorithmic_bytes_per_launch_avg": dom["bytes_alg"] // dom["launches"],
                    "achieved_fused": round(ach_f, 1), "frac_fused": round(ach_f / HBM_PEAK_GBS, 4),
                    "fused_bytes_per_launch_avg": dom["bytes"] // dom["launches"],
                    "path_frac": round(path_b / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if path_ms > 0 else None,
                    "path_ms_per_step": round(path_ms / args.steps, 3),
                    "path_bytes_per_step": path_b // args.steps,
                    "convention": "achieved/frac: SURVEY.md 8(d) algorithmic bytes of this launch; *_fused: all bytes the "
                                  "launch is built to move (differs where work of a neighbouring pass is folded in); "
                                  "path_frac: 8(d) compulsory bytes of the whole MRLA path per step / time of all its "
                                  "kernels (streaming passes + gate / reduce kernels) / peak"}
    gx_desc = None
    if dist_on:
        gx_desc = {"flat": None if exchange is None else
                   (f"{len(exchange.buckets)} all-reduce(s) (RCCL avg) over one flat fp32 gradient buffer"
                    + (", sent from backward as its buckets fill" if schedule == "bucketed_overlap" else ", after backward")),
                   "ddp": "DistributedDataParallel: 32 MB buckets, overlapped with backward"}[dp]
    out = {"metric": f"images/sec fwd+bwd {args.arch} b={args.batch}", "value": round(ips, 1), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"{args.arch} fwd+bwd+SGD, {args.batch} images/GPU of 3x224x224, bf16 autocast, "
                                  f"fp32 master weights, drop_path {args.drop_path}",
                      "global_batch": world * args.batch, "parallelism": f"dp{world}", "launch": launch,
                      "ranks_seen": seen, "weights_finite": R.get("finite"),
                      "miopen": {"find_mode": bool(torch.backends.cudnn.benchmark),          # resnet/train.py:247
                                 "deterministic_solvers_only": bool(torch.backends.cudnn.deterministic),   # train.py:107-110 (--seed)
                                 **({"why": R["miopen_deterministic_why"], "first_attempt": R.get("replay_first_attempt")}
                                    if R.get("miopen_deterministic_why") else {})},
                      # the replayed graph against eagerly launched steps from the same state (max over parameters of the
                      # relative L2 difference of the weights); null when the timed steps were launched eagerly anyway
                      "replay_matches_eager": (R.get("replay") or {}).get("weights_rel_l2"),
                      "replay_check": R.get("replay"),
                      "backend": ({"nccl": "nccl (RCCL)"}.get(args.backend, args.backend) if dist_on else "none (single process)"),
                      **({"gradient_exchange": gx_desc, "gradient_exchange_schedule": schedule,
                          "gradient_exchange_ab_ms": ab_ms, "replicas_in_sync": R.get("in_sync"),
                          "rank_ms_per_step": R.get("rank_ms"), "miopen_find_rank0_first_s": R.get("find_s")} if dist_on else {}),
                      "path": "eager restatement" if args.eager else
                              f"mrla_amd (HIP MRLA tails incl. shortcut add+ReLU, HIP BatchNorm+ReLU(+stem max-pool), HIP MFMA GEMMs for the "
                              f"1x1 convolutions fwd / dgrad / wgrad where eligible, stock 3x3 / 7x7 / strided convolutions; {layout})"},
           "eager_launch_ms_per_step": round(1e3 * dt_eager / args.steps, 3),
           **({"eager_launch_with_kernel_events_ms_per_step": round(1e3 * R["dt_events"] / args.steps, 3)}
              if R.get("dt_events") is not None else {}),
           # what resnet/train.py gets UNCHANGED (its loop launches the step eagerly, :387-409); `value` is the same step
           # replayed from one HIP graph -- mrla_amd.graphed_step(model, optimizer, criterion, (images, target)), INTEGRATION.md
           "eager_launch_images_per_sec": round(world * args.batch * args.steps / dt_eager, 1),
           "roofline": roofline,
           "mrla_kernels": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                **({"GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} if v["bytes"] else {})}
                            for k, v in ks.items()}}
    if args.arch in MODEL_GFLOP_PER_IMAGE:
        tf = ips * MODEL_GFLOP_PER_IMAGE[args.arch] / 1e3
        out["compute_roofline"] = {"model_tflops": round(tf, 1), "peak_bf16_mfma_tflops": MFMA_BF16_PEAK_TFLOPS * world,
                                   "frac": round(tf / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                                   "mfma_util_counter": mfma_counter(args),
                                   "note": "whole-model flops (MIOpen convolutions); the MRLA kernels are HBM/VALU work"}
    if world == 1 and R["legs"] and not args.no_forward_only:
        out["forward_only"] = forward_only(net, x, graph=use_graph)
    if world == 1 and R["legs"] and not dist_on and not args.no_baselines:
        out["eager_rocm"] = eager_rocm(args.arch, args.batch, args.drop_path)
        # like for like: both sides launched kernel by kernel by PyTorch (the eager restatement is never graph-replayed);
        # the graph-replayed product forward against the same denominator is reported beside it, labelled
        fo, den = out.get("forward_only"), out["eager_rocm"]["fwd_images_per_sec"]
        if fo is not None:                       # (--no-forward-only: no ratios)
            fo["vs_eager_rocm"] = round(fo.get("eager_launch_fwd_images_per_sec", fo["fwd_images_per_sec"]) / den, 2)
            if "graph_fwd_images_per_sec" in fo:
                fo["graph_replay_vs_eager_rocm"] = round(fo["graph_fwd_images_per_sec"] / den, 2)
        out["cpu_baseline"] = cpu_baseline(args.arch)
        if not args.no_others and (args.arch, args.batch) == ("resnet50_mrlal", 256):
            # this process goes idle: give its graph pool and cached blocks back first
            R["net"] = net = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["other_configs"] = run_other_configs()
    line = json.dumps(out)
    if emit:
        print(line, flush=True)
    return line


if __name__ == "__main__":
    main()
