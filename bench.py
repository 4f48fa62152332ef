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
`--graph 0`.)  `--dp ddp` runs torch's DistributedDataParallel, launched kernel by kernel -- and with the flat exchange that very configuration
is ALWAYS timed first for the full region (`config.ddp_eager_first`; `--ddp-first 0` skips it): whatever tier finishes, the line
that is printed is never slower than what resnet/train.py:174 unchanged would get.  Rank 0 prints ONE JSON line.
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
                   by a child process of the default N=1 run after the headline (`--no-others` skips the children);
  config.drop_path_0 -- the headline's step with drop_path 0 (the factory default, resnet_mrla_light.py:134) beside train.py's 0.2;
  detection_backbone -- `--arch det_resnet50_mrlal --shape 2x3x800x1344`: the mmdet backbone forward + backward where it runs
                   (mmdetection/mmdet/models/backbones/resnet_mrlal.py:283-293), with the eager restatement beside it.
`--autocast none` times resnet/train.py's own fp32 recipe (:397-409), `fp16` deit/engine.py:37's.
The parts live in benchkit/: common (flags, the step, the contract's timing), core (one rank's measurement), ranks +
dataparallel (N > 1), baselines (the legs beside the headline; the only importer of oracle/ besides --eager), report (the line).
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from benchkit import common  # noqa: E402
from benchkit.common import ROUND, free_port, make_step, parse, sgd, timed  # noqa: E402,F401  (re-exported: scripts/, tests/)
from benchkit.ranks import (CaptureBroken, all_ranks_ok, launch_ranks, measure_exchange_schedules, rank0_first,  # noqa: E402,F401
                            ranks_seen)
from benchkit.report import counters_current, library_identity, pmc_traffic, report  # noqa: E402,F401


def main():
    args = parse()
    common.SGD_FUSED = bool(args.sgd_fused)
    common.AUTOCAST_DTYPE = common.AUTOCAST[args.autocast]
    # dmabuf IPC: RCCL between processes (and CUDA-tensor sharing) fails with `hipIpcGetMemHandle: invalid argument` on this
    # image's driver without it.  Set for EVERY rank before the first GPU call -- the ranks of the driver's own
    # `python -m torch.distributed.run ... bench.py` line never pass through launch_ranks().
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # ProcessGroupNCCL's watchdog thread must never query an event that was last recorded inside a stream capture
    # (hipErrorCapturedEvent -> std::terminate: seen in one run out of two of the one-rank RCCL tests once the captured steps
    # carried collectives on side streams).  Its event cache hands events of captured collectives to later eager ones, and
    # the flight recorder keeps events of captured collectives for the watchdog to retire: both off for this process.
    # (TORCH_NCCL_RETHROW_CUDA_ERRORS stays at its default: asynchronous errors of the timed run must surface.)
    for k, v in (("TORCH_NCCL_CUDA_EVENT_CACHE", "0"), ("TORCH_FR_BUFFER_SIZE", "0")):
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

    from benchkit import core, dataparallel
    run = core.Run(args, rank, local, world, dist_on, seen)
    ddp = None                     # tier 0's (seconds, finished line): plain DistributedDataParallel, timed before anything else
    if run.dp == "flat" and run.split_why:
        ddp = dataparallel.flat_split(run)
    elif run.dp == "flat":
        ddp = dataparallel.flat_schedules(run)
    else:
        run.single_or_ddp()
    run.timed_region_and_report(ddp)


if __name__ == "__main__":
    main()
