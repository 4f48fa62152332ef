"""Shared by every part of the benchmark (bench.py): constants, the command line, the training step, the contract's timing."""
import argparse
import os
import socket
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")

ROUND = "r06"             # profiles/<ROUND>_* are this build's measurements; older rounds are never substituted
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide)
# forward+backward flops per 224x224 image (3 x the hook-counted forward MACs*2 of SURVEY.md section 8d); these are
# almost entirely MIOpen convolution flops, not this build's kernels -- reported for the "fraction of compute roofline"
MODEL_GFLOP_PER_IMAGE = {"resnet50_mrlal": 24.8, "resnet101_mrlab": 48.7}
OTHER_CONFIGS = (("deit_mrlal_tiny_patch16_224", 256), ("resnet101_mrlab", 128))      # BASELINE.json configs 4 and 5
STATUS_ENV = "MRLA_BENCH_STATUS_FILE"
DET_LR = 1e-5             # the detection backbone is timed without a head: a loss on the raw feature maps wants a small step
AUTOCAST = {"bf16": torch.bfloat16, "fp16": torch.float16, "none": None}
DTYPE_NAME = {"bf16": "bf16", "fp16": "fp16", "none": "fp32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet50_mrlal")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--drop-path", type=float, default=0.2, help="resnet/train.py:67 default")
    ap.add_argument("--autocast", choices=sorted(AUTOCAST), default="bf16",
                    help="bf16 (BASELINE's metric); none: fp32, resnet/train.py's own recipe (:397-409, no AMP); fp16: deit/engine.py:37")
    ap.add_argument("--shape", default="", help="det_* backbones: the image batch as BxCxHxW (default 2x3x800x1344)")
    ap.add_argument("--no-baselines", action="store_true", help="skip the cpu_baseline / eager_rocm / other_configs legs")
    ap.add_argument("--no-others", action="store_true",
                    help="skip the child-process legs (other_configs = BASELINE configs 4 and 5, drop_path 0, the detection backbone)")
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
    ap.add_argument("--ddp-first", type=int, default=1,
                    help="N > 1 with the flat exchange: 1 (default) times plain DistributedDataParallel, launched eagerly, for the "
                         "full region BEFORE any other tier and keeps its finished line -- the printed line is never slower")
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ------------------------------------------------------------------------------------------------------------------
AUTOCAST_DTYPE = torch.bfloat16      # (--autocast; None: no autocast, fp32 as resnet/train.py itself trains)


def autocast():
    """The autocast context of the timed model."""
    return torch.autocast("cuda", dtype=AUTOCAST_DTYPE or torch.bfloat16, enabled=AUTOCAST_DTYPE is not None)


def make_step(net, opt, x, y, exchange=None):
    """resnet/train.py:397-409 (forward, criterion, zero_grad, backward, optimizer step) under the chosen autocast.  `step.loss`
    is the latest call's loss tensor (right after a capture: the graph's static loss, which every replay overwrites)."""
    def step():
        with autocast():
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


def sgd(params, lr=0.1):
    """resnet/train.py:199-201: torch.optim.SGD(lr 0.1, momentum 0.9, weight decay 1e-4).  `fused=True` is the same optimizer
    in its single-pass implementation (gradient, weight and momentum buffer read once, weight and buffer written once: 5
    tensor-passes per step instead of foreach's 11); product run and eager baseline both use it."""
    params = list(params)
    if SGD_FUSED:
        try:
            return torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=1e-4, fused=True)
        except (RuntimeError, TypeError, ValueError):
            pass
    return torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=1e-4)


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

