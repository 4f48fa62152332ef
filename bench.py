#!/usr/bin/env python3
"""Headline benchmark: images/sec fwd+bwd of resnet50_mrlal, batch 256 per GPU, bf16 autocast, synthetic
ImageNet-shaped data (BASELINE.json metric / configs[1]; configs[2] for --gpus N > 1).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward + loss + backward + SGD update on one synthetic batch already resident in HBM.  On a single GPU the
timed steps replay the whole step from one HIP graph (`config.launch`; `--graph 0` times PyTorch's kernel-by-kernel
launches instead).  Under torch.distributed the step is captured as well, with the gradient exchange inside the graph as
ONE all-reduce over one flat gradient buffer after backward (`--dp flat`, mrla_amd/distributed.py: FlatGradientExchange;
MRLA_FLAT_OVERLAP=1: bucketed and sent from backward -- measured slower inside the graph; if the capture fails the same step
is launched eagerly); `--dp ddp` runs torch's DistributedDataParallel, launched kernel by kernel.  Rank 0 prints
ONE JSON line.  Besides the contract
keys it carries
  roofline     -- HBM roofline of the dominant MRLA kernel (mrla_light_apply_bwd), timed live with HIP events on
                  the launch stream over `steps` steps launched kernel by kernel (the timed region itself when it is not
                  graph-replayed, else the same steps run once more right after it: events cannot be read out of a
                  replayed graph; `eager_launch_ms_per_step` is that region's step time); algorithmic bytes =
                  6*N*sizeof(bf16) per launch (dOut, x_t, o_prev, y3 in; dx, do out);
  cpu_baseline -- the eager CPU restatement (oracle/eager_models.py, kind "port") forward on the host cores,
                  bounded sample, rank 0 at N=1 only;
  eager_rocm   -- the same restatement run eager on this GPU (the north-star's ">=4x" denominator), N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide)
# forward+backward flops per 224x224 image (3 x the hook-counted forward MACs*2 of SURVEY.md section 8d); these are
# almost entirely MIOpen convolution flops, not this build's kernels -- reported for the "fraction of compute roofline"
MODEL_GFLOP_PER_IMAGE = {"resnet50_mrlal": 24.8, "resnet101_mrlab": 48.7}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="resnet50_mrlal")
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--drop-path", type=float, default=0.2, help="resnet/train.py:67 default")
    ap.add_argument("--no-baselines", action="store_true", help="skip the cpu_baseline / eager_rocm legs")
    ap.add_argument("--eager", action="store_true", help="time the eager restatement instead (diagnostic)")
    ap.add_argument("--channels-last", type=int, default=-1,
                    help="1 / 0: force torch.channels_last on / off; -1: the model class default (on for resnet*_mrlal)")
    ap.add_argument("--benchmark", type=int, default=1,
                    help="torch.backends.cudnn.benchmark for the timed model: 1 as resnet/train.py:247 sets it (MIOpen picks its "
                         "solvers by measuring them during the warm-up steps), 0 for MIOpen's immediate-mode choice")
    ap.add_argument("--graph", type=int, default=-1,
                    help="1: the timed steps replay the whole step (fwd+bwd+SGD) from one HIP graph; 0: launched kernel by "
                         "kernel; -1 (default): 1, except with --dp ddp or a non-RCCL backend")
    ap.add_argument("--backend", default=os.environ.get("MRLA_DIST_BACKEND", "nccl"))
    ap.add_argument("--dp", choices=["auto", "flat", "ddp"], default="auto",
                    help="gradient exchange at N > 1.  flat: mrla_amd.distributed.FlatGradientExchange -- one all-reduce over "
                         "one flat buffer after backward, which lets the whole step (exchange included) "
                         "replay from one HIP graph like the N = 1 point; ddp: torch DistributedDataParallel (bucketed, overlapped with "
                         "backward, launched kernel by kernel); auto: flat unless --graph 0")
    ap.add_argument("--ddp-probe", action="store_true",
                    help="diagnostic on one GPU: a ONE-rank process group + DistributedDataParallel around the model, so "
                         "that the reducer hooks, bucket views and RCCL all-reduce launches of the N > 1 path run (and can "
                         "be graph-captured with --graph 1) without a second GPU")
    return ap.parse_args()


def make_step(net, opt, x, y, exchange=None):
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(net(x).float(), y)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if exchange is not None:
            exchange.reduce()              # the N > 1 gradient average (one flat all-reduce; capturable)
        opt.step()
        return loss
    return step


def sgd(params):
    return torch.optim.SGD(params, lr=0.1, momentum=0.9, weight_decay=1e-4)   # resnet/train.py:199-201


def timed(step, steps, warmup, dist_on):
    from mrla_amd import distributed as D
    for _ in range(warmup):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    return D.max_over_ranks(time.perf_counter() - t0)


def cpu_baseline(arch):
    """Eager CPU restatement, forward only, b=32 fp32, bounded to ~10-30 s."""
    from oracle import eager_models as em
    cores = os.cpu_count() or 1
    threads = min(cores, 128)
    torch.set_num_threads(threads)
    net = getattr(em, "eager_" + arch)().eval()
    xb = torch.randn(32, 3, 224, 224)
    with torch.no_grad():
        net(xb)
        t0 = time.perf_counter()
        n = 0
        while n < 3 or (time.perf_counter() - t0 < 10.0 and n < 20):
            net(xb)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(32 * n / dt, 2), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"forward only (eval, no_grad), fp32, batch 32, {n} iterations after 1 warm-up, "
                      f"torch.set_num_threads({threads}) of {cores} logical CPUs"}


def eager_rocm(arch, batch, drop_path, steps=6):
    from oracle import eager_models as em
    torch.manual_seed(0)
    kw = {"drop_path_rate": drop_path} if arch.startswith("deit") else {"drop_path": drop_path}
    net = getattr(em, "eager_" + arch)(**kw).cuda().train()
    x = torch.randn(batch, 3, 224, 224, device="cuda")
    y = torch.randint(0, 1000, (batch,), device="cuda")
    step = make_step(net, sgd(net.parameters()), x, y)
    dt = timed(step, steps, 3, False)
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


def pmc_traffic(args, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (2*FETCH_SIZE + WRITE_SIZE, gfx950
    correction of MI355X_MICROARCH.md; collected by scripts/pmc_bench.sh on this exact workload), else None."""
    for rnd in ("r03", "r02", "r01"):   # the newest committed PMC pass of this workload
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{args.arch}_b{args.batch}.json")
        try:
            rec = json.load(open(path)).get(kernel.replace("mrla_", ""))
            if rec:
                return int(rec["hbm_bytes_per_launch"])
        except (OSError, ValueError, KeyError):
            continue
    return None


def main():
    args = parse()
    from mrla_amd import distributed as D
    rank, local, world = D.env_world()
    dist_on = world > 1
    local = local % max(1, torch.cuda.device_count())     # (lets a 1-GPU box exercise the N>1 code path over gloo)
    torch.cuda.set_device(local)
    if args.ddp_probe and world == 1:
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(port))
        kw = {"device_id": torch.device("cuda", local)} if args.backend == "nccl" else {}
        torch.distributed.init_process_group(args.backend, rank=0, world_size=1, **kw)
        dist_on = True
    D.init_from_env(args.backend)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    from mrla_amd import functional as Fm
    torch.backends.cudnn.benchmark = bool(args.benchmark)
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
    exchange = None
    if dp == "flat":
        net = net.cuda().train()
        exchange = D.FlatGradientExchange(net.parameters(), overlap=os.environ.get("MRLA_FLAT_OVERLAP", "0") == "1")
    elif use_graph and dist_on:
        # capturing a DDP step (PyTorch's whole-network-capture recipe): the wrapper is built in a side-stream context and
        # at least 11 DDP iterations run eagerly on a side stream before the capture
        side0 = torch.cuda.Stream()
        side0.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side0):
            net = D.wrap_data_parallel(net.cuda().train(), device_ids=[local], force=args.ddp_probe)
        torch.cuda.current_stream().wait_stream(side0)
    else:
        net = D.wrap_data_parallel(net.cuda().train(), device_ids=[local], force=args.ddp_probe)
    gx = torch.Generator(device="cuda").manual_seed(0)
    gy = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(args.batch, 3, 224, 224, device="cuda", generator=gx)
    y = torch.randint(0, 1000, (args.batch,), device="cuda", generator=gy)
    step = make_step(net, sgd(net.parameters()), x, y, exchange)

    # warm-up without the timer
    for _ in range(args.warmup):
        step()
    eager_step = step
    launch = "kernel by kernel (PyTorch eager launches)"
    if use_graph:
        # the whole training step is launch-order static (no host sync inside): capture it once into a HIP graph and
        # replay it -- the same kernels on the same buffers, minus ~1 ms/step of launch gaps
        try:
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(11 if dist_on else 2):
                    eager_step()
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # (thread_local: RCCL's watchdog thread may query events while this thread captures)
            with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if dist_on else {})):
                eager_step()
            step = graph.replay
            launch = "one HIP graph per step (captured fwd+loss+bwd" + ("+gradient all-reduce" if dist_on else "") + "+SGD), replayed"
        except Exception as e:                        # capture not possible here: time the eager launches instead
            print(f"warning: HIP graph capture failed ({type(e).__name__}: {e}); timing eager launches", file=sys.stderr)
            step, use_graph = eager_step, False
    timer = Fm.KernelTimer(["mrla_light_apply_bwd", "mrla_light_stats_bwd", "mrla_light_apply_fwd",
                            "mrla_light_stats_fwd", "mrla_base_attend_fwd", "mrla_base_tail_fwd",
                            "mrla_base_tail_stats_bwd", "mrla_base_attend_bwd", "mrla_base_value_bwd",
                            "mrla_base_pool_value_fwd", "mrla_base_dv_combine", "mrla_base_value_bwd_dv",
                            "mrla_token_apply_fwd", "mrla_token_apply_bwd", "mrla_token_ln_bwd", "mrla_token_base_value_fwd",
                            "mrla_token_base_attend_fwd", "mrla_token_base_attend_bwd", "mrla_token_base_value_bwd",
                            "mrla_light_stats_fwd_fused", "mrla_light_pool_fused", "mrla_light_apply_fwd_fused", "mrla_bn_plane_moments", "mrla_bn_act_fwd",
                            "mrla_bn_plane_dmoments", "mrla_bn_act_bwd",
                            "mrla_conv1x1_fwd", "mrla_conv1x1_bwd_data", "mrla_conv1x1_wgrad", "mrla_bn_relu_pool_fwd",
                            "mrla_bn_relu_pool_dmoments", "mrla_bn_relu_pool_bwd"])
    # the timed region: exactly `steps` steps between barrier + synchronize
    if use_graph:
        dt = timed(step, args.steps, 0, dist_on)
        # per-kernel HIP events cannot be read out of a replayed graph: the same `steps` steps once more, launched kernel by
        # kernel with the events on the launch stream (this second region feeds `roofline` / `mrla_kernels` only)
        Fm.TIMER = timer
        dt_eager = timed(eager_step, args.steps, 0, dist_on)
        Fm.TIMER = None
    else:
        Fm.TIMER = timer
        dt = timed(step, args.steps, 0, dist_on)
        Fm.TIMER = None
        dt_eager = dt
    ips = world * args.batch * args.steps / dt

    if rank == 0:
        ks = timer.summary()
        path_k = {k: v for k, v in ks.items() if not k.startswith(("mrla_bn_", "mrla_conv1x1"))}   # the MRLA path proper
        dom_name = max(path_k, key=lambda k: path_k[k]["ms"]) if path_k else None  # its kernel with the most time
        dom = ks.get(dom_name)
        roofline = None
        if dom:
            ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
            # DeiT keeps its residual stream (and therefore the token MRLA kernels) in fp32 under autocast, as the reference does
            kdt = "fp32" if args.arch.startswith("deit") else "bf16"
            roofline = {"bound": "hbm", "kernel": f"{dom_name}<{kdt}>", "achieved": round(ach, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "traffic": pmc_traffic(args, dom_name),
                        "launches": dom["launches"], "avg_launch_us": round(1e3 * dom["ms"] / dom["launches"], 2),
                        "algorithmic_bytes_per_launch_avg": dom["bytes"] // dom["launches"]}
        out = {"metric": f"images/sec fwd+bwd {args.arch} b={args.batch}", "value": round(ips, 1), "unit": "images/sec",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": f"{args.arch} fwd+bwd+SGD, {args.batch} images/GPU of 3x224x224, bf16 autocast, "
                                      f"fp32 master weights, drop_path {args.drop_path}",
                          "global_batch": world * args.batch, "parallelism": f"dp{world}", "launch": launch,
                          **({"gradient_exchange": {"flat": f"{len(exchange.buckets) if exchange else 0} all-reduce(s) (RCCL avg) over one flat "
                                                            "fp32 gradient buffer" + (", sent from backward as its buckets fill" if exchange and len(exchange.buckets) > 1
                                                                                     else ", after backward"),
                                                    "ddp": "DistributedDataParallel: 32 MB buckets, overlapped with backward"}[dp]}
                             if dist_on else {}),
                          "path": "eager restatement" if args.eager else
                                  f"mrla_amd (HIP MRLA tails incl. shortcut add+ReLU, HIP BatchNorm+ReLU(+stem max-pool), HIP MFMA GEMMs for the "
                                  f"1x1 convolutions fwd / dgrad / wgrad where eligible, stock 3x3 / 7x7 / strided convolutions; {layout})"},
               "eager_launch_ms_per_step": round(1e3 * dt_eager / args.steps, 3),
               "roofline": roofline,
               "mrla_kernels": {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / args.steps, 3),
                                    "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} for k, v in ks.items()}}
        if args.arch in MODEL_GFLOP_PER_IMAGE:
            tf = ips * MODEL_GFLOP_PER_IMAGE[args.arch] / 1e3
            out["compute_roofline"] = {"model_tflops": round(tf, 1), "peak_bf16_mfma_tflops": MFMA_BF16_PEAK_TFLOPS * world,
                                       "frac": round(tf / (MFMA_BF16_PEAK_TFLOPS * world), 4),
                                       "note": "whole-model flops (MIOpen convolutions); the MRLA kernels are HBM/VALU work"}
        if world == 1:
            out["forward_only"] = forward_only(net, x, graph=use_graph)
        if world == 1 and not args.no_baselines:
            out["eager_rocm"] = eager_rocm(args.arch, args.batch, args.drop_path)
            # like for like: both sides launched kernel by kernel by PyTorch (the eager restatement is never graph-replayed);
            # the graph-replayed product forward against the same denominator is reported beside it, labelled
            fo, den = out["forward_only"], out["eager_rocm"]["fwd_images_per_sec"]
            fo["vs_eager_rocm"] = round(fo.get("eager_launch_fwd_images_per_sec", fo["fwd_images_per_sec"]) / den, 2)
            if "graph_fwd_images_per_sec" in fo:
                fo["graph_replay_vs_eager_rocm"] = round(fo["graph_fwd_images_per_sec"] / den, 2)
            out["cpu_baseline"] = cpu_baseline(args.arch)
        print(json.dumps(out), flush=True)
    if dist_on:
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
