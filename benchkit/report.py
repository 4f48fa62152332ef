"""The ONE JSON line: roofline of the dominant MRLA kernel from the per-kernel events, committed counters tied to the library that
is loaded, and the contract keys."""
import json
import os

import torch

from .common import (DTYPE_NAME, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFLOPS, MODEL_GFLOP_PER_IMAGE, ROOT, ROUND)


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


def report(R, emit=True):
    """Rank 0: build the ONE JSON line from a finished measurement R (see main()), print it (emit) and return it."""
    from . import baselines
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
        kdt = "fp32" if args.arch.startswith("deit") else DTYPE_NAME[args.autocast]
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
    shape = tuple(x.shape[1:])
    precision = {"bf16": "bf16 autocast, fp32 master weights", "fp16": "fp16 autocast, fp32 master weights",
                 "none": "fp32 (no autocast: resnet/train.py's own recipe)"}[args.autocast]
    out = {"metric": f"images/sec fwd+bwd {args.arch} b={args.batch}", "value": round(ips, 1), "unit": "images/sec",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[args.autocast],
           "data": "synthetic",
           "config": {"workload": f"{args.arch} {R.get('what', 'fwd+bwd+SGD')}, {args.batch} images/GPU of "
                                  f"{shape[0]}x{shape[1]}x{shape[2]}, {precision}, drop_path {args.drop_path}",
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
    if R.get("ddp_first") is not None:
        out["config"]["ddp_eager_first"] = R["ddp_first"]
    if world == 1 and R["legs"] and not args.no_forward_only and not args.arch.startswith("det_"):
        out["forward_only"] = baselines.forward_only(net, x, graph=use_graph)
    if world == 1 and R["legs"] and not dist_on and args.arch.startswith("det_"):
        out["eager_rocm"] = baselines.eager_rocm_detection(x)
    elif world == 1 and R["legs"] and not dist_on and not args.no_baselines:
        out["eager_rocm"] = baselines.eager_rocm(args.arch, args.batch, args.drop_path)
        # like for like: both sides launched kernel by kernel by PyTorch (the eager restatement is never graph-replayed);
        # the graph-replayed product forward against the same denominator is reported beside it, labelled
        fo, den = out.get("forward_only"), out["eager_rocm"]["fwd_images_per_sec"]
        if fo is not None:                       # (--no-forward-only: no ratios)
            fo["vs_eager_rocm"] = round(fo.get("eager_launch_fwd_images_per_sec", fo["fwd_images_per_sec"]) / den, 2)
            if "graph_fwd_images_per_sec" in fo:
                fo["graph_replay_vs_eager_rocm"] = round(fo["graph_fwd_images_per_sec"] / den, 2)
        out["cpu_baseline"] = baselines.cpu_baseline(args.arch)
        if not args.no_others and (args.arch, args.batch, args.autocast) == ("resnet50_mrlal", 256, "bf16"):
            # this process goes idle: give its graph pool and cached blocks back first
            R["net"] = net = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            out["config"]["drop_path_0"] = baselines.run_drop_path_0(args.arch, args.batch)
            out["other_configs"] = baselines.run_other_configs()
            out["detection_backbone"] = baselines.run_detection_backbone()
    line = json.dumps(out)
    if emit:
        print(line, flush=True)
    return line
