"""The legs beside the headline: the CPU restatement on the host cores, the eager restatement on this GPU, the inference pass,
and the child processes for the other configurations (BASELINE configs 4 and 5, drop_path 0, the detection backbone).
Only this module may import oracle/ (test infrastructure) -- as a BASELINE that is timed beside the product, never as the product."""
import json
import os
import subprocess
import sys
import time

import torch

from .common import BENCH, DET_LR, OTHER_CONFIGS, ROOT, autocast, make_step, sgd, timed


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arch, budget_s=90.0):
    """Eager CPU restatement, forward only, b=32 fp32.  The host's fair number: the thread count is
    swept ({16, 32, 64, 128} and the 8 of the survey container, capped at the logical CPUs; after a common warm-up every count
    gets one untimed and THREE timed iterations, the median counts), channels_last is tried at the best count the same way, and
    >= 3 more iterations are timed with the winner."""
    from oracle import eager_models as em
    cores = os.cpu_count() or 1
    cand = sorted({min(t, cores) for t in (8, 16, 32, 64, 128)})
    net = getattr(em, "eager_" + arch)().eval()
    xb = torch.randn(32, 3, 224, 224)
    t_start = time.perf_counter()
    tried = {}

    def median_of_3(x):
        net(x)                                                           # (this count's own warm-up: thread pool, primitives)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            net(x)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[1]

    with torch.no_grad():
        torch.set_num_threads(cand[len(cand) // 2])
        net(xb)                                                          # warm-up (allocator, oneDNN primitives)
        for t in cand:
            if time.perf_counter() - t_start > budget_s * 0.55 and tried:
                break
            torch.set_num_threads(t)
            tried[str(t)] = round(32 / median_of_3(xb), 2)
        best_t = int(max(tried, key=tried.get))
        torch.set_num_threads(best_t)
        fmt, x_best = "contiguous (NCHW)", xb
        if time.perf_counter() - t_start < budget_s * 0.7:
            net_cl, x_cl = net.to(memory_format=torch.channels_last), xb.contiguous(memory_format=torch.channels_last)
            cl = round(32 / median_of_3(x_cl), 2)
            tried[f"{best_t}+channels_last"] = cl
            if cl > tried[str(best_t)]:
                fmt, x_best = "channels_last", x_cl
            else:
                net.to(memory_format=torch.contiguous_format)
        n, t0 = 0, time.perf_counter()
        while n < 3 or (time.perf_counter() - t_start < budget_s * 0.9 and n < 10):
            net(x_best)
            n += 1
        dt = time.perf_counter() - t0
    return {"value": round(32 * n / dt, 2), "unit": "images/sec", "cores": best_t, "kind": "port",
            "cpu": cpu_model_name(), "logical_cpus": cores, "threads_tried": tried, "memory_format": fmt,
            "sample": f"forward only (eval, no_grad), fp32, batch 32, {n} iterations with torch.set_num_threads({best_t}) "
                      f"(the best of threads_tried: the median of three timed iterations per count after a warm-up), {fmt}"}


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
        with torch.no_grad(), autocast():
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


def eager_rocm_detection(x, steps=6):
    """The eager restatement of the mmdet backbone (oracle/eager_models.py: stock ATen / MIOpen ops) on this GPU, the same
    step as the product's (norm_eval, frozen stem + stage 1, bf16 autocast, SGD): fwd+bwd and forward-only images/sec."""
    from oracle import eager_models as em
    torch.manual_seed(0)
    net = em.EagerDetBackbone(frozen_stages=1, norm_eval=True).cuda().to(memory_format=torch.channels_last).train()
    opt = sgd((p for p in net.parameters() if p.requires_grad), lr=DET_LR)

    def step():
        with autocast():
            loss = sum(m.float().square().mean() for m in net(x))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    dt = timed(step, steps, 3)
    with torch.no_grad(), autocast():
        for _ in range(2):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            net(x)
        torch.cuda.synchronize()
        fw = (time.perf_counter() - t0) / steps
    return {"fwd_bwd_images_per_sec": round(x.shape[0] * steps / dt, 2), "fwd_bwd_ms": round(1e3 * dt / steps, 2),
            "fwd_images_per_sec": round(x.shape[0] / fw, 2),
            "what": "oracle/eager_models.py EagerDetBackbone (channels_last, stock ops) on this GPU, same images / dtype / optimizer"}


def forward_only(net, x, steps=10, graph=True):
    """Inference pass (eval, no_grad, bf16 autocast) of the product network: the numerator of the north-star's
    ">= 4x the eager PyTorch-ROCm forward" target (eager_rocm.fwd_images_per_sec is its denominator).  Timed both as
    PyTorch launches it and (graph=True) replayed from one HIP graph; `fwd_images_per_sec` is the faster of the two."""
    was = net.training
    net.eval()
    res = {"mode": "eval, no_grad, autocast as the timed step"}
    with torch.no_grad(), autocast():
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


def _child(extra, timeout=900):
    """`python bench.py <extra> --no-baselines` as a fresh process (its own MIOpen state and HIP graph; this process is idle
    meanwhile).  Returns (parsed line or None, error text or None, wall seconds)."""
    cmd = [sys.executable, BENCH] + extra + ["--no-baselines"]
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            return None, f"rc {p.returncode}: " + p.stderr.decode(errors="replace")[-400:], time.perf_counter() - t0
        return json.loads(lines[-1]), None, time.perf_counter() - t0
    except (subprocess.TimeoutExpired, ValueError) as e:
        return None, f"{type(e).__name__}: {e}"[:400], time.perf_counter() - t0


def _summary(rec, wall, **more):
    cfg = rec["config"]
    out = {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
           "launch": cfg["launch"], "roofline": rec["roofline"], "replay_matches_eager": cfg.get("replay_matches_eager"),
           "miopen": cfg.get("miopen"), "replay_check": cfg.get("replay_check"), "weights_finite": cfg.get("weights_finite"),
           "eager_launch_ms_per_step": rec.get("eager_launch_ms_per_step"), "workload": cfg["workload"], "wall_s": round(wall, 1)}
    out.update(more)
    return out


def run_other_configs():
    """BASELINE configs 4 and 5 as child processes of the default N = 1 run."""
    out = {}
    for arch, batch in OTHER_CONFIGS:
        rec, err, wall = _child(["--arch", arch, "--batch", str(batch), "--steps", "10", "--warmup", "3"])
        try:
            out[arch] = {"error": err} if rec is None else _summary(rec, wall, batch=batch)
        except KeyError as e:
            out[arch] = {"error": f"KeyError: {e}"}
    return out


def run_drop_path_0(arch, batch):
    """The headline's step with drop_path = 0 -- the reference's FACTORY default (resnet_mrla_light.py:134) beside train.py's 0.2
    (resnet/train.py:67): SURVEY.md section 8(d) asks for both.  10 steps of a child process."""
    rec, err, wall = _child(["--arch", arch, "--batch", str(batch), "--steps", "10", "--warmup", "3", "--drop-path", "0"])
    if rec is None:
        return {"error": err}
    return {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"], "steps": rec["steps"],
            "replay_matches_eager": rec["config"].get("replay_matches_eager"), "launch": rec["config"]["launch"],
            "wall_s": round(wall, 1)}


def run_detection_backbone():
    """SURVEY.md section 8(f) row 2 where it runs: the mmdet backbone (mmdetection/mmdet/models/backbones/resnet_mrlal.py:283-293,
    358-367: norm_eval, frozen stem + stage 1, four output maps) forward + backward on 2 x 3 x 800 x 1344, bf16."""
    rec, err, wall = _child(["--arch", "det_resnet50_mrlal", "--shape", "2x3x800x1344", "--steps", "10", "--warmup", "3"])
    try:
        return {"error": err} if rec is None else _summary(rec, wall, eager_rocm=rec.get("eager_rocm"),
                                                           mrla_kernels=rec.get("mrla_kernels"))
    except KeyError as e:
        return {"error": f"KeyError: {e}"}
